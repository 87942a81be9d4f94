"""ctypes binding of libspp.so -- fails loudly when the HIP library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libspp.so")
_LIB = None

SPP_OK = 0
SPP_ERR_UNSAT = -4
SPP_CIRCUIT_WITHDRAW = 1
SPP_CIRCUIT_AUDIT = 2
SPP_CIRCUIT_WITHDRAW_DEPTH20 = 4     # build only: withdraw over a depth-20 tree (synthetic variant)
SPP_CIRCUIT_WITHDRAW_REFSHAPE = 3   # build only: withdraw padded to the reference's R1CS size (12 452 constraints)
PROOF_LEN = 388


class SppError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libspp error %d: %s" % (code, msg))
        self.code = code


class WithdrawInputs(ctypes.Structure):
    _fields_ = [("root", ctypes.c_uint8 * 32), ("nullifier", ctypes.c_uint8 * 32), ("recipient", ctypes.c_uint8 * 32),
                ("amount", ctypes.c_uint64), ("wa_commitment", ctypes.c_uint8 * 32), ("secret_key", ctypes.c_uint8 * 32),
                ("owner_x", ctypes.c_uint8 * 32), ("owner_y", ctypes.c_uint8 * 32), ("randomness", ctypes.c_uint8 * 32),
                ("index", ctypes.c_uint64), ("siblings", (ctypes.c_uint8 * 32) * 16)]


def load_library():
    """Returns the loaded libspp.so; raises if it has not been built (no fallback path exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("libspp.so not found at %s -- build it with __graft_entry__.build() "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # libspp keeps up to six batches in flight, each on a proving stream plus a side stream for the G2 sum: with the HIP
    # runtime's default of 4 hardware queues per device unrelated streams end up sharing a queue and run one after the other
    # (128-proof batches, four in flight: 3 755 proofs/s with 4 queues, 4 554 with 8).  Read when the runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    L = ctypes.CDLL(LIB_PATH)
    vp, cp, u32, sz, i32 = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_int
    L.spp_last_error.restype = cp
    L.spp_version.restype = cp
    L.spp_circuit_build.argtypes = [i32, vp, cp, ctypes.POINTER(u32)]
    L.spp_circuit_build_acir.argtypes = [cp, sz, i32, cp, ctypes.POINTER(u32)]
    L.spp_init.argtypes = [i32, ctypes.POINTER(vp)]
    L.spp_free_ctx.argtypes = [vp]
    L.spp_free_ctx.restype = None
    L.spp_setup.argtypes = [vp, cp, cp, cp, cp]
    L.spp_load_circuit.argtypes = [vp, cp, cp, i32, ctypes.POINTER(vp)]
    L.spp_free_circuit.argtypes = [vp]
    L.spp_free_circuit.restype = None
    L.spp_circuit_info.argtypes = [vp, ctypes.POINTER(u32)]
    L.spp_circuit_msm_sizes.argtypes = [vp, ctypes.POINTER(u32)]
    L.spp_circuit_msm_windows.argtypes = [vp, ctypes.POINTER(u32)]
    L.spp_circuit_msm_table_rows.argtypes = [vp, ctypes.POINTER(u32)]
    L.spp_circuit_small_rows.argtypes = [vp, ctypes.POINTER(u32)]
    L.spp_pk_msm_sizes.argtypes = [cp, ctypes.POINTER(u32)]
    L.spp_plan_windows.argtypes = [u32, ctypes.POINTER(u32), ctypes.c_double, ctypes.POINTER(u32)]
    L.spp_load_circuit_with_windows.argtypes = [vp, cp, cp, ctypes.POINTER(u32), ctypes.POINTER(vp)]
    L.spp_circuit_table_bytes.argtypes = [vp]
    L.spp_circuit_table_bytes.restype = ctypes.c_uint64
    L.spp_prove_batch.argtypes = [vp, sz, cp, cp, vp, vp, vp]
    L.spp_prove_batch_device.argtypes = [vp, sz, vp, vp, vp, vp, vp]
    L.spp_sync.argtypes = [vp]
    L.spp_last_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.spp_timings.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float)]
    L.spp_msm_kernel_ms.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float)]
    L.spp_commitment_challenge.argtypes = [vp, ctypes.c_size_t, ctypes.c_char_p, vp]
    L.spp_set_serial.argtypes = [vp, i32]
    L.spp_prove_withdraw.argtypes = [vp, ctypes.POINTER(WithdrawInputs), cp, vp, vp]
    L.spp_verify.argtypes = [cp, sz, cp, sz, cp, sz, ctypes.POINTER(i32)]
    L.spp_verify_batch.argtypes = [vp, cp, sz, sz, cp, cp, sz, vp, ctypes.POINTER(ctypes.c_float)]
    L.spp_pairing_check.argtypes = [vp, u32, cp, cp, ctypes.POINTER(i32)]
    L.spp_pairing_check_host.argtypes = [u32, cp, cp, ctypes.POINTER(i32)]
    L.spp_debug_witness.argtypes = [vp, vp, sz]
    L.spp_rlwe_witness_batch.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.spp_rlwe_witness_batch_device.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.spp_ctx_sync.argtypes = [vp]
    L.spp_poseidon_hash_batch.argtypes = [vp, sz, i32, cp, vp]
    L.spp_merkle_root_batch.argtypes = [vp, sz, u32, cp, vp, cp, vp]
    L.spp_merkle_build.argtypes = [vp, sz, u32, cp, sz, vp, vp, vp]
    L.spp_merkle_tree_new.argtypes = [vp, u32, ctypes.POINTER(vp)]
    L.spp_merkle_tree_free.argtypes = [vp]
    L.spp_merkle_tree_free.restype = None
    L.spp_merkle_tree_size.argtypes = [vp]
    L.spp_merkle_tree_size.restype = ctypes.c_uint64
    L.spp_merkle_tree_insert.argtypes = [vp, sz, cp, ctypes.POINTER(ctypes.c_uint64)]
    L.spp_merkle_tree_root.argtypes = [vp, vp]
    L.spp_merkle_tree_proofs.argtypes = [vp, sz, vp, vp]
    L.spp_grumpkin_keygen_batch.argtypes = [vp, sz, cp, vp]
    L.spp_poseidon2_sponge_batch.argtypes = [vp, sz, u32, cp, vp]
    L.spp_audit_inputs_batch.argtypes = [vp, vp, vp, sz, cp, vp, vp, vp, vp]
    L.spp_audit_inputs_batch_device.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, vp]
    L.spp_prove_audit_from_secrets_device.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.spp_shamir_reconstruct.argtypes = [vp, u32, vp, cp, sz, vp, vp]
    L.spp_rlwe_decrypt_batch.argtypes = [vp, vp, sz, vp, vp, vp]
    L.spp_ntt_fr.argtypes = [vp, vp, u32, i32]
    L.spp_msm_g1.argtypes = [vp, cp, cp, sz, i32, vp]
    L.spp_msm_g2.argtypes = [vp, cp, cp, sz, i32, vp]
    L.spp_msm_g1_pippenger.argtypes = [vp, cp, cp, sz, vp]
    L.spp_msm_g2_pippenger.argtypes = [vp, cp, cp, sz, vp]
    L.spp_msm_g1_pippenger_bench.argtypes = [vp, sz, ctypes.c_uint64, cp, i32, vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.spp_msm_g1_pippenger_bench_dist.argtypes = [vp, sz, ctypes.c_uint64, ctypes.c_uint32, cp, i32, vp, ctypes.POINTER(ctypes.c_float),
                                                  ctypes.POINTER(ctypes.c_float)]
    L.spp_msm_g1_pippenger_bench_shard.argtypes = [vp, sz, sz, sz, ctypes.c_uint64, ctypes.c_uint32, cp, i32, vp, ctypes.POINTER(ctypes.c_float),
                                                   ctypes.POINTER(ctypes.c_float)]
    _LIB = L
    return L


def last_error():
    return load_library().spp_last_error().decode()


def check(rc):
    if rc != 0:
        raise SppError(rc, last_error())

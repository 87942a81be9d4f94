"""Thin object layer over the C ABI: Context (one GPU), CircuitHandle (R1CS + pk + tables in HBM)."""
import ctypes
from .lib import load_library, check, SppError, PROOF_LEN, SPP_ERR_UNSAT


def build_circuit(circuit_id, out_path, aux=None):
    """`sunspot compile` equivalent (host only): writes the SPPC container, returns nbConstraints."""
    L = load_library()
    n = ctypes.c_uint32(0)
    auxbuf = None
    if aux is not None:
        auxbuf = (ctypes.c_uint32 * len(aux))(*[int(v) for v in aux])
    check(L.spp_circuit_build(int(circuit_id), auxbuf, out_path.encode(), ctypes.byref(n)))
    return n.value


def verify(vk_bytes, proof_bytes, pw_bytes):
    """`sunspot verify vk proof pw`: True / False; raises SppError on malformed inputs. Host only (no GPU)."""
    L = load_library()
    ok = ctypes.c_int(0)
    check(L.spp_verify(vk_bytes, len(vk_bytes), proof_bytes, len(proof_bytes), pw_bytes, len(pw_bytes), ctypes.byref(ok)))
    return bool(ok.value)


def pairing_check_host(pairs):
    """prod e(P, Q) == 1 with the product's host pairing (no GPU). pairs: list of (g1 64 B, g2 128 B)."""
    L = load_library()
    ok = ctypes.c_int(0)
    check(L.spp_pairing_check_host(len(pairs), b"".join(p for p, _ in pairs), b"".join(q for _, q in pairs), ctypes.byref(ok)))
    return bool(ok.value)


class Context:
    def __init__(self, device=0):
        self.L = load_library()
        h = ctypes.c_void_p()
        check(self.L.spp_init(int(device), ctypes.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.L.spp_free_ctx(self.h)
            self.h = None

    def setup(self, circuit_path, seed32, pk_path, vk_path):
        assert len(seed32) == 32
        check(self.L.spp_setup(self.h, circuit_path.encode(), bytes(seed32), pk_path.encode(), vk_path.encode()))

    def load_circuit(self, circuit_path, pk_path, window_bits=0, bits=None):
        """bits: 7 planned window sizes (plan_windows) instead of window_bits / a budget of this circuit's own."""
        return CircuitHandle(self, circuit_path, pk_path, window_bits, bits)

    def plan_windows(self, pk_paths, budget_bytes=240e9):
        """Window bits for circuits that are to be resident TOGETHER: one greedy split of the budget over the union of their MSM
        sets (spp_plan_windows).  Returns one list of 7 per proving key, in the order of CircuitHandle.msm_sizes()."""
        n = len(pk_paths)
        sizes = (ctypes.c_uint32 * (7 * n))()
        for k, path in enumerate(pk_paths):
            one = (ctypes.c_uint32 * 7)()
            check(self.L.spp_pk_msm_sizes(path.encode(), one))
            sizes[7 * k:7 * k + 7] = list(one)
        bits = (ctypes.c_uint32 * (7 * n))()
        check(self.L.spp_plan_windows(n, sizes, float(budget_bytes), bits))
        return [list(bits[7 * k:7 * k + 7]) for k in range(n)]

    def ntt(self, values, inverse=False):
        """values: list of ints (len 2^k) -> list of ints, natural order."""
        n = len(values)
        logn = n.bit_length() - 1
        assert 1 << logn == n
        buf = ctypes.create_string_buffer(b"".join(int(v).to_bytes(32, "big") for v in values), 32 * n)
        check(self.L.spp_ntt_fr(self.h, ctypes.cast(buf, ctypes.c_void_p), logn, 1 if inverse else 0))
        return [int.from_bytes(buf.raw[32 * i:32 * i + 32], "big") for i in range(n)]

    def msm_g1(self, bases_bytes, scalars, window_bits=8):
        n = len(scalars)
        out = ctypes.create_string_buffer(64)
        sc = b"".join(int(s).to_bytes(32, "big") for s in scalars)
        check(self.L.spp_msm_g1(self.h, bases_bytes, sc, n, int(window_bits), ctypes.cast(out, ctypes.c_void_p)))
        return out.raw

    def msm_g2(self, bases_bytes, scalars, window_bits=8):
        out = ctypes.create_string_buffer(128)
        sc = b"".join(int(s).to_bytes(32, "big") for s in scalars)
        check(self.L.spp_msm_g2(self.h, bases_bytes, sc, len(scalars), int(window_bits), ctypes.cast(out, ctypes.c_void_p)))
        return out.raw

    def msm_g1_pippenger(self, bases_bytes, scalars):
        out = ctypes.create_string_buffer(64)
        sc = b"".join(int(s).to_bytes(32, "big") for s in scalars)
        check(self.L.spp_msm_g1_pippenger(self.h, bases_bytes, sc, len(scalars), ctypes.cast(out, ctypes.c_void_p)))
        return out.raw

    def msm_g2_pippenger(self, bases_bytes, scalars):
        out = ctypes.create_string_buffer(128)
        sc = b"".join(int(s).to_bytes(32, "big") for s in scalars)
        check(self.L.spp_msm_g2_pippenger(self.h, bases_bytes, sc, len(scalars), ctypes.cast(out, ctypes.c_void_p)))
        return out.raw

    def pairing_check(self, pairs):
        """prod e(P, Q) == 1 on the GPU with the batched verifier's device pairing code (spp_pairing_check)."""
        ok = ctypes.c_int(0)
        check(self.L.spp_pairing_check(self.h, len(pairs), b"".join(p for p, _ in pairs), b"".join(q for _, q in pairs), ctypes.byref(ok)))
        return bool(ok.value)

    def verify_batch(self, vk, proofs, pws, want_ms=False):
        """`sunspot verify` for many proofs against one key, on the GPU (spp_verify_batch). proofs / pws: lists of bytes.
        Returns a list of booleans (and the kernel time in ms when want_ms)."""
        count = len(proofs)
        assert count == len(pws)
        pw_len = len(pws[0]) if count else 12
        ok = (ctypes.c_int32 * max(count, 1))()
        ms = ctypes.c_float(0)
        check(self.L.spp_verify_batch(self.h, vk, len(vk), count, b"".join(proofs), b"".join(pws), pw_len,
                                      ctypes.cast(ok, ctypes.c_void_p), ctypes.byref(ms)))
        res = [bool(ok[i]) for i in range(count)]
        return (res, ms.value) if want_ms else res

    def msm_g1_pippenger_bench(self, n, seed=5, scale=None, iters=1, small_permille=0):
        """Returns (result bytes, ms per MSM, ms of the bucket kernel). small_permille: share of byte-sized scalars."""
        out = ctypes.create_string_buffer(64)
        t, k = ctypes.c_float(0), ctypes.c_float(0)
        sb = None if scale is None else int(scale).to_bytes(32, "big")
        check(self.L.spp_msm_g1_pippenger_bench_dist(self.h, n, seed, int(small_permille), sb, iters, ctypes.cast(out, ctypes.c_void_p),
                                                     ctypes.byref(t), ctypes.byref(k)))
        return out.raw, t.value, k.value


    def msm_g1_pippenger_bench_shard(self, n_total, first, count, seed=5, scale=None, iters=1, small_permille=0):
        """The partial sum of points [first, first + count) of the n_total-point synthetic MSM (one rank's share): (bytes, ms, bucket ms)."""
        out = ctypes.create_string_buffer(64)
        t, k = ctypes.c_float(0), ctypes.c_float(0)
        sb = None if scale is None else int(scale).to_bytes(32, "big")
        check(self.L.spp_msm_g1_pippenger_bench_shard(self.h, n_total, first, count, seed, int(small_permille), sb, iters,
                                                      ctypes.cast(out, ctypes.c_void_p), ctypes.byref(t), ctypes.byref(k)))
        return out.raw, t.value, k.value


class CircuitHandle:
    def __init__(self, ctx, circuit_path, pk_path, window_bits=0, bits=None):
        self.ctx = ctx
        self.L = ctx.L
        h = ctypes.c_void_p()
        if bits is not None:
            arr = (ctypes.c_uint32 * 7)(*[int(b) for b in bits])
            check(self.L.spp_load_circuit_with_windows(ctx.h, circuit_path.encode(), pk_path.encode(), arr, ctypes.byref(h)))
        else:
            check(self.L.spp_load_circuit(ctx.h, circuit_path.encode(), pk_path.encode(), int(window_bits), ctypes.byref(h)))
        self.h = h
        info = (ctypes.c_uint32 * 8)()
        check(self.L.spp_circuit_info(self.h, info))
        (self.circuit_id, self.n_public, self.n_secret, self.n_wires, self.n_constraints, self.domain_log,
         self.n_inputs, self.window_bits) = list(info)
        self.pw_len = 12 + 32 * self.n_public

    def close(self):
        if self.h:
            self.L.spp_free_circuit(self.h)
            self.h = None

    def msm_sizes(self):
        s = (ctypes.c_uint32 * 7)()
        check(self.L.spp_circuit_msm_sizes(self.h, s))
        return list(s)

    def msm_windows(self):
        s = (ctypes.c_uint32 * 7)()
        check(self.L.spp_circuit_msm_windows(self.h, s))
        return list(s)

    def small_rows(self):
        """(rows of the matrix evaluation summed as integers, byte-ranged wires they read) -- spp_circuit_small_rows"""
        s = (ctypes.c_uint32 * 2)()
        check(self.L.spp_circuit_small_rows(self.h, s))
        return list(s)

    def msm_table_rows(self):
        """table rows per base and set: 1 = single-row tables walked once per window (window_bits = 0)"""
        s = (ctypes.c_uint32 * 7)()
        check(self.L.spp_circuit_msm_table_rows(self.h, s))
        return list(s)

    @property
    def table_bytes(self):
        return int(self.L.spp_circuit_table_bytes(self.h))

    def prove_batch(self, inputs, rs=None):
        """inputs: list (per proof) of lists of ints; rs: list of (r, s) ints or None (OS randomness).
        Returns (proofs, pws, status) with status[i] == 0 or SPP_ERR_UNSAT."""
        count = len(inputs)
        buf = b"".join(int(v).to_bytes(32, "big") for row in inputs for v in row)
        assert len(buf) == count * self.n_inputs * 32
        rsb = None
        if rs is not None:
            rsb = b"".join(int(r).to_bytes(32, "big") + int(s).to_bytes(32, "big") for r, s in rs)
        proofs = ctypes.create_string_buffer(PROOF_LEN * count)
        pws = ctypes.create_string_buffer(self.pw_len * count)
        status = (ctypes.c_int32 * count)()
        rc = self.L.spp_prove_batch(self.h, count, buf, rsb, ctypes.cast(proofs, ctypes.c_void_p),
                                    ctypes.cast(pws, ctypes.c_void_p), ctypes.cast(status, ctypes.c_void_p))
        if rc != 0 and rc != SPP_ERR_UNSAT:
            check(rc)
        return ([proofs.raw[PROOF_LEN * i:PROOF_LEN * (i + 1)] for i in range(count)],
                [pws.raw[self.pw_len * i:self.pw_len * (i + 1)] for i in range(count)], list(status))

    def commitment_challenge(self, inputs):
        """The commitment challenge the prover derives for each (possibly partial) input row: spp_commitment_challenge."""
        if isinstance(inputs, (bytes, bytearray)):      # rows already serialised: n_inputs x 32 B big-endian each
            buf = bytes(inputs)
            count = len(buf) // (self.n_inputs * 32)
        else:
            count = len(inputs)
            buf = b"".join(int(v).to_bytes(32, "big") for row in inputs for v in row)
        assert len(buf) == count * self.n_inputs * 32
        out = ctypes.create_string_buffer(32 * count)
        check(self.L.spp_commitment_challenge(self.h, count, buf, ctypes.cast(out, ctypes.c_void_p)))
        return [int.from_bytes(out.raw[32 * i:32 * i + 32], "big") for i in range(count)]

    def prove_batch_device(self, count, d_inputs, d_rs, d_proofs, d_pws, d_status):
        """All arguments are raw device pointers (ints), e.g. torch tensors' data_ptr()."""
        check(self.L.spp_prove_batch_device(self.h, count, d_inputs, d_rs, d_proofs, d_pws, d_status))

    def prove_audit_from_secrets_device(self, count, d_pk_a, d_pk_b, d_sk, d_r, d_e1, d_e2, d_rs, d_proofs, d_pws, d_status):
        """Audit proofs from (secret_key, r, e1, e2) resident on the device (spp_prove_audit_from_secrets_device): the input pipeline of
        scripts/generate_audit.py:468-641 runs on the proving stream in front of the solver.  Raw device pointers (ints)."""
        check(self.L.spp_prove_audit_from_secrets_device(self.h, count, d_pk_a, d_pk_b, d_sk, d_r, d_e1, d_e2, d_rs, d_proofs, d_pws, d_status))

    def sync(self):
        check(self.L.spp_sync(self.h))

    def last_timings(self, which=0):
        """which=0: last enqueued batch; 1: the batch before it (see spp_timings in include/spp.h)."""
        ms = (ctypes.c_float * 9)()
        check(self.L.spp_timings(self.h, int(which), ms))
        return list(ms)

    def msm_kernel_ms(self, which=0):
        """durations of the MSM kernel launches of a batch: [commitment, A, B1, K, Z, PoK, G2] (spp_msm_kernel_ms)."""
        ms = (ctypes.c_float * 7)()
        check(self.L.spp_msm_kernel_ms(self.h, int(which), ms))
        return list(ms)

    def set_serial(self, on):
        check(self.L.spp_set_serial(self.h, 1 if on else 0))

    def debug_witness(self):
        buf = ctypes.create_string_buffer(32 * self.n_wires)
        check(self.L.spp_debug_witness(self.h, ctypes.cast(buf, ctypes.c_void_p), self.n_wires))
        return [int.from_bytes(buf.raw[32 * i:32 * i + 32], "big") for i in range(self.n_wires)]

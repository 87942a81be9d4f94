"""ORACLE (test infrastructure only) -- ctypes binding of oracle/liboracle.so (the C restatement)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_load.restype = ctypes.c_void_p
        L.orc_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.orc_setup.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        L.orc_prove.argtypes = [ctypes.c_void_p] + [ctypes.c_char_p] * 3 + [ctypes.c_void_p] * 3
        for f in ("orc_n_inputs", "orc_n_public", "orc_n_wires", "orc_n_constraints"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
            getattr(L, f).restype = ctypes.c_uint32
        L.orc_prove_many.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_check_many.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p]
        L.orc_msm_g1.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_msm_g2.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_ntt.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
        _LIB = L
    return _LIB


def setup(circuit_path, seed32, pk_path, vk_path):
    rc = lib().orc_setup(circuit_path.encode(), bytes(seed32), pk_path.encode(), vk_path.encode())
    if rc != 0:
        raise RuntimeError("orc_setup failed: %d" % rc)


class Prover:
    def __init__(self, circuit_path, pk_path):
        self.h = lib().orc_load(circuit_path.encode(), pk_path.encode())
        if not self.h:
            raise RuntimeError("orc_load failed")
        self.n_inputs = lib().orc_n_inputs(self.h)
        self.n_public = lib().orc_n_public(self.h)
        self.n_wires = lib().orc_n_wires(self.h)
        self.n_constraints = lib().orc_n_constraints(self.h)

    def prove(self, inputs, r, s, want_wires=False):
        """inputs: ints (public then secret). Returns (rc, proof bytes, pw bytes[, wires ints])."""
        assert len(inputs) == self.n_inputs
        buf = b"".join(int(v).to_bytes(32, "big") for v in inputs)
        proof = ctypes.create_string_buffer(388)
        pw = ctypes.create_string_buffer(12 + 32 * self.n_public)
        wires = ctypes.create_string_buffer(32 * self.n_wires) if want_wires else None
        rc = lib().orc_prove(self.h, buf, int(r).to_bytes(32, "big"), int(s).to_bytes(32, "big"),
                             ctypes.cast(proof, ctypes.c_void_p), ctypes.cast(pw, ctypes.c_void_p),
                             ctypes.cast(wires, ctypes.c_void_p) if want_wires else None)
        out = (rc, proof.raw, pw.raw)
        if want_wires:
            w = [int.from_bytes(wires.raw[32 * i:32 * i + 32], "big") for i in range(self.n_wires)]
            out += (w,)
        return out


def prove_many(prover, rows, rs):
    """One proof per OpenMP thread (throughput mode). Returns (rc, proofs, pws)."""
    count = len(rows)
    buf = b"".join(int(v).to_bytes(32, "big") for row in rows for v in row)
    rsb = b"".join(int(r).to_bytes(32, "big") + int(s).to_bytes(32, "big") for r, s in rs)
    proofs = ctypes.create_string_buffer(388 * count)
    pwl = 12 + 32 * prover.n_public
    pws = ctypes.create_string_buffer(pwl * count)
    rc = lib().orc_prove_many(prover.h, count, buf, rsb, ctypes.cast(proofs, ctypes.c_void_p), ctypes.cast(pws, ctypes.c_void_p))
    return rc, [proofs.raw[388 * i:388 * (i + 1)] for i in range(count)], [pws.raw[pwl * i:pwl * (i + 1)] for i in range(count)]


def check_many(prover, rows):
    """Solver + satisfaction check for many input rows (parallel): list of -1 (satisfied) / first unsatisfied constraint / -2."""
    count = len(rows)
    buf = b"".join(int(v).to_bytes(32, "big") for row in rows for v in row)
    out = (ctypes.c_int32 * count)()
    lib().orc_check_many(prover.h, count, buf, ctypes.cast(out, ctypes.c_void_p))
    return list(out)


def set_threads(n):
    lib().orc_set_threads(int(n))


def max_threads():
    return lib().orc_max_threads()

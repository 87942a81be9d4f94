"""Host-side mirrors of the reference's client-side witness-input code, executed by the HIP kernels.

    rlwe_witness(...)        scripts/generate_audit.py:507-584  /  demo-frontend/app/lib/rlwe.ts:157-247
    poseidon_hash2/4(...)    client/merkle.ts:22-38
    ShieldedPoolMerkleTree   client/merkle.ts:146-222
    identity_public_key(...) client/merkle.ts:98-113
    ct_commitment(...)       ct_helper/src/main.nr:15-34
All of them take and return Python ints / lists; field elements cross the C ABI as 32-byte big-endian.
"""
import ctypes
import numpy as np
from .lib import check

TREE_DEPTH = 16
RLWE_N, MSG_SLOTS = 1024, 64


def _be(vals):
    return b"".join(int(v).to_bytes(32, "big") for v in vals)


def _unbe(buf, n):
    return [int.from_bytes(buf[32 * i:32 * i + 32], "big") for i in range(n)]


def rlwe_witness(ctx, pk_a, pk_b, r, e1, e2, msg):
    """Batch form: r, e2 arrays [count,1024]; e1, msg [count,64]. Returns dict of numpy arrays + packed fields."""
    r = np.ascontiguousarray(r, dtype=np.int8).reshape(-1, RLWE_N)
    count = r.shape[0]
    e1 = np.ascontiguousarray(e1, dtype=np.int8).reshape(count, MSG_SLOTS)
    e2 = np.ascontiguousarray(e2, dtype=np.int8).reshape(count, RLWE_N)
    msg = np.ascontiguousarray(msg, dtype=np.uint8).reshape(count, MSG_SLOTS)
    a = np.ascontiguousarray(pk_a, dtype=np.uint32)
    b = np.ascontiguousarray(pk_b, dtype=np.uint32)
    c0 = np.zeros((count, MSG_SLOTS), dtype=np.uint32)
    c1 = np.zeros((count, RLWE_N), dtype=np.uint32)
    k0 = np.zeros((count, MSG_SLOTS), dtype=np.int32)
    k1 = np.zeros((count, RLWE_N), dtype=np.int32)
    packed = np.zeros((count, 157, 32), dtype=np.uint8)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    check(ctx.L.spp_rlwe_witness_batch(ctx.h, p(a), p(b), count, p(r), p(e1), p(e2), p(msg), p(c0), p(c1), p(k0), p(k1), p(packed)))
    pk = [[int.from_bytes(packed[i, f].tobytes(), "big") for f in range(157)] for i in range(count)]
    return dict(c0=c0, c1=c1, k0=k0, k1=k1, c0_packed=[x[:10] for x in pk], c1_packed=[x[10:] for x in pk])


def poseidon_hash_batch(ctx, rows):
    """rows: list of [a, b] or [a, b, c, d] -> list of hashes."""
    if not rows:
        return []
    arity = len(rows[0])
    out = ctypes.create_string_buffer(32 * len(rows))
    check(ctx.L.spp_poseidon_hash_batch(ctx.h, len(rows), arity, _be(v for r in rows for v in r), ctypes.cast(out, ctypes.c_void_p)))
    return _unbe(out.raw, len(rows))


def poseidon_hash2(ctx, a, b):
    return poseidon_hash_batch(ctx, [[a, b]])[0]


def poseidon_hash4(ctx, a, b, c, d):
    return poseidon_hash_batch(ctx, [[a, b, c, d]])[0]


def merkle_roots(ctx, leaves, indices, siblings, depth=TREE_DEPTH):
    count = len(leaves)
    idx = (ctypes.c_uint64 * count)(*[int(i) for i in indices])
    out = ctypes.create_string_buffer(32 * count)
    check(ctx.L.spp_merkle_root_batch(ctx.h, count, depth, _be(leaves), ctypes.cast(idx, ctypes.c_void_p),
                                      _be(s for row in siblings for s in row), ctypes.cast(out, ctypes.c_void_p)))
    return _unbe(out.raw, count)


class ShieldedPoolMerkleTree:
    """client/merkle.ts:146-222 with the tree RESIDENT in HBM and maintained incrementally (spp_merkle_tree_*): insert() is
    O(depth) hashes, getRoot() one read, getProof() `depth` reads -- the reference recomputes up to 2^16 hashes in every
    getRoot / getProof call (merkle.ts:165-176, 198-221)."""

    def __init__(self, ctx, depth=TREE_DEPTH):
        self.ctx, self.depth = ctx, depth
        h = ctypes.c_void_p()
        check(ctx.L.spp_merkle_tree_new(ctx.h, depth, ctypes.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.spp_merkle_tree_free(self.h)
            self.h = None

    # the levels live in HBM: release them when the object goes away (context manager or garbage collection), not only on close()
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: the library may already be gone
            pass

    def __len__(self):
        return int(self.ctx.L.spp_merkle_tree_size(self.h))

    def insert(self, commitment):
        return self.insert_many([commitment])

    def insert_many(self, commitments):
        """appends the leaves in order; returns the index of the first one"""
        first = ctypes.c_uint64(0)
        check(self.ctx.L.spp_merkle_tree_insert(self.h, len(commitments), _be(commitments), ctypes.byref(first)))
        return int(first.value)

    def getRoot(self):
        root = ctypes.create_string_buffer(32)
        check(self.ctx.L.spp_merkle_tree_root(self.h, ctypes.cast(root, ctypes.c_void_p)))
        return int.from_bytes(root.raw, "big")

    def getProofs(self, indices, raw=False):
        nq = len(indices)
        q = (ctypes.c_uint64 * max(nq, 1))(*[int(i) for i in indices])
        sib = ctypes.create_string_buffer(32 * self.depth * max(nq, 1))
        check(self.ctx.L.spp_merkle_tree_proofs(self.h, nq, ctypes.cast(q, ctypes.c_void_p), ctypes.cast(sib, ctypes.c_void_p)))
        if raw:
            return sib.raw[:32 * self.depth * nq]
        return [_unbe(sib.raw[32 * self.depth * i:], self.depth) for i in range(nq)]

    def getProof(self, index):
        return self.getProofs([index])[0]


def merkle_build(ctx, leaves, queries, depth=TREE_DEPTH):
    """One-shot form (spp_merkle_build): all levels recomputed from the leaves, like the reference's getRoot/getProof.
    Returns (root, [siblings per query])."""
    nq = len(queries)
    q = (ctypes.c_uint64 * max(nq, 1))(*[int(i) for i in queries])
    sib = ctypes.create_string_buffer(32 * depth * max(nq, 1))
    root = ctypes.create_string_buffer(32)
    check(ctx.L.spp_merkle_build(ctx.h, len(leaves), depth, _be(leaves), nq, ctypes.cast(q, ctypes.c_void_p),
                                 ctypes.cast(sib, ctypes.c_void_p), ctypes.cast(root, ctypes.c_void_p)))
    return int.from_bytes(root.raw, "big"), [_unbe(sib.raw[32 * depth * i:], depth) for i in range(nq)]


def identity_public_keys(ctx, secret_keys):
    count = len(secret_keys)
    out = ctypes.create_string_buffer(64 * count)
    check(ctx.L.spp_grumpkin_keygen_batch(ctx.h, count, _be(secret_keys), ctypes.cast(out, ctypes.c_void_p)))
    v = _unbe(out.raw, 2 * count)
    return [(v[2 * i], v[2 * i + 1]) for i in range(count)]


def ct_commitments(ctx, packed_rows):
    """packed_rows: list of lists of field elements (157 for the audit ciphertext)."""
    count, n = len(packed_rows), len(packed_rows[0])
    out = ctypes.create_string_buffer(32 * count)
    check(ctx.L.spp_poseidon2_sponge_batch(ctx.h, count, n, _be(v for r in packed_rows for v in r), ctypes.cast(out, ctypes.c_void_p)))
    return _unbe(out.raw, count)


def audit_input_rows(ctx, pk_a, pk_b, secret_keys, r, e1, e2):
    """scripts/generate_audit.py:468-641 for a batch, on the GPU: returns one 3360-element input row per instance
    (wa_commitment, ct_commitment, c0_packed, c1_packed, secret_key, r, e1_sparse, e2, k0, k1 as field elements)."""
    count = len(secret_keys)
    r = np.ascontiguousarray(r, dtype=np.int8).reshape(count, RLWE_N)
    e1 = np.ascontiguousarray(e1, dtype=np.int8).reshape(count, MSG_SLOTS)
    e2 = np.ascontiguousarray(e2, dtype=np.int8).reshape(count, RLWE_N)
    a = np.ascontiguousarray(pk_a, dtype=np.uint32)
    b = np.ascontiguousarray(pk_b, dtype=np.uint32)
    rows = np.zeros((count, 3360 * 32), dtype=np.uint8)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    check(ctx.L.spp_audit_inputs_batch(ctx.h, p(a), p(b), count, _be(secret_keys), p(r), p(e1), p(e2), p(rows)))
    return [_unbe(rows[i].tobytes(), 3360) for i in range(count)]


# ---- auditor side: demo-frontend/app/lib/shamir.ts reconstructSk / rlweDecrypt, scripts/rlwe_decrypt.py ----
def reconstruct_sk(ctx, shares):
    """shares: list of {"x": int, "y": [hex or int] * 1024} (threshold = len(shares)); returns sk mod q (list of ints)."""
    t = len(shares)
    n = len(shares[0]["y"])
    xs = (ctypes.c_uint32 * t)(*[int(s["x"]) for s in shares])
    ys = _be((int(v, 16) if isinstance(v, str) else int(v)) for s in shares for v in s["y"])
    out = (ctypes.c_uint32 * n)()
    check(ctx.L.spp_shamir_reconstruct(ctx.h, t, ctypes.cast(xs, ctypes.c_void_p), ys, n, None, ctypes.cast(out, ctypes.c_void_p)))
    return list(out)


def rlwe_decrypt(ctx, sk_mod_q, c0, c1):
    """Batch: c0 [count,64], c1 [count,1024] -> list of (owner_x, owner_y) and the raw byte slots."""
    c0 = np.ascontiguousarray(c0, dtype=np.uint32).reshape(-1, MSG_SLOTS)
    count = c0.shape[0]
    c1 = np.ascontiguousarray(c1, dtype=np.uint32).reshape(count, RLWE_N)
    sk = np.ascontiguousarray(sk_mod_q, dtype=np.uint32)
    msg = np.zeros((count, MSG_SLOTS), dtype=np.uint8)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    check(ctx.L.spp_rlwe_decrypt_batch(ctx.h, p(sk), count, p(c0), p(c1), p(msg)))
    owners = [(int.from_bytes(msg[i, :32].tobytes(), "little"), int.from_bytes(msg[i, 32:].tobytes(), "little")) for i in range(count)]
    return owners, msg

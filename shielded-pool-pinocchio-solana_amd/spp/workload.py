"""Synthetic batches of DISTINCT, valid proving inputs, produced with the HIP witness-input kernels.

The shapes follow the reference's own callers:
  withdraw  client/payroll-demo.ts:199-352 -- every recipient gets a fresh identity (merkle.ts:98-113), a note
            commitment H4(owner_x, owner_y, amount, randomness) (:126-133) inserted in ONE tree (:146-222); each proof
            then uses its own index / siblings against the common root.
  audit     scripts/generate_audit.py:468-505 with SURVEY 8d Config 3's per-proof parameters: sk_i = 12345 + i,
            rng_i = Random(1000 + i), draw order r[1024], e1[64], e2[1024] (:503-505).
Every row of a batch is different, so the table gathers of the MSM kernels see independent scalars.
Rows are returned as bytes (count * n_inputs * 32, big-endian field elements) ready for spp_prove_batch(_device).
"""
import ctypes
import random
import numpy as np
from .lib import check
from . import witness as W

RLWE_N, MSG_SLOTS = 1024, 64


def withdraw_rows(ctx, count, seed=2, depth=16, first_index=0):
    """count distinct notes in one depth-`depth` tree -> (rows bytes, list of row ints for a few spot checks)."""
    rng = random.Random(seed)
    sks = [rng.randrange(1, 1 << 128) for _ in range(count)]
    amounts = [rng.randrange(1, 1 << 40) for _ in range(count)]
    rnds = [rng.randrange(1 << 250) for _ in range(count)]
    recipients = [rng.randrange(1, 1 << 240) for _ in range(count)]
    owners = W.identity_public_keys(ctx, sks)
    commitments = W.poseidon_hash_batch(ctx, [[o[0], o[1], a, r] for o, a, r in zip(owners, amounts, rnds)])
    idx = list(range(count))
    nulls = W.poseidon_hash_batch(ctx, [[s, i] for s, i in zip(sks, idx)])
    was = W.poseidon_hash_batch(ctx, [[o[0], o[1]] for o in owners])
    tree = W.ShieldedPoolMerkleTree(ctx, depth)          # device-resident, incremental (spp_merkle_tree_*)
    try:
        assert tree.insert_many(commitments) == 0
        root_b = tree.getRoot().to_bytes(32, "big")
        sib_b = tree.getProofs(idx, raw=True)
    finally:
        tree.close()
    be = lambda v: int(v).to_bytes(32, "big")
    out = bytearray()
    for i in range(count):
        out += root_b + be(nulls[i]) + be(recipients[i]) + be(amounts[i]) + be(was[i])
        out += be(sks[i]) + be(owners[i][0]) + be(owners[i][1]) + be(rnds[i]) + be(idx[i])
        out += sib_b[32 * depth * i:32 * depth * (i + 1)]
    return bytes(out)


def audit_noise(first, count, seed_base=1000):
    """(sk list, r, e1, e2 int8 arrays) of proofs first .. first+count-1 of SURVEY 8d Config 3."""
    r = np.zeros((count, RLWE_N), dtype=np.int8)
    e1 = np.zeros((count, MSG_SLOTS), dtype=np.int8)
    e2 = np.zeros((count, RLWE_N), dtype=np.int8)
    for k in range(count):
        rng = random.Random(seed_base + first + k)
        r[k] = [rng.randint(-3, 3) for _ in range(RLWE_N)]
        e1[k] = [rng.randint(-3, 3) for _ in range(MSG_SLOTS)]
        e2[k] = [rng.randint(-3, 3) for _ in range(RLWE_N)]
    return [12345 + first + k for k in range(count)], r, e1, e2


def audit_rows(ctx, pk_a, pk_b, count, first=0, seed_base=1000):
    """rows bytes (count * 3360 * 32) for proofs first .. first+count-1, built on the GPU (spp_audit_inputs_batch)."""
    sks, r, e1, e2 = audit_noise(first, count, seed_base)
    a = np.ascontiguousarray(pk_a, dtype=np.uint32)
    b = np.ascontiguousarray(pk_b, dtype=np.uint32)
    rows = np.zeros((count, 3360 * 32), dtype=np.uint8)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    check(ctx.L.spp_audit_inputs_batch(ctx.h, p(a), p(b), count, b"".join(int(s).to_bytes(32, "big") for s in sks),
                                       p(r), p(e1), p(e2), p(rows)))
    return rows.tobytes()


def row_ints(rows_bytes, n_inputs, i):
    """row i of a rows blob as a list of ints (for handing a sample to a checker)."""
    base = 32 * n_inputs * i
    return [int.from_bytes(rows_bytes[base + 32 * k:base + 32 * k + 32], "big") for k in range(n_inputs)]

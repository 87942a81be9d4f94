"""GPU parity tests: the HIP path (through the C ABI, libspp.so) against the CPU oracle on the same inputs.
Bit-exact comparison everywhere (integer / byte work)."""
import os
import random
import pytest
try:
    import torch  # noqa: F401  (before libspp: both must share ONE HIP runtime; torch's has to be loaded first)
except Exception:  # pragma: no cover
    torch = None

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import spp
    c = spp.Context(0)
    yield c
    c.close()


def test_ntt_matches_oracle(ctx):
    from oracle import native
    import ctypes
    rng = random.Random(5)
    from oracle.bn254 import R
    for logn in (1, 4, 8, 9, 13, 14, 15):      # 2^13 / 2^14 / 2^15 = the proving domains (withdraw, reference-size withdraw, audit)
        n = 1 << logn
        vals = [rng.randrange(R) for _ in range(n)]
        for inverse in (False, True):
            got = ctx.ntt(vals, inverse)
            buf = ctypes.create_string_buffer(b"".join(v.to_bytes(32, "big") for v in vals), 32 * n)
            native.lib().orc_ntt(ctypes.cast(buf, ctypes.c_void_p), logn, 1 if inverse else 0)
            exp = [int.from_bytes(buf.raw[32 * i:32 * i + 32], "big") for i in range(n)]
            assert got == exp, (logn, inverse)
    # round trip at the audit size
    vals = [rng.randrange(R) for _ in range(1 << 15)]
    assert ctx.ntt(ctx.ntt(vals, False), True) == vals


def test_msm_g1_matches_oracle(ctx):
    from oracle import bn254 as B, native
    import ctypes
    rng = random.Random(9)
    pts = []
    p = B.G1_GEN
    for i in range(150):
        p = B.g1_add(p, B.g1_mul(B.G1_GEN, rng.randrange(1, 1 << 64)))
        pts.append(p)
    bases = b"".join(B.g1_to_bytes(q) for q in pts)
    for n, wb in ((0, 6), (1, 6), (7, 4), (150, 6), (150, 8)):
        sc = [rng.randrange(B.R) for _ in range(n)]
        if n >= 7:
            sc[0] = 0
            sc[1] = 1
            sc[2] = B.R - 1          # -1: sign folding
            sc[3] = (B.R - 1) // 2   # largest positive magnitude
            sc[4] = (B.R + 1) // 2
            sc[5] = 255
        got = ctx.msm_g1(bases[:64 * n], sc, wb)
        out = ctypes.create_string_buffer(64)
        native.lib().orc_msm_g1(bases[:64 * n], b"".join(s.to_bytes(32, "big") for s in sc), n, ctypes.cast(out, ctypes.c_void_p))
        assert got == out.raw, (n, wb)


def test_msm_g2_matches_oracle(ctx):
    """The G2 table walk (k_msm_fixed<Fq2>, the MSM behind a proof's Bs) on its own: random twist points and the same scalar
    edge cases as the G1 test, against the oracle's C Pippenger over Fq2 and, for one case, the Python big-int sum."""
    from oracle import bn254 as B, native
    import ctypes
    rng = random.Random(19)
    pts = []
    p = B.G2_GEN
    for i in range(70):
        p = B.g2_add(p, B.g2_mul(B.G2_GEN, rng.randrange(1, 1 << 64)))
        pts.append(p)
    bases = b"".join(B.g2_to_bytes(q) for q in pts)
    for n, wb in ((0, 6), (1, 6), (7, 4), (70, 6), (70, 8)):
        sc = [rng.randrange(B.R) for _ in range(n)]
        if n >= 7:
            sc[0], sc[1], sc[2], sc[3], sc[4], sc[5] = 0, 1, B.R - 1, (B.R - 1) // 2, (B.R + 1) // 2, 255
        got = ctx.msm_g2(bases[:128 * n], sc, wb)
        out = ctypes.create_string_buffer(128)
        native.lib().orc_msm_g2(bases[:128 * n], b"".join(s.to_bytes(32, "big") for s in sc), n, ctypes.cast(out, ctypes.c_void_p))
        assert got == out.raw, (n, wb)
        if n == 7:
            acc = None
            for q, k in zip(pts, sc):
                acc = B.g2_add(acc, B.g2_mul(q, k))
            assert got == B.g2_to_bytes(acc)
    # repeated and cancelling bases: doubling and cancellation paths of XYZZ29G2::madd
    g = B.g2_mul(B.G2_GEN, 0x77)
    neg = B.g2_neg(g)
    mix = [g if i % 3 else neg for i in range(200)]
    sc = [5] * 200
    got = ctx.msm_g2(b"".join(B.g2_to_bytes(q) for q in mix), sc, 8)
    out = ctypes.create_string_buffer(128)
    native.lib().orc_msm_g2(b"".join(B.g2_to_bytes(q) for q in mix), b"".join(s.to_bytes(32, "big") for s in sc), 200, ctypes.cast(out, ctypes.c_void_p))
    assert got == out.raw


def test_pippenger_g2_matches_oracle(ctx):
    """The general-base Pippenger over G2 (shared digit / sort kernels, bucket kernels on the G2 accumulator) against the oracle's
    MSM: random twist points with the scalar edge cases, repeated and cancelling bases inside one bucket, the empty sum, and
    linearity at a size the oracle does not need to see (MSM(k s) = k MSM(s))."""
    from oracle import bn254 as B, native
    import ctypes
    rng = random.Random(23)
    pts = []
    p = B.G2_GEN
    for i in range(90):
        p = B.g2_add(p, B.g2_mul(B.G2_GEN, rng.randrange(1, 1 << 64)))
        pts.append(p)
    bases = b"".join(B.g2_to_bytes(q) for q in pts)

    def oracle(bb, sc):
        out = ctypes.create_string_buffer(128)
        native.lib().orc_msm_g2(bb, b"".join(s.to_bytes(32, "big") for s in sc), len(sc), ctypes.cast(out, ctypes.c_void_p))
        return out.raw
    for n in (0, 1, 7, 90):
        sc = [rng.randrange(B.R) for _ in range(n)]
        if n >= 7:
            sc[0], sc[1], sc[2], sc[3], sc[4], sc[5] = 0, 1, B.R - 1, (B.R - 1) // 2, (B.R + 1) // 2, 255
        assert ctx.msm_g2_pippenger(bases[:128 * n], sc) == oracle(bases[:128 * n], sc), n
    g = B.g2_mul(B.G2_GEN, 0x77)
    mix = b"".join(B.g2_to_bytes(g if i % 3 else B.g2_neg(g)) for i in range(200))
    assert ctx.msm_g2_pippenger(mix, [5] * 200) == oracle(mix, [5] * 200)
    # linearity on 3000 points (bases repeat, scalars do not)
    big = bases * 34
    sc = [rng.randrange(B.R) for _ in range(len(big) // 128)]
    k = rng.randrange(1, B.R)
    a = ctx.msm_g2_pippenger(big, sc)
    b = ctx.msm_g2_pippenger(big, [s * k % B.R for s in sc])
    assert b == B.g2_to_bytes(B.g2_mul(B.g2_from_bytes(a), k))


def test_msm_g1_repeated_and_cancelling_bases(ctx):
    """Many copies of one base (and of its negative) with equal scalars: inside a lane the accumulator meets the very
    point it holds (doubling path of XYZZ29::madd) or its negative (cancellation to infinity and restart)."""
    from oracle import bn254 as B, native
    import ctypes
    rng = random.Random(21)
    g = B.g1_mul(B.G1_GEN, 0x1234567)
    neg = (g[0], (B.P - g[1]) % B.P)
    other = B.g1_mul(B.G1_GEN, 99)
    for pattern in ("same", "cancel", "mixed"):
        n = 1500
        if pattern == "same":
            pts, sc = [g] * n, [5] * n
        elif pattern == "cancel":
            pts, sc = [g if i % 2 == 0 else neg for i in range(n)], [77] * n
            pts[-1], sc[-1] = other, 3                      # leave something non-trivial
        else:
            pts = [rng.choice((g, neg, other)) for _ in range(n)]
            sc = [rng.choice((1, 2, 255, B.R - 1, rng.randrange(B.R))) for _ in range(n)]
        bases = b"".join(B.g1_to_bytes(q) for q in pts)
        got = ctx.msm_g1(bases, sc, 8)
        out = ctypes.create_string_buffer(64)
        native.lib().orc_msm_g1(bases, b"".join(s.to_bytes(32, "big") for s in sc), n, ctypes.cast(out, ctypes.c_void_p))
        assert got == out.raw, pattern
        if pattern == "mixed":
            # bases at infinity (64 zero bytes) contribute nothing, whatever their scalars
            holes = list(range(0, n, 7))
            with_inf = bytearray(bases)
            sc2 = list(sc)
            for i in holes:
                with_inf[64 * i:64 * i + 64] = bytes(64)
            kept = [i for i in range(n) if i not in set(holes)]
            native.lib().orc_msm_g1(b"".join(bases[64 * i:64 * i + 64] for i in kept), b"".join(sc2[i].to_bytes(32, "big") for i in kept),
                                    len(kept), ctypes.cast(out, ctypes.c_void_p))
            assert ctx.msm_g1(bytes(with_inf), sc2, 8) == out.raw


def test_setup_matches_oracle(ctx, withdraw_artifacts, workdir):
    pk2 = os.path.join(workdir, "gpu.pk")
    vk2 = os.path.join(workdir, "gpu.vk")
    ctx.setup(withdraw_artifacts["sppc"], b"\x07" * 32, pk2, vk2)
    assert open(vk2, "rb").read() == open(withdraw_artifacts["vk"], "rb").read()
    assert open(pk2, "rb").read() == open(withdraw_artifacts["pk"], "rb").read()
    assert os.path.getsize(vk2) == 1296   # size of the reference's shielded_pool_verifier.vk


@pytest.fixture(scope="module")
def withdraw_handle(ctx, withdraw_artifacts):
    h = ctx.load_circuit(withdraw_artifacts["sppc"], withdraw_artifacts["pk"], 6)
    yield h
    h.close()


def _withdraw_variants(kat, count):
    """count input vectors: the reference KAT first, then fresh notes inserted in a tree (client/merkle.ts)."""
    from oracle import circuit as C, hashes as H
    rng = random.Random(77)
    rows = [C.withdraw_inputs(kat)]
    tree = H.MerkleTree()
    notes = []
    for i in range(count - 1):
        sk = rng.randrange(1, 1 << 128)
        owner = H.fixed_base_scalar_mul(sk)
        amount = rng.randrange(1, 1 << 40)
        rnd = rng.randrange(1 << 250)
        cm = H.poseidon_hash4(owner[0], owner[1], amount, rnd)
        idx = tree.insert(cm)
        notes.append((sk, owner, amount, rnd, idx))
    root = tree.root()
    for sk, owner, amount, rnd, idx in notes:
        rows.append([root, H.poseidon_hash2(sk, idx), rng.randrange(1, 1 << 240), amount, H.poseidon_hash2(owner[0], owner[1]),
                     sk, owner[0], owner[1], rnd, idx] + tree.proof(idx))
    return rows


def test_withdraw_proof_bytes_match_oracle_and_verify(withdraw_handle, withdraw_artifacts, withdraw_kat):
    from oracle import native, groth16
    rows = _withdraw_variants(withdraw_kat, 5)
    rs = [(1000 + i, 2000 + 7 * i) for i in range(len(rows))]
    proofs, pws, status = withdraw_handle.prove_batch(rows, rs)
    assert status == [0] * len(rows)
    orc = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    vk = open(withdraw_artifacts["vk"], "rb").read()
    for i, row in enumerate(rows):
        rc, proof, pw = orc.prove(row, rs[i][0], rs[i][1])
        assert rc == 0
        assert pws[i] == pw == groth16.public_witness_bytes(row[:5])
        assert proofs[i] == proof, "proof %d differs from the oracle" % i
    assert groth16.verify(vk, proofs[0], pws[0])
    assert groth16.verify(vk, proofs[3], pws[3])
    # the commitment hides: the same inputs under other blinding factors give another commitment point (proof bytes 260..324)
    again, _, st2 = withdraw_handle.prove_batch([rows[0], rows[0]], [rs[0], (rs[0][0] + 1, rs[0][1])])
    assert st2 == [0, 0] and again[0] == proofs[0] and again[1][260:324] != proofs[0][260:324]
    bad = bytearray(proofs[0])
    bad[0] ^= 1                     # client/test-shielded-pool.ts:386-392 corrupts byte 0
    assert not groth16.verify(vk, bytes(bad), pws[0])


def test_full_size_batch_is_consistent_with_small_batches(withdraw_handle, withdraw_artifacts, withdraw_kat):
    """Bench-size batch (4096 proofs per launch sequence) through a size-independent property: lanes that carry the
    same inputs and blinding produce the same bytes wherever they sit in the batch, those bytes are the oracle's, and
    one unsatisfiable row in the middle is refused without disturbing its neighbours."""
    from oracle import native
    rows4 = _withdraw_variants(withdraw_kat, 4)
    rs4 = [(31 + i, 57 + 2 * i) for i in range(4)]
    B_ = 4096
    rows = [rows4[i % 4] for i in range(B_)]
    rs = [rs4[i % 4] for i in range(B_)]
    bad_at = 3027
    rows[bad_at] = [rows[bad_at][0] + 1] + rows[bad_at][1:]
    proofs, pws, status = withdraw_handle.prove_batch(rows, rs)
    assert status[bad_at] != 0 and proofs[bad_at] == bytes(388)
    assert all(st == 0 for i, st in enumerate(status) if i != bad_at)
    for i in range(B_):
        if i != bad_at:
            assert proofs[i] == proofs[i % 4] and pws[i] == pws[i % 4], i
    orc = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    for i in range(4):
        rc, proof, pw = orc.prove(rows4[i], rs4[i][0], rs4[i][1])
        assert rc == 0 and proofs[i] == proof and pws[i] == pw


def test_full_batch_of_distinct_withdraw_rows(ctx, withdraw_handle, withdraw_artifacts):
    """A bench-size batch in which EVERY row is a different note (4096 identities, one tree, every index / sibling path
    different: spp/workload.py, the shape of client/payroll-demo.ts:199-352): all proofs produced, a sample byte-identical
    to the oracle's proofs of the same rows, every proof accepted by the batched verifier, inputs agree with the oracle's
    hashes (pinned by client/prover-params.toml)."""
    from spp import workload
    from oracle import native, groth16, hashes as H
    B_ = 4096
    rows_b = workload.withdraw_rows(ctx, B_, seed=5)
    n_in = withdraw_handle.n_inputs
    assert len(rows_b) == B_ * n_in * 32
    assert len({rows_b[32 * n_in * i + 32:32 * n_in * (i + 1)] for i in range(B_)}) == B_          # all rows distinct (root aside)
    import ctypes
    rs = b"".join((7919 * i + 3).to_bytes(32, "big") + (104729 * i + 5).to_bytes(32, "big") for i in range(B_))
    proofs = ctypes.create_string_buffer(388 * B_)
    pws = ctypes.create_string_buffer(withdraw_handle.pw_len * B_)
    status = (ctypes.c_int32 * B_)()
    rc = withdraw_handle.L.spp_prove_batch(withdraw_handle.h, B_, rows_b, rs, ctypes.cast(proofs, ctypes.c_void_p),
                                           ctypes.cast(pws, ctypes.c_void_p), ctypes.cast(status, ctypes.c_void_p))
    assert rc == 0 and not any(status)
    pl = [proofs.raw[388 * i:388 * (i + 1)] for i in range(B_)]
    wl = [pws.raw[withdraw_handle.pw_len * i:withdraw_handle.pw_len * (i + 1)] for i in range(B_)]
    assert len(set(pl)) == B_
    orc = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    rng = random.Random(99)
    sample = [0, 1, 63, 64, 2047, 2048, B_ - 1] + [rng.randrange(B_) for _ in range(9)]
    for i in sample:
        row = workload.row_ints(rows_b, n_in, i)
        rc, proof, pw = orc.prove(row, 7919 * i + 3, 104729 * i + 5)
        assert rc == 0 and pl[i] == proof and wl[i] == pw, i
        # the row itself is what the reference's client code would compute (oracle/hashes.py restates client/merkle.ts)
        sk, ox, oy, amount, rnd, idx = row[5], row[6], row[7], row[3], row[8], row[9]
        assert H.fixed_base_scalar_mul(sk) == (ox, oy) and idx == i
        assert row[1] == H.poseidon_hash2(sk, idx) and row[4] == H.poseidon_hash2(ox, oy)
        assert H.compute_merkle_root(H.poseidon_hash4(ox, oy, amount, rnd), idx, row[10:26]) == row[0]
    vk = open(withdraw_artifacts["vk"], "rb").read()
    assert all(ctx.verify_batch(vk, pl, wl))
    assert groth16.verify(vk, pl[sample[-1]], wl[sample[-1]])


def test_batch_of_distinct_audit_rows(ctx, audit_artifacts, rlwe_pk):
    """SURVEY 8d Config 3's rows (sk_i = 12345 + i, Random(1000 + i)) built on the GPU, every row distinct: 256 proofs, a
    sample byte-identical to the oracle's, all accepted by the batched verifier; the rows equal the oracle's restatement
    of scripts/generate_audit.py:468-641 for the same parameters."""
    from spp import workload
    from oracle import native, rlwe
    B_ = 256
    rows_b = workload.audit_rows(ctx, rlwe_pk["a"], rlwe_pk["b"], B_, first=40)
    h = ctx.load_circuit(audit_artifacts["sppc"], audit_artifacts["pk"], 6)
    try:
        import ctypes
        # the 1 088 quotient equations (generate_audit.py:539-554) take the integer path of the matrix evaluation
        assert h.small_rows()[0] == 1088 and h.small_rows()[1] > 2112
        rs = b"".join((31 * i + 3).to_bytes(32, "big") + (37 * i + 5).to_bytes(32, "big") for i in range(B_))
        proofs = ctypes.create_string_buffer(388 * B_)
        pws = ctypes.create_string_buffer(h.pw_len * B_)
        status = (ctypes.c_int32 * B_)()
        rc = h.L.spp_prove_batch(h.h, B_, rows_b, rs, ctypes.cast(proofs, ctypes.c_void_p), ctypes.cast(pws, ctypes.c_void_p),
                                 ctypes.cast(status, ctypes.c_void_p))
        assert rc == 0 and not any(status)
        n_in = h.n_inputs
    finally:
        h.close()
    pl = [proofs.raw[388 * i:388 * (i + 1)] for i in range(B_)]
    wl = [pws.raw[76 * i:76 * (i + 1)] for i in range(B_)]
    assert len(set(pl)) == B_
    orc = native.Prover(audit_artifacts["sppc"], audit_artifacts["pk"])
    for i in (0, 77, B_ - 1):
        row = workload.row_ints(rows_b, n_in, i)
        assert row == rlwe.audit_input_vector(rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345 + 40 + i, random.Random(1000 + 40 + i)))
        rc, proof, pw = orc.prove(row, 31 * i + 3, 37 * i + 5)
        assert rc == 0 and pl[i] == proof and wl[i] == pw, i
    assert all(ctx.verify_batch(open(audit_artifacts["vk"], "rb").read(), pl, wl))


def test_audit_batch_refuses_bad_rows_in_place(ctx, audit_artifacts, rlwe_pk):
    """The batch path of the matrix evaluation (run kernel + the integer "small rows" of the 1 088 quotient equations,
    generate_audit.py:539-554) refuses, lane by lane: a quotient off by one, a noise value outside the range-check table, a noise
    value inside the table that breaks its equation, a ciphertext byte changed -- and proves the 124 good rows around them."""
    from spp import workload
    B_ = 128
    rows = bytearray(workload.audit_rows(ctx, rlwe_pk["a"], rlwe_pk["b"], B_, first=900))
    n_in = 3360
    R_ = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    def poke(i, idx, fn):
        off = 32 * (n_in * i + idx)
        v = fn(int.from_bytes(rows[off:off + 32], "big")) % R_
        rows[off:off + 32] = v.to_bytes(32, "big")
    poke(5, 2336 + 7, lambda v: v + 1)          # k1[7] + 1
    poke(70, 160 + 3, lambda v: 300)            # r[3] = 300: not in [-128, 127]
    poke(71, 160 + 3, lambda v: v + 1)          # r[3] + 1: in the table, equation broken
    poke(127, 2 + 10 + 5, lambda v: v ^ 1)      # one bit of a packed ciphertext word
    h = ctx.load_circuit(audit_artifacts["sppc"], audit_artifacts["pk"], 6)
    try:
        import ctypes
        proofs = ctypes.create_string_buffer(388 * B_)
        pws = ctypes.create_string_buffer(76 * B_)
        status = (ctypes.c_int32 * B_)()
        rc = h.L.spp_prove_batch(h.h, B_, bytes(rows), None, ctypes.cast(proofs, ctypes.c_void_p), ctypes.cast(pws, ctypes.c_void_p),
                                 ctypes.cast(status, ctypes.c_void_p))
        assert rc == -4
        bad = {5, 70, 71, 127}
        assert [i for i in range(B_) if status[i] != 0] == sorted(bad)
        pl = [proofs.raw[388 * i:388 * (i + 1)] for i in range(B_) if i not in bad]
        wl = [pws.raw[76 * i:76 * (i + 1)] for i in range(B_) if i not in bad]
    finally:
        h.close()
    assert all(ctx.verify_batch(open(audit_artifacts["vk"], "rb").read(), pl, wl))
    assert all(proofs.raw[388 * i:388 * (i + 1)] == b"\x00" * 388 for i in bad)


def test_batched_verifier_matches_the_single_proof_verifiers(ctx, withdraw_handle, withdraw_artifacts, audit_artifacts, withdraw_kat, rlwe_pk):
    """spp_verify_batch on real proofs of both circuits: accepts what the oracle and the host verifier
    accept, and rejects -- lane by lane -- a flipped byte in each proof element, a wrong public input, a proof checked
    against another statement, a missing commitment count and the all-zero proof of a refused row."""
    import spp
    from oracle import groth16
    rows = _withdraw_variants(withdraw_kat, 5)
    rs = [(900 + i, 1900 + 3 * i) for i in range(5)]
    proofs, pws, status = withdraw_handle.prove_batch(rows, rs)
    assert status == [0] * 5
    vk = open(withdraw_artifacts["vk"], "rb").read()
    cases, expect = [], []
    for pr, pw in zip(proofs, pws):
        cases.append((pr, pw)); expect.append(True)
    def flip(b, i):
        x = bytearray(b); x[i] ^= 1; return bytes(x)
    for off in (0, 40, 70, 130, 200, 259, 270, 330, 387):      # Ar.x, Ar.y, Bs, Bs, Krs, commitment count, Cm, PoK, PoK
        cases.append((flip(proofs[0], off), pws[0])); expect.append(False)
    cases.append((proofs[1], flip(pws[1], 12 + 31))); expect.append(False)     # root
    cases.append((proofs[1], flip(pws[1], 12 + 32 * 3 + 31))); expect.append(False)   # amount
    cases.append((proofs[1], flip(pws[1], 3))); expect.append(False)           # header
    cases.append((proofs[2], pws[3])); expect.append(False)                    # another statement's inputs
    cases.append((bytes(388), pws[0])); expect.append(False)                   # refused row
    # aliased encodings: v + r in a public word (same nullifier, other bytes), coordinate + q in the proof -- refused, not reduced
    from oracle import bn254 as B
    def add_at(b, off, m):
        x = bytearray(b); v = int.from_bytes(x[off:off + 32], "big") + m
        if v >= 1 << 256:
            return None
        x[off:off + 32] = v.to_bytes(32, "big"); return bytes(x)
    n_alias = 0
    for k in range(5):
        a = add_at(pws[4], 12 + 32 * k, B.R)
        if a is not None:
            cases.append((proofs[4], a)); expect.append(False); n_alias += 1
    for off in (0, 32, 64, 192, 260, 324):
        a = add_at(proofs[4], off, B.P)
        if a is not None:
            cases.append((a, pws[4])); expect.append(False); n_alias += 1
    assert n_alias >= 6
    got = ctx.verify_batch(vk, [c[0] for c in cases], [c[1] for c in cases])
    assert got == expect
    for (pr, pw), e in list(zip(cases, expect))[-n_alias:]:
        assert groth16.verify(vk, pr, pw) == e and spp.verify(vk, pr, pw) == e
    for (pr, pw), e in list(zip(cases, expect))[:8]:
        assert groth16.verify(vk, pr, pw) == e and spp.verify(vk, pr, pw) == e
    assert ctx.verify_batch(vk, [], []) == []
    # a batch larger than a wavefront, mixed
    big_p = [proofs[i % 5] if i % 7 else flip(proofs[i % 5], 100) for i in range(150)]
    big_w = [pws[i % 5] for i in range(150)]
    assert ctx.verify_batch(vk, big_p, big_w) == [bool(i % 7) for i in range(150)]
    # audit circuit (2 public inputs, 1104-byte key)
    ha = ctx.load_circuit(audit_artifacts["sppc"], audit_artifacts["pk"], 6)
    try:
        arows = _audit_rows(rlwe_pk, 2)
        ap, aw, ast = ha.prove_batch(arows, [(5, 6), (7, 8)])
        assert ast == [0, 0]
    finally:
        ha.close()
    avk = open(audit_artifacts["vk"], "rb").read()
    assert ctx.verify_batch(avk, ap + [ap[0]], aw + [aw[1]]) == [True, True, False]
    assert groth16.verify(avk, ap[0], aw[0])


def test_large_host_batch_is_chunked_and_pipelined(withdraw_handle, withdraw_kat):
    """spp_prove_batch cuts a host batch larger than 1.5 x 4096 into chunks alternating between the two workspaces: results
    must land at the right offsets (position-dependent blinding), including a refused row in the last, partial chunk."""
    rows4 = _withdraw_variants(withdraw_kat, 4)
    n = 2 * 4096 + 700
    rows = [rows4[i % 4] for i in range(n)]
    rs = [(1 + (i % 5), 2 + (i % 3)) for i in range(n)]
    bad_at = n - 13
    rows[bad_at] = [rows[bad_at][0] + 1] + rows[bad_at][1:]
    proofs, pws, status = withdraw_handle.prove_batch(rows, rs)
    assert status[bad_at] != 0 and sum(1 for v in status if v) == 1
    ref = {}
    small_rows, small_rs, keys = [], [], []
    for i in range(60):                      # every (row, blinding) combination occurs within the first 60 positions
        keys.append((i % 4, rs[i]))
        small_rows.append(rows4[i % 4]); small_rs.append(rs[i])
    sp, sw, sst = withdraw_handle.prove_batch(small_rows, small_rs)
    assert sst == [0] * 60
    for k, pr, pw in zip(keys, sp, sw):
        ref[k] = (pr, pw)
    for i in range(n):
        if i != bad_at:
            assert (proofs[i], pws[i]) == ref[(i % 4, rs[i])], i


def test_load_errors_are_reported(ctx, tmp_path, withdraw_artifacts, audit_artifacts):
    """Error behaviour at the boundary (negative codes + message, mapped to thrown errors by the addon)."""
    import spp
    from spp.lib import SppError
    with pytest.raises(SppError) as e:
        ctx.load_circuit(str(tmp_path / "missing.sppc"), withdraw_artifacts["pk"], 6)
    assert "-3" in str(e.value) or "cannot read" in str(e.value)                    # SPP_ERR_IO
    with pytest.raises(SppError) as e:
        ctx.load_circuit(withdraw_artifacts["sppc"], audit_artifacts["pk"], 6)      # key of another circuit
    assert "does not match" in str(e.value) or "-7" in str(e.value)                 # SPP_ERR_FORMAT
    bad = tmp_path / "trunc.pk"
    bad.write_bytes(open(withdraw_artifacts["pk"], "rb").read()[:1000])
    with pytest.raises(SppError):
        ctx.load_circuit(withdraw_artifacts["sppc"], str(bad), 6)
    with pytest.raises(SppError):
        ctx.load_circuit(withdraw_artifacts["sppc"], withdraw_artifacts["pk"], 3)   # window outside [4,16]


def test_withdraw_witness_matches_oracle(withdraw_handle, withdraw_artifacts, withdraw_kat):
    from oracle import native, circuit as C
    row = C.withdraw_inputs(withdraw_kat)
    withdraw_handle.prove_batch([row], [(5, 6)])
    got = withdraw_handle.debug_witness()
    orc = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    rc, _, _, wires = orc.prove(row, 5, 6, want_wires=True)
    assert rc == 0 and got == wires


def test_unsatisfied_inputs_are_refused(withdraw_handle, withdraw_kat):
    from oracle import circuit as C
    good = C.withdraw_inputs(withdraw_kat)
    wrong_root = list(good); wrong_root[0] += 1
    zero_recipient = list(good); zero_recipient[2] = 0          # main.nr:81
    big_amount = list(good); big_amount[3] = 1 << 64            # amount: pub u64, main.nr:43
    proofs, pws, status = withdraw_handle.prove_batch([good, wrong_root, zero_recipient, big_amount], [(1, 2)] * 4)
    assert status[0] == 0 and status[1] == -4 and status[2] == -4 and status[3] == -4
    assert proofs[1] == b"\x00" * 388


def test_odd_batch_sizes_and_default_window(ctx, withdraw_artifacts, withdraw_kat):
    """P = 1, 3, 65, 70, 1025 (not multiples of the wavefront: above 64 the batch is cut into a 64-aligned body and a tail on the
    other proving stream) and automatically sized per-set window tables (20 GB budget); with SPP_NO_SPLIT-style whole batches
    the bytes are the same (checked for 70 through the device entry point's tail-free twin, the host entry point's chunks)."""
    from oracle import native
    os.environ["SPP_TABLE_BUDGET_GB"] = "20"
    h = ctx.load_circuit(withdraw_artifacts["sppc"], withdraw_artifacts["pk"], 0)
    del os.environ["SPP_TABLE_BUDGET_GB"]
    try:
        assert h.table_bytes <= 20e9 and 8 <= min(h.msm_windows()[:4]) and len(set(h.msm_windows())) > 1
        orc = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
        rows = _withdraw_variants(withdraw_kat, 4)
        for count in (1, 3, 65, 70, 1025):
            batch = [rows[i % len(rows)] for i in range(count)]
            rs = [(i + 1, 3 * i + 2) for i in range(count)]
            proofs, pws, status = h.prove_batch(batch, rs)
            assert status == [0] * count
            for i in sorted({0, count // 2, 63 % count, 64 % count, count - 1}):
                rc, proof, pw = orc.prove(batch[i], rs[i][0], rs[i][1])
                assert proofs[i] == proof and pws[i] == pw, (count, i)
    finally:
        h.close()


def test_reference_shape_withdraw_circuit_matches_oracle(ctx, tmp_path, withdraw_kat):
    """The withdraw statement padded to the reference R1CS's dimensions (12 452 constraints, 2^14): GPU proof bytes ==
    C oracle under the same pk and (r, s), and the pairing check passes."""
    import spp
    from oracle import native, groth16
    sppc, pk, vk = (str(tmp_path / ("wref." + e)) for e in ("sppc", "pk", "vk"))
    assert spp.build_circuit(spp.lib.SPP_CIRCUIT_WITHDRAW_REFSHAPE, sppc) == 12452
    native.setup(sppc, b"\x0b" * 32, pk, vk)
    os.environ["SPP_TABLE_BUDGET_GB"] = "24"
    h = ctx.load_circuit(sppc, pk, 0)
    del os.environ["SPP_TABLE_BUDGET_GB"]
    try:
        assert h.n_constraints == 12452 and h.domain_log == 14
        rows = _withdraw_variants(withdraw_kat, 3)
        rs = [(101 + i, 202 + i) for i in range(3)]
        proofs, pws, status = h.prove_batch(rows, rs)
        assert status == [0, 0, 0]
        orc = native.Prover(sppc, pk)
        for i in range(3):
            rc, proof, pw = orc.prove(rows[i], rs[i][0], rs[i][1])
            assert rc == 0 and proofs[i] == proof and pws[i] == pw
        assert groth16.verify(open(vk, "rb").read(), proofs[0], pws[0])
    finally:
        h.close()


def test_depth20_withdraw_variant_matches_oracle(ctx, tmp_path):
    """The depth-20 variant of the withdraw statement (SURVEY 8d Config 2): GPU proof bytes == C oracle, pairing check."""
    import spp
    from oracle import native, groth16
    from test_host_cpu import _depth20_rows
    sppc, pk, vk = (str(tmp_path / ("w20." + e)) for e in ("sppc", "pk", "vk"))
    spp.build_circuit(spp.lib.SPP_CIRCUIT_WITHDRAW_DEPTH20, sppc)
    native.setup(sppc, b"\x14" * 32, pk, vk)
    os.environ["SPP_TABLE_BUDGET_GB"] = "24"
    h = ctx.load_circuit(sppc, pk, 0)
    del os.environ["SPP_TABLE_BUDGET_GB"]
    try:
        assert h.n_inputs == 30 and h.domain_log == 14
        rows = _depth20_rows(3)
        rs = [(41 + i, 97 + i) for i in range(3)]
        proofs, pws, status = h.prove_batch(rows, rs)
        assert status == [0, 0, 0]
        orc = native.Prover(sppc, pk)
        for i in range(3):
            rc, proof, pw = orc.prove(rows[i], rs[i][0], rs[i][1])
            assert rc == 0 and proofs[i] == proof and pws[i] == pw
        assert groth16.verify(open(vk, "rb").read(), proofs[1], pws[1])
        bad = list(rows[2]); bad[10 + 17] += 1
        assert h.prove_batch([bad], [(1, 2)])[2] != [0]
    finally:
        h.close()


# ---------------------------------------------------------------------------------------------- audit circuit
def _audit_rows(rlwe_pk, count):
    from oracle import rlwe
    rows = []
    for i in range(count):
        # instance 0 = the reference's own run: sk 12345, Random(999) (scripts/generate_audit.py:469-470)
        d = rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345 + i, random.Random(999 + i))
        rows.append(rlwe.audit_input_vector(d))
    return rows


def test_audit_proof_bytes_match_oracle_and_verify(ctx, audit_artifacts, rlwe_pk):
    from oracle import native, groth16
    assert 20000 < audit_artifacts["n_constraints"] < 32768      # README.md:49 quotes ~26K
    h = ctx.load_circuit(audit_artifacts["sppc"], audit_artifacts["pk"], 6)
    try:
        rows = _audit_rows(rlwe_pk, 3)
        rs = [(11 + i, 13 + 5 * i) for i in range(3)]
        proofs, pws, status = h.prove_batch(rows, rs)
        assert status == [0, 0, 0]
        orc = native.Prover(audit_artifacts["sppc"], audit_artifacts["pk"])
        vk = open(audit_artifacts["vk"], "rb").read()
        assert len(vk) == 1104                                   # size of the reference's rlwe_audit.vk
        for i in range(3):
            rc, proof, pw = orc.prove(rows[i], rs[i][0], rs[i][1])
            assert rc == 0 and len(pw) == 76                     # submit_audit.rs:19-21
            assert pws[i] == pw and proofs[i] == proof, "audit proof %d differs from the oracle" % i
        assert groth16.verify(vk, proofs[0], pws[0])
        # noise out of range / tampered ciphertext commitment are refused
        bad_r = list(rows[0]); bad_r[2 + 157 + 1 + 7] = 200
        bad_ct = list(rows[0]); bad_ct[1] += 1
        _, _, st = h.prove_batch([rows[1], bad_r, bad_ct], [(1, 2)] * 3)
        assert st == [0, -4, -4]
    finally:
        h.close()


def test_matrix_evaluation_paths_agree(ctx, audit_artifacts, rlwe_pk, monkeypatch):
    """<A,w>, <B,w>, <C,w> three ways on a 70-proof audit batch (the run kernel's regime): every row in field arithmetic on one lane
    per (run, proof); the 1 088 quotient equations as integer sums (small rows); the 6 720-term lookup row on 16 lanes per proof
    (long rows).  The switches take the paths out one by one; the proofs must not change, and a bad row is refused on every path."""
    from oracle import native
    rows = _audit_rows(rlwe_pk, 3)
    batch = [rows[i % 3] for i in range(70)]
    rs = [(5 + i, 900 + 7 * i) for i in range(70)]
    bad = list(rows[1]); bad[1] += 1
    got = {}
    for name, env in (("all", {}), ("no_small", {"SPP_NO_SMALL_ROWS": "1"}), ("no_long", {"SPP_NO_LONG_ROWS": "1"}),
                      ("neither", {"SPP_NO_SMALL_ROWS": "1", "SPP_NO_LONG_ROWS": "1"})):
        for k in ("SPP_NO_SMALL_ROWS", "SPP_NO_LONG_ROWS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = ctx.load_circuit(audit_artifacts["sppc"], audit_artifacts["pk"], 6)
        try:
            assert (h.small_rows()[0] == 1088) == ("SPP_NO_SMALL_ROWS" not in env)
            proofs, pws, status = h.prove_batch(batch, rs)
            assert status == [0] * 70, name
            got[name] = (proofs, pws)
            _, _, st = h.prove_batch([batch[i] if i != 66 else bad for i in range(70)], rs)
            assert st == [0] * 66 + [-4] + [0] * 3, name
        finally:
            h.close()
    for k in ("SPP_NO_SMALL_ROWS", "SPP_NO_LONG_ROWS"):
        monkeypatch.delenv(k, raising=False)
    assert got["all"] == got["no_small"] == got["no_long"] == got["neither"]
    orc = native.Prover(audit_artifacts["sppc"], audit_artifacts["pk"])
    for i in (0, 64, 69):
        rc, proof, pw = orc.prove(batch[i], *rs[i])
        assert rc == 0 and got["all"][0][i] == proof and got["all"][1][i] == pw, i


def test_compute_h_forms_agree_with_the_oracle(ctx, withdraw_artifacts, audit_artifacts, withdraw_kat, rlwe_pk, monkeypatch):
    """computeH three ways ("computeH in product form"): gnark's seven transforms (SPP_H_MODE=0: h coefficients
    against pk.G1.Z, groth16/bn254/prove.go computeH), six (H bases moved to the evaluation basis on 5*H) and four (the default: h as
    the high half of A*B, the H half of its values folded into the K bases).  All three must emit the bytes of the C oracle, which
    keeps gnark's form: Krs is one group element however it is summed."""
    from oracle import native
    wrows = _withdraw_variants(withdraw_kat, 5)
    arows = _audit_rows(rlwe_pk, 3)
    for art, rows in ((withdraw_artifacts, wrows), (audit_artifacts, arows)):
        rs = [(101 + 7 * i, 33 + i) for i in range(len(rows))]
        orc = native.Prover(art["sppc"], art["pk"])
        want = [orc.prove(rows[i], *rs[i]) for i in range(len(rows))]
        assert all(w[0] == 0 for w in want)
        for mode in ("0", "1", "2"):
            monkeypatch.setenv("SPP_H_MODE", mode)
            h = ctx.load_circuit(art["sppc"], art["pk"], 6)
            try:
                proofs, pws, status = h.prove_batch(rows, rs)
            finally:
                h.close()
            assert status == [0] * len(rows), (mode, status)
            for i in range(len(rows)):
                assert proofs[i] == want[i][1] and pws[i] == want[i][2], (mode, i)
    monkeypatch.delenv("SPP_H_MODE")


# ---------------------------------------------------------------------------------------------- witness-input kernels
def test_rlwe_witness_matches_reference_fixtures(ctx, rlwe_pk, rlwe_vectors):
    """The reference's own values (tests/golden/rlwe_vectors.json, produced by importing scripts/generate_audit.py)."""
    from spp import witness
    r = [v["r"] for v in rlwe_vectors]
    e1 = [v["e1"] for v in rlwe_vectors]
    e2 = [v["e2"] for v in rlwe_vectors]
    msg = [v["msg"] for v in rlwe_vectors]
    out = witness.rlwe_witness(ctx, rlwe_pk["a"], rlwe_pk["b"], r, e1, e2, msg)
    for i, v in enumerate(rlwe_vectors):
        assert out["c0"][i].tolist() == v["c0"] and out["c1"][i].tolist() == v["c1"], v["name"]
        assert out["k0"][i].tolist() == v["k0"] and out["k1"][i].tolist() == v["k1"], v["name"]
        assert [hex(x) for x in out["c0_packed"][i]] == v["c0_packed"]
        assert [hex(x) for x in out["c1_packed"][i]] == v["c1_packed"]


def test_rlwe_witness_random_batch_and_edges(ctx, rlwe_pk):
    import numpy as np
    from spp import witness
    from oracle import rlwe
    rng = np.random.default_rng(4)
    count = 70
    r = rng.integers(-3, 4, size=(count, 1024), dtype=np.int8)
    e1 = rng.integers(-3, 4, size=(count, 64), dtype=np.int8)
    e2 = rng.integers(-3, 4, size=(count, 1024), dtype=np.int8)
    msg = rng.integers(0, 256, size=(count, 64), dtype=np.uint8)
    r[0] = 127; e2[0] = 127; e1[0] = 127; msg[0] = 255          # largest positive values the circuit's range proof admits
    r[1] = -128; e2[1] = -128; e1[1] = -128; msg[1] = 0          # most negative: quotients go far below zero
    r[2] = 0; e1[2] = 0; e2[2] = 0
    out = witness.rlwe_witness(ctx, rlwe_pk["a"], rlwe_pk["b"], r, e1, e2, msg)
    for i in (0, 1, 2, 3, 37, 69):
        c0, c1, k0, k1 = rlwe.rlwe_witness(rlwe_pk["a"], rlwe_pk["b"], r[i].tolist(), e1[i].tolist(), e2[i].tolist(), msg[i].tolist())
        assert out["c0"][i].tolist() == c0 and out["c1"][i].tolist() == c1
        assert out["k0"][i].tolist() == k0 and out["k1"][i].tolist() == k1
        assert out["c0_packed"][i] == rlwe.pack_values(c0) and out["c1_packed"][i] == rlwe.pack_values(c1)


def test_rlwe_witness_degenerate_public_keys(ctx, rlwe_pk):
    """Public keys the NTT kernel treats specially: zero coefficients (the reference's matrix rows hold 0, not q, in the wrapped
    positions: the zero-list correction of rlwe_ntt.hpp), the all-zero key, the identity key a = [1, 0, ...], the largest
    coefficients q - 1, and alternating extreme r -- all against the oracle's schoolbook restatement (generate_audit.py:45-66,
    236-243), plus linearity of c1 in r modulo q at a larger batch."""
    import numpy as np
    from spp import witness
    from oracle import rlwe
    q = rlwe.RLWE_Q
    rng = np.random.default_rng(9)
    a0 = np.array(rlwe_pk["a"], dtype=np.int64)
    b0 = np.array(rlwe_pk["b"], dtype=np.int64)
    keys = []
    a = a0.copy(); a[[0, 1, 500, 1023]] = 0; b = b0.copy(); b[[3, 63, 64, 700]] = 0
    keys.append((a, b))                                                       # a few zeros in both polynomials
    keys.append((np.zeros(1024, dtype=np.int64), np.zeros(1024, dtype=np.int64)))
    ident = np.zeros(1024, dtype=np.int64); ident[0] = 1
    keys.append((ident, np.full(1024, q - 1, dtype=np.int64)))
    half = a0.copy(); half[::2] = 0
    keys.append((half, b0))                                                   # 512 zeros
    count = 6
    r = rng.integers(-3, 4, size=(count, 1024), dtype=np.int8)
    e1 = rng.integers(-3, 4, size=(count, 64), dtype=np.int8)
    e2 = rng.integers(-3, 4, size=(count, 1024), dtype=np.int8)
    msg = rng.integers(0, 256, size=(count, 64), dtype=np.uint8)
    r[0, ::2] = 127; r[0, 1::2] = -128
    r[1] = -128; e2[1] = 127; msg[1] = 255; e1[1] = 127
    for a, b in keys:
        out = witness.rlwe_witness(ctx, a.tolist(), b.tolist(), r, e1, e2, msg)
        for i in range(count):
            c0, c1, k0, k1 = rlwe.rlwe_witness(a.tolist(), b.tolist(), r[i].tolist(), e1[i].tolist(), e2[i].tolist(), msg[i].tolist())
            assert out["c0"][i].tolist() == c0 and out["c1"][i].tolist() == c1, i
            assert out["k0"][i].tolist() == k0 and out["k1"][i].tolist() == k1, i
    # size-independent property at a larger batch: c1(r + r') = c1(r) + c1(r') - e2-terms, modulo q (zero noise)
    n = 3000
    r1 = rng.integers(-60, 61, size=(n, 1024), dtype=np.int8)
    r2 = rng.integers(-60, 61, size=(n, 1024), dtype=np.int8)
    z1 = np.zeros((n, 64), dtype=np.int8); z2 = np.zeros((n, 1024), dtype=np.int8); zm = np.zeros((n, 64), dtype=np.uint8)
    o1 = witness.rlwe_witness(ctx, rlwe_pk["a"], rlwe_pk["b"], r1, z1, z2, zm)
    o2 = witness.rlwe_witness(ctx, rlwe_pk["a"], rlwe_pk["b"], r2, z1, z2, zm)
    o3 = witness.rlwe_witness(ctx, rlwe_pk["a"], rlwe_pk["b"], (r1 + r2).astype(np.int8), z1, z2, zm)
    assert ((o1["c1"].astype(np.int64) + o2["c1"]) % q == o3["c1"]).all() and ((o1["c0"].astype(np.int64) + o2["c0"]) % q == o3["c0"]).all()
    # the integer identity behind the quotient: sum over the three runs of (k q + c) is additive too
    s1 = o1["k1"].astype(np.int64) * q + o1["c1"]; s2 = o2["k1"].astype(np.int64) * q + o2["c1"]; s3 = o3["k1"].astype(np.int64) * q + o3["c1"]
    assert (s1 + s2 == s3).all()


def test_poseidon_merkle_grumpkin_kernels(ctx, withdraw_kat):
    from spp import witness
    from oracle import hashes as H
    from oracle.bn254 import R
    kat = withdraw_kat
    f = lambda k: int(kat[k], 16)
    # the reference's golden vector (client/prover-params.toml)
    assert witness.identity_public_keys(ctx, [f("secret_key")]) == [(f("owner_x"), f("owner_y"))]
    assert witness.poseidon_hash2(ctx, f("owner_x"), f("owner_y")) == f("wa_commitment")
    assert witness.poseidon_hash2(ctx, f("secret_key"), kat["index"]) == f("nullifier")
    cm = witness.poseidon_hash4(ctx, f("owner_x"), f("owner_y"), kat["amount"], f("randomness"))
    sib = [int(s, 16) for s in kat["siblings"]]
    assert witness.merkle_roots(ctx, [cm], [kat["index"]], [sib]) == [f("root")]
    t = witness.ShieldedPoolMerkleTree(ctx)
    assert t.getRoot() == H.default_hashes()[16]                 # empty tree
    t.insert(cm)
    assert t.getRoot() == f("root") and t.getProof(0) == sib     # single leaf: siblings = zero-hash chain
    # random batches against the oracle, incl. zero and p-1 operands
    rng = random.Random(12)
    rows2 = [[rng.randrange(R), rng.randrange(R)] for _ in range(130)] + [[0, 0], [R - 1, R - 1]]
    assert witness.poseidon_hash_batch(ctx, rows2) == [H.poseidon_hash2(*r) for r in rows2]
    rows4 = [[rng.randrange(R) for _ in range(4)] for _ in range(67)]
    assert witness.poseidon_hash_batch(ctx, rows4) == [H.poseidon_hash4(*r) for r in rows4]
    sks = [12345, 1, (1 << 128) - 1, R - 1] + [rng.randrange(R) for _ in range(20)]
    assert witness.identity_public_keys(ctx, sks) == [H.fixed_base_scalar_mul(s) for s in sks]
    # a 37-leaf tree: roots and proofs for several indices, ragged right edge included
    ot = H.MerkleTree()
    gt = witness.ShieldedPoolMerkleTree(ctx)
    for _ in range(37):
        v = rng.randrange(R)
        ot.insert(v)
        gt.insert(v)
    assert gt.getRoot() == ot.root()
    for idx in (0, 17, 35, 36):
        assert gt.getProof(idx) == ot.proof(idx)
    paths = [(ot.leaves[i], i, ot.proof(i)) for i in (3, 36)]
    assert witness.merkle_roots(ctx, [p[0] for p in paths], [p[1] for p in paths], [p[2] for p in paths]) == [ot.root()] * 2


def test_incremental_merkle_tree_interleaved_inserts_and_queries(ctx):
    """the device-resident incremental tree (spp_merkle_tree_*) against oracle/hashes.MerkleTree (which restates
    client/merkle.ts:146-222) on an interleaved sequence of single inserts, batch inserts, root reads and proofs -- including
    proofs for indices that are not inserted yet, capacity growth past the initial 1024 leaves, the one-shot builder
    (spp_merkle_build) as a second reference, a depth-20 tree and the refusal of non-canonical leaves."""
    import spp
    from spp import witness
    from oracle import hashes as H
    from oracle.bn254 import R
    rng = random.Random(44)
    ot = H.MerkleTree()
    gt = witness.ShieldedPoolMerkleTree(ctx)
    try:
        assert gt.getRoot() == ot.root() == H.default_hashes()[16] and len(gt) == 0
        assert gt.getProof(5) == ot.proof(5)                              # empty tree: all defaults
        steps = [1, 1, 1, 2, 5, 1, 64, 1, 100, 333, 1, 700, 1, 1]          # 1211 leaves in the end: crosses 1024
        for k, n in enumerate(steps):
            vals = [rng.randrange(R) for _ in range(n)]
            first = gt.insert_many(vals) if n > 1 else gt.insert(vals[0])
            assert first == len(ot.leaves)
            for v in vals:
                ot.insert(v)
            assert len(gt) == len(ot.leaves) and gt.getRoot() == ot.root(), k
            size = len(ot.leaves)
            qs = sorted({0, size - 1, size // 2, min(size, (1 << 16) - 1), rng.randrange(size), rng.randrange(1 << 16)})
            assert gt.getProofs(qs) == [ot.proof(q) for q in qs], k
        # the one-shot builder agrees with the incremental tree on the same leaves
        root2, sib2 = witness.merkle_build(ctx, ot.leaves, [0, 1000, 1210])
        assert root2 == gt.getRoot() and sib2 == gt.getProofs([0, 1000, 1210])
        # every proof verifies through compute_merkle_root on the GPU
        qs = [rng.randrange(len(ot.leaves)) for _ in range(70)]
        assert witness.merkle_roots(ctx, [ot.leaves[q] for q in qs], qs, gt.getProofs(qs)) == [ot.root()] * 70
        with pytest.raises(spp.SppError):
            gt.insert(R)                                                  # not a canonical field element
        with pytest.raises(spp.SppError):
            gt.getProof(1 << 16)
        assert len(gt) == len(ot.leaves)
    finally:
        gt.close()
    o20 = H.MerkleTree(20)
    g20 = witness.ShieldedPoolMerkleTree(ctx, 20)
    try:
        vals = [rng.randrange(R) for _ in range(9)]
        g20.insert_many(vals)
        for v in vals:
            o20.insert(v)
        assert g20.getRoot() == o20.root() and g20.getProof(8) == o20.proof(8) and len(g20.getProof(8)) == 20
    finally:
        g20.close()


def test_poseidon2_sponge_kernel(ctx, rlwe_pk):
    from spp import witness
    from oracle import hashes as H, rlwe
    from oracle.bn254 import R
    rng = random.Random(3)
    d = rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345, random.Random(999))
    rows = [d["c0_packed"] + d["c1_packed"], [rng.randrange(R) for _ in range(157)]]
    assert witness.ct_commitments(ctx, rows) == [H.poseidon2_sponge(r) for r in rows]
    for n in (1, 2, 3, 4):                                       # ragged absorb lengths
        row = [rng.randrange(R) for _ in range(n)]
        assert witness.ct_commitments(ctx, [row]) == [H.poseidon2_sponge(row)]


# ---------------------------------------------------------------------------------------------- general Pippenger
def test_pippenger_matches_oracle_and_is_linear(ctx):
    from oracle import bn254 as B, native
    import ctypes
    rng = random.Random(21)
    pts = []
    p = B.G1_GEN
    for i in range(300):
        p = B.g1_add(p, B.g1_mul(B.G1_GEN, rng.randrange(1, 1 << 64)))
        pts.append(p)
    bases = b"".join(B.g1_to_bytes(q) for q in pts)
    for n in (0, 1, 2, 300):
        sc = [rng.randrange(B.R) for _ in range(n)]
        if n == 300:
            sc[:8] = [0, 1, B.R - 1, (B.R - 1) // 2, (B.R + 1) // 2, 0x8000, 0x8001, 0xffff]     # window-boundary digits
            sc[8:40] = [5] * 32                                                                 # one crowded bucket
        got = ctx.msm_g1_pippenger(bases[:64 * n], sc)
        out = ctypes.create_string_buffer(64)
        native.lib().orc_msm_g1(bases[:64 * n], b"".join(s.to_bytes(32, "big") for s in sc), n, ctypes.cast(out, ctypes.c_void_p))
        assert got == out.raw, n
        assert got == ctx.msm_g1(bases[:64 * n], sc, 6)                  # table path agrees with the bucket path
    # size-independent property at a size the oracle cannot check: MSM(k * s) == k * MSM(s)  (2^18 synthetic points)
    r1, ms, _ = ctx.msm_g1_pippenger_bench(1 << 18, seed=5)
    k = 0x1234567
    r2, _, _ = ctx.msm_g1_pippenger_bench(1 << 18, seed=5, scale=k)
    assert B.g1_to_bytes(B.g1_mul(B.g1_from_bytes(r1), k)) == r2
    # and against the oracle on a 2^12 prefix-sized instance of the same generator
    r3, _, _ = ctx.msm_g1_pippenger_bench(1 << 12, seed=9)
    x = (9 * 6364136223846793005 + 1442695040888963407) % (1 << 64)
    def nxt():
        nonlocal x
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return x
    ks, ss = [], []
    for _ in range(1 << 12):
        ks.append(nxt() | 1)
        w = []
        for _ in range(4):
            v = nxt(); w += [v & 0xffffffff, v >> 32]
        w[7] &= 0x1fffffff
        ss.append(sum(l << (32 * i) for i, l in enumerate(w)))
    # the generator writes raw limbs into Montgomery storage: the field element is limbs * 2^-256
    rinv = pow(1 << 256, -1, B.R)
    ss = [s * rinv % B.R for s in ss]
    out = ctypes.create_string_buffer(64)
    basesb = b"".join(B.g1_to_bytes(B.g1_mul(B.G1_GEN, kk)) for kk in ks)
    native.lib().orc_msm_g1(basesb, b"".join(s.to_bytes(32, "big") for s in ss), 1 << 12, ctypes.cast(out, ctypes.c_void_p))
    assert r3 == out.raw


def test_pippenger_sort_forms_agree(ctx, monkeypatch):
    """The two-level bucket sort (kernels_pippenger.hip step 3) carries the 7 fine key bits inside the entry while n <= 2^24 and in a
    byte array beyond; SPP_PIP_UNPACKED=1 runs the second form on small inputs.  Same sums, oracle-checked at 300 points."""
    from oracle import bn254 as B, native
    import ctypes
    rng = random.Random(22)
    pts, p = [], B.G1_GEN
    for i in range(300):
        p = B.g1_add(p, B.g1_mul(B.G1_GEN, rng.randrange(1, 1 << 64)))
        pts.append(p)
    bases = b"".join(B.g1_to_bytes(q) for q in pts)
    sc = [rng.randrange(B.R) for _ in range(300)]
    sc[:6] = [0, 1, B.R - 1, 0x8000, 0xffff, 1 << 252]
    sc[100:164] = [77] * 64
    out = ctypes.create_string_buffer(64)
    native.lib().orc_msm_g1(bases, b"".join(s.to_bytes(32, "big") for s in sc), 300, ctypes.cast(out, ctypes.c_void_p))
    packed = ctx.msm_g1_pippenger(bases, sc)
    big_packed, _, _ = ctx.msm_g1_pippenger_bench(1 << 17, seed=3)
    skew_packed, _, _ = ctx.msm_g1_pippenger_bench(1 << 17, seed=3, small_permille=900)
    monkeypatch.setenv("SPP_PIP_UNPACKED", "1")
    assert ctx.msm_g1_pippenger(bases, sc) == packed == out.raw
    assert ctx.msm_g1_pippenger_bench(1 << 17, seed=3)[0] == big_packed
    assert ctx.msm_g1_pippenger_bench(1 << 17, seed=3, small_permille=900)[0] == skew_packed


def test_pippenger_shards_add_up_to_the_whole_msm(ctx):
    """spp_msm_g1_pippenger_bench_shard: the partial sums of three uneven contiguous shares of the synthetic 2^16-point MSM add up to
    the MSM of all points (what spp/multi.py msm_g1_sharded gathers over RCCL on a multi-GPU node), and a share of everything is the
    unsharded entry point's result."""
    n = 1 << 16
    whole, _, _ = ctx.msm_g1_pippenger_bench(n, seed=9)
    assert ctx.msm_g1_pippenger_bench_shard(n, 0, n, seed=9)[0] == whole
    cuts = [0, 20000, 20001, n]
    parts = [ctx.msm_g1_pippenger_bench_shard(n, cuts[i], cuts[i + 1] - cuts[i], seed=9)[0] for i in range(3)]
    assert ctx.msm_g1(b"".join(parts), [1, 1, 1]) == whole


def test_pippenger_skewed_scalars(ctx):
    """Witness-like distributions (SURVEY 8d Config 5): most scalars byte-sized, so a few buckets of window 0 hold
    thousands of points and span many 256-entry segments of the sorted list (k_pip_segments / k_pip_fixup)."""
    from oracle import bn254 as B, native
    import ctypes
    rng = random.Random(33)
    pts = []
    p = B.G1_GEN
    for i in range(200):
        p = B.g1_add(p, B.g1_mul(B.G1_GEN, rng.randrange(1, 1 << 64)))
        pts.append(p)
    n = 8000                 # 5 300 + 300 entries in one bucket: 22 segments, summed by a whole wave in k_pip_fixup (> PIP_FIX_SEQ)
    idx = [rng.randrange(200) for _ in range(n)]
    bases = b"".join(B.g1_to_bytes(pts[i]) for i in idx)
    sc = [7] * 5300 + [B.R - 7] * 300 + [rng.randrange(256) for _ in range(1500)] + [rng.randrange(B.R) for _ in range(900)]
    rng.shuffle(sc)
    out = ctypes.create_string_buffer(64)
    native.lib().orc_msm_g1(bases, b"".join(s.to_bytes(32, "big") for s in sc), n, ctypes.cast(out, ctypes.c_void_p))
    assert ctx.msm_g1_pippenger(bases, sc) == out.raw
    # exactly one bucket, exactly filling segments: 512 copies of the same digit
    sc2 = [3] * 512
    native.lib().orc_msm_g1(bases[:64 * 512], b"".join(s.to_bytes(32, "big") for s in sc2), 512, ctypes.cast(out, ctypes.c_void_p))
    assert ctx.msm_g1_pippenger(bases[:64 * 512], sc2) == out.raw
    # the synthetic generator with 70 % byte-sized scalars: oracle on 2^12, linearity on 2^18
    r3, _, _ = ctx.msm_g1_pippenger_bench(1 << 12, seed=9, small_permille=700)
    x = (9 * 6364136223846793005 + 1442695040888963407) % (1 << 64)
    def nxt():
        nonlocal x
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return x
    rinv = pow(1 << 256, -1, B.R)
    ks, ss = [], []
    for _ in range(1 << 12):
        ks.append(nxt() | 1)
        w = []
        for _ in range(4):
            v = nxt(); w += [v & 0xffffffff, v >> 32]
        w[7] &= 0x1fffffff
        if (nxt() >> 20) % 1000 < 700:
            ss.append(w[0] & 0xff)
        else:
            ss.append(sum(l << (32 * i) for i, l in enumerate(w)) * rinv % B.R)
    assert sum(1 for v in ss if v < 256) > 2500
    basesb = b"".join(B.g1_to_bytes(B.g1_mul(B.G1_GEN, kk)) for kk in ks)
    native.lib().orc_msm_g1(basesb, b"".join(s.to_bytes(32, "big") for s in ss), 1 << 12, ctypes.cast(out, ctypes.c_void_p))
    assert r3 == out.raw
    r1, _, _ = ctx.msm_g1_pippenger_bench(1 << 18, seed=5, small_permille=700)
    k = 0x7654321
    r2, _, _ = ctx.msm_g1_pippenger_bench(1 << 18, seed=5, scale=k, small_permille=700)
    assert B.g1_to_bytes(B.g1_mul(B.g1_from_bytes(r1), k)) == r2


def test_cli_setup_prove_verify_on_gpu(tmp_path, withdraw_kat):
    """The prove_linux.sh pipeline (compile -> setup -> prove -> verify) through the spp CLI."""
    from spp import cli
    from spp.proof_helper import ShieldedPoolInputs, prover_toml
    sppc = str(tmp_path / "shielded_pool_verifier.sppc")
    assert cli.main(["compile", "withdraw", "-o", sppc]) == 0
    assert cli.main(["setup", sppc, "--seed", "11" * 32]) == 0
    fields = ("root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness", "index", "siblings")
    toml = tmp_path / "Prover.toml"
    toml.write_text(prover_toml(ShieldedPoolInputs(**{f: withdraw_kat[f] for f in fields})))
    base = str(tmp_path / "shielded_pool_verifier")
    assert cli.main(["prove", sppc, base + ".pk", str(toml), "--window", "6"]) == 0
    assert os.path.getsize(base + ".proof") == 388 and os.path.getsize(base + ".pw") == 172
    assert cli.main(["verify", base + ".vk", base + ".proof", base + ".pw"]) == 0
    bad = tmp_path / "Bad.toml"
    bad.write_text(toml.read_text().replace('recipient = "0x0000', 'recipient = "0x0001', 1).replace(withdraw_kat["root"], withdraw_kat["nullifier"], 1))
    assert cli.main(["prove", sppc, base + ".pk", str(bad), "--window", "6"]) == 1
    # skip-if-exists (prove_linux.sh:72-79), keyed by the circuit's hash: same circuit + seed -> keys untouched; another
    # seed or --force -> redone; a different circuit file under the same name -> redone
    import json
    pk_before = open(base + ".pk", "rb").read()
    mtime = os.path.getmtime(base + ".pk")
    assert cli.main(["setup", sppc, "--seed", "11" * 32]) == 0 and os.path.getmtime(base + ".pk") == mtime
    assert cli.main(["setup", sppc]) == 0 and os.path.getmtime(base + ".pk") == mtime          # no seed given: existing keys are fine
    assert json.load(open(base + ".setup.json"))["pk_bytes"] == len(pk_before)
    assert cli.main(["setup", sppc, "--seed", "12" * 32]) == 0 and open(base + ".pk", "rb").read() != pk_before
    assert cli.main(["setup", sppc, "--seed", "11" * 32]) == 0 and open(base + ".pk", "rb").read() == pk_before   # deterministic
    assert cli.main(["compile", "withdraw", "-o", sppc]) == 0
    open(sppc, "ab").close()
    meta = json.load(open(base + ".setup.json")); meta["circuit_sha256"] = "0" * 64
    json.dump(meta, open(base + ".setup.json", "w"))
    t0 = os.path.getmtime(base + ".pk")
    import time
    time.sleep(0.05)
    assert cli.main(["setup", sppc, "--seed", "11" * 32]) == 0 and os.path.getmtime(base + ".pk") > t0          # hash mismatch -> redone


def test_cli_audit_pipeline_on_gpu(tmp_path, rlwe_pk, withdraw_kat, capsys):
    """audit_circuit/prove_audit.sh:74-111 through the spp CLI on the GPU: compile -> setup -> prove from the Prover.toml that
    scripts/generate_audit.py:630-641 writes (the reference's own run: sk = 12345, Random(999)) -> verify -> hex dump; proof bytes
    equal the oracle's under the same key and blinding; a Prover.toml with one quotient off by one is refused (exit 1)."""
    import json
    import random
    from spp import cli, generate_proof_hex
    from spp.proof_helper import audit_prover_toml
    from oracle import rlwe, native
    root = tmp_path
    adir, wdir = root / "audit_circuit" / "target", root / "noir_circuit" / "target"
    adir.mkdir(parents=True); wdir.mkdir(parents=True)
    pkj = tmp_path / "rlwe_pk.json"
    pkj.write_text(json.dumps({"a": ["0x%08x" % v for v in rlwe_pk["a"]], "b": ["0x%08x" % v for v in rlwe_pk["b"]]}))
    sppc = str(adir / "rlwe_audit.sppc")
    assert cli.main(["compile", "audit", "--rlwe-pk", str(pkj), "-o", sppc]) == 0
    assert "nbConstraints=" in capsys.readouterr().out                     # what benchmark_all.py:646,664 parses
    assert cli.main(["setup", sppc, "--seed", "21" * 32]) == 0
    d = rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345, random.Random(999))
    d["e1_sparse"] = d["e1"]                                               # the Prover.toml key (generate_audit.py:637)
    toml = root / "audit_circuit" / "Prover.toml"
    toml.write_text(audit_prover_toml(d))
    base = str(adir / "rlwe_audit")
    assert cli.main(["prove", sppc, base + ".pk", str(toml), "--window", "8", "--rs", "77", "99"]) == 0
    assert os.path.getsize(base + ".proof") == 388 and os.path.getsize(base + ".pw") == 76   # submit_audit.rs:18-21
    assert cli.main(["verify", base + ".vk", base + ".proof", base + ".pw"]) == 0
    rc, proof, pw = native.Prover(sppc, base + ".pk").prove(rlwe.audit_input_vector(d), 77, 99)
    assert rc == 0 and open(base + ".proof", "rb").read() == proof and open(base + ".pw", "rb").read() == pw
    bad = dict(d)
    bad["k1"] = list(d["k1"]); bad["k1"][5] += 1
    (root / "audit_circuit" / "Bad.toml").write_text(audit_prover_toml(bad))
    assert cli.main(["prove", sppc, base + ".pk", str(root / "audit_circuit" / "Bad.toml"), "--window", "8"]) == 1
    # the hex dump of client/generate-proof-hex.ts needs the withdraw pair as well
    (wdir / "shielded_pool_verifier.proof").write_bytes(b"\x01" * 388)
    (wdir / "shielded_pool_verifier.pw").write_bytes(b"\x02" * 172)
    code, out, err = generate_proof_hex.render(str(root))
    assert code == 0 and "3. AUDIT PROOF (hex):" in out and "0x" + proof.hex() in out and "0x" + pw.hex() in out


def test_audit_inputs_pipeline_on_gpu(ctx, rlwe_pk):
    """(sk, r, e1, e2) -> full audit input rows on the device == the oracle's restatement of generate_audit.py:468-641."""
    from spp import witness
    from oracle import rlwe
    sks, rs, e1s, e2s, exp = [], [], [], [], []
    for i in range(5):
        d = rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345 + 7 * i, random.Random(999 + i))   # i = 0: the reference's own run
        sks.append(d["secret_key"]); rs.append(d["r"]); e1s.append(d["e1"]); e2s.append(d["e2"])
        exp.append(rlwe.audit_input_vector(d))
    got = witness.audit_input_rows(ctx, rlwe_pk["a"], rlwe_pk["b"], sks, rs, e1s, e2s)
    assert got == exp


def test_audit_proofs_from_secrets_on_the_device(ctx, audit_artifacts, rlwe_pk):
    """spp_prove_audit_from_secrets_device: (secret_key, r, e1, e2) -> proof without leaving the device == the oracle's proof of the
    oracle's restatement of scripts/generate_audit.py:468-641 for the same secrets; 70 instances, two calls in flight (each with its own
    scratch and transformed public key) produce the same bytes."""
    import numpy as np
    import torch
    from spp import workload
    from oracle import native, rlwe
    B_ = 70
    dev = torch.device("cuda", 0)
    sks, r8, e18, e28 = workload.audit_noise(500, B_)
    up = lambda raw: torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    d_a, d_b = up(np.asarray(rlwe_pk["a"], dtype=np.uint32).tobytes()), up(np.asarray(rlwe_pk["b"], dtype=np.uint32).tobytes())
    d_sk = up(b"".join(int(v).to_bytes(32, "big") for v in sks))
    d_r, d_e1, d_e2 = up(r8.tobytes()), up(e18.tobytes()), up(e28.tobytes())
    rs_vals = [(101 * i + 7, 103 * i + 9) for i in range(B_)]
    d_rs = up(b"".join(r.to_bytes(32, "big") + s.to_bytes(32, "big") for r, s in rs_vals))
    h = ctx.load_circuit(audit_artifacts["sppc"], audit_artifacts["pk"], 6)
    try:
        outs = [(torch.zeros(388 * B_, dtype=torch.uint8, device=dev), torch.zeros(76 * B_, dtype=torch.uint8, device=dev),
                 torch.ones(B_, dtype=torch.int32, device=dev)) for _ in range(2)]
        for pr, pw, st in outs:
            h.prove_audit_from_secrets_device(B_, d_a.data_ptr(), d_b.data_ptr(), d_sk.data_ptr(), d_r.data_ptr(), d_e1.data_ptr(), d_e2.data_ptr(),
                                              d_rs.data_ptr(), pr.data_ptr(), pw.data_ptr(), st.data_ptr())
        h.sync()
        assert all(int(o[2].abs().sum().item()) == 0 for o in outs)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        pb, wb = bytes(outs[0][0].cpu().numpy()), bytes(outs[0][1].cpu().numpy())
    finally:
        h.close()
    orc = native.Prover(audit_artifacts["sppc"], audit_artifacts["pk"])
    for i in (0, 33, 64, B_ - 1):
        d = rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345 + 500 + i, random.Random(1000 + 500 + i))
        rc, proof, pw = orc.prove(rlwe.audit_input_vector(d), *rs_vals[i])
        assert rc == 0 and pb[388 * i:388 * (i + 1)] == proof and wb[76 * i:76 * (i + 1)] == pw, i


def test_host_mirror_generate_proof_and_audit_proof(tmp_path, withdraw_artifacts, audit_artifacts, withdraw_kat, rlwe_pk):
    """spp.generateProof / generateAuditProof: same call shape and side effects as client/proof.helper.ts:28-72."""
    import shutil
    import spp
    from spp import proof_helper
    from oracle import rlwe
    os.environ["SPP_TABLE_BUDGET_GB"] = "12"
    try:
        wdir = tmp_path / "noir_circuit"; os.makedirs(wdir / "target")
        shutil.copy(withdraw_artifacts["sppc"], wdir / "target" / "shielded_pool_verifier.sppc")
        shutil.copy(withdraw_artifacts["pk"], wdir / "target" / "shielded_pool_verifier.pk")
        fields = ("root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness", "index", "siblings")
        inputs = spp.ShieldedPoolInputs(**{f: withdraw_kat[f] for f in fields})
        out = spp.generateProof(spp.CircuitConfig(str(wdir), "shielded_pool_verifier"), inputs)
        assert set(out) == {"proof", "publicWitness"} and len(out["proof"]) == 388 and len(out["publicWitness"]) == 172
        assert (wdir / "Prover.toml").read_text() == proof_helper.prover_toml(inputs)
        assert (wdir / "target" / "shielded_pool_verifier.proof").read_bytes() == out["proof"]
        assert spp.verify(open(withdraw_artifacts["vk"], "rb").read(), out["proof"], out["publicWitness"])
        bad = spp.ShieldedPoolInputs(**{**inputs.__dict__, "recipient": "0x0"})
        with pytest.raises(spp.SppError):
            spp.generateProof(spp.CircuitConfig(str(wdir), "shielded_pool_verifier"), bad)
        adir = tmp_path / "audit_circuit"; os.makedirs(adir / "target")
        shutil.copy(audit_artifacts["sppc"], adir / "target" / "rlwe_audit.sppc")
        shutil.copy(audit_artifacts["pk"], adir / "target" / "rlwe_audit.pk")
        d = rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345, random.Random(999))
        ain = dict(secret_key=d["secret_key"], wa_commitment=d["wa_commitment"], ct_commitment=d["ct_commitment"], c0_packed=d["c0_packed"],
                   c1_packed=d["c1_packed"], r=d["r"], e1_sparse=d["e1"], e2=d["e2"], k0=d["k0"], k1=d["k1"])
        out = spp.generateAuditProof(spp.CircuitConfig(str(adir), "rlwe_audit"), ain)
        assert len(out["proof"]) == 388 and len(out["publicWitness"]) == 76
        assert spp.verify(open(audit_artifacts["vk"], "rb").read(), out["proof"], out["publicWitness"])
        toml = (adir / "Prover.toml").read_text()
        assert toml.startswith('secret_key = "0x%064x"\nwa_commitment = ' % 12345) and "\nk1 = [" in toml      # generate_audit.py:630-641
    finally:
        del os.environ["SPP_TABLE_BUDGET_GB"]


def test_extreme_inputs_and_empty_batch(withdraw_handle, withdraw_artifacts):
    """Maximum sizes the circuit admits: amount 2^64-1, leaf index 65535 (all path bits set), secret key r-1 (254 bits,
    top window of the Grumpkin ladder), siblings r-1; plus the empty batch."""
    from oracle import native, groth16, hashes as H
    from oracle.bn254 import R
    sk = R - 1
    owner = H.fixed_base_scalar_mul(sk)
    amount = (1 << 64) - 1
    rnd = R - 2
    idx = (1 << 16) - 1
    sib = [R - 1 - i for i in range(16)]
    cm = H.poseidon_hash4(owner[0], owner[1], amount, rnd)
    row = [H.compute_merkle_root(cm, idx, sib), H.poseidon_hash2(sk, idx), R - 1, amount, H.poseidon_hash2(owner[0], owner[1]),
           sk, owner[0], owner[1], rnd, idx] + sib
    proofs, pws, status = withdraw_handle.prove_batch([row], [(R - 1, R - 2)])     # extreme blinding too
    assert status == [0]
    orc = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    rc, proof, pw = orc.prove(row, R - 1, R - 2)
    assert rc == 0 and proofs[0] == proof and pws[0] == pw
    assert groth16.verify(open(withdraw_artifacts["vk"], "rb").read(), proofs[0], pws[0])
    # blinding 0 (degenerate r = s = 0) still yields a valid proof
    proofs0, pws0, st0 = withdraw_handle.prove_batch([row], [(0, 0)])
    assert st0 == [0] and proofs0[0] == orc.prove(row, 0, 0)[1]
    # empty batch: nothing to do, no error
    assert withdraw_handle.prove_batch([], []) == ([], [], [])
    # OS randomness: two proofs of the same statement differ and both verify
    pa, wa_, sa = withdraw_handle.prove_batch([row, row], None)
    assert sa == [0, 0] and pa[0] != pa[1]
    import spp
    vk = open(withdraw_artifacts["vk"], "rb").read()
    assert spp.verify(vk, pa[0], wa_[0]) and spp.verify(vk, pa[1], wa_[1])


def test_auditor_side_reconstruct_and_decrypt(ctx, rlwe_pk, rlwe_vectors):
    """scripts/rlwe_decrypt.py end to end on the GPU against the reference-derived fixture (tests/golden/rlwe_decrypt.json):
    Shamir shares 1+2 -> sk mod q, every fixture ciphertext decrypts to its message; plus random ciphertexts vs the oracle."""
    import json
    import numpy as np
    from conftest import GOLDEN
    from spp import witness
    from oracle import rlwe, hashes as H
    d = json.load(open(os.path.join(GOLDEN, "rlwe_decrypt.json")))
    sk = witness.reconstruct_sk(ctx, d["shares"])
    assert sk == d["sk_mod_q"]
    owners, msg = witness.rlwe_decrypt(ctx, sk, [v["c0"] for v in rlwe_vectors], [v["c1"] for v in rlwe_vectors])
    for i, v in enumerate(rlwe_vectors):
        assert msg[i].tolist() == v["msg"] == d["decrypt"][i]["msg"], v["name"]
    assert owners[0] == H.fixed_base_scalar_mul(12345)          # RLWE-1 encrypts the key of sk = 12345 (generate_audit.py:470)
    rng = np.random.default_rng(11)
    c0 = rng.integers(0, rlwe.RLWE_Q, size=(33, 64), dtype=np.uint32)
    c1 = rng.integers(0, rlwe.RLWE_Q, size=(33, 1024), dtype=np.uint32)
    skr = rng.integers(0, rlwe.RLWE_Q, size=1024, dtype=np.uint32)    # arbitrary (not small) key: exercises the mod-q folding
    _, got = witness.rlwe_decrypt(ctx, skr, c0, c1)
    for i in (0, 7, 32):
        assert got[i].tolist() == rlwe.rlwe_decrypt(skr.tolist(), c0[i].tolist(), c1[i].tolist())


def test_small_batch_paths_agree_with_the_batch_paths(ctx, withdraw_artifacts, audit_artifacts, rlwe_pk, monkeypatch):
    """Batches up to 1024 proofs take the one-wave-per-proof solver (lane-parallel Poseidon / Poseidon2 / Grumpkin, SOLVE_C rows
    by dependency level), small batches (144 withdraw / 31 audit proofs) also replace s*Ar and r*Bs1 by table sums over the scaled witness, small launches split
    the windows of a base over lanes and evaluate matrix rows 16 lanes at a time.  None of that may change a byte: the same rows
    and (full-size) blinding factors give the same proofs at batch sizes 1, 3, 17 (cooperative), through a handle loaded with
    SPP_NO_COOP=1 (always one lane per proof, lanes multiply the blinding), and inside a batch of 1100 (past the threshold)."""
    from spp import workload
    import ctypes
    rng = random.Random(2024)
    from oracle.bn254 import R

    def prove(h, rows_b, n, rs):
        proofs = ctypes.create_string_buffer(388 * n)
        pws = ctypes.create_string_buffer(h.pw_len * n)
        status = (ctypes.c_int32 * n)()
        rc = h.L.spp_prove_batch(h.h, n, rows_b, rs, ctypes.cast(proofs, ctypes.c_void_p), ctypes.cast(pws, ctypes.c_void_p),
                                 ctypes.cast(status, ctypes.c_void_p))
        assert rc == 0 and not any(status)
        return [proofs.raw[388 * i:388 * (i + 1)] for i in range(n)], [pws.raw[h.pw_len * i:h.pw_len * (i + 1)] for i in range(n)]

    for name, art, big in (("withdraw", withdraw_artifacts, 1100), ("audit", audit_artifacts, 1100)):
        h = ctx.load_circuit(art["sppc"], art["pk"], 6)
        monkeypatch.setenv("SPP_NO_COOP", "1")
        h_lane = ctx.load_circuit(art["sppc"], art["pk"], 6)
        monkeypatch.delenv("SPP_NO_COOP")
        try:
            rows_b = workload.withdraw_rows(ctx, big, seed=11) if name == "withdraw" else \
                workload.audit_rows(ctx, rlwe_pk["a"], rlwe_pk["b"], big, first=7)
            row_len = h.n_inputs * 32
            rs = b"".join(rng.randrange(R).to_bytes(32, "big") + rng.randrange(R).to_bytes(32, "big") for _ in range(big))
            ref_p, ref_w = prove(h_lane, rows_b[:40 * row_len], 40, rs[:40 * 64])
            for n in (1, 3, 17, 40):                    # 40 audit proofs: cooperative solver, 16-lane rows, per-lane blinding
                p, w = prove(h, rows_b[:n * row_len], n, rs[:n * 64])
                assert p == ref_p[:n] and w == ref_w[:n], (name, n)
            p, w = prove(h, rows_b, big, rs)            # > 1024: one lane per proof, blinding multiplied by the lanes
            assert p[:40] == ref_p and w[:40] == ref_w, name
            # 1024 distinct rows, the largest cooperative batch, proof by proof against the batch result
            pc, wc = prove(h, rows_b[:1024 * row_len], 1024, rs[:1024 * 64])
            assert pc == p[:1024] and wc == w[:1024], name
            tail_p, tail_w = prove(h, rows_b[(big - 2) * row_len:], 2, rs[(big - 2) * 64:])
            assert p[-2:] == tail_p and w[-2:] == tail_w, name
            assert all(ctx.verify_batch(open(art["vk"], "rb").read(), p[:64] + p[-64:], w[:64] + w[-64:]))
        finally:
            h.close()
            h_lane.close()

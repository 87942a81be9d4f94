"""CPU: the oracle against every golden vector the reference holds for this path."""
import json
import os
import random

from conftest import GOLDEN


def test_prover_params_kat(withdraw_kat):
    """client/prover-params.toml: Grumpkin key, wa_commitment, nullifier, H4 commitment, zero-hash chain, root."""
    from oracle import hashes as H
    k = withdraw_kat
    f = lambda n: int(k[n], 16)
    pv = H.withdraw_public_values(f("secret_key"), k["amount"], f("randomness"), k["index"], [int(s, 16) for s in k["siblings"]])
    for name in ("owner_x", "owner_y", "wa_commitment", "nullifier", "root"):
        assert pv[name] == f(name), name
    assert H.default_hashes()[:16] == [int(s, 16) for s in k["siblings"]]
    assert H.identity_keypair(f("secret_key"))[1] == (f("owner_x"), f("owner_y"))


def test_noir_unit_test_vector():
    """noir_circuit/src/main.nr:84-130: sk 12345, amount 1000000, randomness 67890, index 0, ZERO siblings.  The hash-level
    part (the values main() recomputes and asserts equal, :60-78); the same vector goes through the product's R1CS in
    tests/test_circuit_soundness.py::test_noir_unit_test_vector_through_the_circuit."""
    from oracle import hashes as H
    owner = H.fixed_base_scalar_mul(12345)
    assert H.identity_keypair(12345)[1] == owner
    cm = H.poseidon_hash4(owner[0], owner[1], 1000000, 67890)
    # compute_merkle_root with index 0: sixteen left-child steps H2(current, 0) (main.nr:11-29)
    cur = cm
    for _ in range(16):
        cur = H.poseidon_hash2(cur, 0)
    assert H.compute_merkle_root(cm, 0, [0] * 16) == cur
    # index bit set -> the sibling goes left (main.nr:19-23)
    nxt = H.poseidon_hash2(7, cm)
    for _ in range(15):
        nxt = H.poseidon_hash2(nxt, 0)
    assert H.compute_merkle_root(cm, 1, [7] + [0] * 15) == nxt
    pv = H.withdraw_public_values(12345, 1000000, 67890, 0, [0] * 16)
    assert pv["root"] == cur and pv["nullifier"] == H.poseidon_hash2(12345, 0) and pv["wa_commitment"] == H.poseidon_hash2(*owner)


def test_poseidon2_kat_and_constants():
    """Literature KAT perm([0,1,2,3]) and the internal diagonal quoted in SURVEY App. B."""
    from oracle import hashes as H
    rc, mu = H.poseidon2_params()
    assert mu == H.P2_MU_EXPECTED
    assert rc[0] == 0x19b849f69450b06848da1d39bd5e4a4302bb86744edc26238b0878e269ed23e5
    assert H.poseidon2_permute([0, 1, 2, 3]) == [
        0x01bd538c2ee014ed5141b29e9ae240bf8db3fe5b9a38629a9647cf8d76c01737,
        0x239b62e7db98aa3a2a8f6a0d2fa1709e7a35959aa6c7034814d9daa90cbac662,
        0x04cbb44c61d928ed06808456bf758cbf0c18d1e15a7b6dbc8245fa7515d5e3cb,
        0x2e11c5cff2a22c64d01304b778d78f6998eff1ab73163a35603f54794c30847a]


def test_rlwe_vectors(rlwe_pk, rlwe_vectors):
    """Vectors produced by importing the reference's scripts/generate_audit.py (tests/golden/make_fixtures.py)."""
    from oracle import rlwe
    assert rlwe_vectors[0]["c1"][:3] == [78874407, 130923686, 120074060]      # SURVEY 8c RLWE-1
    for v in rlwe_vectors:
        c0, c1, k0, k1 = rlwe.rlwe_witness(rlwe_pk["a"], rlwe_pk["b"], v["r"], v["e1"], v["e2"], v["msg"])
        assert (c0, c1, k0, k1) == (v["c0"], v["c1"], v["k0"], v["k1"]), v["name"]
        assert [hex(x) for x in rlwe.pack_values(c0)] == v["c0_packed"]
        assert [hex(x) for x in rlwe.pack_values(c1)] == v["c1_packed"]
        assert [rlwe.format_field(x) for x in k0[:8]] == v["k0_fmt"]
        q = rlwe.RLWE_Q
        assert rlwe.negacyclic_mul_mod_q(rlwe_pk["a"], [x % q for x in v["r"]]).tolist() == v["ar"]
        assert rlwe.negacyclic_matrix_row_mod_q(rlwe_pk["a"], 1000)[:16].tolist() == v["row_a_1000"]
        assert rlwe.negacyclic_matrix_row_mod_q(rlwe_pk["b"], 5)[:16].tolist() == v["row_b_5"]


def test_pack_and_format_kat():
    from oracle import rlwe
    k = json.load(open(os.path.join(GOLDEN, "pack_kat.json")))
    assert [hex(v) for v in rlwe.pack_values(k["in"])] == k["out"]
    assert rlwe.encode_field_to_bytes(int(k["bytes_in"], 16), 8) == k["bytes_out"]
    for v, s in k["fmt"]:
        assert rlwe.format_field(v) == s


def test_c_oracle_matches_python_oracle(rlwe_pk, rlwe_vectors):
    """The C restatement (liboracle.so) against the Python one: hashes, RLWE, NTT round trip, MSM."""
    import ctypes
    import numpy as np
    from oracle import native, hashes as H, bn254 as B
    L = native.lib()
    rng = random.Random(8)
    for arity in (2, 4):
        vals = [rng.randrange(B.R) for _ in range(arity)]
        out = ctypes.create_string_buffer(32)
        L.orc_poseidon_hash(b"".join(v.to_bytes(32, "big") for v in vals), arity, out)
        assert int.from_bytes(out.raw, "big") == H.poseidon_hash(vals)
    out = ctypes.create_string_buffer(128)
    L.orc_poseidon2_permute(b"".join(v.to_bytes(32, "big") for v in (0, 1, 2, 3)), out)
    assert [int.from_bytes(out.raw[32 * i:32 * i + 32], "big") for i in range(4)] == H.poseidon2_permute([0, 1, 2, 3])
    v = rlwe_vectors[0]
    a = np.array(rlwe_pk["a"], dtype=np.uint32); b = np.array(rlwe_pk["b"], dtype=np.uint32)
    r = np.array(v["r"], dtype=np.int32); e1 = np.array(v["e1"], dtype=np.int32); e2 = np.array(v["e2"], dtype=np.int32)
    m = np.array(v["msg"], dtype=np.uint32)
    c0 = np.zeros(64, dtype=np.uint32); c1 = np.zeros(1024, dtype=np.uint32)
    k0 = np.zeros(64, dtype=np.int64); k1 = np.zeros(1024, dtype=np.int64)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    L.orc_rlwe_witness(p(a), p(b), p(r), p(e1), p(e2), p(m), p(c0), p(c1), p(k0), p(k1))
    assert c0.tolist() == v["c0"] and c1.tolist() == v["c1"] and k0.tolist() == v["k0"] and k1.tolist() == v["k1"]
    pts = [B.g1_mul(B.G1_GEN, rng.randrange(1, B.R)) for _ in range(9)]
    sc = [rng.randrange(B.R) for _ in range(9)]
    out = ctypes.create_string_buffer(64)
    L.orc_msm_g1(b"".join(B.g1_to_bytes(q) for q in pts), b"".join(s.to_bytes(32, "big") for s in sc), 9, ctypes.cast(out, ctypes.c_void_p))
    assert out.raw == B.g1_to_bytes(B.g1_msm(pts, sc))


def test_pairing_and_hash_to_field_kats():
    from oracle import bn254 as B
    a, b = 1234567, 7654321
    assert B.pairing_product_is_one([(B.g1_mul(B.G1_GEN, a), B.g2_mul(B.G2_GEN, b)), (B.g1_neg(B.g1_mul(B.G1_GEN, a * b)), B.G2_GEN)])
    assert not B.pairing_product_is_one([(B.g1_mul(B.G1_GEN, a), B.g2_mul(B.G2_GEN, b)), (B.g1_neg(B.g1_mul(B.G1_GEN, a * b + 1)), B.G2_GEN)])
    # RFC 9380 K.1 expand_message_xmd(SHA-256), msg = "", len 0x20
    assert B.expand_message_xmd(b"", b"QUUX-V01-CS02-with-expander-SHA256-128", 0x20).hex() == \
        "68a985b87eb6b46952128911f2a4412bbc302a9d759667f87f7a21d803f07235"


def test_auditor_side_oracle(rlwe_vectors):
    """Oracle restatement of scripts/rlwe_decrypt.py against the fixture generated by importing it."""
    from oracle import rlwe
    d = json.load(open(os.path.join(GOLDEN, "rlwe_decrypt.json")))
    sk = rlwe.reconstruct_sk([s["x"] for s in d["shares"]], [[int(y, 16) for y in s["y"]] for s in d["shares"]])
    assert sk == d["sk_mod_q"] and all(v in (0, 1, 2, 3) or v >= rlwe.RLWE_Q - 3 for v in sk)     # noise-bounded key
    for v, e in zip(rlwe_vectors, d["decrypt"]):
        m = rlwe.rlwe_decrypt(sk, v["c0"], v["c1"])
        assert m == e["msg"] == v["msg"]
        assert rlwe.decode_owner(m) == (sum(b << (8 * i) for i, b in enumerate(v["msg"][:32])), sum(b << (8 * i) for i, b in enumerate(v["msg"][32:])))
    # rounding: exact ties go to the even neighbour, as Python's round()
    assert [round(x / 2) for x in (1, 3, 5, -1, -3)] == [0, 2, 2, 0, -2]


def test_commitment_mask_has_a_domain_of_its_own():
    """the hiding mask of the commitment (OP_MASK, fr.Hash of the blinding factors r || s) and the commitment challenge
    (fr.Hash(Cm.x || Cm.y), gnark's tag "bsb22-commitment") are different protocol values: they are hashed under different
    domain-separation tags, so the mask is not the challenge of the same 64 bytes; the C restatement derives the same mask."""
    from oracle import circuit as C, bn254
    r, s = 0x1234567890abcdef, 0xfedcba0987654321
    msg = r.to_bytes(32, "big") + s.to_bytes(32, "big")
    assert bn254.DST_MASK != bn254.DST_COMMITMENT and len(bn254.DST_MASK) == len(bn254.DST_COMMITMENT) == 16
    assert C.mask_value(r, s) == bn254.hash_to_fr(msg, bn254.DST_MASK)[0]
    assert C.mask_value(r, s) != bn254.hash_to_fr(msg, bn254.DST_COMMITMENT)[0]

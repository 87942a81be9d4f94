"""ORACLE (test infrastructure only) -- Groth16 + BSB22 verifier (pairing check) and format parsers.

Restates what `sunspot verify vk proof pw` (noir_circuit/prove_linux.sh:86-87, audit_circuit/
prove_audit.sh:98-99, scripts/generate_audit.py:687-691) checks -- gnark 0.14 groth16/bn254 Verify
(third-party, absent from the reference): commitment challenge = fr.Hash(commitment, "bsb22-commitment"),
Pedersen proof of knowledge, then e(Ar,Bs) = e(alpha,beta) e(kSum,gamma) e(Krs,delta).
Formats: proof 388 B (withdraw.rs:13), public witness 12+32n B (withdraw.rs:14-16), vk.
"""
from . import bn254 as B


def parse_vk(data):
    off = 0

    def g1():
        nonlocal off
        p = B.g1_from_bytes(data[off:off + 64])
        off += 64
        return p

    def g2():
        nonlocal off
        p = B.g2_from_bytes(data[off:off + 128])
        off += 128
        return p

    def u32():
        nonlocal off
        v = int.from_bytes(data[off:off + 4], "big")
        off += 4
        return v
    vk = dict(alpha1=g1(), beta1=g1(), beta2=g2(), gamma2=g2(), delta1=g1(), delta2=g2())
    nk = u32()
    vk["K"] = [g1() for _ in range(nk)]
    nlists = u32()
    vk["committed_public"] = []
    for _ in range(nlists):
        ln = u32()
        vk["committed_public"].append([u32() for _ in range(ln)])
    nkeys = u32()
    assert nkeys == 1
    vk["ped_G"] = g2()
    vk["ped_GSigmaNeg"] = g2()
    assert off == len(data), (off, len(data))
    return vk


def parse_proof(data):
    assert len(data) == 388
    pr = dict(Ar=B.g1_from_bytes(data[0:64]), Bs=B.g2_from_bytes(data[64:192]), Krs=B.g1_from_bytes(data[192:256]))
    assert int.from_bytes(data[256:260], "big") == 1
    pr["commitment"] = B.g1_from_bytes(data[260:324])
    pr["pok"] = B.g1_from_bytes(data[324:388])
    return pr


def parse_public_witness(data):
    npub = int.from_bytes(data[0:4], "big")
    nsec = int.from_bytes(data[4:8], "big")
    nvec = int.from_bytes(data[8:12], "big")
    assert nsec == 0 and nvec == npub and len(data) == 12 + 32 * npub
    return [int.from_bytes(data[12 + 32 * i:44 + 32 * i], "big") for i in range(npub)]


def public_witness_bytes(values):
    n = len(values)
    return n.to_bytes(4, "big") + (0).to_bytes(4, "big") + n.to_bytes(4, "big") + b"".join(
        int(v % B.R).to_bytes(32, "big") for v in values)


def _g2_times_r(pt):
    """[r]pt without reducing the scalar (B.g2_mul reduces mod r)."""
    acc, add, k = None, pt, B.R
    while k:
        if k & 1:
            acc = B.g2_add(acc, add)
        add = B.g2_add(add, add)
        k >>= 1
    return acc


def verify(vk_bytes, proof_bytes, pw_bytes):
    vk = parse_vk(vk_bytes)
    try:
        pr = parse_proof(proof_bytes)
    except Exception:
        return False
    pub = parse_public_witness(pw_bytes)
    if len(pub) + 2 != len(vk["K"]):
        return False
    # canonical encodings only: gnark's readers refuse a public word >= r or a coordinate >= q instead of reducing it
    if any(v >= B.R for v in pub):
        return False
    for off in list(range(0, 256, 32)) + list(range(260, 388, 32)):
        if int.from_bytes(proof_bytes[off:off + 32], "big") >= B.P:
            return False
    for p in (pr["Ar"], pr["Krs"], pr["commitment"], pr["pok"]):
        if not B.g1_is_on_curve(p):
            return False
    if not B.g2_is_on_curve(pr["Bs"]):
        return False
    if _g2_times_r(pr["Bs"]) is not None:      # order-r subgroup (the twist has a large cofactor)
        return False
    # Pedersen proof of knowledge of the commitment
    # gnark-crypto pedersen.VerifyingKey.Verify: e(commitment, GSigmaNeg) * e(pok, G) == 1
    if not B.pairing_product_is_one([(pr["commitment"], vk["ped_GSigmaNeg"]), (pr["pok"], vk["ped_G"])]):
        return False
    challenge = B.hash_to_fr(B.g1_to_bytes(pr["commitment"]), B.DST_COMMITMENT, 1)[0]
    ksum = vk["K"][0]
    for v, k in zip(pub + [challenge], vk["K"][1:]):
        ksum = B.g1_add(ksum, B.g1_mul(k, v))
    ksum = B.g1_add(ksum, pr["commitment"])
    return B.pairing_product_is_one([
        (pr["Ar"], pr["Bs"]),
        (B.g1_neg(vk["alpha1"]), vk["beta2"]),
        (B.g1_neg(ksum), vk["gamma2"]),
        (B.g1_neg(pr["Krs"]), vk["delta2"]),
    ])

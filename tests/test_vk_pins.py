"""The only gnark-made curve / pairing data the reference holds are its two verifying keys
(noir_circuit/target/shielded_pool_verifier.vk, audit_circuit/target/rlwe_audit.vk; layout SURVEY App. A.3), committed
as data under tests/golden/reference_{withdraw,audit}.vk.  They pin: the raw point encodings (G1 X|Y, G2 X.A1|X.A0|Y.A1|Y.A0),
the twist and its order-r subgroup, and the pairing itself (a Groth16 key satisfies e(beta1, G2) = e(G1, beta2) and
e(delta1, G2) = e(G1, delta2)) -- in the oracle (oracle/bn254.py), in the product's host pairing (csrc/pairing.hpp, through
spp_pairing_check_host) and in the device pairing code of the batched verifier (csrc/pairing_fast.hpp on the GPU, through
spp_pairing_check).  No proof made by gnark exists in the reference, so the Groth16 equation itself stays unpinned."""
import os
import pytest
from conftest import GOLDEN

KEYS = [("reference_withdraw.vk", 1296, 7), ("reference_audit.vk", 1104, 4)]


def _load(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


def _pairs(vk_bytes):
    """(G1, G2) byte pairs whose pairing product must be one, and one that must not."""
    from oracle import bn254 as B
    g1, g2 = B.g1_to_bytes(B.G1_GEN), B.g2_to_bytes(B.G2_GEN)
    neg_g1 = B.g1_to_bytes(B.g1_neg(B.G1_GEN))
    alpha1, beta1, beta2, gamma2, delta1, delta2 = (vk_bytes[0:64], vk_bytes[64:128], vk_bytes[128:256], vk_bytes[256:384],
                                                  vk_bytes[384:448], vk_bytes[448:576])
    good = [[(beta1, g2), (neg_g1, beta2)], [(delta1, g2), (neg_g1, delta2)]]
    bad = [[(alpha1, g2), (neg_g1, beta2)], [(delta1, g2), (neg_g1, gamma2)]]
    return good, bad


@pytest.mark.parametrize("name,size,nk", KEYS)
def test_oracle_decodes_reference_vk(name, size, nk):
    from oracle import bn254 as B, groth16
    data = _load(name)
    assert len(data) == size
    vk = groth16.parse_vk(data)                      # asserts the trailer: u32 1 | u32 0 | u32 1 | G | GSigmaNeg, no bytes left
    assert len(vk["K"]) == nk and vk["committed_public"] == [[]]
    for p in [vk["alpha1"], vk["beta1"], vk["delta1"]] + vk["K"]:
        assert p is not None and B.g1_is_on_curve(p)
    for q in (vk["beta2"], vk["gamma2"], vk["delta2"], vk["ped_G"], vk["ped_GSigmaNeg"]):
        assert q is not None and B.g2_is_on_curve(q)
        assert groth16._g2_times_r(q) is None        # order-r subgroup of the twist
    # every coordinate is a canonical field element
    for off in range(0, 576, 32):
        assert int.from_bytes(data[off:off + 32], "big") < B.P
    good, bad = _pairs(data)
    dec = lambda prs: [(B.g1_from_bytes(p), B.g2_from_bytes(q)) for p, q in prs]
    for prs in good:
        assert B.pairing_product_is_one(dec(prs))
    assert not B.pairing_product_is_one(dec(bad[0]))
    # re-encoding reproduces the file byte for byte
    assert B.g1_to_bytes(vk["alpha1"]) == data[0:64] and B.g2_to_bytes(vk["beta2"]) == data[128:256]


@pytest.mark.parametrize("name,size,nk", KEYS)
def test_product_host_pairing_on_reference_vk(name, size, nk):
    import spp
    data = _load(name)
    good, bad = _pairs(data)
    for prs in good:
        assert spp.pairing_check_host(prs)
    for prs in bad:
        assert not spp.pairing_check_host(prs)
    # the Pedersen key of the commitment: both G2 points are valid subgroup points for the host code as well
    from oracle import bn254 as B
    g1 = B.g1_to_bytes(B.G1_GEN)
    ped_off = 580 + 64 * nk + 12
    assert not spp.pairing_check_host([(g1, data[ped_off:ped_off + 128])])            # e(G1, G) != 1, but the point was accepted
    swapped = bytearray(data[128:256]); swapped[0:32], swapped[32:64] = swapped[32:64], swapped[0:32]   # X.A0 | X.A1: not on the twist
    assert not spp.pairing_check_host([(data[64:128], bytes(swapped)), (g1, data[128:256])])


def test_audit_vk_twin_is_identical():
    """audit_circuit.vk is byte-identical to rlwe_audit.vk, so one fixture covers both."""
    ref = "/root/reference/audit_circuit/target/audit_circuit.vk"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present on this machine")
    assert open(ref, "rb").read() == _load("reference_audit.vk")


@pytest.mark.gpu
@pytest.mark.parametrize("name,size,nk", KEYS)
def test_device_pairing_on_reference_vk(name, size, nk):
    import spp
    ctx = spp.Context(0)
    try:
        data = _load(name)
        good, bad = _pairs(data)
        for prs in good:
            assert ctx.pairing_check(prs)
            assert ctx.pairing_check(prs[::-1])          # each point once through the projective-line path, once through a line table
        for prs in bad:
            assert not ctx.pairing_check(prs)
        from oracle import bn254 as B
        g1 = B.g1_to_bytes(B.G1_GEN)
        ped_off = 580 + 64 * nk + 12
        for k in (0, 1):                                 # Pedersen G and GSigmaNeg: accepted as subgroup points, pairing not one
            q = data[ped_off + 128 * k:ped_off + 128 * (k + 1)]
            assert not ctx.pairing_check([(g1, q)])
            neg = B.g1_to_bytes(B.g1_neg(B.G1_GEN))
            assert ctx.pairing_check([(g1, q), (neg, q)])
    finally:
        ctx.close()

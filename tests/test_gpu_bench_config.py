"""The configuration bench.py measures, byte-compared with the oracle (VERDICT r2 "do this" item 2).

Every other parity test loads small tables (window_bits = 6 or a budget of a few GB).  bench.py loads with window_bits = 0 and
the default 240 GB budget: single-row tables of 15-16-bit windows, byte offsets beyond 2^37, one pass per window and the Horner
combine (kernels_msm.hip).  These tests run exactly that configuration on exactly bench.py's rows (spp/workload.py) and compare
sampled proofs byte for byte with the C oracle (oracle/c/groth16.c: Jacobian Pippenger on 4x64-bit limbs, nothing shared with the
device code); every proof of the batch goes through the batched pairing verifier.
Reference behaviour: `sunspot prove` at scripts/generate_audit.py:680 / client/proof.helper.ts:64 and the byte layouts the on-chain
program accepts (shielded_pool_program/src/instructions/submit_audit.rs:18-21, withdraw.rs:13-16)."""
import random

import pytest
import torch  # noqa: F401  (before libspp: both must share ONE HIP runtime, torch's is the one that has to be loaded first)

pytestmark = pytest.mark.gpu
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


@pytest.fixture(scope="module")
def ctx():
    import spp
    c = spp.Context(0)
    yield c
    c.close()


def _prove_resident(h, rows_b, rs, B):
    """spp_prove_batch_device on buffers resident in HBM: the call bench.py times."""
    dev = torch.device("cuda", 0)
    inp = torch.frombuffer(bytearray(rows_b), dtype=torch.uint8).to(dev)
    rst = torch.frombuffer(bytearray(rs), dtype=torch.uint8).to(dev)
    pr = torch.zeros(388 * B, dtype=torch.uint8, device=dev)
    pw = torch.zeros(h.pw_len * B, dtype=torch.uint8, device=dev)
    st = torch.ones(B, dtype=torch.int32, device=dev)
    for _ in range(2):   # twice: both pipelined workspaces produce the same bytes
        h.prove_batch_device(B, inp.data_ptr(), rst.data_ptr(), pr.data_ptr(), pw.data_ptr(), st.data_ptr())
    h.sync()
    assert int(st.abs().sum().item()) == 0
    prb, pwb = bytes(pr.cpu().numpy()), bytes(pw.cpu().numpy())
    return [prb[388 * i:388 * (i + 1)] for i in range(B)], [pwb[h.pw_len * i:h.pw_len * (i + 1)] for i in range(B)]


def _blinding(B, seed):
    rng = random.Random(seed)
    vals = [(rng.randrange(1, R), rng.randrange(1, R)) for _ in range(B)]   # full-size r, s as bench.py draws them
    return vals, b"".join(r.to_bytes(32, "big") + s.to_bytes(32, "big") for r, s in vals)


def test_audit_bench_configuration_matches_the_oracle(ctx, audit_artifacts, rlwe_pk, monkeypatch):
    from spp import workload
    from oracle import native
    monkeypatch.delenv("SPP_TABLE_BUDGET_GB", raising=False)
    B = 2048
    rows_b = workload.audit_rows(ctx, rlwe_pk["a"], rlwe_pk["b"], B)
    h = ctx.load_circuit(audit_artifacts["sppc"], audit_artifacts["pk"], 0)
    try:
        bits, trows = h.msm_windows(), h.msm_table_rows()
        assert h.table_bytes > 150e9, "the default budget was not used"
        assert min(bits[:4] + [bits[6]]) >= 14 and trows[:4] == [1, 1, 1, 1] and trows[6] == 1, (bits, trows)
        rs_vals, rs = _blinding(B, 2048)
        pl, wl = _prove_resident(h, rows_b, rs, B)
        n_in = h.n_inputs
    finally:
        h.close()
    assert len(set(pl)) == B
    orc = native.Prover(audit_artifacts["sppc"], audit_artifacts["pk"])
    rng = random.Random(7)
    for i in [0, 1, 63, 64, 1023, 1024, B - 1] + [rng.randrange(B) for _ in range(3)]:
        rc, proof, pw = orc.prove(workload.row_ints(rows_b, n_in, i), *rs_vals[i])
        assert rc == 0 and pl[i] == proof and wl[i] == pw, i
    assert all(ctx.verify_batch(open(audit_artifacts["vk"], "rb").read(), pl, wl))


def test_withdraw_bench_configuration_matches_the_oracle(ctx, withdraw_artifacts, monkeypatch):
    from spp import workload
    from oracle import native
    monkeypatch.delenv("SPP_TABLE_BUDGET_GB", raising=False)
    B = 4096
    rows_b = workload.withdraw_rows(ctx, B)
    h = ctx.load_circuit(withdraw_artifacts["sppc"], withdraw_artifacts["pk"], 0)
    try:
        bits, trows = h.msm_windows(), h.msm_table_rows()
        assert min(bits[:4] + [bits[6]]) >= 15 and trows[:4] == [1, 1, 1, 1] and trows[6] == 1, (bits, trows)
        rs_vals, rs = _blinding(B, 4096)
        pl, wl = _prove_resident(h, rows_b, rs, B)
        n_in = h.n_inputs
    finally:
        h.close()
    assert len(set(pl)) == B
    orc = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    rng = random.Random(8)
    for i in [0, 1, 63, 64, 2047, 2048, B - 1] + [rng.randrange(B) for _ in range(5)]:
        rc, proof, pw = orc.prove(workload.row_ints(rows_b, n_in, i), *rs_vals[i])
        assert rc == 0 and pl[i] == proof and wl[i] == pw, i
    assert all(ctx.verify_batch(open(withdraw_artifacts["vk"], "rb").read(), pl, wl))


def test_audit_and_withdraw_coresident_match_the_oracle(ctx, audit_artifacts, withdraw_artifacts, rlwe_pk):
    """One GPU serving both circuits (the relayer's pair: demo-frontend/app/api/relay/withdraw/route.ts:238-276): the windows of
    BOTH circuits' MSM sets are planned under one 240 GB budget (spp_plan_windows: a greedy split over the union of the sets),
    both handles are loaded and prove alternating batches while co-resident; sampled proofs of each == the oracle's bytes, every
    proof passes the batched verifier."""
    from spp import workload
    from oracle import native
    plans = ctx.plan_windows([audit_artifacts["pk"], withdraw_artifacts["pk"]], 236e9)
    ha = ctx.load_circuit(audit_artifacts["sppc"], audit_artifacts["pk"], bits=plans[0])
    hw = None
    try:
        hw = ctx.load_circuit(withdraw_artifacts["sppc"], withdraw_artifacts["pk"], bits=plans[1])
        assert ha.msm_windows() == plans[0] and hw.msm_windows() == plans[1]
        assert ha.table_bytes + hw.table_bytes <= 236e9 and min(plans[0][:4] + plans[1][:4]) >= 13, (plans, ha.table_bytes, hw.table_bytes)
        Ba, Bw = 512, 1024
        rows_a = workload.audit_rows(ctx, rlwe_pk["a"], rlwe_pk["b"], Ba, first=7000)
        rows_w = workload.withdraw_rows(ctx, Bw, seed=77)
        rs_a, rsb_a = _blinding(Ba, 11)
        rs_w, rsb_w = _blinding(Bw, 12)
        dev = torch.device("cuda", 0)
        up = lambda raw: torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        bufs = {}
        for name, h, rows, rsb, B in (("a", ha, rows_a, rsb_a, Ba), ("w", hw, rows_w, rsb_w, Bw)):
            bufs[name] = (up(rows), up(rsb), torch.zeros(388 * B, dtype=torch.uint8, device=dev), torch.zeros(h.pw_len * B, dtype=torch.uint8, device=dev),
                          torch.ones(B, dtype=torch.int32, device=dev))
        for _ in range(2):   # alternating batches, all in flight together
            for name, h, B in (("a", ha, Ba), ("w", hw, Bw)):
                i_, r_, p_, w_, s_ = bufs[name]
                h.prove_batch_device(B, i_.data_ptr(), r_.data_ptr(), p_.data_ptr(), w_.data_ptr(), s_.data_ptr())
        ha.sync(); hw.sync()
        res = {}
        for name, h, B in (("a", ha, Ba), ("w", hw, Bw)):
            _, _, p_, w_, s_ = bufs[name]
            assert int(s_.abs().sum().item()) == 0
            pb, wb = bytes(p_.cpu().numpy()), bytes(w_.cpu().numpy())
            res[name] = ([pb[388 * i:388 * (i + 1)] for i in range(B)], [wb[h.pw_len * i:h.pw_len * (i + 1)] for i in range(B)])
        na, nw = ha.n_inputs, hw.n_inputs
    finally:
        ha.close()
        if hw is not None:
            hw.close()
    orc_a = native.Prover(audit_artifacts["sppc"], audit_artifacts["pk"])
    orc_w = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    for i in (0, 255, Ba - 1):
        rc, proof, pw = orc_a.prove(workload.row_ints(rows_a, na, i), *rs_a[i])
        assert rc == 0 and res["a"][0][i] == proof and res["a"][1][i] == pw, ("audit", i)
    for i in (0, 64, 513, Bw - 1):
        rc, proof, pw = orc_w.prove(workload.row_ints(rows_w, nw, i), *rs_w[i])
        assert rc == 0 and res["w"][0][i] == proof and res["w"][1][i] == pw, ("withdraw", i)
    assert all(ctx.verify_batch(open(audit_artifacts["vk"], "rb").read(), *res["a"]))
    assert all(ctx.verify_batch(open(withdraw_artifacts["vk"], "rb").read(), *res["w"]))

"""the reference's own compiled artefacts of the withdraw circuit, read by the product's host code
(spp/acir.py, spp/ccs.py) and pinned to the numbers SURVEY App. A.4 / A.5 record for them.

  tests/golden/reference_withdraw_acir.json   bytecode + abi of noir_circuit/target/shielded_pool_verifier.json
  tests/golden/reference_withdraw.ccs         noir_circuit/target/shielded_pool_verifier.ccs (gnark 0.14 R1CS container)
Both are data files the reference holds (copied by tests/golden/make_fixtures.py; no source text).

The strongest pin here: the reference's OWN constraint system (6 148 AssertZero opcodes, the Grumpkin black box, 25 range checks)
is executed on client/prover-params.toml and on fresh notes built by the oracle -- it accepts exactly what the repository's
R1CS accepts, so the statement proved here is the statement the reference's circuit states."""
import json
import os
import random
import pytest
from conftest import GOLDEN


@pytest.fixture(scope="module")
def program():
    from spp import acir
    return acir.load_program(os.path.join(GOLDEN, "reference_withdraw_acir.json"))


def test_acir_program_matches_the_recorded_facts(program):
    from spp import acir
    c = program.main
    assert program.noir_version.startswith("1.0.0-beta.18")
    assert (c.name, c.current_witness_index, len(c.opcodes)) == ("main", 23643, 6180)
    assert c.histogram() == {"AssertZero": 6148, "RANGE": 25, "BrilligCall": 6, "MultiScalarMul": 1}
    az = [op[1] for op in c.opcodes if op[0] == "AssertZero"]
    assert sum(len(e.mul_terms) for e in az) == 4688 and sum(len(e.linear) for e in az) == 44714
    assert sum(1 for e in az if len(e.mul_terms) == 1 and len(e.linear) == 1) == 4552      # Poseidon S-box steps
    assert sum(1 for e in az if len(e.linear) == 61) == 55                                  # MDS rows
    ranges = {}
    for op in c.opcodes:
        if op[0] == "RANGE":
            ranges[op[2]] = ranges.get(op[2], 0) + 1
    assert ranges == {64: 1, 126: 6, 128: 2, 1: 16}
    assert c.opcodes[0] == ("RANGE", ("witness", 3), 64)                                    # amount: u64
    msm = [op for op in c.opcodes if op[0] == "MultiScalarMul"][0]
    assert msm[1] == [("constant", 1), ("constant", 17631683881184975370165255887551781615748388533673675138860), ("constant", 0)]   # Grumpkin G
    # ABI order = TOML key order of client/proof.helper.ts:34-50; parameters sit on witnesses 0..25, public ones first
    names = [n for n, _ in program.parameter_witnesses()]
    assert names == ["root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness", "index", "siblings"]
    assert c.public_parameters == [0, 1, 2, 3, 4] and c.private_parameters == list(range(5, 26)) and c.return_values == []
    assert program.parameter_witnesses()[-1][1] == list(range(10, 26))
    rows, wide = acir.r1cs_rows(c)
    assert len(rows) + len(wide) == 6148 and len(wide) == 51


def test_reference_acir_accepts_the_reference_kat_and_refuses_what_main_asserts(program, withdraw_kat):
    """`nargo execute` on the reference's own inputs, by the host executor: every constraint of the reference's circuit holds
    for client/prover-params.toml; each assertion of noir_circuit/src/main.nr:38-82 fires on the matching mutation."""
    from spp import acir
    from oracle import circuit as C
    good = C.withdraw_inputs(withdraw_kat)
    w = acir.execute(program, good)
    assert len(w) == 6189 and max(w) == 23643           # 5 public + 6184 secret witnesses of the gnark system (App. A.4)
    assert [w[i] for i in range(26)] == [v % acir.R for v in good]
    for pos, val in ((0, good[0] + 1), (1, good[1] + 1), (2, 0), (3, 1 << 64), (4, good[4] + 1), (5, good[5] + 1), (6, good[6] + 1),
                     (8, good[8] + 1), (9, 1), (9, 1 << 16), (12, good[12] + 1)):
        bad = list(good)
        bad[pos] = val
        with pytest.raises(acir.UnsatisfiedConstraint):
            acir.execute(program, bad)
    other = list(good); other[2] = good[2] + 1          # the recipient is free (only != 0)
    acir.execute(program, other)


def test_reference_acir_and_repository_r1cs_accept_the_same_statements(program, withdraw_artifacts):
    """Fresh notes (oracle hashes, pinned by the KAT) in a tree: accepted by the reference's ACIR and by the repository's
    R1CS; one flipped input each: refused by both."""
    from spp import acir
    from oracle import hashes as H, native
    rng = random.Random(4)
    tree = H.MerkleTree()
    notes = []
    for _ in range(6):
        sk = rng.randrange(1, 1 << 128)
        owner = H.fixed_base_scalar_mul(sk)
        amount, rnd = rng.randrange(1, 1 << 63), rng.randrange(1 << 253)
        idx = tree.insert(H.poseidon_hash4(owner[0], owner[1], amount, rnd))
        notes.append((sk, owner, amount, rnd, idx))
    root = tree.root()
    p = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    rows = []
    for sk, owner, amount, rnd, idx in notes:
        rows.append([root, H.poseidon_hash2(sk, idx), rng.randrange(1, 1 << 240), amount, H.poseidon_hash2(owner[0], owner[1]),
                     sk, owner[0], owner[1], rnd, idx] + tree.proof(idx))
    assert native.check_many(p, rows) == [-1] * len(rows)
    for row in rows:
        acir.execute(program, row)
    for k, row in enumerate(rows):
        bad = list(row)
        pos = (0, 1, 4, 5, 9, 10 + k)[k]
        bad[pos] = (bad[pos] + 1) % acir.R
        assert native.check_many(p, [bad])[0] >= 0
        with pytest.raises(acir.UnsatisfiedConstraint):
            acir.execute(program, bad)


def _patched_program(doc, patch):
    """the reference program with `patch(raw bytearray, Program)` applied to its decoded bytecode"""
    import base64
    import gzip
    from spp import acir
    prog = acir.load_program(doc)
    raw = bytearray(prog.raw)
    patch(raw, prog)
    d = dict(doc)
    d["bytecode"] = base64.b64encode(gzip.compress(bytes(raw))).decode()
    return acir.load_program(d)


def test_brillig_helpers_are_identified_by_content_not_by_id_or_shape(program, withdraw_kat):
    """execute() used to pick the host helper by function id, to_blob() by call shape.  Both now ask
    Program.brillig_kind: the unconstrained section must be the reference's own (sha256), and every call site must fit its helper.
    (1) call sites with their function ids permuted, (2) a helper body that differs in one byte -- e.g. a 1 -> 1 function that is
    not the inverse -- are unsupported-program errors in BOTH consumers, not wrong hints."""
    import json
    import struct
    from spp import acir
    from oracle import circuit as C
    doc = json.load(open(os.path.join(GOLDEN, "reference_withdraw_acir.json")))
    row = C.withdraw_inputs(withdraw_kat)
    assert program.unconstrained_sha256 == acir.REFERENCE_UNCONSTRAINED_SHA256
    calls = [(i, op) for i, op in enumerate(program.main.opcodes) if op[0] == "BrilligCall"]
    assert sorted({op[1] for _, op in calls}) == [0, 1, 2] and len(program.main.brillig_id_offsets) == len(calls)
    assert {program.brillig_kind(i, op[1], op[2], op[3]) for i, op in calls} == {"divmod", "inverse", "radix"}

    def swap_ids(raw, prog):        # ids 0 <-> 1 at every call site: the divmod sites now name the inverse helper and vice versa
        for off in prog.main.brillig_id_offsets:
            (fid,) = struct.unpack_from("<I", raw, off)
            if fid in (0, 1):
                struct.pack_into("<I", raw, off, 1 - fid)
    swapped = _patched_program(doc, swap_ids)
    with pytest.raises(acir.AcirFormatError, match="shape"):
        acir.execute(swapped, row)
    with pytest.raises(acir.AcirFormatError, match="shape"):
        acir.to_blob(swapped)

    def other_body(raw, prog):      # one byte of a helper's bytecode: no longer the functions this module knows
        raw[len(raw) - 40] ^= 1
    changed = _patched_program(doc, other_body)
    assert changed.unconstrained_sha256 != acir.REFERENCE_UNCONSTRAINED_SHA256
    with pytest.raises(acir.AcirFormatError, match="not the reference's"):
        acir.execute(changed, row)
    with pytest.raises(acir.AcirFormatError, match="not the reference's"):
        acir.to_blob(changed)
    # the untouched program still executes and lowers
    assert acir.execute(program, row) and len(acir.to_blob(program)) > 1000


def test_witness_stack_round_trip_and_input_extraction(program, withdraw_kat, tmp_path):
    """target/<name>.gz (client/proof.helper.ts:58-66): writer and reader are inverse; the ABI inputs come back out of a
    full witness in Prover.toml order -- the row spp_prove_batch takes."""
    from spp import acir
    from oracle import circuit as C
    good = C.withdraw_inputs(withdraw_kat)
    w = acir.execute(program, good)
    path = str(tmp_path / "shielded_pool_verifier.gz")
    acir.write_witness_stack(path, w)
    back = acir.read_witness_stack(path)
    assert back == w
    assert acir.abi_input_row(program, back) == [v % acir.R for v in good]
    del back[7]
    with pytest.raises(acir.AcirFormatError):
        acir.abi_input_row(program, back)


def test_ccs_container_matches_the_recorded_facts():
    from spp import ccs
    c = ccs.load_ccs(os.path.join(GOLDEN, "reference_withdraw.ccs"))
    assert c.file_size == 576547 and c.header0 == c.file_size - 32 and c.header_opaque == (0, 14, 0)
    assert (c.levels_len, c.instructions_len, c.calldata_len, c.cbor_len) == (19532, 15396, 394954, 94465)
    assert c.cbor_offset == 429946 and c.coeff_offset == 524411 and c.trailing == 0
    m = c.meta
    assert m["GnarkVersion"] == "0.14.0" and int(m["ScalarField"], 16) == ccs.R and m["Type"] == 1
    assert (c.n_constraints, c.n_internal, len(c.public), len(c.secret)) == (12452, 6749, 6, 6184)
    assert c.public == ["1", "root", "nullifier", "recipient", "amount", "wa_commitment"]      # order of withdraw.rs:74-90
    assert c.secret[0] == "__witness_5" and c.secret[-1] == "__witness_23643"                   # = the ACIR witnesses
    assert 6 + 6184 + 6749 == 12939 and 12452 <= 1 << 14                                        # wires; FFT domain 2^14
    ci = m["CommitmentInfo"].value
    assert len(ci) == 1 and ci[0]["CommitmentIndex"] == 12426 and len(ci[0]["PrivateCommitted"]) == 490      # one BSB22 commitment
    assert ci[0]["NbPublicCommitted"] == 0 and ci[0]["PublicAndCommitmentCommitted"] == []
    hints = sorted(v.rsplit("/", 1)[-1] for v in m["MHintsDependencies"].values())
    assert hints == sorted(["rangecheck.DecomposeHint", "emulated.mulHint", "hints.Randomize", "sw-grumpkin.decompose", "logderivarg.countHint",
                            "solver.InvZeroHint", "sw-grumpkin.decomposeScalar", "bits.nBits", "cs.Bsb22CommitmentComputePlaceholder"])
    assert len(c.coefficients_mont) == 1629
    assert c.coefficients_mont[1] == (1 << 256) % ccs.R                  # entry 1 = R mod r (Montgomery one)
    assert [ccs.coefficient(c, i) for i in range(5)] == [0, 1, 2, ccs.R - 1, ccs.R - 2]
    assert all(v < ccs.R for v in c.coefficients_mont)
    # the secret witnesses of the gnark system are exactly the witnesses the ACIR executor solves
    from spp import acir
    prog = acir.load_program(os.path.join(GOLDEN, "reference_withdraw_acir.json"))
    kat = json.load(open(os.path.join(GOLDEN, "withdraw_kat.json")))
    from oracle import circuit as C
    w = acir.execute(prog, C.withdraw_inputs(kat))
    assert sorted(int(s.split("_")[-1]) for s in c.secret) == sorted(k for k in w if k >= 5)


def test_ccs_streams_decode_to_the_whole_constraint_system():
    """The three compressed streams (levels, instructions, calldata) of the reference's .ccs decode to a consistent system:
    12 493 instructions = 12 452 R1C rows + 41 hint calls of the 9 kinds the CBOR body names, 657 solver levels that hold every
    instruction exactly once, constraint / wire / calldata offsets that add up to the recorded dimensions."""
    from spp import ccs
    import collections
    c = ccs.load_ccs(os.path.join(GOLDEN, "reference_withdraw.ccs"))
    s = ccs.decode_system(c)
    n = len(s.blueprint)
    assert n == 12493 and len(s.rows) == c.n_constraints == 12452 and len(s.hints) == 41 and s.n_wires == 12939
    assert len(s.levels) == 657 and [len(l) for l in s.levels[:6]] == [6250, 4643, 398, 6, 2, 492]
    flat = [k for l in s.levels for k in l]
    assert sorted(flat) == list(range(n)) and all(a < b for l in s.levels for a, b in zip(l, l[1:]))
    assert collections.Counter(s.blueprint) == {ccs.BLUEPRINT_R1C: 12452, ccs.BLUEPRINT_HINT: 41}
    # offsets: constraints count the R1C rows, wires end at the wire count, calldata offsets are the running record lengths
    assert s.constraint_offset[0] == 0 and s.constraint_offset[-1] == 12451 and s.wire_offset[0] == 6190 and s.wire_offset[-1] == 12939
    assert all(a <= b for a, b in zip(s.wire_offset, s.wire_offset[1:]))
    assert s.calldata_offset[0] == 0 and len(s.calldata) == 262332
    assert all(s.calldata_offset[k] + s.calldata[s.calldata_offset[k]] == (s.calldata_offset[k + 1] if k + 1 < n else len(s.calldata)) for k in range(n))
    by_name = collections.Counter(h[2].rsplit("/", 1)[-1] for h in s.hints)
    assert by_name == {"rangecheck.DecomposeHint": 27, "solver.InvZeroHint": 6, "bits.nBits": 2, "sw-grumpkin.decomposeScalar": 1,
                       "sw-grumpkin.decompose": 1, "emulated.mulHint": 1, "logderivarg.countHint": 1, "hints.Randomize": 1,
                       "cs.Bsb22CommitmentComputePlaceholder": 1}
    # every term names an existing coefficient-table entry and wire; hint outputs are fresh internal wires, in order
    assert all(wi < s.n_wires for row in s.rows for side in row for _, wi in side)
    assert all(6190 <= h[4] < h[5] <= 12939 for h in s.hints) and [h[4] for h in s.hints] == sorted(h[4] for h in s.hints)
    # the commitment: its 490 committed wires are the inputs of the placeholder hint, its value lands on one fresh wire
    ci = c.meta["CommitmentInfo"].value[0]
    bsb = [h for h in s.hints if h[2].endswith("Bsb22CommitmentComputePlaceholder")][0]
    assert [t[0][1] for t in bsb[3][1:]] == ci["PrivateCommitted"] and bsb[5] - bsb[4] == 1
    # the Grumpkin scalar is decomposed against the BN254 base field (4 x 64-bit limbs) with the curve's cube root of unity
    dec = [h for h in s.hints if h[2].endswith("decomposeScalar")][0]
    q = sum(t[0][0] << (64 * i) for i, t in enumerate(dec[3][9:13]))
    assert q == 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47


def test_reference_r1cs_accepts_the_acir_executor_witness(program, withdraw_kat):
    """gnark's solver loop over the decoded reference system, fed with the secret wires the ACIR executor solves (the .ccs names
    them __witness_<i>): all 41 hint calls run, 6 458 rows are checked outright and 5 994 more define a wire -- all 12 452 rows hold
    and all 12 939 wires are assigned; a changed witness or public input breaks rows.  This pins the stream decoding, the coefficient
    table, the wire numbering, the hint restatements and the executor against the reference's own constraint system."""
    from spp import ccs, acir
    from oracle import circuit as C
    c = ccs.load_ccs(os.path.join(GOLDEN, "reference_withdraw.ccs"))
    s = ccs.decode_system(c)
    row = C.withdraw_inputs(withdraw_kat)
    w = acir.execute(program, row)
    secret = {"__witness_%d" % k: v for k, v in w.items()}
    wires, st = ccs.solve_partial(s, c, row[:5], secret)
    assert st["rows_unsatisfied"] == [] and st["rows_checked"] == 6458 and st["rows_solved"] == 5994 and st["rows_skipped"] == 0
    assert st["hints_run"] == 41 and st["hints_skipped"] == [] and st["wires_known"] == 12939 and None not in wires
    # the lookup argument holds for whatever challenge the commitment hint returns
    _, st2 = ccs.solve_partial(s, c, row[:5], secret, challenge=987654321)
    assert st2["rows_unsatisfied"] == [] and st2["rows_checked"] == 6458
    # a witness that is not the executor's: one hash state flipped
    bad = dict(secret)
    k = sorted(w)[3000]
    bad["__witness_%d" % k] = (w[k] + 1) % ccs.R
    _, st3 = ccs.solve_partial(s, c, row[:5], bad)
    assert st3["rows_unsatisfied"]
    # public inputs are bound as well
    _, st4 = ccs.solve_partial(s, c, [row[0] + 1] + row[1:5], secret)
    assert st4["rows_unsatisfied"]


def test_reference_r1cs_binds_only_the_low_half_of_the_grumpkin_scalar(program):
    """A property of the reference's gnark system that the decoding brings out: its Grumpkin multiplication is fed with the LOW
    128-bit limb of the secret key only (the decomposition hints and the emulated product read wire 28 = __witness_27; the ACIR
    MultiScalarMul takes (witness 27, witness 34) = (lo, hi)).  For a key >= 2^128 the ACIR program -- and this repository's own
    R1CS -- derive the public key of the whole scalar, and the two rows of the gnark system that compare its result with the ACIR
    outputs (1074, 1075) cannot hold; keys below 2^128 (what client/prover-params.toml uses: 127 bits) are unaffected."""
    from spp import ccs, acir
    c = ccs.load_ccs(os.path.join(GOLDEN, "reference_withdraw.ccs"))
    s = ccs.decode_system(c)
    msm = [op for op in program.main.opcodes if op[0] == "MultiScalarMul"][0]
    assert msm[2] == [("witness", 27), ("witness", 34)] and c.secret[28 - 6] == "__witness_27"
    dec = [h for h in s.hints if h[2].endswith("decomposeScalar")][0]
    assert dec[3][6] == [(1, 28)]
    from oracle import hashes as H
    import random
    rng = random.Random(9)
    for sk, ok in ((rng.randrange(1 << 127, 1 << 128), True), ((1 << 128) + 5, False), (rng.randrange(1 << 250, 1 << 253), False)):
        tree = H.MerkleTree()
        owner = H.fixed_base_scalar_mul(sk)
        amount, rnd = 7, rng.randrange(1 << 250)
        idx = tree.insert(H.poseidon_hash4(owner[0], owner[1], amount, rnd))
        row = [tree.root(), H.poseidon_hash2(sk, idx), 1, amount, H.poseidon_hash2(owner[0], owner[1]), sk, owner[0], owner[1], rnd, idx] + tree.proof(idx)
        w = acir.execute(program, row)                       # the ACIR program accepts all three
        _, st = ccs.solve_partial(s, c, row[:5], {"__witness_%d" % k: v for k, v in w.items()})
        assert st["rows_unsatisfied"] == ([] if ok else [1074, 1075]), sk.bit_length()


def _reference_system(tmp):
    from spp import ccs
    c = ccs.load_ccs(os.path.join(GOLDEN, "reference_withdraw.ccs"))
    s = ccs.decode_system(c)
    sppc = os.path.join(str(tmp), "shielded_pool_verifier_ccs.sppc")
    assert ccs.to_sppc(s, c, sppc) == 12452
    return c, s, sppc


def _fresh_note_inputs(seed):
    """withdraw inputs of a note that is not the reference KAT (client/merkle.ts semantics from oracle/hashes.py)"""
    import random
    from oracle import hashes as H
    rng = random.Random(seed)
    tree = H.MerkleTree()
    sk, amount, rnd = rng.randrange(1, 1 << 128), rng.randrange(1, 1 << 40), rng.randrange(1 << 250)
    for _ in range(3):
        tree.insert(rng.randrange(1 << 250))
    owner = H.fixed_base_scalar_mul(sk)
    idx = tree.insert(H.poseidon_hash4(owner[0], owner[1], amount, rnd))
    tree.insert(rng.randrange(1 << 250))
    return [tree.root(), H.poseidon_hash2(sk, idx), rng.randrange(1, 1 << 240), amount, H.poseidon_hash2(owner[0], owner[1]),
            sk, owner[0], owner[1], rnd, idx] + tree.proof(idx)


def test_reference_r1cs_is_proved_by_the_oracle(program, withdraw_kat, tmp_path):
    """SURVEY 8f-1 on the CPU side: the reference's OWN gnark constraint system (decoded from its .ccs) goes through setup, the
    witness is completed by gnark's solver loop with every one of its 41 hint calls (the three whose code is in neither tree --
    Sunspot's Grumpkin scalar decomposition and gnark's emulated product -- restated from the rows that consume their outputs),
    all 12 452 rows hold, the oracle proves and the pairing verifier accepts; the key has the reference's vk size (1 296 B: 7
    public-side points).  A witness that breaks the statement is refused."""
    from spp import ccs, acir
    from oracle import circuit as C, native, groth16
    c, s, sppc = _reference_system(tmp_path)
    oc = C.Circuit(sppc)
    assert (oc.n_public, oc.n_wires, oc.n_constraints, oc.domain_log, oc.challenge_wire, len(oc.committed)) == (6, 12939, 12452, 14, 12426, 490)
    pk, vk = str(tmp_path / "ref.pk"), str(tmp_path / "ref.vk")
    native.setup(sppc, b"\x07" * 32, pk, vk)
    assert os.path.getsize(vk) == os.path.getsize(os.path.join(GOLDEN, "reference_withdraw.vk")) == 1296
    pr = native.Prover(sppc, pk)
    cw = ccs.challenge_wire(s)

    def chal(partial_row):      # the oracle is the checker here: its solver runs the commitment step and shows the wire
        return pr.prove(partial_row, 1, 2, want_wires=True)[3][cw]
    for k, row in enumerate((C.withdraw_inputs(withdraw_kat), _fresh_note_inputs(5))):
        w = acir.execute(program, row)
        secret = {"__witness_%d" % i: v for i, v in w.items()}
        wires, st = ccs.solve_partial(s, c, row[:5], secret, challenge_fn=lambda ws: chal([0 if v is None else v for v in ws[1:]]))
        assert st["rows_unsatisfied"] == [] and st["rows_skipped"] == 0 and st["hints_run"] == 41 and st["wires_known"] == 12939
        assert st["rows_checked"] + st["rows_solved"] == 12452
        rc, proof, pw = pr.prove(wires[1:], 1000 + k, 2000 + k)
        assert rc == 0 and pw == groth16.public_witness_bytes(row[:5])
        assert groth16.verify(open(vk, "rb").read(), proof, pw)
        bad = list(wires[1:])
        bad[9000] = (bad[9000] + 1) % ccs.R
        assert pr.prove(bad, 1, 2)[0] == 1
    # the scalar decomposition: every scalar has a pair in range
    import random
    rng = random.Random(3)
    for sc in [0, 1, ccs.R - 1, (1 << 128) - 1, 1 << 128] + [rng.randrange(ccs.R) for _ in range(300)]:
        s1, s2 = ccs.glv_split(sc, ccs.GLV_LAMBDA)
        assert 0 <= s1 < 1 << 127 and 0 <= s2 < 1 << 127 and (s1 - ccs.GLV_LAMBDA * s2 - sc) % ccs.Q_BASE == 0


@pytest.mark.gpu
def test_reference_r1cs_is_proved_on_the_gpu(program, withdraw_kat, tmp_path):
    """The same on the product side: GPU setup bytes equal the oracle's, spp_commitment_challenge returns the challenge the oracle's
    solver derives, the proofs of the reference KAT and of a fresh note are byte-identical to the oracle's under the same blinding,
    the host and the batched GPU verifier accept them, a broken witness is refused in place."""
    import spp
    from spp import ccs, acir
    from oracle import circuit as C, native, groth16
    c, s, sppc = _reference_system(tmp_path)
    pk, vk, opk, ovk = (str(tmp_path / n) for n in ("ref.pk", "ref.vk", "oref.pk", "oref.vk"))
    ctx = spp.Context(0)
    try:
        ctx.setup(sppc, b"\x07" * 32, pk, vk)
        native.setup(sppc, b"\x07" * 32, opk, ovk)
        assert open(pk, "rb").read() == open(opk, "rb").read() and open(vk, "rb").read() == open(ovk, "rb").read()
        h = ctx.load_circuit(sppc, pk, 6)
        try:
            assert h.n_inputs == 12938
            orc = native.Prover(sppc, opk)
            cw = ccs.challenge_wire(s)
            rows, fulls = [C.withdraw_inputs(withdraw_kat), _fresh_note_inputs(8)], []
            for row in rows:
                w = acir.execute(program, row)
                secret = {"__witness_%d" % i: v for i, v in w.items()}
                seen = []

                def chal(partial_row):
                    got = h.commitment_challenge([partial_row])[0]
                    seen.append((got, orc.prove(partial_row, 1, 2, want_wires=True)[3][cw]))
                    return got
                fulls.append(ccs.reference_witness(s, c, row[:5], secret, chal))
                assert len(seen) == 1 and seen[0][0] == seen[0][1]
            bad = list(fulls[0])
            bad[7000] = (bad[7000] + 1) % ccs.R
            rs = [(31, 57), (2 ** 200 + 5, 2 ** 199 + 9), (3, 4)]
            proofs, pws, status = h.prove_batch(fulls + [bad], rs)
            assert status[:2] == [0, 0] and status[2] != 0 and proofs[2] == bytes(388)
            vkb = open(vk, "rb").read()
            for i in range(2):
                rc, proof, pw = orc.prove(fulls[i], rs[i][0], rs[i][1])
                assert rc == 0 and proofs[i] == proof and pws[i] == pw == groth16.public_witness_bytes(rows[i][:5])
                assert spp.verify(vkb, proofs[i], pws[i]) and groth16.verify(vkb, proofs[i], pws[i])
            assert ctx.verify_batch(vkb, proofs[:2], pws[:2]) == [True, True]
        finally:
            h.close()
    finally:
        ctx.close()


def test_solver_plan_of_the_reference_r1cs(program):
    """spp/ccs.py plan_solver: from the 26 ABI inputs every one of the 12 939 wires follows by gnark's own rule (a row with one
    unknown wire defines it) plus the 41 hint calls and the ACIR program's unconstrained helpers; the emitted container holds
    the same 12 452 rows and a program of the new solver instructions only."""
    from spp import ccs
    c = ccs.load_ccs(os.path.join(GOLDEN, "reference_withdraw.ccs"))
    s = ccs.decode_system(c)
    steps, _ = ccs.plan_solver(s, c, program)
    kinds = {}
    for st in steps:
        kinds[st[0]] = kinds.get(st[0], 0) + 1
    assert kinds["hint"] == 41 and kinds["acir"] == 7 and kinds["row"] > 12000          # 6 Brillig calls + the MultiScalarMul
    defined = {st[2] for st in steps if st[0] == "row"}
    assert len(defined) == kinds["row"]                       # one wire per defining row


@pytest.mark.gpu
def test_reference_r1cs_is_solved_and_proved_on_the_gpu_from_the_26_inputs(program, withdraw_kat, tmp_path):
    """the reference's own constraint system with its SOLVER on the device (spp/ccs.py to_sppc_solved): the
    container takes the 26 withdraw inputs of client/proof.helper.ts:34-50.  (1) every wire the device computes == gnark's
    solver loop restated in spp/ccs.py solve_partial (fed by this repository's ACIR executor) on the reference KAT and on fresh
    notes with keys on both sides of 2^127 (the scalar decomposition's non-trivial half); (2) same verifying key and byte-identical
    proofs as the ORACLE proving the all-inputs container of the same system from solve_partial's witness, same seed, blinding
    and mask; (3) a batch of 64 distinct notes verifies; bad inputs are refused in place."""
    import spp
    from spp import ccs, acir
    from oracle import circuit as C, native, groth16
    c, s, pure = _reference_system(tmp_path)
    solved = str(tmp_path / "solved.sppc")
    n, ss = ccs.to_sppc_solved(s, c, program, solved)
    assert n == 12452
    pk, vk, opk, ovk = (str(tmp_path / x) for x in ("s.pk", "s.vk", "o.pk", "o.vk"))
    seed = b"\x0b" * 32
    ctx = spp.Context(0)
    try:
        ctx.setup(solved, seed, pk, vk)
        native.setup(pure, seed, opk, ovk)
        assert open(vk, "rb").read() == open(ovk, "rb").read()          # the same statement under the same toxic waste
        orc = native.Prover(pure, opk)
        h = ctx.load_circuit(solved, pk, 6)
        try:
            assert h.n_inputs == 26 and h.n_wires == s.n_wires + 1 and h.n_constraints == 12452
            rows = [C.withdraw_inputs(withdraw_kat)] + [_fresh_note_inputs(k) for k in range(20, 25)]
            assert any(r[5] >> 127 for r in rows) and any(not (r[5] >> 127) for r in rows)
            for i, row in enumerate(rows):
                r_, s_ = 1000003 * i + 17, (1 << 200) + 29 * i
                proofs, pws, status = h.prove_batch([row], [(r_, s_)])
                assert status == [0]
                dev = h.debug_witness()
                wa = acir.execute(program, row)
                ref, st = ccs.solve_partial(s, c, row[:5], {"__witness_%d" % k: v for k, v in wa.items()}, challenge=dev[ss.challenge_wire],
                                            randomizer=C.mask_value(r_, s_))
                assert not st["rows_unsatisfied"] and not st["rows_skipped"] and not st["hints_skipped"]
                assert [dev[ss.perm[g]] for g in range(s.n_wires)] == ref, i        # every wire
                rc, proof, pw = orc.prove(ref[1:], r_, s_)
                assert rc == 0 and proofs[0] == proof and pws[0] == pw, i           # same bytes as the oracle on the all-inputs system
                assert groth16.verify(open(vk, "rb").read(), proofs[0], pws[0])
            batch = [_fresh_note_inputs(100 + k) for k in range(64)]
            bad = list(batch[7]); bad[3] += 1                                       # amount
            batch[7] = bad
            proofs, pws, status = h.prove_batch(batch, [(3 * k + 1, 5 * k + 2) for k in range(64)])
            assert [k for k in range(64) if status[k] != 0] == [7]
            good = [k for k in range(64) if k != 7]
            assert all(ctx.verify_batch(open(vk, "rb").read(), [proofs[k] for k in good], [pws[k] for k in good]))
        finally:
            h.close()
    finally:
        ctx.close()


def test_fixtures_equal_the_reference_files():
    ref = "/root/reference/noir_circuit/target"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present on this machine")
    assert open(os.path.join(ref, "shielded_pool_verifier.ccs"), "rb").read() == open(os.path.join(GOLDEN, "reference_withdraw.ccs"), "rb").read()
    j = json.load(open(os.path.join(ref, "shielded_pool_verifier.json")))
    f = json.load(open(os.path.join(GOLDEN, "reference_withdraw_acir.json")))
    assert j["bytecode"] == f["bytecode"] and j["abi"] == f["abi"] and "file_map" not in f and "debug_symbols" not in f


def test_cli_execute_writes_a_nargo_style_witness(tmp_path, withdraw_kat, capsys):
    """`spp execute` = the `nargo execute` step of proof.helper.ts:55 for the reference's compiled circuit: Prover.toml in,
    target/<name>.gz out; unsatisfiable inputs exit 1."""
    from spp import cli, acir
    from spp.proof_helper import ShieldedPoolInputs, prover_toml
    k = withdraw_kat
    inp = ShieldedPoolInputs(**{f: k[f] for f in ("root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x",
                                                  "owner_y", "randomness", "index", "siblings")})
    toml = tmp_path / "Prover.toml"
    toml.write_text(prover_toml(inp))
    out = tmp_path / "shielded_pool_verifier.gz"
    assert cli.main(["execute", os.path.join(GOLDEN, "reference_withdraw_acir.json"), str(toml), "-o", str(out)]) == 0
    assert "successfully solved" in capsys.readouterr().out
    w = acir.read_witness_stack(str(out))
    assert len(w) == 6189 and w[3] == k["amount"]
    bad = tmp_path / "Bad.toml"
    bad.write_text(prover_toml(inp).replace(k["nullifier"], k["root"]))
    assert cli.main(["execute", os.path.join(GOLDEN, "reference_withdraw_acir.json"), str(bad), "-o", str(out)]) == 1


@pytest.mark.gpu
def test_cli_prove_from_nargo_files_on_gpu(tmp_path, withdraw_kat, withdraw_artifacts):
    """sunspot's argument order: `spp prove <acir.json> <witness.gz> <sppc> <pk>` -- the inputs come out of the nargo witness."""
    import shutil
    import spp
    from spp import cli, acir
    from oracle import circuit as C, groth16
    prog = acir.load_program(os.path.join(GOLDEN, "reference_withdraw_acir.json"))
    gz = str(tmp_path / "w.gz")
    acir.write_witness_stack(gz, acir.execute(prog, C.withdraw_inputs(withdraw_kat)))
    sppc = str(tmp_path / "shielded_pool_verifier.sppc")
    shutil.copy(withdraw_artifacts["sppc"], sppc)
    assert cli.main(["prove", os.path.join(GOLDEN, "reference_withdraw_acir.json"), gz, sppc, withdraw_artifacts["pk"], "--window", "6"]) == 0
    proof = open(str(tmp_path / "shielded_pool_verifier.proof"), "rb").read()
    pw = open(str(tmp_path / "shielded_pool_verifier.pw"), "rb").read()
    assert len(proof) == 388 and pw == groth16.public_witness_bytes(C.withdraw_inputs(withdraw_kat)[:5])
    assert spp.verify(open(withdraw_artifacts["vk"], "rb").read(), proof, pw)


@pytest.mark.gpu
def test_cli_on_the_files_sunspot_itself_takes(tmp_path, withdraw_kat):
    """`sunspot setup <ccs>` and `sunspot prove <acir> <witness> <ccs> <pk>` (noir_circuit/prove_linux.sh:72-83, proof.helper.ts:58-64)
    with the reference's own .json and .ccs and a nargo-style witness file: keys, proof and public witness land next to the .ccs."""
    import shutil
    import spp
    from spp import cli, acir
    from oracle import circuit as C, groth16
    ccs_path = str(tmp_path / "shielded_pool_verifier.ccs")
    shutil.copy(os.path.join(GOLDEN, "reference_withdraw.ccs"), ccs_path)
    acir_path = os.path.join(GOLDEN, "reference_withdraw_acir.json")
    gz = str(tmp_path / "shielded_pool_verifier.gz")
    row = C.withdraw_inputs(withdraw_kat)
    acir.write_witness_stack(gz, acir.execute(acir.load_program(acir_path), row))
    assert cli.main(["setup", ccs_path, "--seed", "11" * 32]) == 0
    base = str(tmp_path / "shielded_pool_verifier")
    assert os.path.getsize(base + ".vk") == 1296
    assert cli.main(["prove", acir_path, gz, ccs_path, base + ".pk", "--window", "6"]) == 0
    proof, pw = open(base + ".proof", "rb").read(), open(base + ".pw", "rb").read()
    assert len(proof) == 388 and pw == groth16.public_witness_bytes(row[:5])
    assert cli.main(["verify", base + ".vk", base + ".proof", base + ".pw"]) == 0
    assert groth16.verify(open(base + ".vk", "rb").read(), proof, pw)
    # a witness of other inputs is refused
    bad = acir.read_witness_stack(gz)
    bad[sorted(bad)[2500]] = 1
    acir.write_witness_stack(gz, bad)
    assert cli.main(["prove", acir_path, gz, ccs_path, base + ".pk", "--window", "6"]) == 1

"""`sunspot compile <acir>` (noir_circuit/prove_linux.sh:66-70) for the reference's OWN compiled circuit: spp/acir.py decodes
noir_circuit/target/shielded_pool_verifier.json (fixture: bytecode + abi), csrc/circuit_acir.cpp lowers its 6 180 opcodes to an
R1CS + solver program in the SPPC container, and that circuit goes through the same setup / prover / verifier as the
hand-written ones.  What is checked without the product's solver: the oracle's interpreter and C solver run the compiled
system; the reference's ACIR, executed opcode by opcode (spp.acir.execute), accepts and refuses the same rows."""
import json
import os
import random
import pytest
from conftest import GOLDEN


@pytest.fixture(scope="module")
def acir_circuit(tmp_path_factory):
    from spp import acir
    from oracle import native
    d = tmp_path_factory.mktemp("acir")
    prog = acir.load_program(os.path.join(GOLDEN, "reference_withdraw_acir.json"))
    sppc, pk, vk = (str(d / ("shielded_pool_verifier." + e)) for e in ("sppc", "pk", "vk"))
    n = acir.compile_to_sppc(prog, sppc)
    native.setup(sppc, b"\x0b" * 32, pk, vk)
    return dict(prog=prog, sppc=sppc, pk=pk, vk=vk, n_constraints=n)


def _notes(count, seed):
    from oracle import hashes as H
    rng = random.Random(seed)
    tree = H.MerkleTree()
    notes = []
    for _ in range(count):
        sk = rng.randrange(1, 1 << 128)
        owner = H.fixed_base_scalar_mul(sk)
        amount, rnd = rng.randrange(1, 1 << 63), rng.randrange(1 << 253)
        notes.append((sk, owner, amount, rnd, tree.insert(H.poseidon_hash4(owner[0], owner[1], amount, rnd))))
    root = tree.root()
    return [[root, H.poseidon_hash2(sk, idx), rng.randrange(1, 1 << 240), amount, H.poseidon_hash2(owner[0], owner[1]),
             sk, owner[0], owner[1], rnd, idx] + tree.proof(idx) for sk, owner, amount, rnd, idx in notes]


def test_reference_acir_compiles_to_a_small_r1cs(acir_circuit):
    from oracle import circuit as C
    c = C.Circuit(acir_circuit["sppc"])
    assert c.id == 1 and (c.n_public - 1, c.n_secret) == (5, 21)            # the withdraw ABI: usable by generateProof as is
    assert c.n_constraints == acir_circuit["n_constraints"] and 6000 < c.n_constraints < 8192 and c.domain_log == 13
    # 12 452 constraints in the reference's gnark R1CS (emulated-field Grumpkin), 7 752 in the hand-written circuit.cpp
    assert len(c.committed) > 0                                             # one BSB22 commitment: the 388-byte proof layout holds


def test_compiled_circuit_accepts_and_refuses_like_the_reference_acir(acir_circuit, withdraw_kat):
    from spp import acir
    from oracle import circuit as C, native
    prog = acir_circuit["prog"]
    c = C.Circuit(acir_circuit["sppc"])
    p = native.Prover(acir_circuit["sppc"], acir_circuit["pk"])
    good = C.withdraw_inputs(withdraw_kat)
    w = C.solve(c, good, lambda w_: 0x77aa55)
    assert C.first_unsatisfied(c, w) == -1 and w[1:27] == [v % acir.R for v in good]
    # the main.nr:84-130 test vector and fresh notes
    from oracle import hashes as H
    owner = H.fixed_base_scalar_mul(12345)
    cm = H.poseidon_hash4(owner[0], owner[1], 1000000, 67890)
    noir_vec = [H.compute_merkle_root(cm, 0, [0] * 16), H.poseidon_hash2(12345, 0), 0x1234, 1000000, H.poseidon_hash2(*owner),
                12345, owner[0], owner[1], 67890, 0] + [0] * 16
    rows = [good, noir_vec] + _notes(5, 31)
    assert native.check_many(p, rows) == [-1] * len(rows)
    for row in rows:
        acir.execute(prog, row)
    # every mutation: both refuse (the C solver of the compiled system, and ACVM-style execution of the ACIR)
    rng = random.Random(5)
    bad_rows = []
    for row in rows:
        for pos in (0, 1, 3, 4, 5, 6, 7, 8, 9, 10 + rng.randrange(16)):
            b = list(row)
            b[pos] = (b[pos] + 1 + rng.randrange(3)) % acir.R
            bad_rows.append(b)
    b = list(good); b[2] = 0; bad_rows.append(b)                 # recipient == 0
    b = list(good); b[3] = 1 << 64; bad_rows.append(b)           # amount: u64
    res = native.check_many(p, bad_rows)
    assert all(r >= 0 for r in res), [i for i, r in enumerate(res) if r < 0]
    for b in bad_rows[::7]:
        with pytest.raises(acir.UnsatisfiedConstraint):
            acir.execute(prog, b)
    other = list(good); other[2] += 1                            # the recipient is free (!= 0 only), in both
    assert native.check_many(p, [other]) == [-1]
    acir.execute(prog, other)


def test_compiled_circuit_every_input_and_every_wire_is_pinned(acir_circuit, withdraw_kat):
    """Soundness of the LOWERING: every input +-1 refused; every wire of the compiled R1CS pinned by some row, except the
    wires ACIR itself leaves free on this witness -- inverse hints of a value that is zero (Brillig outputs are unconstrained
    by design; the is-zero gadget of the following opcodes binds them only when the value is non-zero)."""
    from oracle import circuit as C, native
    from oracle.bn254 import R
    from test_circuit_soundness import _free_wires
    p = native.Prover(acir_circuit["sppc"], acir_circuit["pk"])
    good = C.withdraw_inputs(withdraw_kat)
    rows = []
    for i in range(len(good)):
        for d in (1, -1):
            x = list(good); x[i] = (x[i] + d) % R
            rows.append(x)
    res = native.check_many(p, rows)
    accepted = sorted(set(k // 2 for k in range(len(rows)) if res[k] == -1))
    assert accepted == [2]                                       # recipient: only != 0 (main.nr:80-81)
    c = C.Circuit(acir_circuit["sppc"])
    w = C.solve(c, good, lambda w_: 0xabc123)
    free = _free_wires(c, w, [1, random.Random(2).randrange(2, R)])
    inv_hints = set()      # free by design: inverse hints of a zero value, and the commitment's random mask (no constraint touches it)
    prog, pc = c.program, 0
    while prog[pc] != C.OP_END:
        op = prog[pc]
        if op == C.OP_INV_H:
            if C._dot(c.H, prog[pc + 1], w) == 0:
                inv_hints.add(prog[pc + 2])
            pc += 3
        elif op == C.OP_MASK:
            inv_hints.add(prog[pc + 1])
            assert prog[pc + 1] in c.committed
            pc += 2
        else:
            pc += {C.OP_SOLVE_C: 2, C.OP_SOLVE_A: 2, C.OP_BATCH_DIV: 3, C.OP_BITS: 4, C.OP_LIMBS8: 4, C.OP_COUNT8: 4, C.OP_POSEIDON: 4,
                   C.OP_POSEIDON2: 3, C.OP_COMMIT: 1}.get(op) or (5 + prog[pc + 4])
    assert set(free) <= inv_hints, sorted(set(free) - inv_hints)[:10]


def test_compiled_circuit_oracle_proof_verifies(acir_circuit, withdraw_kat):
    import spp
    from oracle import native, groth16, circuit as C
    p = native.Prover(acir_circuit["sppc"], acir_circuit["pk"])
    row = C.withdraw_inputs(withdraw_kat)
    rc, proof, pw = p.prove(row, 41, 43)
    vk = open(acir_circuit["vk"], "rb").read()
    assert rc == 0 and len(proof) == 388 and len(vk) == 1296 and pw == groth16.public_witness_bytes(row[:5])
    assert groth16.verify(vk, proof, pw) and spp.verify(vk, proof, pw)
    bad = bytearray(proof); bad[0] ^= 1
    assert not spp.verify(vk, bytes(bad), pw)


def test_cli_compile_from_acir_json(tmp_path, capsys):
    from spp import cli
    out = str(tmp_path / "c.sppc")
    assert cli.main(["compile", os.path.join(GOLDEN, "reference_withdraw_acir.json"), "-o", out]) == 0
    line = capsys.readouterr().out.strip()
    assert line.startswith("nbConstraints=") and int(line.split("=")[1]) > 6000          # the line benchmark_all.py:646,664 parses


def test_unsupported_programs_are_refused_with_a_reason(acir_circuit):
    import copy
    from spp import acir
    prog = copy.copy(acir_circuit["prog"])
    prog.main = copy.copy(prog.main)
    prog.main.opcodes = list(prog.main.opcodes)
    k = next(i for i, op in enumerate(prog.main.opcodes) if op[0] == "MultiScalarMul")
    op = prog.main.opcodes[k]
    prog.main.opcodes[k] = (op[0], [("constant", 2), op[1][1], op[1][2]], op[2], op[3], op[4])      # another base point
    with pytest.raises(acir.AcirFormatError):
        acir.to_blob(prog)


def test_lowering_refuses_an_output_that_is_already_defined(acir_circuit, tmp_path):
    """a helper / black-box output must be a fresh witness.  A blob whose inverse helper writes to an INPUT witness (or
    whose MultiScalarMul writes to an already solved one) is refused by spp_circuit_build_acir instead of silently re-mapping it."""
    import copy
    import ctypes
    from spp import acir
    from spp.lib import load_library, last_error
    prog = copy.copy(acir_circuit["prog"])
    prog.main = copy.copy(prog.main)
    prog.main.opcodes = list(prog.main.opcodes)
    L = load_library()
    out = str(tmp_path / "bad.sppc")
    n = ctypes.c_uint32(0)
    k = next(i for i, op in enumerate(prog.main.opcodes) if op[0] == "BrilligCall" and prog.brillig_kind(i, op[1], op[2], op[3]) == "inverse")
    op = prog.main.opcodes[k]
    prog.main.opcodes[k] = (op[0], op[1], op[2], [("simple", 3)], op[4])                    # witness 3 = the public `amount`
    blob = acir.to_blob(prog)
    assert L.spp_circuit_build_acir(blob, len(blob), 1, out.encode(), ctypes.byref(n)) != 0 and "already defined" in last_error()
    prog.main.opcodes[k] = op
    k = next(i for i, op_ in enumerate(prog.main.opcodes) if op_[0] == "MultiScalarMul")
    op = prog.main.opcodes[k]
    prog.main.opcodes[k] = (op[0], op[1], op[2], op[3], (op[4][0], 5, op[4][2]))              # y output onto the secret key's witness
    blob = acir.to_blob(prog)
    assert L.spp_circuit_build_acir(blob, len(blob), 1, out.encode(), ctypes.byref(n)) != 0 and "already defined" in last_error()


@pytest.mark.gpu
def test_gpu_proves_the_reference_acir_circuit(acir_circuit, withdraw_kat, tmp_path):
    """The reference's compiled circuit on the GPU: setup bytes == oracle setup, 64 distinct notes proved, a sample byte-identical
    to the oracle's proofs, all accepted by the batched verifier; unsatisfying rows refused; generateProof (Python mirror) works
    on it unchanged because the container carries the withdraw ABI."""
    import ctypes
    import shutil
    import spp
    from spp import workload, proof_helper
    from oracle import native, groth16, circuit as C
    ctx = spp.Context(0)
    try:
        pk2, vk2 = str(tmp_path / "g.pk"), str(tmp_path / "g.vk")
        ctx.setup(acir_circuit["sppc"], b"\x0b" * 32, pk2, vk2)
        assert open(pk2, "rb").read() == open(acir_circuit["pk"], "rb").read() and open(vk2, "rb").read() == open(acir_circuit["vk"], "rb").read()
        h = ctx.load_circuit(acir_circuit["sppc"], acir_circuit["pk"], 6)
        try:
            B_ = 64
            rows_b = workload.withdraw_rows(ctx, B_, seed=9)
            rows = [workload.row_ints(rows_b, h.n_inputs, i) for i in range(B_)]
            rows[17][1] += 1                                      # one bad nullifier in the middle
            rs = [(3 + 2 * i, 5 + 7 * i) for i in range(B_)]
            proofs, pws, status = h.prove_batch(rows, rs)
            assert [i for i, s in enumerate(status) if s != 0] == [17] and proofs[17] == bytes(388)
            orc = native.Prover(acir_circuit["sppc"], acir_circuit["pk"])
            for i in (0, 1, 16, 18, 63):
                rc, proof, pw = orc.prove(rows[i], *rs[i])
                assert rc == 0 and proofs[i] == proof and pws[i] == pw, i
            vk = open(acir_circuit["vk"], "rb").read()
            ok = ctx.verify_batch(vk, proofs, pws)
            assert ok == [i != 17 for i in range(B_)]
            assert groth16.verify(vk, proofs[5], pws[5])
            assert h.debug_witness()[:27] == [1] + [v % C.R for v in rows[0]]
        finally:
            h.close()
    finally:
        ctx.close()
    # drop-in: generateProof on a circuit directory whose .sppc came from `spp compile <acir.json>`
    cdir = tmp_path / "noir_circuit"
    os.makedirs(cdir / "target")
    shutil.copy(acir_circuit["sppc"], cdir / "target" / "shielded_pool_verifier.sppc")
    shutil.copy(acir_circuit["pk"], cdir / "target" / "shielded_pool_verifier.pk")
    fields = ("root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness", "index", "siblings")
    out = proof_helper.generateProof(proof_helper.CircuitConfig(str(cdir), "shielded_pool_verifier"),
                                     proof_helper.ShieldedPoolInputs(**{f: withdraw_kat[f] for f in fields}))
    assert groth16.verify(open(acir_circuit["vk"], "rb").read(), out["proof"], out["publicWitness"])

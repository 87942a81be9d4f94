"""Soundness evidence for the two R1CS the product builds (csrc/circuit.cpp, csrc/circuit_audit.cpp) that does NOT
come from the product's own solver: HIP == oracle parity says nothing about a wire the constraints leave free, so

  1. per-wire perturbation sweep: take a satisfying witness, change ONE wire (by +1 and by a random amount), keep every
     other wire, and require some constraint to fail -- for EVERY wire of both circuits (the constant wire excepted).
     A wire that can move alone is under-constrained.  Today no wire of either circuit can;
  2. input sweep: every input +-1 (re-solved by the oracle's C solver) must be refused -- all 26 withdraw inputs (the
     recipient excepted: the statement only asks recipient != 0, main.nr:80-81), all 3360 audit inputs;
  3. the reference's own circuit test vector (noir_circuit/src/main.nr:84-130: sk 12345, amount 1000000, randomness 67890,
     index 0, zero siblings, recipient 0x1234) goes through the circuit and is satisfied, with the public values the
     Noir test computes.
"""
import random
import pytest


def _index(circ):
    """wire -> list of (matrix id, row, coefficient)"""
    from oracle.bn254 import R
    idx = [[] for _ in range(circ.n_wires)]
    for m_id, m in enumerate((circ.A, circ.B, circ.C)):
        rp, wires, coeffs = m.rowptr, m.wires, m.coeffs
        for k in range(circ.n_constraints):
            for t in range(rp[k], rp[k + 1]):
                idx[wires[t]].append((m_id, k, coeffs[t] % R))
    return idx


def _free_wires(circ, w, deltas):
    """the wires for which SOME delta of `deltas`, applied to that wire alone, leaves every constraint satisfied"""
    from oracle import circuit as C
    from oracle.bn254 import R
    a, b, c = C.evaluate(circ, w)
    assert all(a[k] * b[k] % R == c[k] for k in range(circ.n_constraints))
    idx = _index(circ)
    free = []
    for wire in range(1, circ.n_wires):
        for d in deltas:
            rows = {}
            for m_id, k, cf in idx[wire]:
                v = rows.setdefault(k, [a[k], b[k], c[k]])
                v[m_id] = (v[m_id] + cf * d) % R
            if all(v[0] * v[1] % R == v[2] for v in rows.values()):
                free.append(wire)
                break
    return free


def _mask_wires(circ):
    """The commitment's random mask (OP_MASK; gnark: hints.Randomize inside api.Commit): a committed wire that no constraint touches,
    on purpose -- it only has to make the Pedersen commitment hiding."""
    from oracle import circuit as C
    out, prog, pc = [], circ.program, 0
    while prog[pc] != C.OP_END:
        op = prog[pc]
        if op == C.OP_MASK:
            out.append(prog[pc + 1])
        pc += {C.OP_SOLVE_C: 2, C.OP_SOLVE_A: 2, C.OP_BATCH_DIV: 3, C.OP_BITS: 4, C.OP_LIMBS8: 4, C.OP_COUNT8: 4, C.OP_POSEIDON: 4,
               C.OP_POSEIDON2: 3, C.OP_COMMIT: 1, C.OP_INV_H: 3, C.OP_MASK: 2}.get(op) or (5 + prog[pc + 4])
    assert len(out) == 1 and out[0] in circ.committed
    return out


def test_withdraw_every_wire_is_constrained(withdraw_artifacts, withdraw_kat):
    from oracle import circuit as C
    from oracle.bn254 import R
    circ = C.Circuit(withdraw_artifacts["sppc"])
    rng = random.Random(11)
    w = C.solve(circ, C.withdraw_inputs(withdraw_kat), lambda w_: 0xabcdef0123)
    free = _free_wires(circ, w, [1, rng.randrange(2, R)])
    assert free == _mask_wires(circ), "under-constrained withdraw wires: %r" % free[:20]


def test_audit_every_wire_is_constrained(audit_artifacts, rlwe_pk):
    from oracle import circuit as C, rlwe
    from oracle.bn254 import R
    circ = C.Circuit(audit_artifacts["sppc"])
    rng = random.Random(12)
    d = rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345, random.Random(999))
    w = C.solve(circ, rlwe.audit_input_vector(d), lambda w_: 0x1234567)
    free = _free_wires(circ, w, [1, rng.randrange(2, R)])
    assert free == _mask_wires(circ), "under-constrained audit wires: %r" % free[:20]


def test_withdraw_every_input_plus_minus_one_is_refused(withdraw_artifacts, withdraw_kat):
    from oracle import native, circuit as C
    from oracle.bn254 import R
    p = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    good = C.withdraw_inputs(withdraw_kat)
    rows = [good]
    for i in range(len(good)):
        for d in (1, -1):
            x = list(good)
            x[i] = (x[i] + d) % R
            rows.append(x)
    res = native.check_many(p, rows)
    assert res[0] == -1
    accepted = sorted(set((k - 1) // 2 for k in range(1, len(rows)) if res[k] == -1))
    # input 2 = recipient: main.nr:80-81 only asserts recipient != 0 -- it is a free PUBLIC input, bound to the proof through
    # the public witness (a proof checked against another recipient fails: test_product_verifier_agrees_with_oracle)
    assert accepted == [2], "withdraw inputs that can move by one: %r" % accepted
    zero_recipient = list(good); zero_recipient[2] = 0
    assert native.check_many(p, [zero_recipient])[0] >= 0


def test_audit_every_input_plus_minus_one_is_refused(audit_artifacts, rlwe_pk):
    from oracle import native, rlwe
    from oracle.bn254 import R
    p = native.Prover(audit_artifacts["sppc"], audit_artifacts["pk"])
    good = rlwe.audit_input_vector(rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345, random.Random(999)))
    accepted = []
    assert native.check_many(p, [good]) == [-1]
    step = 420                                   # 8 rounds of 840 rows: bounded memory (840 x 3360 x 32 B)
    for lo in range(0, len(good), step):
        rows = []
        for i in range(lo, min(lo + step, len(good))):
            for d in (1, -1):
                x = list(good)
                x[i] = (x[i] + d) % R
                rows.append(x)
        res = native.check_many(p, rows)
        accepted += [lo + k // 2 for k in range(len(rows)) if res[k] == -1]
    assert accepted == [], "audit inputs that can move by one: %r" % accepted[:20]


def test_noir_unit_test_vector_through_the_circuit(withdraw_artifacts):
    """noir_circuit/src/main.nr:84-130 (#[test] test_shielded_pool_babyjubjub): the vector the reference's own circuit
    test runs, pushed through the product's R1CS by the oracle's interpreter AND its C solver."""
    from oracle import circuit as C, hashes as H, native
    sk, amount, randomness, index, recipient = 12345, 1000000, 67890, 0, 0x1234
    owner = H.fixed_base_scalar_mul(sk)                               # main.nr:96-105
    wa = H.poseidon_hash2(owner[0], owner[1])                         # :108
    commitment = H.poseidon_hash4(owner[0], owner[1], amount, randomness)   # :111
    siblings = [0] * 16                                               # :113
    root = H.compute_merkle_root(commitment, index, siblings)         # :114
    nullifier = H.poseidon_hash2(sk, index)                           # :115
    row = [root, nullifier, recipient, amount, wa, sk, owner[0], owner[1], randomness, index] + siblings
    circ = C.Circuit(withdraw_artifacts["sppc"])
    w = C.solve(circ, row, lambda w_: 0x5eed)
    assert C.first_unsatisfied(circ, w) == -1
    assert w[1:6] == [root, nullifier, recipient, amount, wa]
    p = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    assert native.check_many(p, [row]) == [-1]
    # all-zero siblings are NOT the empty-tree defaults: the root differs from the one of a one-leaf tree
    t = H.MerkleTree()
    t.insert(commitment)
    assert t.root() != root and H.compute_merkle_root(commitment, 0, t.proof(0)) == t.root()
    # and the assertions of main(): a wrong nullifier / zero recipient are refused
    bad = list(row); bad[1] = (bad[1] + 1)
    bad0 = list(row); bad0[2] = 0
    assert native.check_many(p, [bad, bad0]) != [-1, -1] and -1 not in native.check_many(p, [bad, bad0])

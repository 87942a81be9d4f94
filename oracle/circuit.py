"""ORACLE (test infrastructure only) -- reader + interpreter for the SPPC circuit container.

The circuit FILE is data produced by the product's builder (csrc/circuit.cpp); this module is an
independent interpreter of it: it solves every wire with Python ints and checks <A,w>*<B,w> == <C,w>
row by row, so a wrong solver program, a wrong gadget or a wrong GPU solver shows up as an unsatisfied
row or as a public value differing from oracle/hashes.py (which is pinned by the reference's
client/prover-params.toml). Semantics of the circuits follow noir_circuit/src/main.nr:38-82 and
scripts/generate_audit.py:405-463 of the reference.
"""
import struct
from .bn254 import R, inv
from . import hashes

OP_END, OP_SOLVE_C, OP_SOLVE_A, OP_BATCH_DIV, OP_BITS, OP_LIMBS8, OP_COUNT8, OP_POSEIDON, OP_POSEIDON2, OP_COMMIT, OP_GRUMPKIN, OP_INV_H, OP_MASK = range(13)


class Sparse:
    __slots__ = ("rowptr", "wires", "coeffs")

    def row(self, k):
        a, b = self.rowptr[k], self.rowptr[k + 1]
        return self.wires[a:b], self.coeffs[a:b]

    @property
    def rows(self):
        return len(self.rowptr) - 1


class Circuit:
    def __init__(self, path):
        data = open(path, "rb").read()
        off = [0]

        def u32s(n):
            v = struct.unpack_from("<%dI" % n, data, off[0])
            off[0] += 4 * n
            return v
        magic, version = u32s(2)
        assert magic == 0x43505053 and version == 2
        (self.id, self.n_public, self.n_secret, self.n_wires, self.n_constraints, self.domain_log,
         self.challenge_wire, ncoef, ncommitted, nprog, naux) = u32s(11)
        self.coeffs = []
        for _ in range(ncoef):
            limbs = u32s(8)
            self.coeffs.append(sum(l << (32 * i) for i, l in enumerate(limbs)))
        mats = []
        for _ in range(4):
            rows, nnz = u32s(2)
            m = Sparse()
            m.rowptr = u32s(rows + 1)
            flat = u32s(2 * nnz)
            m.wires = flat[0::2]
            m.coeffs = [self.coeffs[c] for c in flat[1::2]]
            mats.append(m)
        self.A, self.B, self.C, self.H = mats
        self.committed = list(u32s(ncommitted))
        self.program = list(u32s(nprog))
        self.aux = []
        for _ in range(naux):
            limbs = u32s(8)
            self.aux.append(sum(l << (32 * i) for i, l in enumerate(limbs)))
        assert off[0] == len(data)
        self.n = 1 << self.domain_log

    def n_inputs(self):
        return self.n_public - 1 + self.n_secret


def _dot(m, k, w):
    wires, coeffs = m.row(k)
    s = 0
    for wi, c in zip(wires, coeffs):
        s += c * w[wi]
    return s % R


def _poseidon_native(state, out, w):
    """Writes x^2, x^3, x^4, x^5 of every S-box in gadget order; returns next free wire."""
    t = len(state)
    rf, rp, rc, mds = hashes.poseidon_params(t)
    s = list(state)
    for rnd in range(rf + rp):
        s = [(s[i] + rc[rnd * t + i]) % R for i in range(t)]
        full = rnd < rf // 2 or rnd >= rf // 2 + rp
        for i in (range(t) if full else (0,)):
            x = s[i]
            x2 = x * x % R
            x3 = x2 * x % R
            x4 = x3 * x % R
            x5 = x4 * x % R
            w[out], w[out + 1], w[out + 2], w[out + 3] = x2, x3, x4, x5
            out += 4
            s[i] = x5
        s = [sum(mds[i][j] * s[j] for j in range(t)) % R for i in range(t)]
    return out


def _poseidon2_native(state, out, w):
    rc, mu = hashes.poseidon2_params()
    s = hashes._p2_external(state)
    k = 0

    def sbox(x):
        nonlocal out
        x2 = x * x % R            # four wires per S-box, as Poseidon: x^2, x^3, x^4, x^5 (B side of every row = x)
        x3 = x2 * x % R
        x4 = x3 * x % R
        x5 = x4 * x % R
        w[out], w[out + 1], w[out + 2], w[out + 3] = x2, x3, x4, x5
        out += 4
        return x5
    for _ in range(4):
        s = [sbox((s[i] + rc[k + i]) % R) for i in range(4)]
        k += 4
        s = hashes._p2_external(s)
    for _ in range(56):
        s[0] = sbox((s[0] + rc[k]) % R)
        k += 1
        tot = sum(s) % R
        s = [(mu[i] * s[i] + tot) % R for i in range(4)]
    for _ in range(4):
        s = [sbox((s[i] + rc[k + i]) % R) for i in range(4)]
        k += 4
        s = hashes._p2_external(s)
    return out


def mask_value(r, s):
    """the commitment's random mask (OP_MASK): fr.Hash(r || s, "spp-commit-mask1") of the proof's blinding factors -- a tag of its own,
    so that the mask and the commitment challenge (tag "bsb22-commitment") never come out of one random-oracle domain"""
    from .bn254 import hash_to_fr, DST_MASK
    return hash_to_fr((r % R).to_bytes(32, "big") + (s % R).to_bytes(32, "big"), DST_MASK)[0]


def solve(circ, inputs, challenge_fn, rs=(0, 0)):
    """inputs: public (without the constant) then secret values. challenge_fn(w) -> X is called at
    OP_COMMIT with the partially solved witness (committed wires are all known by then).  rs: the proof's blinding
    factors (the commitment mask is derived from them)."""
    assert len(inputs) == circ.n_inputs(), (len(inputs), circ.n_inputs())
    w = [0] * circ.n_wires
    w[0] = 1
    for i, v in enumerate(inputs):
        w[1 + i] = v % R
    prog = circ.program
    pc = 0
    while True:
        op = prog[pc]
        if op == OP_END:
            break
        if op == OP_SOLVE_C:
            k = prog[pc + 1]
            pc += 2
            wires, coeffs = circ.C.row(k)
            out = wires[-1]
            assert coeffs[-1] == 1
            rest = sum(c * w[wi] for wi, c in zip(wires[:-1], coeffs[:-1]))
            w[out] = (_dot(circ.A, k, w) * _dot(circ.B, k, w) - rest) % R
        elif op == OP_SOLVE_A:
            k = prog[pc + 1]
            pc += 2
            wires, coeffs = circ.A.row(k)
            assert len(wires) == 1 and coeffs[0] == 1
            den = _dot(circ.B, k, w)
            w[wires[0]] = _dot(circ.C, k, w) * inv(den, R) % R if den else 0
        elif op == OP_BATCH_DIV:
            k0, n = prog[pc + 1], prog[pc + 2]
            pc += 3
            for k in range(k0, k0 + n):
                wires, coeffs = circ.A.row(k)
                assert len(wires) == 1 and coeffs[0] == 1
                den = _dot(circ.B, k, w)
                w[wires[0]] = _dot(circ.C, k, w) * inv(den, R) % R if den else 0
        elif op == OP_BITS:
            h, nbits, out0 = prog[pc + 1:pc + 4]
            pc += 4
            v = _dot(circ.H, h, w)
            for i in range(nbits):
                w[out0 + i] = (v >> i) & 1
        elif op == OP_LIMBS8:
            h, n, out0 = prog[pc + 1:pc + 4]
            pc += 4
            v = _dot(circ.H, h, w)
            for i in range(n):
                w[out0 + i] = (v >> (8 * i)) & 0xFF
        elif op == OP_COUNT8:
            h0, n, out0 = prog[pc + 1:pc + 4]
            pc += 4
            for j in range(256):
                w[out0 + j] = 0
            for i in range(n):
                v = _dot(circ.H, h0 + i, w)
                if v < 256:
                    w[out0 + v] += 1
        elif op == OP_POSEIDON:
            t, h0, out0 = prog[pc + 1:pc + 4]
            pc += 4
            _poseidon_native([_dot(circ.H, h0 + i, w) for i in range(t)], out0, w)
        elif op == OP_POSEIDON2:
            h0, out0 = prog[pc + 1:pc + 3]
            pc += 3
            _poseidon2_native([_dot(circ.H, h0 + i, w) for i in range(4)], out0, w)
        elif op == OP_GRUMPKIN:
            # slopes of the fixed-base ladder acc <- acc + T_j[digit_j] (acc_0 = O), then + N
            bit0, nbits, aux_off, nl = prog[pc + 1:pc + 5]
            lam_wires = prog[pc + 5:pc + 5 + nl]
            pc += 5 + nl
            aux = circ.aux
            acc = (aux[aux_off], aux[aux_off + 1])
            bits = [w[bit0 + i] if i < nbits else 0 for i in range(256)]
            pts = []
            for j in range(64):
                d = bits[4 * j] + 2 * bits[4 * j + 1] + 4 * bits[4 * j + 2] + 8 * bits[4 * j + 3]
                o = aux_off + 4 + (j * 16 + d) * 2
                pts.append((aux[o], aux[o + 1]))
            pts.append((aux[aux_off + 2], aux[aux_off + 3]))
            for j, s_pt in enumerate(pts):
                den = (s_pt[0] - acc[0]) % R
                w[lam_wires[j]] = (s_pt[1] - acc[1]) * inv(den, R) % R if den else 0
                acc = hashes.grumpkin_add(acc, s_pt)
        elif op == OP_INV_H:
            h, out = prog[pc + 1:pc + 3]
            pc += 3
            v = _dot(circ.H, h, w)
            w[out] = inv(v, R) if v else 0
        elif op == OP_MASK:
            w[prog[pc + 1]] = mask_value(*rs)
            pc += 2
        elif op == OP_COMMIT:
            pc += 1
            w[circ.challenge_wire] = challenge_fn(w) % R
        else:
            raise ValueError("bad opcode %d at %d" % (op, pc))
    return w


def evaluate(circ, w):
    """(a, b, c) vectors over the constraints."""
    a = [_dot(circ.A, k, w) for k in range(circ.n_constraints)]
    b = [_dot(circ.B, k, w) for k in range(circ.n_constraints)]
    c = [_dot(circ.C, k, w) for k in range(circ.n_constraints)]
    return a, b, c


def first_unsatisfied(circ, w):
    a, b, c = evaluate(circ, w)
    for k in range(circ.n_constraints):
        if a[k] * b[k] % R != c[k]:
            return k
    return -1


def withdraw_inputs(kat):
    """Input vector for the withdraw circuit from a prover-params style dict
    (client/proof.helper.ts:6-21 field names)."""
    def f(v):
        return int(v, 16) if isinstance(v, str) else int(v)
    vals = [f(kat[k]) for k in ("root", "nullifier", "recipient", "amount", "wa_commitment",
                                "secret_key", "owner_x", "owner_y", "randomness", "index")]
    vals += [f(s) for s in kat["siblings"]]
    return vals

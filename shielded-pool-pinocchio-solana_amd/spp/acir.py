"""ACIR / nargo artefact ingestion: the files the reference's proving pipeline passes between `nargo execute`
and `sunspot prove` (client/proof.helper.ts:58-66: target/<name>.json, target/<name>.gz).

    load_program(path | dict)         target/<name>.json  -- {noir_version, abi, bytecode = base64(gzip(bincode Program))}
    read_witness_stack(path)          target/<name>.gz    -- gzip(bincode WitnessStack)      (what `nargo execute` writes)
    write_witness_stack(path, map)    the inverse (tests, and tools that want to hand a witness to other ACIR consumers)
    abi_input_row(program, witness)   the circuit's input wires in ABI order = the Prover.toml order of proof.helper.ts:34-50
    r1cs_rows(circuit)                the AssertZero opcodes as R1CS rows  (qM * w_a) * w_b = -(linear part)

Format (noir 1.0.0-beta.18, decoded from the reference's own shielded_pool_verifier.json): bincode with
fixed-width little-endian integers --
  Program   = u64 nFunctions . Circuit* . u64 nUnconstrained . BrilligBytecode*      (the Brillig functions are not decoded)
  Circuit   = String name . u32 current_witness_index . u64 nOpcodes . Opcode* . BTreeSet<Witness> private_parameters
              . BTreeSet<Witness> public_parameters . BTreeSet<Witness> return_values . assert_messages ...
  Opcode    = u32 tag: 0 AssertZero(Expression) | 1 BlackBoxFuncCall | 2 MemoryOp | 3 MemoryInit | 4 BrilligCall | 5 Call
  Expression= Vec<(F, W, W)> mul_terms . Vec<(F, W)> linear . F constant ;  F = u64 32 . 32 bytes big-endian ; W = u32
  BlackBox  = u32 tag: 3 RANGE {FunctionInput, u32 bits} | 8 MultiScalarMul {Vec<FunctionInput> points, Vec<FunctionInput>
              scalars, FunctionInput predicate, (W, W, W) outputs} ;  FunctionInput = u32 tag (0 constant F | 1 witness W)
  BrilligCall = u32 id . Vec<BrilligInputs> (u32 tag: 0 Single Expression | 1 Array Vec<Expression> | 2 MemoryArray u32)
              . Vec<BrilligOutputs> (u32 tag: 0 Simple W | 1 Array Vec<W>) . Option<Expression> predicate (u8 tag)
Only the variants that occur in the reference's circuits are implemented; anything else raises AcirFormatError naming the tag
and the byte offset.  This module is pure host code (no GPU): it turns files into input rows for spp_prove_batch.
"""
import base64
import gzip
import json
import struct

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class AcirFormatError(ValueError):
    pass


class _Reader:
    def __init__(self, data):
        self.d, self.o = data, 0

    def u8(self):
        v = self.d[self.o]
        self.o += 1
        return v

    def u32(self):
        v = struct.unpack_from("<I", self.d, self.o)[0]
        self.o += 4
        return v

    def u64(self):
        v = struct.unpack_from("<Q", self.d, self.o)[0]
        self.o += 8
        return v

    def field(self):
        n = self.u64()
        if n != 32:
            raise AcirFormatError("field element of %d bytes at offset %d" % (n, self.o - 8))
        v = int.from_bytes(self.d[self.o:self.o + 32], "big")
        self.o += 32
        return v

    def string(self):
        n = self.u64()
        s = self.d[self.o:self.o + n].decode()
        self.o += n
        return s

    def vec(self, item):
        return [item() for _ in range(self.u64())]


class Expression:
    __slots__ = ("mul_terms", "linear", "constant")

    def __init__(self, mul_terms, linear, constant):
        self.mul_terms, self.linear, self.constant = mul_terms, linear, constant

    def evaluate(self, w):
        s = self.constant
        for c, a, b in self.mul_terms:
            s += c * w[a] * w[b]
        for c, a in self.linear:
            s += c * w[a]
        return s % R


def _expression(r):
    mul = r.vec(lambda: (r.field(), r.u32(), r.u32()))
    lin = r.vec(lambda: (r.field(), r.u32()))
    return Expression(mul, lin, r.field())


def _function_input(r):
    tag = r.u32()
    if tag == 0:
        return ("constant", r.field())
    if tag == 1:
        return ("witness", r.u32())
    raise AcirFormatError("FunctionInput tag %d at offset %d" % (tag, r.o - 4))


def _opcode(r):
    at = r.o
    tag = r.u32()
    if tag == 0:
        return ("AssertZero", _expression(r))
    if tag == 1:
        bb = r.u32()
        if bb == 3:
            return ("RANGE", _function_input(r), r.u32())
        if bb == 8:
            points = r.vec(lambda: _function_input(r))
            scalars = r.vec(lambda: _function_input(r))
            predicate = _function_input(r)
            return ("MultiScalarMul", points, scalars, predicate, (r.u32(), r.u32(), r.u32()))
        raise AcirFormatError("BlackBoxFuncCall %d at offset %d is not used by the reference's circuits" % (bb, at))
    if tag == 4:
        r.brillig_id_offsets = getattr(r, "brillig_id_offsets", []) + [r.o]
        fid = r.u32()

        def binput():
            t = r.u32()
            if t == 0:
                return ("single", _expression(r))
            if t == 1:
                return ("array", r.vec(lambda: _expression(r)))
            if t == 2:
                return ("memory", r.u32())
            raise AcirFormatError("BrilligInputs tag %d at offset %d" % (t, r.o - 4))

        def boutput():
            t = r.u32()
            if t == 0:
                return ("simple", r.u32())
            if t == 1:
                return ("array", r.vec(r.u32))
            raise AcirFormatError("BrilligOutputs tag %d at offset %d" % (t, r.o - 4))
        inputs = r.vec(binput)
        outputs = r.vec(boutput)
        predicate = _expression(r) if r.u8() else None
        return ("BrilligCall", fid, inputs, outputs, predicate)
    raise AcirFormatError("opcode tag %d at offset %d is not used by the reference's circuits" % (tag, at))


class Circuit:
    def __init__(self, r):
        self.name = r.string()
        self.current_witness_index = r.u32()
        self.opcodes = r.vec(lambda: _opcode(r))
        self.private_parameters = r.vec(r.u32)
        self.public_parameters = r.vec(r.u32)
        self.return_values = r.vec(r.u32)
        self.end_of_parameters = r.o      # assert messages and the Brillig functions follow (not decoded)
        self.brillig_id_offsets = list(getattr(r, "brillig_id_offsets", []))   # byte offsets of the BrilligCall function ids

    def histogram(self):
        h = {}
        for op in self.opcodes:
            h[op[0]] = h.get(op[0], 0) + 1
        return h


class Program:
    def __init__(self, doc):
        self.noir_version = doc.get("noir_version")
        self.abi = doc["abi"]
        self.raw = gzip.decompress(base64.b64decode(doc["bytecode"]))
        r = _Reader(self.raw)
        n = r.u64()
        if n != 1:
            raise AcirFormatError("%d ACIR functions (the reference's programs have one)" % n)
        self.main = Circuit(r)
        # what follows the main function: its assert messages and the Brillig ("unconstrained") functions.  Their bytecode is never
        # interpreted here; it is IDENTIFIED: see brillig_kind
        import hashlib
        self.unconstrained_sha256 = hashlib.sha256(self.raw[self.main.end_of_parameters:]).hexdigest()

    def brillig_kind(self, index, fid, inputs, outputs):
        """What the Brillig function `fid` computes, for BOTH consumers (execute and to_blob): 'divmod' (a, b) -> (a // b, a % b),
        'inverse' x -> 1/x (0 for 0), 'radix' (x, n, radix) -> n little-endian digits.  The functions are identified by CONTENT: a
        program is accepted only when its unconstrained section is byte-identical to that of the reference's compiled withdraw circuit
        (noir_circuit/target/shielded_pool_verifier.json = tests/golden/reference_withdraw_acir.json, noir 1.0.0-beta.18), whose three
        helpers are known; every call site must also have the shape its helper takes.  Anything else -- another compiler version's
        helper bodies, a helper this module has no host implementation of, a call site whose shape does not fit -- is an
        AcirFormatError here rather than a silently wrong hint and an 'inputs do not satisfy the circuit' later."""
        if self.unconstrained_sha256 != REFERENCE_UNCONSTRAINED_SHA256:
            raise AcirFormatError("opcode %d: the program's unconstrained (Brillig) functions are not the reference's (sha256 %s...): Brillig "
                                  "bytecode is not interpreted, only programs carrying the helper functions of noir_circuit/target/"
                                  "shielded_pool_verifier.json are supported" % (index, self.unconstrained_sha256[:16]))
        kind = _REFERENCE_BRILLIG.get(fid)
        if kind is None:
            raise AcirFormatError("opcode %d: Brillig function %d has no host implementation" % (index, fid))
        shape = (len(inputs), [o[0] for o in outputs], all(i[0] == "single" for i in inputs))
        want = {"divmod": (2, ["simple", "simple"], True), "inverse": (1, ["simple"], True), "radix": (3, ["array"], True)}[kind]
        if shape != want:
            raise AcirFormatError("opcode %d: call of Brillig function %d (%s) with %d inputs and outputs %s does not have that helper's shape"
                                  % (index, fid, kind, shape[0], shape[1]))
        return kind

    def parameter_witnesses(self):
        """[(name, [witness indices])] in ABI order: parameters occupy consecutive witnesses from 0 (nargo's ABI encoding)."""
        out, nxt = [], 0
        for p in self.abi["parameters"]:
            n = p["type"]["length"] if p["type"]["kind"] == "array" else 1
            out.append((p["name"], list(range(nxt, nxt + n))))
            nxt += n
        return out


def load_program(src):
    return Program(src if isinstance(src, dict) else json.load(open(src)))


# ---- witness stacks (target/<name>.gz) ----
def write_witness_stack(path, witness_map, index=0):
    """gzip(bincode WitnessStack{stack: [StackItem{index, witness: BTreeMap<Witness, F>}]}) -- the layout `nargo execute` writes."""
    body = struct.pack("<Q", 1) + struct.pack("<I", index) + struct.pack("<Q", len(witness_map))
    for k in sorted(witness_map):
        body += struct.pack("<I", k) + struct.pack("<Q", 32) + int(witness_map[k] % R).to_bytes(32, "big")
    with open(path, "wb") as f:
        f.write(gzip.compress(body))


def read_witness_stack(path):
    """Returns the witness map {index: value} of the LAST stack item (the main function), as sunspot / bb consume it."""
    r = _Reader(gzip.decompress(open(path, "rb").read()))
    items = r.u64()
    if items == 0:
        raise AcirFormatError("empty witness stack")
    wmap = None
    for _ in range(items):
        r.u32()
        wmap = {}
        for _ in range(r.u64()):
            k = r.u32()
            wmap[k] = r.field()
    if r.o != len(r.d):
        raise AcirFormatError("%d trailing bytes in the witness stack" % (len(r.d) - r.o))
    return wmap


def abi_input_row(program, witness_map):
    """Input wires of the circuit (public first, ABI order) from a solved nargo witness: exactly the row spp_prove_batch takes
    for the same statement (Prover.toml order of client/proof.helper.ts:34-50)."""
    row = []
    for name, ws in program.parameter_witnesses():
        for w in ws:
            if w not in witness_map:
                raise AcirFormatError("witness %d (parameter %s) missing from the witness stack" % (w, name))
            row.append(witness_map[w] % R)
    return row


def r1cs_rows(circuit):
    """AssertZero opcodes as R1CS rows (A, B, C) with <A,w> * <B,w> = <C,w>, wires = ACIR witness indices + 1 (wire 0 = one):
    an opcode qM*w_a*w_b + sum q_i*w_i + c = 0 with ONE product is one row A = qM*w_a, B = w_b, C = -(sum q_i*w_i + c).
    Opcodes with several products are returned in `wide` (they need helper wires).  How sunspot lowers ACIR to gnark's R1CS."""
    rows, wide = [], []
    for i, op in enumerate(circuit.opcodes):
        if op[0] != "AssertZero":
            continue
        e = op[1]
        neg = [((-c) % R, w + 1) for c, w in e.linear]
        if e.constant:
            neg.append(((-e.constant) % R, 0))
        if len(e.mul_terms) == 0:
            rows.append(([], [], neg))      # purely linear: 0 * 0 = <C,w>
        elif len(e.mul_terms) == 1:
            c, a, b = e.mul_terms[0]
            rows.append(([(c, a + 1)], [(1, b + 1)], neg))
        else:
            wide.append(i)
    return rows, wide


# ---- witness generation for the reference's ACIR (what `nargo execute` does, client/proof.helper.ts:55) ----
# Grumpkin (Noir's embedded curve): y^2 = x^3 - 17 over BN254 Fr (client/merkle.ts:47-74)
_GK_GEN = (1, 17631683881184975370165255887551781615748388533673675138860)


def _gk_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % R == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, R) % R
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, R) % R
    x = (lam * lam - a[0] - b[0]) % R
    return (x, (lam * (a[0] - x) - a[1]) % R)


def _gk_mul(pt, k):
    acc = None
    while k:
        if k & 1:
            acc = _gk_add(acc, pt)
        pt = _gk_add(pt, pt)
        k >>= 1
    return acc


# sha256 of everything that follows the main function's parameters in the reference's compiled program (assert messages + the three
# Brillig functions) and what those functions compute (ids as the program numbers them)
REFERENCE_UNCONSTRAINED_SHA256 = "ebff9086f0484343725d1dc91d3cc5b1f88edd0018d64bcc4a38c136536d95ef"
_REFERENCE_BRILLIG = {0: "divmod", 1: "inverse", 2: "radix"}


class UnsatisfiedConstraint(ValueError):
    def __init__(self, index, what):
        super().__init__("ACIR opcode %d: %s" % (index, what))
        self.opcode_index = index


def _brillig(kind, args, n_out):
    """Host implementations of the reference's unconstrained helpers, by kind (Program.brillig_kind): the constraints that follow
    each call pin the results.  Brillig bytecode is not interpreted."""
    if kind == "divmod":                      # field -> 128-bit limbs
        a, b = args[0], args[1]
        if b == 0:
            raise AcirFormatError("Brillig divmod by zero")
        return [a // b, a % b]
    if kind == "inverse":                     # is-zero and != gadgets
        return [pow(args[0], -1, R) if args[0] % R else 0]
    if kind == "radix":                       # (value, number of limbs, radix): index -> 16 path bits
        x, radix = args[0], args[2]
        out = []
        for _ in range(n_out):
            out.append(x % radix)
            x //= radix
        return out
    raise AcirFormatError("Brillig helper %r has no host implementation" % (kind,))


def execute(program, input_row):
    """Solves every witness of the program's main function from the ABI input row (public first): ACVM's loop -- an
    AssertZero with one unknown witness defines it, with none it is checked; RANGE is checked; MultiScalarMul is the Grumpkin
    fixed-base multiplication; BrilligCall runs the host helper.  Returns {witness: value}; raises UnsatisfiedConstraint with
    the opcode index when the inputs do not satisfy the circuit (the `nargo execute` failure of proof.helper.ts:55)."""
    c = program.main
    w = {}
    flat = [x for _, ws in program.parameter_witnesses() for x in ws]
    if len(flat) != len(input_row):
        raise ValueError("expected %d inputs" % len(flat))
    for k, v in zip(flat, input_row):
        w[k] = int(v) % R

    def value(e):
        s = e.constant
        for cf, a, b in e.mul_terms:
            s += cf * w[a] * w[b]
        for cf, a in e.linear:
            s += cf * w[a]
        return s % R

    for idx, op in enumerate(c.opcodes):
        kind = op[0]
        if kind == "AssertZero":
            e = op[1]
            unknown = set()
            for _, a, b in e.mul_terms:
                unknown.update(x for x in (a, b) if x not in w)
            unknown.update(a for _, a in e.linear if a not in w)
            if not unknown:
                if value(e) != 0:
                    raise UnsatisfiedConstraint(idx, "assertion failed")
                continue
            if len(unknown) != 1:
                raise AcirFormatError("opcode %d has %d unknown witnesses" % (idx, len(unknown)))
            u = unknown.pop()
            coeff, rest = 0, e.constant
            for cf, a, b in e.mul_terms:
                if a == u and b == u:
                    raise AcirFormatError("opcode %d is quadratic in its unknown" % idx)
                if a == u:
                    coeff += cf * w[b]
                elif b == u:
                    coeff += cf * w[a]
                else:
                    rest += cf * w[a] * w[b]
            for cf, a in e.linear:
                if a == u:
                    coeff += cf
                else:
                    rest += cf * w[a]
            coeff %= R
            if coeff == 0:
                if rest % R:
                    raise UnsatisfiedConstraint(idx, "assertion failed")
                w[u] = 0
            else:
                w[u] = (-rest) * pow(coeff, -1, R) % R
        elif kind == "RANGE":
            _, (t, v), bits = op
            val = v if t == "constant" else w[v]
            if val >> bits:
                raise UnsatisfiedConstraint(idx, "value does not fit %d bits" % bits)
        elif kind == "MultiScalarMul":
            _, points, scalars, predicate, outs = op
            get = lambda fi: fi[1] if fi[0] == "constant" else w[fi[1]]
            acc = None
            for k in range(len(points) // 3):
                px, py, inf = (get(points[3 * k + i]) for i in range(3))
                lo, hi = get(scalars[2 * k]), get(scalars[2 * k + 1])
                if not inf:
                    acc = _gk_add(acc, _gk_mul((px, py), lo + (hi << 128)))
            w[outs[0]], w[outs[1]], w[outs[2]] = (0, 0, 1) if acc is None else (acc[0], acc[1], 0)
        elif kind == "BrilligCall":
            _, fid, inputs, outputs, predicate = op
            flat_out = [x for o in outputs for x in ([o[1]] if o[0] == "simple" else o[1])]
            program.brillig_kind(idx, fid, inputs, outputs)   # identify the helper even when the predicate is off
            if predicate is not None and value(predicate) == 0:
                for x in flat_out:
                    w[x] = 0
                continue
            args = []
            for i in inputs:
                if i[0] == "single":
                    args.append(value(i[1]))
                elif i[0] == "array":
                    args.extend(value(x) for x in i[1])
                else:
                    raise AcirFormatError("opcode %d: Brillig memory inputs are not supported" % idx)
            for x, v in zip(flat_out, _brillig(program.brillig_kind(idx, fid, inputs, outputs), args, len(flat_out))):
                w[x] = v % R
    return w


# ---- ACIR -> R1CS (`sunspot compile <acir>`): flat opcode blob for csrc/circuit_acir.cpp ----
def _f32(v):
    return int(v % R).to_bytes(32, "little")


def _expr_bytes(e):
    out = struct.pack("<I", len(e.mul_terms))
    for c, a, b in e.mul_terms:
        out += _f32(c) + struct.pack("<II", a, b)
    out += struct.pack("<I", len(e.linear))
    for c, a in e.linear:
        out += _f32(c) + struct.pack("<I", a)
    return out + _f32(e.constant)


def to_blob(program):
    """The decoded opcode list in the flat layout csrc/circuit_acir.cpp reads (see its header).  Brillig calls are identified by
    Program.brillig_kind (content of the unconstrained section + call shape; the same classifier execute() uses) -- the functions
    themselves are unconstrained helpers whose outputs the following opcodes constrain:
      (expr, constant power of two) -> (q, r)            quotient / remainder    (field -> 128-bit limbs)
      (expr) -> (x)                                       inverse, 0 for 0        (is-zero / != gadgets)
      (expr, constant n, constant 2) -> [n witnesses]     little-endian bits      (index -> path bits)
    and MultiScalarMul must be over the Grumpkin generator with an always-true predicate (fixed_base_scalar_mul).
    Anything else raises AcirFormatError."""
    c = program.main
    pub, sec = c.public_parameters, c.private_parameters
    if pub != list(range(len(pub))) or sec != list(range(len(pub), len(pub) + len(sec))):
        raise AcirFormatError("parameters must occupy witnesses 0..n-1, public ones first")
    if c.return_values:
        raise AcirFormatError("return values are not supported")
    ops = []
    for i, op in enumerate(c.opcodes):
        k = op[0]
        if k == "AssertZero":
            ops.append(struct.pack("<I", 0) + _expr_bytes(op[1]))
        elif k == "RANGE":
            _, (t, w), bits = op
            if t != "witness":
                raise AcirFormatError("opcode %d: RANGE on a constant" % i)
            ops.append(struct.pack("<III", 1, w, bits))
        elif k == "MultiScalarMul":
            _, points, scalars, predicate, outs = op
            if points != [("constant", _GK_GEN[0]), ("constant", _GK_GEN[1]), ("constant", 0)] or predicate != ("constant", 1) \
                    or len(scalars) != 2 or any(t != "witness" for t, _ in scalars):
                raise AcirFormatError("opcode %d: only fixed-base MultiScalarMul over the Grumpkin generator is supported" % i)
            ops.append(struct.pack("<IIIIII", 2, scalars[0][1], scalars[1][1], outs[0], outs[1], outs[2]))
        elif k == "BrilligCall":
            _, fid, inputs, outputs, predicate = op
            if predicate is not None or any(t != "single" for t, _ in inputs):
                raise AcirFormatError("opcode %d: predicated / array-input Brillig calls are not supported" % i)
            exprs = [e for _, e in inputs]
            const = lambda e: not e.mul_terms and not e.linear
            kind = program.brillig_kind(i, fid, inputs, outputs)     # the same identification execute() uses
            if kind == "divmod" and const(exprs[1]) and exprs[1].constant and exprs[1].constant & (exprs[1].constant - 1) == 0:
                ops.append(struct.pack("<I", 3) + _expr_bytes(exprs[0]) + struct.pack("<III", exprs[1].constant.bit_length() - 1, outputs[0][1], outputs[1][1]))
            elif kind == "inverse":
                ops.append(struct.pack("<I", 4) + _expr_bytes(exprs[0]) + struct.pack("<I", outputs[0][1]))
            elif kind == "radix" and const(exprs[1]) and const(exprs[2]) and exprs[2].constant == 2 and exprs[1].constant == len(outputs[0][1]):
                outs = outputs[0][1]
                ops.append(struct.pack("<I", 5) + _expr_bytes(exprs[0]) + struct.pack("<I%dI" % len(outs), len(outs), *outs))
            else:
                raise AcirFormatError("opcode %d: Brillig %s call with arguments the lowering does not cover (divisor not a constant power "
                                      "of two / radix not 2)" % (i, kind))
        else:
            raise AcirFormatError("opcode %d: %s is not supported" % (i, k))
    return struct.pack("<IIII", 0x31524341, len(pub), len(sec), len(ops)) + b"".join(ops)


def compile_to_sppc(program_or_path, out_path, circuit_id=None):
    """`sunspot compile <acir>`: writes the SPPC container for the program, returns nbConstraints.  circuit_id defaults to
    SPP_CIRCUIT_WITHDRAW when the program has the withdraw circuit's ABI (5 public + 21 private inputs), else the generic id."""
    import ctypes
    from .lib import load_library, check
    prog = program_or_path if isinstance(program_or_path, Program) else load_program(program_or_path)
    blob = to_blob(prog)
    if circuit_id is None:
        circuit_id = 1 if (len(prog.main.public_parameters), len(prog.main.private_parameters)) == (5, 21) else 5
    n = ctypes.c_uint32(0)
    check(load_library().spp_circuit_build_acir(blob, len(blob), int(circuit_id), out_path.encode(), ctypes.byref(n)))
    return n.value

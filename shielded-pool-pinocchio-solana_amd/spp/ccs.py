"""Reader for gnark's R1CS container `.ccs` (noir_circuit/target/shielded_pool_verifier.ccs, what `sunspot
compile` writes and `sunspot prove <acir> <witness> <ccs> <pk>` reads, client/proof.helper.ts:64).

Layout (gnark 0.14, decoded from the reference's own file):
  [0x00] u64 LE  = file size - 32          [0x08] u64 x3 = 0, 14, 0 (opaque here)
  [0x20] u64 LE x4 = byte lengths of: levels stream, instructions stream, calldata stream, CBOR body
  [0x40] the three streams (compressed uint32 sequences), then the CBOR map
         {Type, GnarkVersion, ScalarField, NbConstraints, NbInternalVariables, Public[], Secret[], Blueprints[],
          CommitmentInfo, MHintsDependencies, GkrInfo, Logs, DebugInfo, MDebug, SymbolTable ...}
  then   u64 LE count . count x 32 B   the coefficient table, little-endian Montgomery limbs (entry 1 = R mod r)
This module decodes all of it: header, CBOR body, coefficient table and the three integer streams, i.e. the reference's
whole constraint system -- 12 493 instructions (12 452 R1C rows + 41 hint calls of 9 kinds) in 657 solver levels.

Stream formats (third-party github.com/ronanh/intcomp as called by gnark 0.14; neither is in the reference tree, so the
layout below was worked out from the reference's own file and is pinned by the invariants tests/test_acir_ccs.py checks:
every instruction id exactly once over the levels, constraint offsets ending at NbConstraints, wire offsets ending at the
wire count, calldata offsets equal to the running sum of the calldata record lengths):
  levels        u64 nLevels, then per level: u64 nWords, one uint32 chunk (sorted instruction ids)
  instructions  four chunks, each u64 nWords + chunk: blueprint id, constraint offset, wire offset (uint32 chunks) and the
                calldata offset (uint64 chunk, nWords counts 8-byte words)
  calldata      u64 count, then `count` LEB128 values: per instruction [len, ...]
  chunk         [bin-packed section] [var-byte section], both optional
    bin-packed  count (multiple of 128; 256 for uint64), section length in words, first value, then per block of 128 (256)
                a header word with four bytes (most significant first), one per group of 32 (64) values: bit 7 = zig-zag
                deltas, bits 0-6 = width b; followed by the groups' deltas packed LSB-first in b words each
    var-byte    count, section length, the values' deltas (the first one from zero, not from the packed section) as LEB128
                bytes stored most-significant-byte-first in each word, padded with 0x80; one trailing word = the length
R1C calldata: [len, nL, nR, nO, (coefficient id, wire id) x (nL + nR + nO)]; hint calldata: [len, hint id, nInputs,
{nTerms, (coefficient id, wire id | 0xffffffff = constant) x nTerms} x nInputs, first output wire, end].

`solve_partial` runs gnark's solver loop over the decoded system on a witness of the secret wires (the ACIR witnesses):
an R1C row with one unknown wire defines it, the standard hints are restated here; the three hints whose code is not
available (emulated.mulHint and Sunspot's two Grumpkin scalar-decomposition hints, one call each) leave their outputs and
whatever depends on them unknown.  Every row that can be evaluated is checked.
"""
import struct

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MONT_R_INV = pow(1 << 256, -1, R)


class Tagged:
    def __init__(self, tag, value):
        self.tag, self.value = tag, value

    def __repr__(self):
        return "Tagged(%d, %r)" % (self.tag, self.value)


def cbor_decode(data, off=0):
    """Minimal CBOR (RFC 8949) decoder: the major types gnark's encoder emits.  Returns (value, next offset)."""
    ib = data[off]
    major, info = ib >> 5, ib & 31
    off += 1
    if info < 24:
        arg = info
    elif info == 24:
        arg = data[off]; off += 1
    elif info == 25:
        arg = struct.unpack_from(">H", data, off)[0]; off += 2
    elif info == 26:
        arg = struct.unpack_from(">I", data, off)[0]; off += 4
    elif info == 27:
        arg = struct.unpack_from(">Q", data, off)[0]; off += 8
    elif info == 31 and major in (2, 3, 4, 5):
        arg = None          # indefinite length
    else:
        raise ValueError("CBOR additional info %d at %d" % (info, off - 1))
    if major == 0:
        return arg, off
    if major == 1:
        return -1 - arg, off
    if major in (2, 3):
        if arg is None:
            chunks = []
            while data[off] != 0xFF:
                v, off = cbor_decode(data, off)
                chunks.append(v)
            v = (b"" if major == 2 else "").join(chunks)
            return v, off + 1
        raw = bytes(data[off:off + arg])
        return (raw if major == 2 else raw.decode()), off + arg
    if major == 4:
        out = []
        if arg is None:
            while data[off] != 0xFF:
                v, off = cbor_decode(data, off)
                out.append(v)
            return out, off + 1
        for _ in range(arg):
            v, off = cbor_decode(data, off)
            out.append(v)
        return out, off
    if major == 5:
        out = {}
        n = arg
        while (n is None and data[off] != 0xFF) or (n is not None and n > 0):
            k, off = cbor_decode(data, off)
            v, off = cbor_decode(data, off)
            out[k if not isinstance(k, (list, dict)) else repr(k)] = v
            if n is not None:
                n -= 1
        return out, (off + 1 if arg is None else off)
    if major == 6:
        v, off = cbor_decode(data, off)
        return Tagged(arg, v), off
    # major 7: simple values and floats
    if info == 20:
        return False, off
    if info == 21:
        return True, off
    if info in (22, 23):
        return None, off
    if info == 25:
        return struct.unpack(">e", struct.pack(">H", arg))[0], off
    if info == 26:
        return struct.unpack(">f", struct.pack(">I", arg))[0], off
    if info == 27:
        return struct.unpack(">d", struct.pack(">Q", arg))[0], off
    return ("simple", arg), off


class Ccs:
    pass


def load_ccs(path):
    d = open(path, "rb").read()
    c = Ccs()
    c.file_size = len(d)
    c.header0 = struct.unpack_from("<Q", d, 0)[0]
    if c.header0 != len(d) - 32:
        raise ValueError("not a gnark .ccs container: header %d, file size %d" % (c.header0, len(d)))
    c.header_opaque = struct.unpack_from("<3Q", d, 8)
    c.levels_len, c.instructions_len, c.calldata_len, c.cbor_len = struct.unpack_from("<4Q", d, 0x20)
    o = 0x40
    c.levels_raw = d[o:o + c.levels_len]; o += c.levels_len
    c.instructions_raw = d[o:o + c.instructions_len]; o += c.instructions_len
    c.calldata_raw = d[o:o + c.calldata_len]; o += c.calldata_len
    c.cbor_offset = o
    body, end = cbor_decode(d, o)
    if end != o + c.cbor_len:
        raise ValueError("CBOR body ends at %d, header says %d" % (end, o + c.cbor_len))
    c.meta = body
    o = end
    n = struct.unpack_from("<Q", d, o)[0]
    o += 8
    c.coeff_offset = o - 8
    c.coefficients_mont = [int.from_bytes(d[o + 32 * i:o + 32 * i + 32], "little") for i in range(n)]
    o += 32 * n
    c.trailing = len(d) - o
    c.n_constraints = body.get("NbConstraints")
    c.n_internal = body.get("NbInternalVariables")
    c.public = body.get("Public")
    c.secret = body.get("Secret")
    return c


def coefficient(c, i):
    """canonical value of coefficient table entry i (stored as a * 2^256 mod r)"""
    return c.coefficients_mont[i] * MONT_R_INV % R


# ---------------------------------------------------------------------------------------------------------------------
# the three integer streams
# ---------------------------------------------------------------------------------------------------------------------
def _unzigzag(v):
    return (v >> 1) ^ -(v & 1)


def _leb128(bs, i):
    v = sh = 0
    while True:
        b = bs[i]
        i += 1
        v |= (b & 0x7F) << sh
        sh += 7
        if not b & 0x80:
            return v, i


def decode_chunk(words, bits=32):
    """One intcomp chunk (list of 32- or 64-bit words) -> list of values.  See the module docstring for the layout."""
    blk = 128 if bits == 32 else 256
    grp = blk // 4
    mask_v = (1 << bits) - 1
    half = 0xFFFFFFFF
    pos, out = 0, []
    first = words[0] & half if bits == 64 else words[0]
    if words and first >= blk and first % blk == 0:
        if bits == 64:
            count, length, prev, p = words[0] & half, words[0] >> 32, words[1], 2
        else:
            count, length, prev, p = words[0], words[1], words[2], 3
        for _ in range(count // blk):
            header = words[p]
            p += 1
            for sh in (24, 16, 8, 0):
                hb = (header >> sh) & 0xFF
                nb, zz = hb & 0x7F, hb >> 7
                nw = nb * grp // bits
                acc = 0
                for i in range(nw):
                    acc |= words[p + i] << (bits * i)
                p += nw
                m = (1 << nb) - 1
                for k in range(grp):
                    d = (acc >> (k * nb)) & m
                    prev = (prev + (_unzigzag(d) if zz else d)) & mask_v
                    out.append(prev)
        if p != length:
            raise ValueError("bin-packed section ends at word %d, header says %d" % (p, length))
        pos = p
    if pos < len(words):
        if bits == 64:
            count, length = words[pos] & half, words[pos] >> 32
        else:
            count, length = words[pos], words[pos + 1]
        data0 = pos + (1 if bits == 64 else 2)
        bs = b"".join(struct.pack(">Q" if bits == 64 else ">I", x) for x in words[data0:pos + length])
        i, prev = 0, 0
        for _ in range(count):
            d, i = _leb128(bs, i)
            prev = (prev + d) & mask_v
            out.append(prev)
        if any(b != 0x80 for b in bs[i:]) or pos + length + 1 != len(words) or words[pos + length] != length:
            raise ValueError("var-byte section: bad padding or trailer")
    return out


def _u32s(b):
    return list(struct.unpack_from("<%dI" % (len(b) // 4), b, 0))


def decode_levels(c):
    """list of levels, each the sorted instruction ids that gnark's solver may run in parallel"""
    w = _u32s(c.levels_raw)
    n = w[0] | w[1] << 32
    pos, levels = 2, []
    for _ in range(n):
        k = w[pos] | w[pos + 1] << 32
        levels.append(decode_chunk(w[pos + 2:pos + 2 + k]))
        pos += 2 + k
    if pos != len(w):
        raise ValueError("levels stream: %d words left" % (len(w) - pos))
    return levels


def decode_instructions(c):
    """(blueprint id, constraint offset, wire offset, calldata offset) columns, one entry per instruction"""
    w = _u32s(c.instructions_raw)
    pos, cols = 0, []
    for _ in range(3):
        k = w[pos] | w[pos + 1] << 32
        cols.append(decode_chunk(w[pos + 2:pos + 2 + k]))
        pos += 2 + k
    k = w[pos] | w[pos + 1] << 32
    q = [w[pos + 2 + 2 * i] | w[pos + 3 + 2 * i] << 32 for i in range(k)]
    cols.append(decode_chunk(q, 64))
    if pos + 2 + 2 * k != len(w) or len({len(x) for x in cols}) != 1:
        raise ValueError("instructions stream: inconsistent columns")
    return cols


def decode_calldata(c):
    n = struct.unpack_from("<Q", c.calldata_raw, 0)[0]
    out, i = [], 8
    for _ in range(n):
        v, i = _leb128(c.calldata_raw, i)
        out.append(v)
    if i != len(c.calldata_raw):
        raise ValueError("calldata stream: %d bytes left" % (len(c.calldata_raw) - i))
    return out


CONST_WIRE = 0xFFFFFFFF
BLUEPRINT_HINT, BLUEPRINT_R1C = 0, 1


class System:
    """the decoded constraint system: rows[k] = (L, R, O) term lists [(coefficient, wire)], hints = [(instruction, hint id, name,
    inputs [[(coefficient, wire | CONST_WIRE)]], first output wire, end)], levels, per-instruction columns"""


def decode_system(c):
    s = System()
    s.levels = decode_levels(c)
    s.blueprint, s.constraint_offset, s.wire_offset, s.calldata_offset = decode_instructions(c)
    cd = s.calldata = decode_calldata(c)
    coef = [v * MONT_R_INV % R for v in c.coefficients_mont]
    names = c.meta["MHintsDependencies"]
    s.rows, s.row_instruction, s.hints = [], [], []
    s.kind = []          # per instruction: ("r1c", row index) or ("hint", index into hints)
    for k, (bp, off) in enumerate(zip(s.blueprint, s.calldata_offset)):
        ln = cd[off]
        if bp == BLUEPRINT_R1C:
            nl, nr, no = cd[off + 1:off + 4]
            if ln != 4 + 2 * (nl + nr + no) or s.constraint_offset[k] != len(s.rows):
                raise ValueError("instruction %d: bad R1C record" % k)
            p = off + 4
            terms = [(coef[cd[p + 2 * i]], cd[p + 2 * i + 1]) for i in range(nl + nr + no)]
            s.rows.append((terms[:nl], terms[nl:nl + nr], terms[nl + nr:]))
            s.row_instruction.append(k)
            s.kind.append(("r1c", len(s.rows) - 1))
        elif bp == BLUEPRINT_HINT:
            hid, nin = cd[off + 1], cd[off + 2]
            p, ins = off + 3, []
            for _ in range(nin):
                nt = cd[p]
                ins.append([(coef[cd[p + 1 + 2 * i]], cd[p + 2 + 2 * i]) for i in range(nt)])
                p += 1 + 2 * nt
            o0, o1 = cd[p], cd[p + 1]
            if p + 2 != off + ln or o1 != s.wire_offset[k]:      # the wire-offset column holds the offset AFTER the instruction
                raise ValueError("instruction %d: bad hint record" % k)
            s.hints.append((k, hid, names[hid], ins, o0, o1))
            s.kind.append(("hint", len(s.hints) - 1))
        else:
            raise ValueError("instruction %d: blueprint %d" % (k, bp))
    s.n_wires = len(c.public) + len(c.secret) + c.n_internal
    return s


# ---------------------------------------------------------------------------------------------------------------------
# gnark's solver loop on the decoded system
# ---------------------------------------------------------------------------------------------------------------------
UNSUPPORTED_HINTS = ()

# Grumpkin's group order is the BN254 base field modulus; LAMBDA is the eigenvalue of its endomorphism (a cube root of unity mod Q).
Q_BASE = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
GLV_BITS = 127
GLV_LAMBDA = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe


def _limbs(v, n, bits):
    return [(v >> (bits * i)) & ((1 << bits) - 1) for i in range(n)]


def glv_split(s, lam, q=Q_BASE, bits=GLV_BITS):
    """(s1, s2) with 0 <= s1, s2 < 2^bits and s1 - lam * s2 = s (mod q) -- what the rows after `sw-grumpkin.decomposeScalar` demand of
    its outputs (s_limbs + lam * s2 - s1 = 0 mod q through emulated.mulHint, both halves recomposed from `bits` bits).  Sunspot's own
    hint is not in the reference tree; ANY pair with these properties satisfies the system, so this one is found from scratch: a
    reduced basis of the lattice {(x, y): x = lam * y mod q} by the extended Euclid on (q, lam), Babai rounding of (s, 0), then the
    neighbouring lattice points until both coordinates are in range (the box holds 2^254 / q = 5.3 lattice points on average)."""
    r0, r1, t0, t1 = q, lam % q, 0, 1
    rows = []
    while r1:
        k = r0 // r1
        r0, r1, t0, t1 = r1, r0 - k * r1, t1, t0 - k * t1
        rows.append((r0, t0))                      # r0 = t0 * lam (mod q)
    lim = 1 << (q.bit_length() // 2)
    i = next(j for j, (r, _) in enumerate(rows) if r < lim)
    v1, v2 = rows[i], rows[i + 1]                  # consecutive remainders: |det| = q, both vectors of norm ~ sqrt(q)
    det = v1[0] * v2[1] - v2[0] * v1[1]
    if det < 0:
        v1, v2, det = v2, v1, -det
    # (s, 0) = b1 v1 + b2 v2 over the rationals, rounded to the nearest integers
    b1 = (2 * s * v2[1] + det) // (2 * det)
    b2 = (-2 * s * v1[1] + det) // (2 * det)
    for radius in range(0, 6):
        for i1 in range(-radius, radius + 1):
            for i2 in range(-radius, radius + 1):
                if max(abs(i1), abs(i2)) != radius:
                    continue
                x = s - (b1 + i1) * v1[0] - (b2 + i2) * v2[0]
                y = -(b1 + i1) * v1[1] - (b2 + i2) * v2[1]
                if 0 <= x < 1 << bits and 0 <= y < 1 << bits:
                    assert (x - lam * y - s) % q == 0
                    return x, y
    raise ValueError("no decomposition in range")


def _emulated_mul_hint(ins, n_out):
    """gnark std/math/emulated mulHint as this system calls it: inputs [limb bits, modulus limbs n, limbs of a, limbs of the
    quotient, modulus limbs..., a limbs..., b limbs...]; outputs quotient limbs, remainder limbs (n), carry limbs.  The rows that
    consume them (decoded from the .ccs) check  a(X) b(X) = k(X) p(X) + r(X) + (2^bits - X) c(X)  at a random X, so: k = a*b div p,
    r = a*b mod p as limbs, and c by dividing the difference by (2^bits - X) coefficient by coefficient (field elements)."""
    bits, n, na, nq = ins[0], ins[1], ins[2], ins[3]
    p_l = ins[4:4 + n]
    a_l = ins[4 + n:4 + n + na]
    b_l = ins[4 + n + na:]
    B = 1 << bits
    val = lambda ls: sum(v << (bits * i) for i, v in enumerate(ls))
    pv, av, bv = val(p_l), val(a_l), val(b_l)
    quo, rem = divmod(av * bv, pv)
    k_l, r_l = _limbs(quo, nq, bits), _limbs(rem, n, bits)
    if quo >> (bits * nq):
        raise ValueError("quotient does not fit")
    ncarry = n_out - nq - n
    d = [0] * (ncarry + 1)
    for i, x in enumerate(a_l):
        for j, y in enumerate(b_l):
            d[i + j] += x * y
    for i, x in enumerate(k_l):
        for j, y in enumerate(p_l):
            d[i + j] -= x * y
    for i, x in enumerate(r_l):
        d[i] -= x
    carries, c = [], 0
    for i in range(ncarry):                          # d_i = B c_i - c_{i-1}
        num = d[i] + c
        if num % B:
            raise ValueError("carry %d is not integral" % i)
        c = num // B
        carries.append(c % R)
    if d[ncarry] + c != 0:
        raise ValueError("the product identity does not close")
    return k_l + r_l + carries


def _hint_outputs(name, ins, n_out, challenge, randomizer=0x5EED):
    """The standard gnark hints of this system, restated from their documented behaviour (gnark 0.14: std/rangecheck,
    std/math/bits, constraint/solver, std/internal/logderivarg, internal/hints, frontend/cs)."""
    short = name.rsplit("/", 1)[-1]
    if short == "bits.nBits":                               # bits of the value, least significant first
        return [(ins[0] >> i) & 1 for i in range(n_out)]
    if short == "rangecheck.DecomposeHint":                 # inputs: total width, limb width, value -> limbs
        width = ins[1]
        return [(ins[2] >> (width * i)) & ((1 << width) - 1) for i in range(n_out)]
    if short == "solver.InvZeroHint":                       # 1/x, or 0 for 0
        return [pow(v, -1, R) if v else 0 for v in ins][:n_out]
    if short == "logderivarg.countHint":                    # inputs: table size, row width, table rows, queries -> multiplicities
        nb_table, nb_vals = ins[0], ins[1]
        table = {}
        flat = ins[2:]
        for i in range(nb_table):
            table.setdefault(tuple(flat[i * nb_vals:(i + 1) * nb_vals]), i)
        counts = [0] * nb_table
        q = flat[nb_table * nb_vals:]
        for i in range(len(q) // nb_vals):
            counts[table[tuple(q[i * nb_vals:(i + 1) * nb_vals])]] += 1
        return counts[:n_out]
    if short == "sw-grumpkin.decompose":                    # the scalar as 64-bit limbs
        return _limbs(ins[0], n_out, 64)
    if short == "sw-grumpkin.decomposeScalar":              # inputs: six layout words, the scalar, limb count, limb bits, modulus limbs
        nl, bits = ins[7], ins[8]
        q = sum(v << (bits * i) for i, v in enumerate(ins[9:9 + nl]))
        s1, s2 = glv_split(ins[6], GLV_LAMBDA, q)
        return _limbs(s1, nl, bits) + _limbs(s2, nl, bits)
    if short == "emulated.mulHint":
        return _emulated_mul_hint(ins, n_out)
    if short == "hints.Randomize":                          # the random mask api.Commit adds to the committed wires (hiding): no row
        return [randomizer] * n_out                         # uses it, and being committed it must not move with the challenge
    if short == "cs.Bsb22CommitmentComputePlaceholder":     # the commitment challenge: any value satisfies the rows
        return [challenge] * n_out
    return None


def solve_partial(system, c, public_inputs, secret_by_name, challenge=0x5EED, challenge_fn=None, randomizer=None):
    """-> (wires list with None for unknown, stats dict).  public_inputs: the five values after the constant wire;
    secret_by_name: {"__witness_<i>": value}.  challenge_fn(wires) -> the commitment challenge (what gnark's
    Bsb22CommitmentComputePlaceholder hint returns: the hash of the Pedersen commitment to the committed wires, which only the
    holder of the proving key can compute -- spp_commitment_challenge); without it the fixed `challenge` is used, which satisfies
    the rows just as well but not a verifier.  randomizer: the value of gnark's hints.Randomize (the commitment's hiding mask);
    None draws a fresh one from the OS."""
    if randomizer is None:
        import secrets
        randomizer = secrets.randbelow(R)
    w = [None] * system.n_wires
    w[0] = 1
    for i, v in enumerate(public_inputs):
        w[1 + i] = v % R
    base = len(c.public)
    for i, nm in enumerate(c.secret):
        w[base + i] = secret_by_name[nm] % R
    stats = {"rows_checked": 0, "rows_solved": 0, "rows_unsatisfied": [], "rows_skipped": 0, "hints_run": 0, "hints_skipped": []}

    def lin(terms):
        s, unknown = 0, []
        for cf, wi in terms:
            if w[wi] is None:
                unknown.append((cf, wi))
            else:
                s += cf * w[wi]
        return s % R, unknown

    for level in system.levels:
        for k in level:
            kind, idx = system.kind[k]
            if kind == "hint":
                _, _, name, ins, o0, o1 = system.hints[idx]
                vals, ok = [], True
                for terms in ins:
                    s = 0
                    for cf, wi in terms:
                        if wi == CONST_WIRE:
                            s += cf
                        elif w[wi] is None:
                            ok = False
                        else:
                            s += cf * w[wi]
                    vals.append(s % R)
                if ok and challenge_fn is not None and name.endswith("Bsb22CommitmentComputePlaceholder"):
                    outs = [challenge_fn(w) % R]
                else:
                    outs = _hint_outputs(name, vals, o1 - o0, challenge, randomizer) if ok else None
                if outs is None:
                    stats["hints_skipped"].append((k, name.rsplit("/", 1)[-1]))
                    continue
                for i, v in enumerate(outs):
                    w[o0 + i] = v % R
                stats["hints_run"] += 1
                continue
            L, Rr, O = system.rows[idx]
            (l, ul), (r, ur), (o, uo) = lin(L), lin(Rr), lin(O)
            unknown = {wi for _, wi in ul + ur + uo}
            if not unknown:
                stats["rows_checked"] += 1
                if l * r % R != o:
                    stats["rows_unsatisfied"].append(idx)
                continue
            if len(unknown) > 1:
                stats["rows_skipped"] += 1
                continue
            wi = next(iter(unknown))
            cf = lambda u: sum(c_ for c_, x in u if x == wi) % R
            if uo and not ul and not ur:          # l * r = o0 + cf * x
                if cf(uo) == 0:
                    stats["rows_skipped"] += 1
                    continue
                w[wi] = (l * r - o) * pow(cf(uo), -1, R) % R
            elif ul and not ur and not uo:        # (l0 + cf * x) * r = o
                if r == 0 or cf(ul) == 0:
                    stats["rows_skipped"] += 1
                    continue
                w[wi] = (o * pow(r, -1, R) - l) * pow(cf(ul), -1, R) % R
            elif ur and not ul and not uo:
                if l == 0 or cf(ur) == 0:
                    stats["rows_skipped"] += 1
                    continue
                w[wi] = (o * pow(l, -1, R) - r) * pow(cf(ur), -1, R) % R
            else:                                  # the unknown on two sides: not a shape gnark's solver accepts either
                stats["rows_skipped"] += 1
                continue
            stats["rows_solved"] += 1
    stats["wires_known"] = sum(v is not None for v in w)
    return w, stats


# ---------------------------------------------------------------------------------------------------------------------
# the decoded system as a circuit of this repository's prover
# ---------------------------------------------------------------------------------------------------------------------
SPPC_MAGIC, SPPC_VERSION = 0x43505053, 2
CIRCUIT_CCS = 6
OP_END, OP_COMMIT = 0, 9


def challenge_wire(system):
    return [h for h in system.hints if h[2].endswith("Bsb22CommitmentComputePlaceholder")][0][4]


def to_sppc(system, c, path):
    """Writes the reference's R1CS as an SPPC container (csrc/circuit.cpp) so that `spp setup` / `spp_prove_batch` -- and the
    oracle -- take it like any other circuit: same rows, same wire numbering (constant, 5 public, 6 184 ACIR witnesses, 6 749
    internal), same coefficient table, the same 490 committed wires and commitment wire.  Every wire after the public ones is
    declared an INPUT and the solver program is the commitment step alone: the witness is completed on the host by gnark's own
    solver loop (`reference_witness`), the library checks every row, commits, and proves."""
    coef_index = {}
    coeffs = []

    def cid(v):
        if v not in coef_index:
            coef_index[v] = len(coeffs)
            coeffs.append(v)
        return coef_index[v]

    def sparse(side):
        rowptr, flat = [0], []
        for row in system.rows:
            for cf, wi in row[side]:
                flat += [wi, cid(cf)]
            rowptr.append(len(flat) // 2)
        return struct.pack("<2I", len(system.rows), len(flat) // 2) + struct.pack("<%dI" % len(rowptr), *rowptr) + \
            struct.pack("<%dI" % len(flat), *flat)

    mats = [sparse(0), sparse(1), sparse(2), struct.pack("<3I", 0, 0, 0)]      # A, B, C, no hint rows
    committed = c.meta["CommitmentInfo"].value[0]["PrivateCommitted"]
    program = [OP_COMMIT, OP_END]
    n_public = len(c.public)
    n = len(system.rows)
    domain_log = max(1, (n - 1).bit_length())
    head = struct.pack("<13I", SPPC_MAGIC, SPPC_VERSION, CIRCUIT_CCS, n_public, system.n_wires - n_public, system.n_wires, n, domain_log,
                       challenge_wire(system), len(coeffs), len(committed), len(program), 0)
    body = b"".join(v.to_bytes(32, "little") for v in coeffs) + b"".join(mats) + \
        struct.pack("<%dI" % len(committed), *committed) + struct.pack("<%dI" % len(program), *program)
    with open(path, "wb") as f:
        f.write(head + body)
    return n


# ---------------------------------------------------------------------------------------------------------------------
# the decoded system WITH its solver: 26 withdraw inputs -> every wire on the device
# ---------------------------------------------------------------------------------------------------------------------
OP_BITS, OP_INV_H, OP_MASK = 4, 11, 12
OP_SOLVE_ROW, OP_LIMBS, OP_COUNTN, OP_GK_MUL, OP_GLV, OP_EMUL = 13, 14, 15, 16, 17, 18
NONE32 = 0xFFFFFFFF
GRUMPKIN_GY = 17631683881184975370165255887551781615748388533673675138860
LIMBS_REVERSED = 1 << 31


def _glv_constants(lam=GLV_LAMBDA, q=Q_BASE):
    """(v1, v2, det) of glv_split -- the reduced lattice basis depends on (q, lambda) alone -- as the 28 words OP_GLV carries"""
    r0, r1, t0, t1 = q, lam % q, 0, 1
    rows = []
    while r1:
        k = r0 // r1
        r0, r1, t0, t1 = r1, r0 - k * r1, t1, t0 - k * t1
        rows.append((r0, t0))
    lim = 1 << (q.bit_length() // 2)
    i = next(j for j, (r, _) in enumerate(rows) if r < lim)
    v1, v2 = rows[i], rows[i + 1]
    det = v1[0] * v2[1] - v2[0] * v1[1]
    if det < 0:
        v1, v2, det = v2, v1, -det
    words = []
    for v in (v1[0], v1[1], v2[0], v2[1]):
        assert abs(v) < 1 << 128
        words += [(abs(v) >> (32 * i)) & 0xFFFFFFFF for i in range(4)] + [1 if v < 0 else 0]
    words += [(det >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    return words


class SolvedSystem:
    """what to_sppc_solved derives: the wire permutation (gnark wire -> container wire), the solving steps and the container parts"""


def plan_solver(system, c, program):
    """The order in which every wire of the reference's system follows from the 26 ABI inputs: gnark's own rule -- a row with ONE
    unknown wire defines it -- applied to ALL wires, the 6 184 ACIR witnesses included (in `sunspot prove` those come from `nargo
    execute`; here the rows sunspot compiled from the same ACIR opcodes define them), plus the unconstrained helpers: the six Brillig
    calls and the MultiScalarMul black box of the ACIR program, the 41 hint calls of the gnark system.  Returns the list of steps
    ('acir', opcode index) | ('hint', hint index) | ('row', row, gnark wire, side) in a valid order; raises when a wire stays unknown."""
    npub = len(c.public)
    sec_index = {nm: npub + i for i, nm in enumerate(c.secret)}

    def gw(acir_witness):        # gnark wire of an ACIR witness (None: the system does not use it)
        flat_pub = len(program.main.public_parameters)
        if acir_witness < flat_pub:
            return 1 + acir_witness
        return sec_index.get("__witness_%d" % acir_witness)

    known = set(range(npub))
    for _, ws in program.parameter_witnesses():
        for x in ws:
            if gw(x) is not None:
                known.add(gw(x))

    def expr_wires(e):
        ws = set()
        for _, a, b in e.mul_terms:
            ws |= {a, b}
        for _, a in e.linear:
            ws.add(a)
        return ws
    acir_ops = []
    for i, op in enumerate(program.main.opcodes):
        if op[0] == "BrilligCall":
            ins = set()
            for inp in op[2]:
                for e in ([inp[1]] if inp[0] == "single" else inp[1]):
                    ins |= expr_wires(e)
            outs = [x for o in op[3] for x in ([o[1]] if o[0] == "simple" else o[1])]
            if any(gw(x) is not None for x in outs):
                acir_ops.append((i, ins, outs))
        elif op[0] == "MultiScalarMul":
            acir_ops.append((i, {x[1] for x in op[2]}, list(op[4])))
    steps = []
    pending_acir, pending_hints = list(acir_ops), list(range(len(system.hints)))
    work = set(range(len(system.rows)))
    wires_of = [set(w for _, w in L + Rr + O) for (L, Rr, O) in system.rows]
    progress = True
    while progress:
        progress = False
        for op in list(pending_acir):
            i, ins, outs = op
            if all(gw(x) in known for x in ins):
                for x in outs:
                    if gw(x) is not None:
                        known.add(gw(x))
                steps.append(("acir", i))
                pending_acir.remove(op)
                progress = True
        for hi in list(pending_hints):
            _, _, _, ins, o0, o1 = system.hints[hi]
            if all(w == CONST_WIRE or w in known for terms in ins for _, w in terms):
                known.update(range(o0, o1))
                steps.append(("hint", hi))
                pending_hints.remove(hi)
                progress = True
        for k in sorted(work):
            unk = wires_of[k] - known
            if not unk:
                work.discard(k)
                continue
            if len(unk) != 1:
                continue
            w = next(iter(unk))
            sides = [any(x == w for _, x in side) for side in system.rows[k]]
            if sum(sides) != 1:
                continue
            known.add(w)
            work.discard(k)
            steps.append(("row", k, w, sides.index(True)))
            progress = True
    if len(known) != system.n_wires or pending_hints:
        raise ValueError("the solver plan leaves %d wires unknown" % (system.n_wires - len(known)))
    return steps, gw


def to_sppc_solved(system, c, program, path, circuit_id=CIRCUIT_CCS):
    """The reference's R1CS as a container that carries its SOLVER: inputs are the 26 values of the withdraw ABI (5 public + 21
    private, the order of client/proof.helper.ts:34-50), every other wire -- the 6 163 remaining ACIR witnesses and the 6 749 internal
    wires -- is computed on the device by the program emitted here (plan_solver; new instructions OP_SOLVE_ROW .. OP_EMUL of
    csrc/circuit.hpp).  Same rows as `to_sppc` (terms of a row reordered so that the wire a row defines comes last on its side,
    duplicate wires of a side merged), wires renumbered: constant, 5 public, the 21 private inputs, then the rest in gnark's
    order (+ one scratch wire no row touches, for hint outputs the system does not use).  Setup under the same seed gives the same
    verifying key and, for the same (r, s) and mask, the same proof bytes as the all-inputs container -- what the tests compare.
    circuit_id = 1 makes it a drop-in for generateProof / spp_prove_withdraw.  Returns (n_constraints, SolvedSystem)."""
    steps, gw = plan_solver(system, c, program)
    npub = len(c.public)
    abi_secret = [gw(x) for _, ws in program.parameter_witnesses() for x in ws][npub - 1:]
    assert len(abi_secret) == 21 and all(w is not None for w in abi_secret)
    order = list(range(npub)) + abi_secret + [w for w in range(npub, system.n_wires) if w not in set(abi_secret)]
    perm = {g: i for i, g in enumerate(order)}
    trash = system.n_wires                       # one wire no constraint touches
    n_wires = system.n_wires + 1
    coef_index, coeffs = {}, []

    def cid(v):
        v %= R
        if v not in coef_index:
            coef_index[v] = len(coeffs)
            coeffs.append(v)
        return coef_index[v]

    def merged(terms, last=None):
        acc = {}
        for cf, wi in terms:
            acc[wi] = (acc.get(wi, 0) + cf) % R
        items = [(wi, cf) for wi, cf in acc.items() if wi != last]
        if last is not None:
            items.append((last, acc[last]))
        return items
    defined = {}                                  # row -> (gnark wire, side)
    for st in steps:
        if st[0] == "row":
            defined[st[1]] = (st[2], st[3])
    rows_out = []
    for k, row in enumerate(system.rows):
        w, side = defined.get(k, (None, None))
        rows_out.append([merged(row[sd], w if sd == side else None) for sd in range(3)])
    hrows = []

    def hrow(terms):
        acc = {}
        for cf, wi in terms:
            key = 0 if wi == CONST_WIRE else perm[wi]
            acc[key] = (acc.get(key, 0) + cf) % R
        hrows.append([(wi, cf) for wi, cf in acc.items() if cf])
        return len(hrows) - 1

    def contiguous(gwires):
        idx = [perm[g] for g in gwires]
        if idx != list(range(idx[0], idx[0] + len(idx))):
            raise ValueError("hint outputs are not consecutive wires")
        return idx[0]
    prog = []
    chal = None
    for st in steps:
        if st[0] == "row":
            _, k, w, side = st
            cf = rows_out[k][side][-1][1]
            if cf == 0:
                raise ValueError("row %d: the wire it defines has coefficient 0" % k)
            inv_ci = NONE32 if cf == 1 else cid(pow(cf, -1, R))
            odiv = NONE32
            if side != 2:
                other = rows_out[k][1 - side]
                if all(wi == 0 for wi, _ in other):
                    val = sum(cf_ for _, cf_ in other) % R
                    if val == 0:
                        raise ValueError("row %d divides by the constant 0" % k)
                    odiv = cid(pow(val, -1, R))
            prog += [OP_SOLVE_ROW, k, side, inv_ci, odiv]
        elif st[0] == "acir":
            op = program.main.opcodes[st[1]]

            def lin(e):
                if e.mul_terms:
                    raise ValueError("ACIR opcode %d: a helper input with a product term" % st[1])
                return [(cf, gw(a)) for cf, a in e.linear] + [(e.constant, CONST_WIRE)]
            if op[0] == "MultiScalarMul":
                (_, lo), (_, hi) = op[2]
                ox, oy, oi = (perm[gw(x)] if gw(x) is not None else trash for x in op[4])
                prog += [OP_GK_MUL, hrow([(1, gw(lo))]), hrow([(1, gw(hi))]), cid(GRUMPKIN_GY), ox, oy, oi]
                continue
            kind = program.brillig_kind(st[1], op[1], op[2], op[3])
            exprs = [i[1] for i in op[2]]
            outs = [x for o in op[3] for x in ([o[1]] if o[0] == "simple" else o[1])]
            if kind == "divmod":
                if exprs[1].mul_terms or exprs[1].linear or exprs[1].constant != 1 << 128:
                    raise ValueError("ACIR opcode %d: quotient / remainder by something else than 2^128" % st[1])
                q_w, r_w = perm[gw(outs[0])], perm[gw(outs[1])]
                if q_w + 1 == r_w:      # limb 0 (the remainder) goes to the HIGHER wire
                    prog += [OP_LIMBS, hrow(lin(exprs[0])), 2, 128 | LIMBS_REVERSED, q_w]
                elif r_w + 1 == q_w:
                    prog += [OP_LIMBS, hrow(lin(exprs[0])), 2, 128, r_w]
                else:
                    raise ValueError("ACIR opcode %d: quotient and remainder are not neighbouring wires" % st[1])
            elif kind == "inverse":
                prog += [OP_INV_H, hrow(lin(exprs[0])), perm[gw(outs[0])] if gw(outs[0]) is not None else trash]
            else:                       # radix 2
                if exprs[2].constant != 2 or exprs[1].constant != len(outs):
                    raise ValueError("ACIR opcode %d: radix decomposition other than bits" % st[1])
                prog += [OP_BITS, hrow(lin(exprs[0])), len(outs), contiguous([gw(x) for x in outs])]
        else:
            _, _, name, ins, o0, o1 = system.hints[st[1]]
            short = name.rsplit("/", 1)[-1]
            n_out = o1 - o0
            const = lambda terms: sum(cf for cf, _ in terms) % R if all(w == CONST_WIRE for _, w in terms) else None
            if short == "bits.nBits":
                prog += [OP_BITS, hrow(ins[0]), n_out, contiguous(range(o0, o1))]
            elif short == "rangecheck.DecomposeHint":
                width = const(ins[1])
                prog += [OP_LIMBS, hrow(ins[2]), n_out, width, contiguous(range(o0, o1))]
            elif short == "sw-grumpkin.decompose":
                prog += [OP_LIMBS, hrow(ins[0]), n_out, 64, contiguous(range(o0, o1))]
            elif short == "solver.InvZeroHint":
                assert n_out == 1 and len(ins) == 1
                prog += [OP_INV_H, hrow(ins[0]), perm[o0]]
            elif short == "logderivarg.countHint":
                size, width = const(ins[0]), const(ins[1])
                if width != 1 or [const(t) for t in ins[2:2 + size]] != list(range(size)) or n_out != size or size > 256:
                    raise ValueError("countHint over a table that is not 0 .. size-1")
                q = ins[2 + size:]
                h0 = hrow(q[0])
                for t in q[1:]:
                    hrow(t)
                prog += [OP_COUNTN, h0, len(q), contiguous(range(o0, o1)), size]
            elif short == "sw-grumpkin.decomposeScalar":
                nl, bits = const(ins[7]), const(ins[8])
                q = sum(const(t) << (bits * i) for i, t in enumerate(ins[9:9 + nl]))
                if (nl, bits, q, n_out) != (4, 64, Q_BASE, 8):
                    raise ValueError("decomposeScalar with an unexpected layout")
                prog += [OP_GLV, hrow(ins[6]), contiguous(range(o0, o1))] + _glv_constants()
            elif short == "emulated.mulHint":
                bits, n, na, nq = (const(ins[i]) for i in range(4))
                p_l = [const(t) for t in ins[4:4 + n]]
                b_l = [const(t) for t in ins[4 + n + na:]]
                if (bits, n, na, nq, b_l, n_out) != (64, 4, 6, 4, [1], 14) or sum(v << (64 * i) for i, v in enumerate(p_l)) != Q_BASE:
                    raise ValueError("emulated.mulHint with an unexpected layout")
                a_rows = ins[4 + n:4 + n + na]
                h0 = hrow(a_rows[0])
                for t in a_rows[1:]:
                    hrow(t)
                qinv = pow(Q_BASE, -1, 1 << 256)
                prog += [OP_EMUL, h0, contiguous(range(o0, o1))] + [(Q_BASE >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + \
                        [(qinv >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
            elif short == "hints.Randomize":
                prog += [OP_MASK, perm[o0]]
            elif short == "cs.Bsb22CommitmentComputePlaceholder":
                prog += [OP_COMMIT]
                chal = perm[o0]
            else:
                raise ValueError("hint %s has no device implementation" % short)
    prog.append(OP_END)

    def sparse(rows):
        rowptr, flat = [0], []
        for terms in rows:
            for wi, cf in terms:
                flat += [wi, cid(cf)]
            rowptr.append(len(flat) // 2)
        return struct.pack("<2I", len(rows), len(flat) // 2) + struct.pack("<%dI" % len(rowptr), *rowptr) + struct.pack("<%dI" % len(flat), *flat)
    mats = [sparse([[(perm[wi], cf) for wi, cf in r[sd]] for r in rows_out]) for sd in range(3)] + [sparse(hrows)]
    committed = [perm[w] for w in c.meta["CommitmentInfo"].value[0]["PrivateCommitted"]]
    n = len(system.rows)
    domain_log = max(1, (n - 1).bit_length())
    head = struct.pack("<13I", SPPC_MAGIC, SPPC_VERSION, circuit_id, npub, 21, n_wires, n, domain_log, chal, len(coeffs), len(committed),
                       len(prog), 0)
    body = b"".join(v.to_bytes(32, "little") for v in coeffs) + b"".join(mats) + struct.pack("<%dI" % len(committed), *committed) + \
        struct.pack("<%dI" % len(prog), *prog)
    with open(path, "wb") as f:
        f.write(head + body)
    ss = SolvedSystem()
    ss.perm, ss.order, ss.steps, ss.program, ss.hrows, ss.rows, ss.trash, ss.challenge_wire = perm, order, steps, prog, hrows, rows_out, trash, chal
    return n, ss


def reference_witness(system, c, public_inputs, secret_by_name, challenge_of_row):
    """The full assignment of the reference's system as the input row of the container `to_sppc` writes (wires 1 .. n-1).
    challenge_of_row(row with unknown wires as 0) -> commitment challenge (CircuitHandle.commitment_challenge on the GPU)."""
    def chal(w):
        return challenge_of_row([0 if v is None else v for v in w[1:]])
    wires, st = solve_partial(system, c, public_inputs, secret_by_name, challenge_fn=chal)
    if st["rows_unsatisfied"] or st["rows_skipped"] or st["hints_skipped"]:
        raise ValueError("the inputs do not satisfy the reference's constraint system: %r" % {k: v for k, v in st.items() if v})
    return wires[1:]


_WORKER_CACHE = {}


def complete_witness_worker(args):
    """Process-pool worker (bench.py, batch drivers): (ccs path, acir path, input row, challenge | None, mask) -> the witness row of the
    container `to_sppc` writes, 32 B big-endian per wire.  challenge None: a placeholder is used -- the committed wires, hence the
    commitment, do not depend on it -- and the caller repeats the call with the value spp_commitment_challenge returns and the
    SAME mask (a fresh random field element per proof: it is one of the committed wires)."""
    ccs_path, acir_path, row, challenge, randomizer = args
    key = (ccs_path, acir_path)
    if key not in _WORKER_CACHE:
        from . import acir
        c = load_ccs(ccs_path)
        _WORKER_CACHE[key] = (c, decode_system(c), acir.load_program(acir_path), acir)
    c, system, program, acir = _WORKER_CACHE[key]
    w = acir.execute(program, row)
    secret = {"__witness_%d" % k: v for k, v in w.items()}
    wires, st = solve_partial(system, c, row[:len(c.public) - 1], secret, challenge=0x5EED if challenge is None else challenge,
                              randomizer=randomizer)
    if st["rows_unsatisfied"] or st["rows_skipped"] or st["hints_skipped"]:
        raise ValueError("the inputs do not satisfy the reference's constraint system")
    return b"".join(v.to_bytes(32, "big") for v in wires[1:])

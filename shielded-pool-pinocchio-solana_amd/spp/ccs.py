"""Reader for gnark's R1CS container `.ccs` (noir_circuit/target/shielded_pool_verifier.ccs, what `sunspot
compile` writes and `sunspot prove <acir> <witness> <ccs> <pk>` reads, client/proof.helper.ts:64).

Layout (gnark 0.14, decoded from the reference's own file):
  [0x00] u64 LE  = file size - 32          [0x08] u64 x3 = 0, 14, 0 (opaque here)
  [0x20] u64 LE x4 = byte lengths of: levels stream, instructions stream, calldata stream, CBOR body
  [0x40] the three streams (compressed uint32 sequences), then the CBOR map
         {Type, GnarkVersion, ScalarField, NbConstraints, NbInternalVariables, Public[], Secret[], Blueprints[],
          CommitmentInfo, MHintsDependencies, GkrInfo, Logs, DebugInfo, MDebug, SymbolTable ...}
  then   u64 LE count . count x 32 B   the coefficient table, little-endian Montgomery limbs (entry 1 = R mod r)
This module decodes the header, the CBOR body and the coefficient table -- everything that pins the DIMENSIONS and wire
naming of the reference's constraint system.  The three integer streams (constraint levels, instructions, calldata) are
returned as raw bytes: their compression (github.com/ronanh/intcomp, third-party, absent from the reference) is not decoded,
so this repository cannot yet re-evaluate the reference's own R1CS; see DESIGN.md section 8.
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

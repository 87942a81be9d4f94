"""ORACLE (test infrastructure only) -- BN254 fields, G1/G2, optimal-ate pairing, hash-to-field.

This file is part of the CPU restatement used ONLY by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg as the checker. The product path (libspp.so, HIP) never imports it.

The reference (Ham3798/shielded-pool-pinocchio-solana) holds no prover code: proving is
`execSync("sunspot prove ...")` (client/proof.helper.ts:64). Sunspot (reilabs/sunspot @5fd6223)
wraps gnark 0.14.0 groth16/bn254 -- neither is present under /root/reference, so the algorithm
restated here is the published one (Groth16 + BSB22 Pedersen commitment, RFC 9380
expand_message_xmd with SHA-256), anchored on the reference's byte formats:
  proof 388 B   shielded_pool_program/src/instructions/withdraw.rs:13
  .pw  12+32n B shielded_pool_program/src/instructions/withdraw.rs:14-16
  vk layout     noir_circuit/target/shielded_pool_verifier.vk
Groth16 proof BYTES are "parity unpinned" at the reference level (no pk, no .proof fixtures);
everything upstream (Poseidon, Merkle, Grumpkin, RLWE) is pinned by golden vectors.

Plain Python big-int arithmetic: slow, obviously correct.
"""
import hashlib

# BN254 (alt_bn128) parameters -- scripts/generate_audit.py:34, client/merkle.ts:47-48
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # Fr (scalar field)
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # Fq (base field)
G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
FR_GENERATOR = 5          # multiplicative generator of Fr (2-adicity 28)
FR_TWO_ADICITY = 28


def inv(a, m):
    return pow(a % m, -1, m)


# ----------------------------------------------------------------------------- G1 (affine, None = infinity)
def g1_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - 3) % P == 0


def g1_neg(pt):
    return None if pt is None else (pt[0], (-pt[1]) % P)


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * inv(2 * y1, P) % P
    else:
        lam = (y2 - y1) * inv(x2 - x1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def _jac_dbl(X, Y, Z, p):
    if Y == 0 or Z == 0:
        return (1, 1, 0)
    A = X * X % p
    B = Y * Y % p
    C = B * B % p
    D = 2 * ((X + B) * (X + B) - A - C) % p
    E = 3 * A % p
    F = E * E % p
    X3 = (F - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def _jac_add_affine(X1, Y1, Z1, x2, y2, p):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % p
    U2 = x2 * Z1Z1 % p
    S2 = y2 * Z1 * Z1Z1 % p
    H = (U2 - X1) % p
    r = (S2 - Y1) % p
    if H == 0:
        if r == 0:
            return _jac_dbl(X1, Y1, Z1, p)
        return (1, 1, 0)
    HH = H * H % p
    HHH = H * HH % p
    V = X1 * HH % p
    X3 = (r * r - HHH - 2 * V) % p
    Y3 = (r * (V - X3) - Y1 * HHH) % p
    Z3 = Z1 * H % p
    return (X3, Y3, Z3)


def g1_mul(pt, k):
    """Scalar multiplication (Jacobian double-and-add), k reduced mod R."""
    k %= R
    if pt is None or k == 0:
        return None
    X, Y, Z = 1, 1, 0
    for bit in bin(k)[2:]:
        X, Y, Z = _jac_dbl(X, Y, Z, P)
        if bit == '1':
            X, Y, Z = _jac_add_affine(X, Y, Z, pt[0], pt[1], P)
    if Z == 0:
        return None
    zi = inv(Z, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def g1_msm(points, scalars):
    acc = None
    for pt, s in zip(points, scalars):
        acc = g1_add(acc, g1_mul(pt, s))
    return acc


# ----------------------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1)
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_scalar(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    d = inv(a[0] * a[0] + a[1] * a[1], P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


G2_B = f2_mul((3, 0), f2_inv((9, 1)))  # twist: y^2 = x^3 + 3/(9+u)


def g2_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), G2_B)) == (0, 0)


def g2_neg(pt):
    return None if pt is None else (pt[0], f2_neg(pt[1]))


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        lam = f2_mul(f2_scalar(f2_mul(x1, x1), 3), f2_inv(f2_scalar(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_mul(pt, k):
    k %= R
    acc = None
    add = pt
    while k:
        if k & 1:
            acc = g2_add(acc, add)
        add = g2_add(add, add)
        k >>= 1
    return acc


# ----------------------------------------------------------------------------- Fq12 = Fq[w]/(w^12 - 18 w^6 + 82)
# (w^6 = 9 + u, so u = w^6 - 9; (w^6-9)^2 = -1  =>  w^12 = 18 w^6 - 82)
def f12_one():
    return [1] + [0] * 11


def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [v % P for v in t[:12]]


def f12_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def f12_scalar(a, k):
    return [x * k % P for x in a]


def _poly_deg(p_):
    d = len(p_) - 1
    while d and p_[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    """Extended Euclid over Fq[w] against the modulus polynomial."""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [82, 0, 0, 0, 0, 0, (-18) % P, 0, 0, 0, 0, 0, 1]
    while _poly_deg(low):
        # r = high / low (poly division, rounded)
        dl, dh = _poly_deg(low), _poly_deg(high)
        r = [0] * 13
        temp = list(high)
        for i in range(dh - dl, -1, -1):
            q = temp[dl + i] * inv(low[dl], P) % P
            r[i] = q
            for c in range(dl + 1):
                temp[c + i] = (temp[c + i] - low[c] * q) % P
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] -= lm[i] * r[j]
                new[i + j] -= low[i] * r[j]
        nm = [x % P for x in nm]
        new = [x % P for x in new]
        lm, low, hm, high = nm, new, lm, low
    li = inv(low[0], P)
    return [x * li % P for x in lm[:12]]


def f12_pow(a, e):
    result = f12_one()
    base = a
    while e:
        if e & 1:
            result = f12_mul(result, base)
        base = f12_mul(base, base)
        e >>= 1
    return result


def _twist(pt):
    """G2 point over Fq2 -> point over Fq12 on y^2 = x^3 + 3."""
    (x0, x1), (y0, y1) = pt
    xc = [(x0 - 9 * x1) % P, x1]
    yc = [(y0 - 9 * y1) % P, y1]
    nx = [0] * 12
    ny = [0] * 12
    # nx = (xc0 + xc1 w^6) * w^2 ; ny = (yc0 + yc1 w^6) * w^3
    nx[2], nx[8] = xc[0], xc[1]
    ny[3], ny[9] = yc[0], yc[1]
    return (nx, ny)


def _cast_g1(pt):
    return ([pt[0]] + [0] * 11, [pt[1]] + [0] * 11)


def _f12_is_zero(a):
    return all(v == 0 for v in a)


def _pt12_double(pt):
    x, y = pt
    lam = f12_mul(f12_scalar(f12_mul(x, x), 3), f12_inv(f12_scalar(y, 2)))
    nx = f12_sub(f12_sub(f12_mul(lam, lam), x), x)
    ny = f12_sub(f12_mul(lam, f12_sub(x, nx)), y)
    return (nx, ny)


def _pt12_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if y1 == y2:
            return _pt12_double(a)
        return None
    lam = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    nx = f12_sub(f12_sub(f12_mul(lam, lam), x1), x2)
    ny = f12_sub(f12_mul(lam, f12_sub(x1, nx)), y1)
    return (nx, ny)


def _linefunc(p1, p2, t):
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(f12_scalar(f12_mul(x1, x1), 3), f12_inv(f12_scalar(y1, 2)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    return f12_sub(xt, x1)


ATE_LOOP_COUNT = 29793968203157093288
LOG_ATE_LOOP_COUNT = 63


def miller_loop(q2, p1):
    """Miller loop of the optimal ate pairing e(p1 in G1, q2 in G2) (no final exponentiation)."""
    if q2 is None or p1 is None:
        return f12_one()
    Q = _twist(q2)
    Pt = _cast_g1(p1)
    Rr = Q
    f = f12_one()
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f12_mul(f12_mul(f, f), _linefunc(Rr, Rr, Pt))
        Rr = _pt12_double(Rr)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, _linefunc(Rr, Q, Pt))
            Rr = _pt12_add(Rr, Q)
    Q1 = (f12_pow(Q[0], P), f12_pow(Q[1], P))
    nQ2 = (f12_pow(Q1[0], P), [(-v) % P for v in f12_pow(Q1[1], P)])
    f = f12_mul(f, _linefunc(Rr, Q1, Pt))
    Rr = _pt12_add(Rr, Q1)
    f = f12_mul(f, _linefunc(Rr, nQ2, Pt))
    return f


def final_exponentiation(f):
    return f12_pow(f, (P ** 12 - 1) // R)


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 for pairs [(g1 point, g2 point), ...]."""
    f = f12_one()
    for p1, q2 in pairs:
        f = f12_mul(f, miller_loop(q2, p1))
    return final_exponentiation(f) == f12_one()


# ----------------------------------------------------------------------------- serialisation (gnark raw)
def fe_be(x):
    return int(x).to_bytes(32, 'big')


def g1_to_bytes(pt):
    """Uncompressed 64 B: X || Y big-endian; infinity = 64 zero bytes."""
    if pt is None:
        return b'\x00' * 64
    return fe_be(pt[0]) + fe_be(pt[1])


def g1_from_bytes(b):
    if b == b'\x00' * 64:
        return None
    return (int.from_bytes(b[:32], 'big'), int.from_bytes(b[32:64], 'big'))


def g2_to_bytes(pt):
    """Uncompressed 128 B: X.A1 || X.A0 || Y.A1 || Y.A0."""
    if pt is None:
        return b'\x00' * 128
    (x0, x1), (y0, y1) = pt
    return fe_be(x1) + fe_be(x0) + fe_be(y1) + fe_be(y0)


def g2_from_bytes(b):
    if b == b'\x00' * 128:
        return None
    x1, x0, y1, y0 = (int.from_bytes(b[i * 32:(i + 1) * 32], 'big') for i in range(4))
    return ((x0, x1), (y0, y1))


# ----------------------------------------------------------------------------- hash to field (RFC 9380 expand_message_xmd / SHA-256)
def expand_message_xmd(msg, dst, len_in_bytes):
    b_in_bytes, s_in_bytes = 32, 64
    ell = (len_in_bytes + b_in_bytes - 1) // b_in_bytes
    assert ell <= 255 and len(dst) <= 255
    dst_prime = dst + bytes([len(dst)])
    msg_prime = b'\x00' * s_in_bytes + msg + len_in_bytes.to_bytes(2, 'big') + b'\x00' + dst_prime
    b0 = hashlib.sha256(msg_prime).digest()
    bi = hashlib.sha256(b0 + b'\x01' + dst_prime).digest()
    out = bi
    for i in range(2, ell + 1):
        bi = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:len_in_bytes]


def hash_to_fr(msg, dst, count=1):
    """gnark-crypto fr.Hash: L = 48 bytes per element, big-endian, reduced mod R."""
    L = 48
    u = expand_message_xmd(msg, dst, count * L)
    return [int.from_bytes(u[i * L:(i + 1) * L], 'big') % R for i in range(count)]


DST_COMMITMENT = b'bsb22-commitment'   # string present in audit_circuit/target/audit_verifier.so
DST_MASK = b'spp-commit-mask1'         # the commitment's hiding mask (OP_MASK): its own domain
DST_FOLD = b'G16-BSB22'

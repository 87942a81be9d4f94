"""Interval-arithmetic certificate for csrc/f29.hpp (test infrastructure).

Re-derives, with exact integers, the bounds the comments in f29.hpp state: for every multiplication the 64-bit column
sums cannot overflow, for every subtraction the lifted constant dominates the subtrahend limb by limb, and the value
bounds assumed for the accumulator (X < 5.1p, Y < 1.2p, ZZ/ZZZ < 1.1p) are reproduced by one more addition (so they
hold inductively).  Constants are parsed from the generated header so the check sees what the kernels compile.
"""
import os
import re

M29 = (1 << 29) - 1
RP = 1 << 261
HDR = os.path.join(os.path.dirname(__file__), "..", "..", "shielded-pool-pinocchio-solana_amd", "csrc", "bn254_consts.hpp")


def parse(struct):
    txt = open(HDR).read()
    body = txt.split("struct %s {" % struct)[1].split("\n};")[0]
    out = {}
    for name, vals in re.findall(r"uint32_t (\w+)\(int i\) \{ constexpr uint32_t v\[\d\] = \{([^}]*)\}", body):
        out[name] = [int(v.strip().rstrip("u"), 16) for v in vals.split(",")]
    return out


class V:
    """limb bounds l[0..8] (inclusive maxima) and an exclusive bound on the integer value"""
    def __init__(self, limbs, value):
        self.l, self.v = list(limbs), value

    @staticmethod
    def normalised(value):
        return V([M29] * 8 + [value >> 232], value)


class Ctx:
    def __init__(self, struct):
        self.c = parse(struct)
        mod = self.c["MOD"]
        self.p = sum(w << (32 * i) for i, w in enumerate(mod))
        self.p9 = [(self.p >> (29 * i)) & M29 for i in range(9)]
        self.log = []

    def const(self, name):
        limbs = self.c[name]
        val = sum(x << (29 * i) for i, x in enumerate(limbs))
        assert val % self.p == 0, name
        return limbs, val

    def columns(self, pairs, what):
        col = [0] * 18
        for a, b in pairs:
            for i in range(9):
                for j in range(9):
                    col[i + j] += a.l[i] * b.l[j]
        for k in range(9):                       # reduction step k adds m*p9[j] to column k+j, then carries
            for j in range(9):
                col[k + j] += M29 * self.p9[j]
            assert col[k] < 1 << 64, "%s: column %d overflows" % (what, k)
            col[k + 1] += col[k] >> 29
        for k in range(9, 17):
            assert col[k] < 1 << 64, "%s: column %d overflows" % (what, k)
            col[k + 1] += col[k] >> 29
        assert col[17] < 1 << 32, "%s: top limb overflows" % what
        value = sum(a.v * b.v for a, b in pairs) // RP + self.p + 1
        self.log.append((what, value / self.p))
        return V.normalised(value)

    def mul(self, a, b, what):
        return self.columns([(a, b)], what)

    def sqr(self, a, what):
        assert all(x < 1 << 31 for x in a.l)
        return self.columns([(a, a)], what)

    def mul2(self, a, b, c, d, what):
        return self.columns([(a, b), (c, d)], what)

    def _dominates(self, climbs, sub, what):
        for i in range(9):
            assert climbs[i] >= sub[i], "%s: constant limb %d (%d) below subtrahend bound (%d)" % (what, i, climbs[i], sub[i])

    def sub_norm(self, a, b, cname, what):
        cl, cv = self.const(cname)
        self._dominates(cl, b.l, what)
        assert all(a.l[i] + cl[i] + 8 < 1 << 32 for i in range(9)), what
        return V.normalised(a.v + cv)

    def sub3_norm(self, a, b, c2, cname, what):
        cl, cv = self.const(cname)
        self._dominates(cl, [b.l[i] + 2 * c2.l[i] for i in range(9)], what)
        assert all(a.l[i] + cl[i] + 8 < 1 << 32 for i in range(9)), what
        return V.normalised(a.v + cv)

    def sub_lazy(self, a, b, cname, what):
        cl, cv = self.const(cname)
        self._dominates(cl, b.l, what)
        return V([a.l[i] + cl[i] for i in range(9)], a.v + cv)

    def neg_lazy(self, b, cname, what):
        cl, cv = self.const(cname)
        self._dominates(cl, b.l, what)
        return V(cl, cv + 1)


def check_madd(struct="FqParams"):
    """XYZZ29::madd, line by line; returns the value bounds (in units of p) it reproduces."""
    cx = Ctx(struct)
    p = cx.p
    X = V.normalised(int(5.1 * p))
    Y = V.normalised(int(1.2 * p))
    ZZ = V.normalised(int(1.1 * p))
    ZZZ = V.normalised(int(1.1 * p))
    x2 = V.normalised(2 * p)                      # table words: "almost Montgomery" range [0, 2p)
    y2 = cx.neg_lazy(V.normalised(2 * p), "SUBC_4P_1", "negated y2")    # the larger of the two forms of y2
    kin = V.normalised(p)
    # first addition into an empty accumulator
    X0 = cx.mul(x2, kin, "X0")
    Y0 = cx.mul(y2, kin, "Y0")
    assert X0.v <= X.v and Y0.v <= Y.v
    U2 = cx.mul(x2, ZZ, "U2")
    S2 = cx.mul(y2, ZZZ, "S2")
    Pp = cx.sub_norm(U2, X, "SUBC_6P_1", "P")
    Rr = cx.sub_norm(S2, Y, "SUBC_2P_1", "R")
    assert Pp.v <= 8 * p and Rr.v <= 4 * p        # is_zero_mod_p(7) / (3) cover every multiple below the bound
    PP = cx.sqr(Pp, "PP")
    PPP = cx.mul(Pp, PP, "PPP")
    Q = cx.mul(X, PP, "Q")
    R2 = cx.sqr(Rr, "R2")
    X3 = cx.sub3_norm(R2, PPP, Q, "SUBC_4P_3", "X3")
    T = cx.sub_lazy(Q, X3, "SUBC_6P_1", "T")
    Yn = cx.neg_lazy(Y, "SUBC_2P_1", "Yn")
    Y3 = cx.mul2(Rr, T, Yn, PPP, "Y3")
    ZZ3 = cx.mul(ZZ, PP, "ZZ3")
    ZZZ3 = cx.mul(ZZZ, PPP, "ZZZ3")
    assert X3.v <= X.v, ("X grows", X3.v / p)
    assert Y3.v <= Y.v, ("Y grows", Y3.v / p)
    assert ZZ3.v <= ZZ.v and ZZZ3.v <= ZZZ.v
    # conversions out: to_fp / scaled_to_fp multiply by a constant < p and must land below 2p with limbs that pack
    for nm, v in (("X", X), ("Y", Y), ("ZZ", ZZ)):
        o = cx.mul(v, V.normalised(p), nm + " out")
        assert o.v < 2 * p
    return dict(cx.log)


def check_madd_g2():
    """XYZZ29G2::madd (Fq2 components), same line order as f29.hpp."""
    cx = Ctx("FqParams")
    p = cx.p
    norm = V.normalised

    def add_lazy(a, b):
        return V([a.l[i] + b.l[i] for i in range(9)], a.v + b.v)

    def mul(a, b, ca, what):          # F29x2::mul<CA>
        na1 = cx.neg_lazy(a[1], ca, what + " -a1")
        return (cx.mul2(a[0], b[0], na1, b[1], what + ".c0"), cx.mul2(a[0], b[1], a[1], b[0], what + ".c1"))

    def sqr(a, ca, what):             # F29x2::sqr<CA>
        s = add_lazy(a[0], a[1])
        d = cx.sub_lazy(a[0], a[1], ca, what + " a0-a1")
        t = add_lazy(a[0], a[0])
        return (cx.mul(s, d, what + ".c0"), cx.mul(t, a[1], what + ".c1"))

    def sub_norm(a, b, c, what):
        return tuple(cx.sub_norm(a[i], b[i], c, what) for i in range(2))

    X = (norm(int(5.6 * p)),) * 2
    Y = (norm(int(1.5 * p)),) * 2
    ZZ = (norm(int(1.3 * p)),) * 2
    ZZZ = (norm(int(1.3 * p)),) * 2
    x2 = (norm(2 * p),) * 2
    y2 = (norm(2 * p),) * 2
    kin = norm(p)
    # first addition
    yn = cx.neg_lazy(y2[0], "SUBC_4P_1", "y2 negated")
    assert cx.mul(x2[0], kin, "X0").v <= X[0].v and cx.mul(yn, kin, "Y0").v <= Y[0].v
    U2 = mul(x2, ZZ, "SUBC_4P_1", "U2")
    Pp = sub_norm(U2, X, "SUBC_6P_1", "P")
    S2 = mul(y2, ZZZ, "SUBC_4P_1", "S2")
    S2n = tuple(cx.neg_lazy(c, "SUBC_2P_1", "-S2") for c in S2)          # the larger of the two forms
    Rr = sub_norm(S2n, Y, "SUBC_2P_1", "R")
    assert Pp[0].v <= 8 * p and Rr[0].v <= 5 * p                          # is_zero_mod_p<7> / <4>
    PP = sqr(Pp, "SUBC_8P_1", "PP")
    Q = mul(X, PP, "SUBC_6P_1", "Q")
    PPP = mul(Pp, PP, "SUBC_8P_1", "PPP")
    ZZ3 = mul(ZZ, PP, "SUBC_2P_1", "ZZ3")
    ZZZ3 = mul(ZZZ, PPP, "SUBC_2P_1", "ZZZ3")
    R2 = sqr(Rr, "SUBC_6P_1", "R2")
    X3 = tuple(cx.sub3_norm(R2[i], PPP[i], Q[i], "SUBC_4P_3", "X3") for i in range(2))
    T = sub_norm(Q, X3, "SUBC_6P_1", "T")
    nR1 = cx.neg_lazy(Rr[1], "SUBC_6P_1", "-R1")
    nY0 = cx.neg_lazy(Y[0], "SUBC_2P_1", "-Y0")
    nY1 = cx.neg_lazy(Y[1], "SUBC_2P_1", "-Y1")
    Y3 = (cx.columns([(Rr[0], T[0]), (nR1, T[1]), (nY0, PPP[0]), (Y[1], PPP[1])], "Y3.c0"),
          cx.columns([(Rr[0], T[1]), (Rr[1], T[0]), (nY0, PPP[1]), (nY1, PPP[0])], "Y3.c1"))
    for nm, new, old in (("X", X3, X), ("Y", Y3, Y), ("ZZ", ZZ3, ZZ), ("ZZZ", ZZZ3, ZZZ)):
        for i in range(2):
            assert new[i].v <= old[i].v, (nm, i, new[i].v / p)
    for nm, v in (("X", X[0]), ("Y", Y[0]), ("ZZ", ZZ[0])):
        assert cx.mul(v, norm(p), nm + " out").v < 2 * p
    return dict(cx.log)


if __name__ == "__main__":
    print("G2", {k: round(v, 3) for k, v in check_madd_g2().items()})
    for s in ("FqParams", "FrParams"):
        print(s, {k: round(v, 3) for k, v in check_madd(s).items()})

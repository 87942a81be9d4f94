// f29.hpp -- BN254 field elements in an unsaturated 9 x 29-bit limb form for the MSM / NTT inner loops (host + gfx950).
//
// Fp (bn254.hpp) keeps 8 saturated 32-bit words in memory and re-slices both operands into 29-bit limbs for every
// product (and packs the result back): ~130 of the ~300 instructions of a multiplication are that re-slicing and the
// carry chains of the surrounding add/sub.  F29 stays in the 29-bit form between operations:
//   * value = sum l[k] * 2^(29k); limbs 0..7 are "normalised" when < 2^29, limb 8 carries whatever is left
//     (values here stay below 2^257, so it never exceeds 2^25);
//   * Montgomery radix R' = 2^261 (nine reduction steps of 29 bits): a product of values < 8p reduces to < 1.4 p, so
//     nothing is ever compared against p;
//   * subtraction adds a multiple of p whose limbs are pre-lifted above the subtrahend's (Pm::SUBC_kP_m: k*p with
//     every low limb >= m*(2^29-1)), so limbs never go negative and no borrow chain exists; the carry sweep that
//     brings limbs back under 2^29 is fused into the same pass where the result feeds a squaring;
//   * unsigned 64-bit column sums: 9 * A * B + 9 * 2^58 + carries < 2^64 needs A * B <= 1.5 * 2^60 for the limb
//     bounds A, B of the two operands -- every call site below states its bounds; tests/test_host_cpu.py re-derives
//     them by interval arithmetic (tests/host/f29_bounds.py) and tests/host/field_check.cpp checks the arithmetic
//     against Fp on random and extremal inputs.
// Reference behaviour served: the G1/G2 multi-scalar multiplications and NTTs of `sunspot prove`
// (noir_circuit/prove_linux.sh:83, scripts/generate_audit.py:680); the algorithm restated by oracle/c/groth16.c.
#pragma once
#include "bn254.hpp"

#if defined(__HIPCC__)
#define SPP_HD_COLD __host__ __device__ __attribute__((noinline))
#else
#define SPP_HD_COLD __attribute__((noinline))
#endif

namespace spp {

template <class Pm>
struct F29 {
  uint32_t l[9];
  using Base = Fp<Pm>;
  typedef uint32_t (*ConstFn)(int);
  static constexpr uint32_t M = (1u << 29) - 1u;
  static constexpr uint32_t INV = Pm::INV32 & M;   // -p^-1 mod 2^29
  static SPP_HD constexpr uint32_t P9(int k) { return Base::P9(k); }

  template <ConstFn C>
  static SPP_HD F29 konst() {
    F29 r;
    SPP_UNROLL for (int i = 0; i < 9; i++) r.l[i] = C(i);
    return r;
  }
  // the same integer as 8 x 32-bit words (no domain change); words must be < 2^256
  static SPP_HD F29 from_words(const uint32_t w[8]) {
    F29 r;
    Base::to9(w, r.l);
    return r;
  }
  // normalised limbs, value < 2^256 -> 8 x 32-bit words
  SPP_HD void to_words(uint32_t w[8]) const {
    SPP_UNROLL for (int k = 0; k < 8; k++) {
      const int bit = 32 * k, i = bit / 29, o = bit % 29;
      uint32_t v = l[i] >> o;
      if (i + 1 < 9) v |= l[i + 1] << (29 - o);
      if (i + 2 < 9 && 58 - o < 32) v |= l[i + 2] << (58 - o);
      w[k] = v;
    }
  }
  // carry sweep: limbs 0..7 < 2^29 afterwards (input limbs < 2^32, value unchanged)
  SPP_HD F29 norm() const {
    F29 r;
    uint32_t carry = 0;
    SPP_UNROLL for (int k = 0; k < 8; k++) {
      const uint32_t t = l[k] + carry;
      r.l[k] = t & M;
      carry = t >> 29;
    }
    r.l[8] = l[8] + carry;
    return r;
  }

  // ---- column products -----------------------------------------------------------------------------
  static SPP_HD void clear(uint64_t (&c)[18]) {
    SPP_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
  }
  static SPP_HD void mac(uint64_t (&c)[18], const F29& a, const F29& b) {
    SPP_UNROLL for (int i = 0; i < 9; i++) {
      SPP_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
  }
  static SPP_HD void mac_sqr(uint64_t (&c)[18], const F29& a) {   // limbs of a < 2^31 (doubled in 32 bits)
    uint32_t d[9];
    SPP_UNROLL for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
    SPP_UNROLL for (int i = 0; i < 9; i++) {
      c[2 * i] += (uint64_t)a.l[i] * a.l[i];
      SPP_UNROLL for (int j = i + 1; j < 9; j++) c[i + j] += (uint64_t)a.l[i] * d[j];
    }
  }
  // Montgomery reduction by R' = 2^261: nine steps clear 29 bits each; result normalised, < columns/R' + p
  static SPP_HD F29 reduce(uint64_t (&c)[18]) {
    SPP_UNROLL for (int k = 0; k < 9; k++) {
      const uint32_t m = ((uint32_t)c[k] * INV) & M;
      SPP_UNROLL for (int j = 0; j < 9; j++) c[k + j] += (uint64_t)m * P9(j);
      c[k + 1] += c[k] >> 29;
    }
    F29 r;
    SPP_UNROLL for (int k = 0; k < 8; k++) {
      r.l[k] = (uint32_t)c[9 + k] & M;
      c[10 + k] += c[9 + k] >> 29;
    }
    r.l[8] = (uint32_t)c[17];
    return r;
  }
  friend SPP_HD F29 operator*(const F29& a, const F29& b) {
    uint64_t c[18];
    clear(c);
    mac(c, a, b);
    return reduce(c);
  }
  SPP_HD F29 sqr() const {
    uint64_t c[18];
    clear(c);
    mac_sqr(c, *this);
    return reduce(c);
  }
  // a*b + c*d with one reduction (limb bounds: 9*(A*B + C*D) + 9*2^58 < 2^64)
  static SPP_HD F29 mul2(const F29& a, const F29& b, const F29& cc, const F29& d) {
    uint64_t c[18];
    clear(c);
    mac(c, a, b);
    mac(c, cc, d);
    return reduce(c);
  }

  // ---- additive operations (C = lifted multiple of p; see gen_consts.py `lifted`) ---------------------
  // a - b + C, limbs left as they fall (each < a + C)
  template <ConstFn C>
  static SPP_HD F29 sub_lazy(const F29& a, const F29& b) {
    F29 r;
    SPP_UNROLL for (int k = 0; k < 9; k++) r.l[k] = a.l[k] + (C(k) - b.l[k]);
    return r;
  }
  template <ConstFn C>
  static SPP_HD F29 neg_lazy(const F29& b) {
    F29 r;
    SPP_UNROLL for (int k = 0; k < 9; k++) r.l[k] = C(k) - b.l[k];
    return r;
  }
  // a - b + C with the carry sweep fused in
  template <ConstFn C>
  static SPP_HD F29 sub_norm(const F29& a, const F29& b) {
    F29 r;
    uint32_t carry = 0;
    SPP_UNROLL for (int k = 0; k < 8; k++) {
      const uint32_t t = a.l[k] - b.l[k] + C(k) + carry;
      r.l[k] = t & M;
      carry = t >> 29;
    }
    r.l[8] = a.l[8] - b.l[8] + C(8) + carry;
    return r;
  }
  // a - b - 2c + C, normalised
  template <ConstFn C>
  static SPP_HD F29 sub3_norm(const F29& a, const F29& b, const F29& c2) {
    F29 r;
    uint32_t carry = 0;
    SPP_UNROLL for (int k = 0; k < 8; k++) {
      const uint32_t t = a.l[k] - b.l[k] - (c2.l[k] << 1) + C(k) + carry;
      r.l[k] = t & M;
      carry = t >> 29;
    }
    r.l[8] = a.l[8] - b.l[8] - (c2.l[8] << 1) + C(8) + carry;
    return r;
  }
  friend SPP_HD F29 add_lazy(const F29& a, const F29& b) {
    F29 r;
    SPP_UNROLL for (int k = 0; k < 9; k++) r.l[k] = a.l[k] + b.l[k];
    return r;
  }
  friend SPP_HD F29 add_norm(const F29& a, const F29& b) {
    F29 r;
    uint32_t carry = 0;
    SPP_UNROLL for (int k = 0; k < 8; k++) {
      const uint32_t t = a.l[k] + b.l[k] + carry;
      r.l[k] = t & M;
      carry = t >> 29;
    }
    r.l[8] = a.l[8] + b.l[8] + carry;
    return r;
  }

  // normalised value == k*p for some 0 <= k <= KMAX ?  The normalised form is unique, so this is a limb comparison;
  // the low limbs of 0, p, .., KMAX*p are distinct (p odd), so the low limb both filters (all but ~KMAX/2^29 of the
  // non-zero cases leave here) and names the only multiple left to compare.
  template <uint32_t KMAX>
  SPP_HD bool is_zero_mod_p() const {
    uint32_t kk = 0xffffffffu;
    SPP_UNROLL for (uint32_t k = 0; k <= KMAX; k++) {
      if (l[0] == ((k * P9(0)) & M)) kk = k;
    }
    if (kk == 0xffffffffu) return false;
    uint32_t diff = 0, carry = 0;
    SPP_UNROLL for (int i = 0; i < 8; i++) {
      const uint32_t t = kk * P9(i) + carry;     // KMAX * 2^29 < 2^32
      diff |= l[i] ^ (t & M);
      carry = t >> 29;
    }
    diff |= l[8] ^ (kk * P9(8) + carry);
    return diff == 0;
  }

  // ---- domain changes ---------------------------------------------------------------------------------
  // Fp (x*R, words) -> x*R'
  static SPP_HD F29 from_fp(const Base& a) { return from_words(a.l) * konst<Pm::K29_IN>(); }
  // x*R' (value < 8p) -> Fp in [0, 2p)
  SPP_HD Base to_fp() const {
    uint32_t w[8];
    SPP_UNROLL for (int i = 0; i < 8; i++) w[i] = Pm::ONE(i);
    const F29 t = *this * from_words(w);   // x*R' * R / R' = x*R, < 1.05 p
    Base r;
    t.to_words(r.l);
    return r;
  }
  // zz*R'^2/R (the scaled domain of XYZZ29::ZZ/ZZZ) -> Fp
  SPP_HD Base scaled_to_fp() const {
    const F29 t = *this * konst<Pm::K29_ZZ_OUT>();
    Base r;
    t.to_words(r.l);
    return r;
  }
};

// --------------------------------------------------------------------------------------------------------------
// XYZZ accumulator over F29 for "acc += table point" (mixed addition, 8M + 2S with 9 reductions).
//   X, Y      : x*R', y*R'                     (X normalised < 5.1 p, Y < 1.2 p)
//   ZZ, ZZZ   : zz*R'^2/R, zzz*R'^2/R          (< 1.1 p): a table coordinate x2*R (plain Fp words, < 2p) times ZZ
//               gives x2*zz*R' directly, so the table stays in the Fp format every other kernel uses.
// Limb/value bounds per line are in the comments (A x B = limb bounds of the two mul operands, in units of 2^29).
// --------------------------------------------------------------------------------------------------------------
template <class Pm>
struct XYZZ29 {
  using F = F29<Pm>;
  using B = Fp<Pm>;
  F X, Y, ZZ, ZZZ;
  bool inf;

  static SPP_HD XYZZ29 infinity() {
    XYZZ29 r;
    SPP_UNROLL for (int i = 0; i < 9; i++) r.X.l[i] = r.Y.l[i] = r.ZZ.l[i] = r.ZZZ.l[i] = 0;
    r.inf = true;
    return r;
  }
  SPP_HD XYZZ<B> to_xyzz() const {
    if (inf) return XYZZ<B>::infinity();
    return {X.to_fp(), Y.to_fp(), ZZ.scaled_to_fp(), ZZZ.scaled_to_fp()};
  }
  static SPP_HD XYZZ29 from_xyzz(const XYZZ<B>& q) {
    XYZZ29 r;
    r.inf = q.is_inf();
    r.X = F::from_fp(q.X);
    r.Y = F::from_fp(q.Y);
    // zz*R -> zz*R' -> zz*R'^2/R (K29_IN as a plain factor); rare path (doubling fallback)
    r.ZZ = F::from_fp(q.ZZ) * F::template konst<Pm::K29_IN>();
    r.ZZZ = F::from_fp(q.ZZZ) * F::template konst<Pm::K29_IN>();
    return r;
  }

  // this += (x2, +-y2); e = table entry (Fp words, not infinity)
  SPP_HD void madd(const Affine<B>& e, bool negate) {
    F x2 = F::from_words(e.x.l);                                         // limbs < 1, value < 2p
    F y2 = F::from_words(e.y.l);
    if (negate) y2 = F::template neg_lazy<Pm::SUBC_4P_1>(y2);            // limbs < 2 (2^30), value <= 4p
    if (inf) {
      X = x2 * F::template konst<Pm::K29_IN>();
      Y = y2 * F::template konst<Pm::K29_IN>();
      ZZ = F::template konst<Pm::K29_IN>();
      ZZZ = ZZ;
      inf = false;
      return;
    }
    // statement order keeps few values alive at once (x2 dies first, then U2, P, PP, ...)
    const F U2 = x2 * ZZ;                                                // 1 x 1
    const F Pp = F::template sub_norm<Pm::SUBC_6P_1>(U2, X);             // normalised, < 7.1 p
    if (Pp.template is_zero_mod_p<7>()) {                                // same x: doubling or cancellation (rare)
      const F S2 = y2 * ZZZ;
      const F Rr = F::template sub_norm<Pm::SUBC_2P_1>(S2, Y);
      if (Rr.template is_zero_mod_p<3>()) {
        XYZZ<B> t = to_xyzz();
        t.dbl_inplace();
        const XYZZ29 d = from_xyzz(t);
        X = d.X;
        Y = d.Y;
        ZZ = d.ZZ;
        ZZZ = d.ZZZ;
      } else {
        inf = true;
      }
      return;
    }
    const F PP = Pp.sqr();                                               // 1 x 1 -> < 1.3 p
    const F Q = X * PP;                                                  // < 1.05 p
    const F PPP = Pp * PP;                                               // < 1.06 p
    ZZ = ZZ * PP;
    const F S2 = y2 * ZZZ;                                               // 2 x 1
    const F Rr = F::template sub_norm<Pm::SUBC_2P_1>(S2, Y);             // normalised, < 3.1 p
    ZZZ = ZZZ * PPP;
    const F R2 = Rr.sqr();                                               // < 1.06 p
    X = F::template sub3_norm<Pm::SUBC_4P_3>(R2, PPP, Q);                // normalised, < 5.1 p
    const F T = F::template sub_lazy<Pm::SUBC_6P_1>(Q, X);               // limbs < 3, < 7.1 p
    const F Yn = F::template neg_lazy<Pm::SUBC_2P_1>(Y);                 // limbs < 2, <= 2p
    Y = F::mul2(Rr, T, Yn, PPP);                                         // 1x3 + 2x1 -> < 1.2 p
  }
};

// --------------------------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2+1) over F29 and the G2 accumulator.  Every component is a sum of products reduced once:
//   mul : c0 = a0*b0 + (C - a1)*b1,  c1 = a0*b1 + a1*b0            (4 products, 2 reductions)
//   sqr : c0 = (a0 + a1)*(a0 - a1 + C),  c1 = (2*a0)*a1            (2 products, 2 reductions)
// Operands are normalised (limbs < 2^29); the negations/sums made on the fly have limbs < 2^30 or 3*2^29, which keeps
// the column sums under 2^64 (sum of limb-bound products <= 6 * 2^58 per column term: tests/host/f29_bounds.py).
// CNEG_x = lifted multiple of p that dominates the component being negated (template parameter per call site).
// --------------------------------------------------------------------------------------------------------------
struct F29x2 {
  using F = F29<FqParams>;
  using Pm = FqParams;
  F c0, c1;

  static SPP_HD F29x2 from_words(const Fq2& a) { return {F::from_words(a.c0.l), F::from_words(a.c1.l)}; }
  // (a0 + a1 u)(b0 + b1 u); CA dominates a1
  template <F::ConstFn CA>
  static SPP_HD F29x2 mul(const F29x2& a, const F29x2& b) {
    const F na1 = F::template neg_lazy<CA>(a.c1);
    return {F::mul2(a.c0, b.c0, na1, b.c1), F::mul2(a.c0, b.c1, a.c1, b.c0)};
  }
  // by a real constant (components scale independently)
  SPP_HD F29x2 mul_real(const F& k) const { return {c0 * k, c1 * k}; }
  // CA dominates a1
  template <F::ConstFn CA>
  SPP_HD F29x2 sqr() const {
    const F s = add_lazy(c0, c1);
    const F d = F::template sub_lazy<CA>(c0, c1);
    const F t = add_lazy(c0, c0);
    return {s * d, t * c1};
  }
  template <F::ConstFn C>
  static SPP_HD F29x2 sub_norm(const F29x2& a, const F29x2& b) {
    return {F::template sub_norm<C>(a.c0, b.c0), F::template sub_norm<C>(a.c1, b.c1)};
  }
  template <F::ConstFn C>
  static SPP_HD F29x2 sub3_norm(const F29x2& a, const F29x2& b, const F29x2& c2) {
    return {F::template sub3_norm<C>(a.c0, b.c0, c2.c0), F::template sub3_norm<C>(a.c1, b.c1, c2.c1)};
  }
  template <F::ConstFn C>
  static SPP_HD F29x2 neg_lazy(const F29x2& a) {
    return {F::template neg_lazy<C>(a.c0), F::template neg_lazy<C>(a.c1)};
  }
  template <uint32_t KMAX>
  SPP_HD bool is_zero_mod_p() const {
    return c0.template is_zero_mod_p<KMAX>() && c1.template is_zero_mod_p<KMAX>();
  }
  SPP_HD Fq2 to_fp() const { return {c0.to_fp(), c1.to_fp()}; }
  SPP_HD Fq2 scaled_to_fp() const { return {c0.scaled_to_fp(), c1.scaled_to_fp()}; }
  static SPP_HD F29x2 from_fp(const Fq2& a) { return {F::from_fp(a.c0), F::from_fp(a.c1)}; }
};

// G2 accumulator: same formulas and domains as XYZZ29 (X, Y in the R' domain, ZZ/ZZZ scaled by R'^2/R).
// Value bounds per component: X < 5.6 p, Y < 1.5 p, ZZ/ZZZ < 1.3 p (certificate: check_madd_g2).
struct XYZZ29G2 {
  using E = F29x2;
  using F = F29<FqParams>;
  using Pm = FqParams;
  E X, Y, ZZ, ZZZ;
  bool inf;

  static SPP_HD XYZZ29G2 infinity() {
    XYZZ29G2 r;
    SPP_UNROLL for (int i = 0; i < 9; i++) {
      r.X.c0.l[i] = r.X.c1.l[i] = r.Y.c0.l[i] = r.Y.c1.l[i] = 0;
      r.ZZ.c0.l[i] = r.ZZ.c1.l[i] = r.ZZZ.c0.l[i] = r.ZZZ.c1.l[i] = 0;
    }
    r.inf = true;
    return r;
  }
  SPP_HD XYZZ<Fq2> to_xyzz() const {
    if (inf) return XYZZ<Fq2>::infinity();
    return {X.to_fp(), Y.to_fp(), ZZ.scaled_to_fp(), ZZZ.scaled_to_fp()};
  }
  static SPP_HD XYZZ29G2 from_xyzz(const XYZZ<Fq2>& q) {
    XYZZ29G2 r;
    r.inf = q.is_inf();
    const F k = F::template konst<Pm::K29_IN>();
    r.X = E::from_fp(q.X);
    r.Y = E::from_fp(q.Y);
    r.ZZ = E::from_fp(q.ZZ).mul_real(k);
    r.ZZZ = E::from_fp(q.ZZZ).mul_real(k);
    return r;
  }

  // this += (x2, +-y2); e = table entry (Fq2 words, not infinity)
  SPP_HD void madd(const Affine<Fq2>& e, bool negate) {
    const E x2 = E::from_words(e.x);                                     // limbs < 1, value < 2p
    const E y2 = E::from_words(e.y);
    if (inf) {
      const F k = F::template konst<Pm::K29_IN>();
      X = x2.mul_real(k);
      const E yn = E::template neg_lazy<Pm::SUBC_4P_1>(y2);               // limbs < 2, <= 4p
      E ys;
      SPP_UNROLL for (int i = 0; i < 9; i++) {
        ys.c0.l[i] = negate ? yn.c0.l[i] : y2.c0.l[i];
        ys.c1.l[i] = negate ? yn.c1.l[i] : y2.c1.l[i];
      }
      Y = ys.mul_real(k);                                                // < 1.03 p
      SPP_UNROLL for (int i = 0; i < 9; i++) {
        ZZ.c0.l[i] = k.l[i];
        ZZ.c1.l[i] = 0;
      }
      ZZZ = ZZ;
      inf = false;
      return;
    }
    const E U2 = E::template mul<Pm::SUBC_4P_1>(x2, ZZ);                 // < 1.1 p
    const E Pp = E::template sub_norm<Pm::SUBC_6P_1>(U2, X);             // normalised, < 7.1 p
    E S2 = E::template mul<Pm::SUBC_4P_1>(y2, ZZZ);                      // < 1.1 p
    if (negate) S2 = E::template neg_lazy<Pm::SUBC_2P_1>(S2);            // limbs < 2, <= 2p
    const E Rr = E::template sub_norm<Pm::SUBC_2P_1>(S2, Y);             // normalised, < 4.1 p
    if (Pp.template is_zero_mod_p<7>()) {                                // same x: doubling or cancellation (rare)
      if (Rr.template is_zero_mod_p<4>()) {
        XYZZ<Fq2> t = to_xyzz();
        t.dbl_inplace();
        const XYZZ29G2 d = from_xyzz(t);
        X = d.X;
        Y = d.Y;
        ZZ = d.ZZ;
        ZZZ = d.ZZZ;
      } else {
        inf = true;
      }
      return;
    }
    const E PP = Pp.template sqr<Pm::SUBC_8P_1>();                       // < 2.3 p
    const E Q = E::template mul<Pm::SUBC_6P_1>(X, PP);                   // < 1.2 p
    const E PPP = E::template mul<Pm::SUBC_8P_1>(Pp, PP);                // < 1.3 p
    ZZ = E::template mul<Pm::SUBC_2P_1>(ZZ, PP);
    ZZZ = E::template mul<Pm::SUBC_2P_1>(ZZZ, PPP);
    const E R2 = Rr.template sqr<Pm::SUBC_6P_1>();                       // < 1.6 p
    X = E::template sub3_norm<Pm::SUBC_4P_3>(R2, PPP, Q);                // normalised, < 5.6 p
    const E T = E::template sub_norm<Pm::SUBC_6P_1>(Q, X);               // normalised, < 7.2 p
    // Y3 = R*T - Y*PPP: eight products, two reductions
    const F nR1 = F::template neg_lazy<Pm::SUBC_6P_1>(Rr.c1);
    const F nY0 = F::template neg_lazy<Pm::SUBC_2P_1>(Y.c0);
    const F nY1 = F::template neg_lazy<Pm::SUBC_2P_1>(Y.c1);
    uint64_t c[18];
    F::clear(c);
    F::mac(c, Rr.c0, T.c0);
    F::mac(c, nR1, T.c1);
    F::mac(c, nY0, PPP.c0);
    F::mac(c, Y.c1, PPP.c1);
    const F y0 = F::reduce(c);
    F::clear(c);
    F::mac(c, Rr.c0, T.c1);
    F::mac(c, Rr.c1, T.c0);
    F::mac(c, nY0, PPP.c1);
    F::mac(c, nY1, PPP.c0);
    Y.c0 = y0;
    Y.c1 = F::reduce(c);
  }
};

}  // namespace spp

// BN254 field and curve arithmetic for gfx950 (and for the host side of libspp).
//
// 254-bit prime fields in Montgomery form on 8 x 32-bit limbs: the natural word of the CDNA4 VALU
// (v_mad_u64_u32 does 32x32+64 -> 64 in one instruction; there is no 64x64 multiplier). Everything is
// written so that, after full unrolling, limbs live in VGPRs and the modulus limbs fold to literals.
// The same code compiles for the host (g++) -- used there only for one-time work (circuit constants,
// setup scalars); the proving hot path runs it on the GPU.
//
// Replaces: the gnark-crypto fr/fp/G1/G2 arithmetic that `sunspot prove`
// (client/proof.helper.ts:64 of the reference) executes on the CPU.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "bn254_consts.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SPP_HD __host__ __device__ __forceinline__
#define SPP_UNROLL _Pragma("unroll")
#else
#define SPP_HD inline
#define SPP_UNROLL
#endif

namespace spp {

// add / subtract with carry: clang's multiprecision builtins lower to v_addc_co_u32 / v_subb_co_u32 chains
// on gfx950 (a 64-bit emulation costs ~4x the instructions); plain C for other host compilers.
SPP_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t& carry) {
#if defined(__clang__)
  unsigned co;
  uint32_t r = __builtin_addc(a, b, carry, &co);
  carry = co;
  return r;
#else
  uint64_t t = (uint64_t)a + b + carry;
  carry = (uint32_t)(t >> 32);
  return (uint32_t)t;
#endif
}
SPP_HD uint32_t subb32(uint32_t a, uint32_t b, uint32_t& borrow) {
#if defined(__clang__)
  unsigned bo;
  uint32_t r = __builtin_subc(a, b, borrow, &bo);
  borrow = bo;
  return r;
#else
  uint64_t t = (uint64_t)a - b - borrow;
  borrow = (uint32_t)(t >> 63);
  return (uint32_t)t;
#endif
}

// --------------------------------------------------------------------------------------------------
// Fp<Params>: element of GF(p) in Montgomery form (value * 2^256 mod p), kept in the redundant range [0, 2p)
// ("almost Montgomery"): with p < 2^254 a product of two such values reduces to < 1.76 p without the final
// conditional subtraction, so mul/sqr never compare against p; add/sub fold back below 2p; only equality tests and
// canonical output (to_canonical) finish the reduction.
// --------------------------------------------------------------------------------------------------
template <class Pm>
struct Fp {
  uint32_t l[8];

  static SPP_HD Fp zero() {
    Fp r;
    SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
  }
  static SPP_HD Fp one() {
    Fp r;
    SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = Pm::ONE(i);
    return r;
  }
  static SPP_HD Fp r2() {
    Fp r;
    SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = Pm::R2(i);
    return r;
  }
  static SPP_HD Fp r3() {
    Fp r;
    SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = Pm::R3(i);
    return r;
  }
  static SPP_HD constexpr uint32_t modulus_word(int i) { return Pm::MOD(i); }
  SPP_HD bool is_zero() const {   // value is 0 or p
    uint32_t o = 0, q = 0;
    SPP_UNROLL for (int i = 0; i < 8; i++) {
      o |= l[i];
      q |= l[i] ^ Pm::MOD(i);
    }
    return o == 0 || q == 0;
  }
  SPP_HD Fp canonical() const {   // same element, limbs < p
    Fp r = *this;
    cond_sub(r.l);
    return r;
  }
  SPP_HD bool operator==(const Fp& b) const {
    const Fp x = canonical(), y = b.canonical();
    uint32_t o = 0;
    SPP_UNROLL for (int i = 0; i < 8; i++) o |= x.l[i] ^ y.l[i];
    return o == 0;
  }
  SPP_HD bool operator!=(const Fp& b) const { return !(*this == b); }

  // raw limbs >= modulus ?
  static SPP_HD bool geq_mod(const uint32_t* a) {
    // compute a - p, look at the borrow
    uint32_t br = 0;
    SPP_UNROLL for (int i = 0; i < 8; i++) (void)subb32(a[i], Pm::MOD(i), br);
    return br == 0;
  }
  // r = a - p if a >= p (a < 2p)
  static SPP_HD void cond_sub(uint32_t* a) {
    uint32_t t[8];
    uint32_t br = 0;
    SPP_UNROLL for (int i = 0; i < 8; i++) t[i] = subb32(a[i], Pm::MOD(i), br);
    SPP_UNROLL for (int i = 0; i < 8; i++) a[i] = br ? a[i] : t[i];
  }

  // a -= 2p if a >= 2p (a < 4p < 2^256)
  static SPP_HD void cond_sub_2p(uint32_t* a) {
    uint32_t t[8];
    uint32_t br = 0;
    SPP_UNROLL for (int i = 0; i < 8; i++) t[i] = subb32(a[i], Pm::TWOP(i), br);
    SPP_UNROLL for (int i = 0; i < 8; i++) a[i] = br ? a[i] : t[i];
  }

  friend SPP_HD Fp operator+(const Fp& a, const Fp& b) {
    Fp r;
    uint32_t c = 0;
    SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = addc32(a.l[i], b.l[i], c);
    // a, b < 2p < 2^255 so a+b < 2^256: no carry out of limb 7
    cond_sub_2p(r.l);
    return r;
  }
  friend SPP_HD Fp operator-(const Fp& a, const Fp& b) {
    Fp r;
    uint32_t br = 0;
    SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = subb32(a.l[i], b.l[i], br);
    // add 2p back when the difference went negative (mask form: no divergent branch): result in [0, 2p)
    const uint32_t mask = 0u - br;
    uint32_t c = 0;
    SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = addc32(r.l[i], Pm::TWOP(i) & mask, c);
    return r;
  }
  SPP_HD Fp neg() const {
    if (is_zero()) return zero();
    Fp r;                                  // 2p - a in (0, 2p)
    uint32_t br = 0;
    SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = subb32(Pm::TWOP(i), l[i], br);
    return r;
  }
  SPP_HD Fp dbl() const { return *this + *this; }

  // ---- Montgomery multiplication -----------------------------------------------------------------
  // Measured on gfx950 (tests/micro/valu_rates.hip): v_mad_u64_u32 issues at the same rate as v_mul_lo_u32 or a
  // 64-bit add (~4.5 cycles per wave-instruction), so what matters is the instruction count around the 32x32
  // products.  A word-serial CIOS on saturated 32-bit limbs needs a carry fix-up (and register-pair shuffles)
  // after every product: ~600 instructions.  Instead the operands are re-sliced into 9 limbs of 29 bits: the 81
  // partial products then accumulate into 17 independent 64-bit columns with one v_mad_u64_u32 each and no
  // carries at all (9 * 2^58 * 2 < 2^63), and the Montgomery reduction adds m_k * p the same way.  Eight
  // reduction steps clear 29 bits each and a ninth clears 24, so the result is still a*b/2^256: the memory format
  // and every constant stay those of the 8 x 32-bit representation.
  static constexpr uint32_t M29 = (1u << 29) - 1u;
  static SPP_HD constexpr uint32_t P9(int k) {
    // limb k (29 bits) of the modulus
    const int bit = 29 * k, w = bit / 32, o = bit % 32;
    uint64_t v = Pm::MOD(w);
    if (w + 1 < 8) v |= (uint64_t)Pm::MOD(w + 1) << 32;
    return (uint32_t)(v >> o) & M29;
  }
  static SPP_HD void to9(const uint32_t w[8], uint32_t o[9]) {
    o[0] = w[0] & M29;
    o[1] = ((w[0] >> 29) | (w[1] << 3)) & M29;
    o[2] = ((w[1] >> 26) | (w[2] << 6)) & M29;
    o[3] = ((w[2] >> 23) | (w[3] << 9)) & M29;
    o[4] = ((w[3] >> 20) | (w[4] << 12)) & M29;
    o[5] = ((w[4] >> 17) | (w[5] << 15)) & M29;
    o[6] = ((w[5] >> 14) | (w[6] << 18)) & M29;
    o[7] = ((w[6] >> 11) | (w[7] << 21)) & M29;
    o[8] = w[7] >> 8;
    hide24(o[8]);
  }
  // Values the compiler can prove to be < 2^24 make it select v_mul_u32_u24 / v_mul_hi_u32_u24 and drop the
  // masking AND; on gfx950 (ROCm 7.2) the high half then came back wrong (tests/micro/device_arith_check.hip,
  // dbg_mul.hip).  An empty asm hides the range so the product stays a v_mad_u64_u32.
  static SPP_HD void hide24(uint32_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#else
    (void)v;
#endif
  }
  // c[0..17]: column sums of the double-width product -> reduced, repacked result
  static SPP_HD Fp reduce_columns(uint64_t (&c)[18]) {
    SPP_UNROLL for (int k = 0; k < 8; k++) {
      const uint32_t m = ((uint32_t)c[k] * Pm::INV32) & M29;
      SPP_UNROLL for (int j = 0; j < 9; j++) c[k + j] += (uint64_t)m * P9(j);
      c[k + 1] += c[k] >> 29;
    }
    {
      uint32_t m = ((uint32_t)c[8] * Pm::INV32) & ((1u << 24) - 1u);
      hide24(m);
      SPP_UNROLL for (int j = 0; j < 9; j++) c[8 + j] += (uint64_t)m * P9(j);
    }
    // result = (c[8] >> 24) + c[9]*2^5 + c[10]*2^34 + ... + c[17]*2^237 ; normalise the columns to 29 bits
    const uint32_t lo5 = (uint32_t)(c[8] >> 24) & 31u;
    c[9] += c[8] >> 29;
    uint32_t n[9];
    SPP_UNROLL for (int k = 0; k < 8; k++) {
      n[k] = (uint32_t)c[9 + k] & M29;
      c[10 + k] += c[9 + k] >> 29;
    }
    n[8] = (uint32_t)c[17];
    Fp r;
    r.l[0] = lo5 | (n[0] << 5);
    r.l[1] = (n[0] >> 27) | (n[1] << 2) | (n[2] << 31);
    r.l[2] = (n[2] >> 1) | (n[3] << 28);
    r.l[3] = (n[3] >> 4) | (n[4] << 25);
    r.l[4] = (n[4] >> 7) | (n[5] << 22);
    r.l[5] = (n[5] >> 10) | (n[6] << 19);
    r.l[6] = (n[6] >> 13) | (n[7] << 16);
    r.l[7] = (n[7] >> 16) | (n[8] << 13);
    return r;   // < 1.76 p for inputs < 2p (sum-of-two-products callers: see Fq2): no conditional subtraction
  }
  friend SPP_HD Fp operator*(const Fp& a, const Fp& b) {
    uint32_t a9[9], b9[9];
    to9(a.l, a9);
    to9(b.l, b9);
    uint64_t c[18];
    SPP_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
    SPP_UNROLL for (int i = 0; i < 9; i++) {
      SPP_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a9[i] * b9[j];
    }
    return reduce_columns(c);
  }
  SPP_HD Fp sqr() const {
    uint32_t a9[9], d9[9];
    to9(l, a9);
    SPP_UNROLL for (int i = 0; i < 9; i++) d9[i] = a9[i] << 1;
    uint64_t c[18];
    SPP_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
    SPP_UNROLL for (int i = 0; i < 9; i++) {
      c[2 * i] += (uint64_t)a9[i] * a9[i];
      SPP_UNROLL for (int j = i + 1; j < 9; j++) c[i + j] += (uint64_t)a9[i] * d9[j];
    }
    return reduce_columns(c);
  }

  // multiply by a small unsigned constant (k < 2^16) via double-and-add on the bits of k
  SPP_HD Fp mul_small(uint32_t k) const {
    Fp acc = zero();
    Fp base = *this;
    while (k) {
      if (k & 1) acc = acc + base;
      base = base.dbl();
      k >>= 1;
    }
    return acc;
  }

  // canonical (non-Montgomery) limbs
  SPP_HD void to_canonical(uint32_t out[8]) const {
    Fp o;
    SPP_UNROLL for (int i = 0; i < 8; i++) o.l[i] = (i == 0);
    Fp c = *this * o;  // a*R * 1 / R = a, <= p
    cond_sub(c.l);
    SPP_UNROLL for (int i = 0; i < 8; i++) out[i] = c.l[i];
  }
  // from canonical limbs (must be < p)
  static SPP_HD Fp from_canonical(const uint32_t in[8]) {
    Fp a;
    SPP_UNROLL for (int i = 0; i < 8; i++) a.l[i] = in[i];
    return a * r2();
  }
  // from arbitrary 256-bit limbs (reduced mod p; input < 2^256 < 6p)
  static SPP_HD Fp from_u256(const uint32_t in[8]) {
    Fp a;
    SPP_UNROLL for (int i = 0; i < 8; i++) a.l[i] = in[i];
    for (int k = 0; k < 6; k++) cond_sub(a.l);  // 2^256 < 6p
    return a * r2();
  }
  static SPP_HD Fp from_u64(uint64_t v) {
    uint32_t c[8];
    SPP_UNROLL for (int i = 0; i < 8; i++) c[i] = 0;
    c[0] = (uint32_t)v;
    c[1] = (uint32_t)(v >> 32);
    return from_canonical(c);
  }
  // 32-byte big-endian canonical
  SPP_HD void to_bytes_be(uint8_t out[32]) const {
    uint32_t c[8];
    to_canonical(c);
    SPP_UNROLL for (int i = 0; i < 8; i++) {
      uint32_t w = c[7 - i];
      out[4 * i + 0] = (uint8_t)(w >> 24);
      out[4 * i + 1] = (uint8_t)(w >> 16);
      out[4 * i + 2] = (uint8_t)(w >> 8);
      out[4 * i + 3] = (uint8_t)w;
    }
  }
  static SPP_HD Fp from_bytes_be(const uint8_t in[32]) {
    uint32_t c[8];
    SPP_UNROLL for (int i = 0; i < 8; i++) {
      c[7 - i] = ((uint32_t)in[4 * i] << 24) | ((uint32_t)in[4 * i + 1] << 16) | ((uint32_t)in[4 * i + 2] << 8) |
                 (uint32_t)in[4 * i + 3];
    }
    return from_u256(c);
  }

  // Inverse by the binary extended Euclid of Kaliski ("almost Montgomery inverse", 1995), several bits per step: with u = p,
  // v = a, r = 0, s = 1 the loop keeps  a*r = -v*2^k  and  a*s = u*2^k  (mod p), u and v odd; the larger of u, v is replaced by
  // their difference with its trailing zeros shifted out, the opposite cofactor is shifted up as far.  About 180 subtract-and-
  // shift steps of ~60 instructions against 380 multiplications of ~300 for a^(p-2): one lane alone (the single-proof path:
  // to_affine, the recipient inverse, the two Grumpkin batch inversions, ...) gets its inverse in ~30 us instead of ~240.
  // Ends with x = a^-1 * 2^k (254 <= k <= 508) for the WORD a = (value)*R, i.e. x = value^-1 * 2^k / R; two products with
  // powers of two (2^(512-k) in two halves, each < 2^130 < p) and one with R^3 give value^-1 * R.   0 -> 0 as a^(p-2) does.
  SPP_HD Fp inv() const {
    uint32_t u[8], v[8], r[8], s[8];
    SPP_UNROLL for (int i = 0; i < 8; i++) {
      u[i] = Pm::MOD(i);
      v[i] = l[i];
      r[i] = 0;
      s[i] = (i == 0);
    }
    cond_sub(v);
    {
      uint32_t o = 0;
      SPP_UNROLL for (int i = 0; i < 8; i++) o |= v[i];
      if (o == 0) return zero();
    }
    auto shr = [](uint32_t (&w)[8], uint32_t t) {   // 0 < t < 32
      SPP_UNROLL for (int i = 0; i < 7; i++) w[i] = (uint32_t)((((uint64_t)w[i + 1] << 32) | w[i]) >> t);
      w[7] >>= t;
    };
    auto shl = [](uint32_t (&w)[8], uint32_t t) {   // 0 < t < 32
      SPP_UNROLL for (int i = 7; i > 0; i--) w[i] = (uint32_t)((((uint64_t)w[i] << 32) | w[i - 1]) >> (32 - t));
      w[0] <<= t;
    };
    auto add = [](uint32_t (&a)[8], const uint32_t (&b)[8]) {
      uint32_t c = 0;
      SPP_UNROLL for (int i = 0; i < 8; i++) a[i] = addc32(a[i], b[i], c);
    };
    // x >> (its trailing zeros), y << as many; returns the count.  x != 0.
    auto strip = [&](uint32_t (&x)[8], uint32_t (&y)[8]) {
      uint32_t total = 0;
      while ((x[0] & 1u) == 0) {
        const uint32_t t = x[0] ? (uint32_t)__builtin_ctz(x[0]) : 31u;
        shr(x, t);
        shl(y, t);
        total += t;
      }
      return total;
    };
    uint32_t k = strip(v, r);   // r = 0: only v moves
    for (;;) {
      uint32_t d[8];
      uint32_t br = 0;
      SPP_UNROLL for (int i = 0; i < 8; i++) d[i] = subb32(u[i], v[i], br);
      if (br) {          // v > u:  v <- (v - u) / 2^t, s <- s + r, r <- r * 2^t
        uint32_t c = 1;
        SPP_UNROLL for (int i = 0; i < 8; i++) d[i] = addc32(~d[i], 0u, c);
        add(s, r);
        SPP_UNROLL for (int i = 0; i < 8; i++) v[i] = d[i];
        k += strip(v, r);
      } else {
        uint32_t o = 0;
        SPP_UNROLL for (int i = 0; i < 8; i++) o |= d[i];
        if (o == 0) {    // u == v (== gcd == 1): the last step, v <- 0, s <- s + r, r <- 2r
          shl(r, 1);
          k += 1;
          break;
        }
        add(r, s);       // u > v:  u <- (u - v) / 2^t, r <- r + s, s <- s * 2^t
        SPP_UNROLL for (int i = 0; i < 8; i++) u[i] = d[i];
        k += strip(u, s);
      }
    }
    // r < 2p holds a * r = -2^k: x = p - (r mod p)
    cond_sub(r);
    Fp x;
    {
      uint32_t brw = 0;
      SPP_UNROLL for (int i = 0; i < 8; i++) x.l[i] = subb32(Pm::MOD(i), r[i], brw);
    }
    const uint32_t m = 512u - k, m1 = m >> 1, m2 = m - m1;   // 4 <= m <= 258
    Fp p1 = zero(), p2 = zero();
    SPP_UNROLL for (int i = 0; i < 8; i++) {
      p1.l[i] = (m1 >> 5) == (uint32_t)i ? 1u << (m1 & 31) : 0u;
      p2.l[i] = (m2 >> 5) == (uint32_t)i ? 1u << (m2 & 31) : 0u;
    }
    return ((x * p1) * p2) * r3();
  }
  // a^(p-2) (Fermat); uniform control flow: the exponent is a compile-time constant.  Kept as the cross-check of inv().
  SPP_HD Fp inv_fermat() const {
    Fp result = one();
    Fp base = *this;
    for (int w = 0; w < 8; w++) {
      uint32_t e = Pm::MODM2(w);
      for (int b = 0; b < 32; b++) {
        if ((e >> b) & 1) result = result * base;
        base = base.sqr();
      }
    }
    return result;
  }
  SPP_HD Fp pow_u64(uint64_t e) const {
    Fp result = one();
    Fp base = *this;
    while (e) {
      if (e & 1) result = result * base;
      base = base.sqr();
      e >>= 1;
    }
    return result;
  }
};

// true iff the 32-byte big-endian integer is below the modulus: the canonical encodings gnark's readers accept
// (a public-witness word >= r, or a point coordinate >= q, is refused, never reduced -- otherwise v and v + r would be two
// byte strings for one nullifier)
template <class P>
SPP_HD bool be_is_canonical(const uint8_t in[32]) {
  for (int i = 0; i < 8; i++) {
    const uint32_t w = ((uint32_t)in[4 * i] << 24) | ((uint32_t)in[4 * i + 1] << 16) | ((uint32_t)in[4 * i + 2] << 8) | in[4 * i + 3];
    const uint32_t m = P::MOD(7 - i);
    if (w < m) return true;
    if (w > m) return false;
  }
  return false;   // equal to the modulus
}
using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// canonical limbs > (p-1)/2 ?
template <class Pm>
SPP_HD bool canonical_gt_half(const uint32_t c[8]) {
  // HALF - c < 0  <=>  c > HALF
  uint32_t br = 0;
  SPP_UNROLL for (int i = 0; i < 8; i++) (void)subb32(Pm::HALF(i), c[i], br);
  return br != 0;
}
// out = p - c  (c canonical, nonzero)
template <class Pm>
SPP_HD void canonical_negate(const uint32_t c[8], uint32_t out[8]) {
  uint32_t br = 0;
  SPP_UNROLL for (int i = 0; i < 8; i++) out[i] = subb32(Pm::MOD(i), c[i], br);
}

// --------------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2+1)
// --------------------------------------------------------------------------------------------------
struct Fq2 {
  Fq c0, c1;
  static SPP_HD Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
  static SPP_HD Fq2 one() { return {Fq::one(), Fq::zero()}; }
  SPP_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  SPP_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
  SPP_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
  friend SPP_HD Fq2 operator+(const Fq2& a, const Fq2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  friend SPP_HD Fq2 operator-(const Fq2& a, const Fq2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  SPP_HD Fq2 neg() const { return {c0.neg(), c1.neg()}; }
  SPP_HD Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  // Lazy reduction on the 29-bit column form of Fq (see Fp::operator*): both components are sums of two products,
  // accumulated as 162 v_mad_u64_u32 into the same 17 columns (2 * 9 * 2^58 + reduction < 2^63) and reduced ONCE
  // each -- two Montgomery reductions instead of three multiplications' worth, and no intermediate add/sub.
  //   c0 = a0*b0 + a1*(-b1),  c1 = a0*b1 + a1*b0
  friend SPP_HD Fq2 operator*(const Fq2& a, const Fq2& b) {
    uint32_t a0[9], a1[9], b0[9], b1[9], n1[9];
    Fq::to9(a.c0.l, a0);
    Fq::to9(a.c1.l, a1);
    Fq::to9(b.c0.l, b0);
    Fq::to9(b.c1.l, b1);
    const Fq nb1 = b.c1.neg();
    Fq::to9(nb1.l, n1);
    uint64_t c[18];
    SPP_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
    SPP_UNROLL for (int i = 0; i < 9; i++) {
      SPP_UNROLL for (int j = 0; j < 9; j++) {
        c[i + j] += (uint64_t)a0[i] * b0[j];
        c[i + j] += (uint64_t)a1[i] * n1[j];
      }
    }
    Fq2 r;
    r.c0 = Fq::reduce_columns(c);   // sum of two products of values < 2p: < 2.51 p
    Fq::cond_sub_2p(r.c0.l);
    SPP_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
    SPP_UNROLL for (int i = 0; i < 9; i++) {
      SPP_UNROLL for (int j = 0; j < 9; j++) {
        c[i + j] += (uint64_t)a0[i] * b1[j];
        c[i + j] += (uint64_t)a1[i] * b0[j];
      }
    }
    r.c1 = Fq::reduce_columns(c);
    Fq::cond_sub_2p(r.c1.l);
    return r;
  }
  //   c0 = a0^2 + a1*(-a1),  c1 = (2 a0) * a1
  SPP_HD Fq2 sqr() const {
    uint32_t a0[9], a1[9], n1[9], d0[9];
    Fq::to9(c0.l, a0);
    Fq::to9(c1.l, a1);
    const Fq na1 = c1.neg();
    Fq::to9(na1.l, n1);
    SPP_UNROLL for (int i = 0; i < 9; i++) d0[i] = a0[i] << 1;
    uint64_t c[18];
    SPP_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
    SPP_UNROLL for (int i = 0; i < 9; i++) {
      c[2 * i] += (uint64_t)a0[i] * a0[i];
      SPP_UNROLL for (int j = i + 1; j < 9; j++) c[i + j] += (uint64_t)a0[i] * d0[j];
      SPP_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a1[i] * n1[j];
    }
    Fq2 r;
    r.c0 = Fq::reduce_columns(c);   // sum of two products of values < 2p: < 2.51 p
    Fq::cond_sub_2p(r.c0.l);
    SPP_UNROLL for (int k = 0; k < 18; k++) c[k] = 0;
    SPP_UNROLL for (int i = 0; i < 9; i++) {
      SPP_UNROLL for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)d0[i] * a1[j];
    }
    r.c1 = Fq::reduce_columns(c);
    Fq::cond_sub_2p(r.c1.l);
    return r;
  }
  SPP_HD Fq2 inv() const {
    Fq d = (c0.sqr() + c1.sqr()).inv();
    return {c0 * d, (c1 * d).neg()};
  }
  SPP_HD Fq2 mul_small(uint32_t k) const { return {c0.mul_small(k), c1.mul_small(k)}; }
};

// --------------------------------------------------------------------------------------------------
// Short-Weierstrass curves y^2 = x^3 + b (a = 0) over F: G1 (F=Fq), G2 (F=Fq2), Grumpkin (F=Fr).
// Affine points (infinity encoded as x=y=0, never on these curves) and extended-Jacobian "XYZZ"
// accumulators: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ = 0. Mixed addition is 8M+2S.
// --------------------------------------------------------------------------------------------------
template <class F>
struct Affine {
  F x, y;
  static SPP_HD Affine infinity() { return {F::zero(), F::zero()}; }
  SPP_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  SPP_HD Affine neg() const { return {x, y.neg()}; }
};

template <class F>
struct XYZZ {
  F X, Y, ZZ, ZZZ;
  static SPP_HD XYZZ infinity() { return {F::one(), F::one(), F::zero(), F::zero()}; }
  SPP_HD bool is_inf() const { return ZZ.is_zero(); }
  static SPP_HD XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return infinity();
    return {p.x, p.y, F::one(), F::one()};
  }
  SPP_HD XYZZ neg() const { return {X, Y.neg(), ZZ, ZZZ}; }

  // this = 2*this
  SPP_HD void dbl_inplace() {
    if (is_inf()) return;
    F U = Y.dbl();
    F V = U.sqr();
    F W = U * V;
    F S = X * V;
    F X2 = X.sqr();
    F M = X2.dbl() + X2;
    F X3 = M.sqr() - S.dbl();
    F Y3 = M * (S - X3) - W * Y;
    ZZ = V * ZZ;
    ZZZ = W * ZZZ;
    X = X3;
    Y = Y3;
  }
  // this += p (affine, not infinity unless flagged by caller)
  SPP_HD void madd(const Affine<F>& p) {
    if (p.is_inf()) return;
    if (is_inf()) {
      X = p.x;
      Y = p.y;
      ZZ = F::one();
      ZZZ = F::one();
      return;
    }
    F U2 = p.x * ZZ;
    F S2 = p.y * ZZZ;
    F Pp = U2 - X;
    F Rr = S2 - Y;
    if (Pp.is_zero()) {
      if (Rr.is_zero()) {
        dbl_inplace();
      } else {
        *this = infinity();
      }
      return;
    }
    F PP = Pp.sqr();
    F PPP = Pp * PP;
    F Q = X * PP;
    F X3 = Rr.sqr() - PPP - Q.dbl();
    F Y3 = Rr * (Q - X3) - Y * PPP;
    ZZ = ZZ * PP;
    ZZZ = ZZZ * PPP;
    X = X3;
    Y = Y3;
  }
  // this += q
  SPP_HD void add(const XYZZ& q) {
    if (q.is_inf()) return;
    if (is_inf()) {
      *this = q;
      return;
    }
    F U1 = X * q.ZZ;
    F U2 = q.X * ZZ;
    F S1 = Y * q.ZZZ;
    F S2 = q.Y * ZZZ;
    F Pp = U2 - U1;
    F Rr = S2 - S1;
    if (Pp.is_zero()) {
      if (Rr.is_zero()) {
        dbl_inplace();
      } else {
        *this = infinity();
      }
      return;
    }
    F PP = Pp.sqr();
    F PPP = Pp * PP;
    F Q = U1 * PP;
    F X3 = Rr.sqr() - PPP - Q.dbl();
    F Y3 = Rr * (Q - X3) - S1 * PPP;
    ZZ = ZZ * q.ZZ * PP;
    ZZZ = ZZZ * q.ZZZ * PPP;
    X = X3;
    Y = Y3;
  }
  SPP_HD Affine<F> to_affine() const {
    if (is_inf()) return Affine<F>::infinity();
    F i = (ZZ * ZZZ).inv();
    F izz = i * ZZZ;   // 1/ZZ
    F izzz = i * ZZ;   // 1/ZZZ
    return {X * izz, Y * izzz};
  }
};

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;
using GkAffine = Affine<Fr>;   // Grumpkin: coordinates in Fr
using GkXYZZ = XYZZ<Fr>;

// variable-base scalar multiplication, scalar as 8 canonical limbs (MSB-first double-and-add)
template <class F>
SPP_HD XYZZ<F> scalar_mul(const Affine<F>& p, const uint32_t k[8]) {
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (int w = 7; w >= 0; w--) {
    for (int b = 31; b >= 0; b--) {
      acc.dbl_inplace();
      if ((k[w] >> b) & 1) acc.madd(p);
    }
  }
  return acc;
}

}  // namespace spp

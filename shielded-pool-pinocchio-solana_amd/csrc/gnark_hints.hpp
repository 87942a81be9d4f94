// gnark_hints.hpp -- the hints of the reference's own gnark constraint system (noir_circuit/target/shielded_pool_verifier.ccs,
// SURVEY 8f-1) that need integer arithmetic wider than a field element, as device code for the solver (kernels_solve.hip):
//   sw-grumpkin.decomposeScalar   s -> (s1, s2), 0 <= s1, s2 < 2^127, s1 - lambda * s2 = s (mod q)      [dev_glv_split]
//   emulated.mulHint (b = 1)      limbs a_i -> quotient, remainder and carries of  a(X) = k(X) p(X) + r(X) + (2^64 - X) c(X) [dev_emulated_reduce]
//   the ACIR MultiScalarMul black box over the Grumpkin generator                                         [dev_grumpkin_mul]
// Neither hint's source is in the reference tree (Sunspot / gnark are third-party): both restate spp/ccs.py
// (glv_split, _emulated_mul_hint), which derives them from the rows of the .ccs that consume their outputs, and are checked
// wire for wire against it (tests/test_acir_ccs.py).  One lane works on one proof; nothing here is on a throughput path (three
// calls per proof), so the code is plain loops over 32-bit words.
#pragma once
#include "bn254.hpp"

namespace spp {

// signed integers of NW 32-bit words, two's complement
template <int NW>
struct BigS {
  uint32_t w[NW];
  SPP_HD void zero() {
    for (int i = 0; i < NW; i++) w[i] = 0;
  }
  SPP_HD bool neg() const { return (w[NW - 1] >> 31) != 0; }
  SPP_HD void add(const BigS& o) {
    uint64_t c = 0;
    for (int i = 0; i < NW; i++) {
      c += (uint64_t)w[i] + o.w[i];
      w[i] = (uint32_t)c;
      c >>= 32;
    }
  }
  SPP_HD void sub(const BigS& o) {
    uint64_t b = 0;
    for (int i = 0; i < NW; i++) {
      const uint64_t d = (uint64_t)w[i] - o.w[i] - b;
      w[i] = (uint32_t)d;
      b = (d >> 32) & 1;
    }
  }
  SPP_HD void negate() {
    uint64_t c = 1;
    for (int i = 0; i < NW; i++) {
      c += (uint32_t)~w[i];
      w[i] = (uint32_t)c;
      c >>= 32;
    }
  }
  // this < o (signed)
  SPP_HD bool lt(const BigS& o) const {
    BigS t = *this;
    t.sub(o);          // no overflow for the magnitudes used here (top bits spare)
    return t.neg();
  }
  // this += m * o for a small signed m
  SPP_HD void add_small_mul(const BigS& o, int m) {
    BigS t = o;
    if (m < 0) {
      t.negate();
      m = -m;
    }
    for (int i = 0; i < m; i++) add(t);
  }
  // arithmetic shift right by 64 bits
  SPP_HD void sar64() {
    const uint32_t fill = neg() ? 0xffffffffu : 0u;
    for (int i = 0; i + 2 < NW; i++) w[i] = w[i + 2];
    w[NW - 2] = fill;
    w[NW - 1] = fill;
  }
  SPP_HD bool low64_zero() const { return (w[0] | w[1]) == 0; }
};
typedef BigS<12> Big384;

// unsigned na-word x nb-word product accumulated into out (nw words, truncating)
SPP_HD inline void big_mul_acc(uint32_t* out, int nw, const uint32_t* a, int na, const uint32_t* b, int nb) {
  for (int i = 0; i < na; i++) {
    uint64_t c = 0;
    for (int j = 0; j < nb && i + j < nw; j++) {
      c += (uint64_t)a[i] * b[j] + out[i + j];
      out[i + j] = (uint32_t)c;
      c >>= 32;
    }
    for (int k = i + nb; c && k < nw; k++) {
      c += out[k];
      out[k] = (uint32_t)c;
      c >>= 32;
    }
  }
}

// a signed 128-bit constant (sign + 4 magnitude words) as Big384
SPP_HD inline Big384 big_from_s128(const uint32_t mag[4], bool negative) {
  Big384 r;
  r.zero();
  for (int i = 0; i < 4; i++) r.w[i] = mag[i];
  if (negative) r.negate();
  return r;
}

// ---- sw-grumpkin.decomposeScalar ---------------------------------------------------------------------------------------------
// kc: constants laid out by spp/ccs.py (to_sppc_solved): v1x, v1y, v2x, v2y as (4 magnitude words, 1 sign word) each, then det
// (8 words, positive).  s: the scalar, < 2^128 (4 words).  The search order is that of spp/ccs.py glv_split, so the outputs are
// the same pair.  Returns false when no pair is in range (the rows that consume the outputs then fail).
SPP_HD inline bool dev_glv_split(const uint32_t* kc, const uint32_t s[4], uint32_t s1[4], uint32_t s2[4]) {
  Big384 v[4];
  for (int k = 0; k < 4; k++) v[k] = big_from_s128(kc + 5 * k, kc[5 * k + 4] != 0);
  Big384 det2;   // 2 * det
  det2.zero();
  for (int i = 0; i < 8; i++) det2.w[i] = kc[20 + i];
  Big384 det = det2;
  det2.add(det);
  Big384 S;
  S.zero();
  for (int i = 0; i < 4; i++) S.w[i] = s[i];
  // b1 = floor((2 s v2y + det) / (2 det)), b2 = floor((-2 s v1y + det) / (2 det))
  int b[2];
  for (int t = 0; t < 2; t++) {
    const uint32_t* mag = kc + 5 * (t == 0 ? 3 : 1);
    const bool sneg = (kc[5 * (t == 0 ? 3 : 1) + 4] != 0) != (t == 1);   // sign of v2y, or of -v1y
    Big384 num;
    num.zero();
    big_mul_acc(num.w, 12, s, 4, mag, 4);   // s * |v|
    Big384 twice = num;
    num.add(twice);                         // 2 s |v|
    if (sneg) num.negate();
    num.add(det);
    int q = 0;
    for (int it = 0; it < 64 && !num.lt(det2) ; it++) {
      num.sub(det2);
      q++;
    }
    for (int it = 0; it < 64 && num.neg(); it++) {
      num.add(det2);
      q--;
    }
    b[t] = q;
  }
  Big384 lim;   // 2^127
  lim.zero();
  lim.w[3] = 0x80000000u;
  for (int radius = 0; radius < 6; radius++) {
    for (int i1 = -radius; i1 <= radius; i1++) {
      for (int i2 = -radius; i2 <= radius; i2++) {
        const int a1 = i1 < 0 ? -i1 : i1, a2 = i2 < 0 ? -i2 : i2;
        if ((a1 > a2 ? a1 : a2) != radius) continue;
        Big384 x = S, y;
        y.zero();
        x.add_small_mul(v[0], -(b[0] + i1));
        x.add_small_mul(v[2], -(b[1] + i2));
        y.add_small_mul(v[1], -(b[0] + i1));
        y.add_small_mul(v[3], -(b[1] + i2));
        if (!x.neg() && !y.neg() && x.lt(lim) && y.lt(lim)) {
          for (int i = 0; i < 4; i++) {
            s1[i] = x.w[i];
            s2[i] = y.w[i];
          }
          return true;
        }
      }
    }
  }
  for (int i = 0; i < 4; i++) s1[i] = s2[i] = 0;
  return false;
}

// ---- emulated.mulHint with b = [1]: reduce a(2^64) modulo q -------------------------------------------------------------------
// a[i]: canonical words (8) of the na limb expressions (each an Fr value, < r < q).  qc: q (8 words) then q^-1 mod 2^256 (8 words).
// Outputs: k (4 x 64-bit limbs as 8 words), r (8 words), carries c_0 .. c_{nc-1} as signed Big384 (|c| < 2^200).
// Restates spp/ccs.py _emulated_mul_hint for bits = 64, n = nq = 4.
template <int NA, int NC>
SPP_HD inline void dev_emulated_reduce(const uint32_t (*a)[8], const uint32_t* qc, uint32_t kq[8], uint32_t rem[8], Big384 (&carry)[NC]) {
  // remainder through Fq: Horner over the limbs with radix 2^64
  uint32_t rad[8] = {0, 0, 1, 0, 0, 0, 0, 0};
  const Fq radix = Fq::from_canonical(rad);
  Fq acc = Fq::zero();
  for (int i = NA - 1; i >= 0; i--) acc = acc * radix + Fq::from_canonical(a[i]);
  acc.to_canonical(rem);
  // low 256 bits of a(2^64), minus the remainder, times q^-1 mod 2^256: the quotient (it has to fit 256 bits)
  uint32_t low[8];
  for (int i = 0; i < 8; i++) low[i] = 0;
  for (int i = 0; i < NA && 2 * i < 8; i++) {
    uint64_t c = 0;
    for (int j = 0; 2 * i + j < 8; j++) {
      c += (uint64_t)low[2 * i + j] + (j < 8 ? a[i][j] : 0u);
      low[2 * i + j] = (uint32_t)c;
      c >>= 32;
    }
  }
  uint64_t br = 0;
  for (int i = 0; i < 8; i++) {
    const uint64_t d = (uint64_t)low[i] - rem[i] - br;
    low[i] = (uint32_t)d;
    br = (d >> 32) & 1;
  }
  for (int i = 0; i < 8; i++) kq[i] = 0;
  big_mul_acc(kq, 8, low, 8, qc + 8, 8);
  // carries: d_i = a_i - sum_{x + y = i} k_x p_y - r_i ;  c_i = (d_i + c_{i-1}) / 2^64
  Big384 c;
  c.zero();
  for (int i = 0; i < NC; i++) {
    Big384 d;
    d.zero();
    if (i < NA)
      for (int j = 0; j < 8; j++) d.w[j] = a[i][j];
    for (int x = 0; x < 4; x++) {
      const int y = i - x;
      if (y < 0 || y >= 4) continue;
      Big384 t;
      t.zero();
      big_mul_acc(t.w, 12, kq + 2 * x, 2, qc + 2 * y, 2);
      d.sub(t);
    }
    if (i < 4) {
      Big384 t;
      t.zero();
      t.w[0] = rem[2 * i];
      t.w[1] = rem[2 * i + 1];
      d.sub(t);
    }
    d.add(c);
    d.sar64();          // exact when the identity closes (the low 64 bits are zero)
    c = d;
    carry[i] = c;
  }
}
// a signed Big384 (|v| < r) as a field element
SPP_HD inline Fr fr_from_bigs(const Big384& v) {
  Big384 m = v;
  const bool n = m.neg();
  if (n) m.negate();
  uint32_t w[8];
  for (int i = 0; i < 8; i++) w[i] = m.w[i];
  const Fr f = Fr::from_canonical(w);
  return n ? f.neg() : f;
}

// ---- Grumpkin fixed-base multiplication (ACIR MultiScalarMul over the generator) -----------------------------------------------
// y^2 = x^3 - 17 over Fr, G = (1, gy).  k = lo + 2^128 * hi as 8 canonical words.  Jacobian double-and-add from the top bit; returns
// false for the point at infinity (k = 0 mod the group order).
SPP_HD inline bool dev_grumpkin_mul(const uint32_t k[8], const Fr& gy, Fr* ox, Fr* oy) {
  const Fr gx = Fr::one();
  Fr X = Fr::zero(), Y = Fr::zero(), Z = Fr::zero();
  bool inf = true;
  for (int bit = 255; bit >= 0; bit--) {
    if (!inf) {   // dbl-2009-l (a = 0)
      const Fr A = X.sqr(), B = Y.sqr(), C = B.sqr();
      const Fr t = (X + B).sqr() - A - C;
      const Fr D = t.dbl();
      const Fr E = A.dbl() + A;
      const Fr X3 = E.sqr() - D.dbl();
      const Fr Z3 = (Y * Z).dbl();
      Y = E * (D - X3) - C.dbl().dbl().dbl();
      X = X3;
      Z = Z3;
      if (Z.is_zero()) inf = true;
    }
    if ((k[bit >> 5] >> (bit & 31)) & 1) {
      if (inf) {
        X = gx;
        Y = gy;
        Z = Fr::one();
        inf = false;
      } else {   // mixed addition
        const Fr Z2 = Z.sqr();
        const Fr U2 = gx * Z2, S2 = gy * Z2 * Z;
        const Fr H = U2 - X, Rr = S2 - Y;
        if (H.is_zero()) {
          if (Rr.is_zero()) {   // acc == G: double G (cannot happen on the way to k >= 2, kept for completeness)
            const Fr A = X.sqr(), B = Y.sqr(), C = B.sqr();
            const Fr t = (X + B).sqr() - A - C;
            const Fr D = t.dbl();
            const Fr E = A.dbl() + A;
            const Fr X3 = E.sqr() - D.dbl();
            const Fr Z3 = (Y * Z).dbl();
            Y = E * (D - X3) - C.dbl().dbl().dbl();
            X = X3;
            Z = Z3;
          } else {
            inf = true;
          }
        } else {
          const Fr H2 = H.sqr(), H3 = H2 * H, V = X * H2;
          const Fr X3 = Rr.sqr() - H3 - V.dbl();
          Y = Rr * (V - X3) - Y * H3;
          X = X3;
          Z = Z * H;
        }
      }
    }
  }
  if (inf) return false;
  const Fr zi = Z.inv(), zi2 = zi.sqr();
  *ox = X * zi2;
  *oy = Y * zi2 * zi;
  return true;
}

}  // namespace spp

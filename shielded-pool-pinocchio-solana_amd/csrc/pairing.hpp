// Host-side optimal-ate pairing check on BN254 for spp_verify (the `sunspot verify vk proof pw` step of
// noir_circuit/prove_linux.sh:86-87 and audit_circuit/prove_audit.sh:98-99).  Verification is not on the hot path
// (one product of four Miller loops per proof), so this is plain, compact host C++ over csrc/bn254.hpp:
//   Fq12 = Fq[w]/(w^12 - 18 w^6 + 82)  (w^6 = 9 + u), schoolbook products;
//   the G2 point stays on the twist E'(Fq2): tangent/chord slopes are computed in Fq2 and every line
//   l(P) = yP - (lambda xP) w + (lambda xT - yT) w^3  is embedded sparsely;
//   final exponentiation without inversion:  f^((p^12-1)/r) = 1  <=>  (f^(p^6))^E = f^E,  E = (p^2+1)(p^4-p^2+1)/r.
#pragma once
#include <vector>
#include "bn254.hpp"

namespace spp {

struct Fq12 {
  Fq c[12];
  static Fq12 zero() { Fq12 r; for (auto& x : r.c) x = Fq::zero(); return r; }
  static Fq12 one() { Fq12 r = zero(); r.c[0] = Fq::one(); return r; }
  bool operator==(const Fq12& o) const { for (int i = 0; i < 12; i++) if (c[i] != o.c[i]) return false; return true; }
};

inline Fq12 f12_mul(const Fq12& a, const Fq12& b) {
  Fq t[23];
  for (auto& x : t) x = Fq::zero();
  for (int i = 0; i < 12; i++) {
    if (a.c[i].is_zero()) continue;
    for (int j = 0; j < 12; j++)
      if (!b.c[j].is_zero()) t[i + j] = t[i + j] + a.c[i] * b.c[j];
  }
  const Fq k18 = Fq::from_u64(18), k82 = Fq::from_u64(82);
  for (int k = 22; k >= 12; k--) {
    if (t[k].is_zero()) continue;
    t[k - 6] = t[k - 6] + t[k] * k18;
    t[k - 12] = t[k - 12] - t[k] * k82;
  }
  Fq12 r;
  for (int i = 0; i < 12; i++) r.c[i] = t[i];
  return r;
}
// embed c * w^k, c = (a + b u) in Fq2, u = w^6 - 9
inline void f12_add_fq2_term(Fq12& f, const Fq2& c, int k) {
  Fq nine_b = c.c1.mul_small(9);
  f.c[k] = f.c[k] + (c.c0 - nine_b);
  f.c[k + 6] = f.c[k + 6] + c.c1;
}

struct PairingConsts {
  Fq2 g13, g12;        // xi^((p-1)/3), xi^((p-1)/2): Frobenius on twist coordinates
  Fq12 wfrob[12];      // w^(k p) = w^k * (xi^((p-1)/6))^k, for Frobenius on Fq12
};
inline Fq2 fq2_pow(const Fq2& base, const uint32_t* e, int nlimbs) {
  Fq2 acc = Fq2::one(), b = base;
  for (int w = 0; w < nlimbs; w++)
    for (int i = 0; i < 32; i++) {
      if ((e[w] >> i) & 1) acc = acc * b;
      b = b.sqr();
    }
  return acc;
}
inline PairingConsts make_pairing_consts() {
  PairingConsts pc;
  // (p-1)/6 as limbs: p-1 divided by 6
  uint32_t pm1[8];
  for (int i = 0; i < 8; i++) pm1[i] = FqParams::MOD(i);
  pm1[0] -= 1;
  uint32_t e6[8];
  uint64_t rem = 0;
  for (int i = 7; i >= 0; i--) {
    uint64_t cur = (rem << 32) | pm1[i];
    e6[i] = (uint32_t)(cur / 6);
    rem = cur % 6;
  }
  Fq2 xi{Fq::from_u64(9), Fq::one()};
  Fq2 g16 = fq2_pow(xi, e6, 8);         // xi^((p-1)/6)
  pc.g13 = g16.sqr();
  pc.g12 = pc.g13 * g16;
  Fq2 gk = Fq2::one();
  for (int k = 0; k < 12; k++) {
    Fq12 t = Fq12::zero();
    f12_add_fq2_term(t, gk, 0);           // gk as an Fq12 element
    Fq12 wk = Fq12::zero();
    wk.c[k] = Fq::one();
    pc.wfrob[k] = f12_mul(t, wk);
    gk = gk * g16;
  }
  return pc;
}
inline const PairingConsts& pairing_consts() {
  static const PairingConsts pc = make_pairing_consts();   // thread-safe one-time initialisation
  return pc;
}
inline Fq12 f12_frobenius(const Fq12& a) {
  const PairingConsts& pc = pairing_consts();
  Fq12 r = Fq12::zero();
  for (int k = 0; k < 12; k++) {
    if (a.c[k].is_zero()) continue;
    for (int i = 0; i < 12; i++) r.c[i] = r.c[i] + pc.wfrob[k].c[i] * a.c[k];
  }
  return r;
}
inline Fq12 f12_conj6(const Fq12& a) {   // a^(p^6): w -> -w
  Fq12 r = a;
  for (int k = 1; k < 12; k += 2) r.c[k] = a.c[k].neg();
  return r;
}
inline Fq12 f12_pow_limbs(const Fq12& base, const uint32_t* e, int nlimbs) {
  Fq12 acc = Fq12::one(), b = base;
  for (int w = 0; w < nlimbs; w++)
    for (int i = 0; i < 32; i++) {
      if ((e[w] >> i) & 1) acc = f12_mul(acc, b);
      b = f12_mul(b, b);
    }
  return acc;
}

// l(P) for the line through T (and Q, or tangent at T when Q == nullptr) on the twist; advances T
inline Fq12 line_and_step(G2Affine& T, const G2Affine* Q, const G1Affine& P) {
  Fq2 lam;
  if (Q == nullptr) {
    Fq2 x2 = T.x.sqr();
    lam = (x2.dbl() + x2) * T.y.dbl().inv();
  } else {
    lam = (Q->y - T.y) * (Q->x - T.x).inv();
  }
  Fq12 l = Fq12::zero();
  l.c[0] = P.y;
  Fq2 lx{lam.c0 * P.x, lam.c1 * P.x};
  f12_add_fq2_term(l, lx.neg(), 1);
  f12_add_fq2_term(l, lam * T.x - T.y, 3);
  Fq2 x3 = lam.sqr() - T.x - (Q ? Q->x : T.x);
  Fq2 y3 = lam * (T.x - x3) - T.y;
  T = {x3, y3};
  return l;
}

inline Fq12 miller_loop(const G1Affine& P, const G2Affine& Q) {
  if (P.is_inf() || Q.is_inf()) return Fq12::one();
  const PairingConsts& pc = pairing_consts();
  const uint64_t ate_lo = 0x9d797039be763ba8ull;   // 29793968203157093288 = 2^64 + ate_lo
  G2Affine T = Q;
  Fq12 f = Fq12::one();
  for (int i = 63; i >= 0; i--) {
    f = f12_mul(f12_mul(f, f), line_and_step(T, nullptr, P));
    if ((ate_lo >> i) & 1) f = f12_mul(f, line_and_step(T, &Q, P));
  }
  auto conj = [](const Fq2& a) { return Fq2{a.c0, a.c1.neg()}; };
  G2Affine Q1{conj(Q.x) * pc.g13, conj(Q.y) * pc.g12};
  G2Affine Q2{conj(Q1.x) * pc.g13, conj(Q1.y) * pc.g12};
  Q2.y = Q2.y.neg();
  f = f12_mul(f, line_and_step(T, &Q1, P));
  f = f12_mul(f, line_and_step(T, &Q2, P));
  return f;
}

// prod e(P_i, Q_i) == 1 ?
inline bool pairing_product_is_one(const std::vector<std::pair<G1Affine, G2Affine>>& pairs) {
  Fq12 f = Fq12::one();
  for (auto& pq : pairs) f = f12_mul(f, miller_loop(pq.first, pq.second));
  static const uint32_t HARD[24] = {0xccdf42b1u, 0xe81bb482u, 0xf49c36d4u, 0x5abf5cc4u, 0x1da014fdu, 0xf1154e7eu, 0x87cdbacfu, 0xdcc7b44cu,
                                    0x954bcf8au, 0xaaa441e3u, 0xd5095f23u, 0x6b887d56u, 0xf3fd90c6u, 0x79581e16u, 0xd189227du, 0x3b1b1355u,
                                    0x61876f6bu, 0x4e529a58u, 0xd5b12278u, 0x6c0eb522u, 0x83177fafu, 0x331ec151u, 0x0b0759adu, 0x01baaa71u};
  auto powE = [&](const Fq12& x) {
    Fq12 y = f12_mul(f12_frobenius(f12_frobenius(x)), x);   // x^(p^2+1)
    return f12_pow_limbs(y, HARD, 24);                      // ^((p^4-p^2+1)/r)
  };
  return powE(f12_conj6(f)) == powE(f);
}

inline bool g1_on_curve(const G1Affine& p) {
  if (p.is_inf()) return true;
  return p.y.sqr() == p.x.sqr() * p.x + Fq::from_u64(3);
}
inline bool g2_on_curve(const G2Affine& p) {
  if (p.is_inf()) return true;
  Fq2 b = Fq2{Fq::from_u64(3), Fq::zero()} * Fq2{Fq::from_u64(9), Fq::one()}.inv();
  return p.y.sqr() == p.x.sqr() * p.x + b;
}

}  // namespace spp

// Host-side preparation for the batched verifier (pairing_fast.hpp): Frobenius constants, and the per-key line tables
// of the G2 points that belong to a verifying key.  Built with the single-proof host pairing (pairing.hpp), i.e. with
// the code the batched path is tested against.
#pragma once
#include <vector>
#include "pairing.hpp"
#include "pairing_fast.hpp"

namespace spp {

inline PairingFastConsts make_pairing_fast_consts() {
  const PairingConsts& pc = pairing_consts();
  PairingFastConsts f;
  for (int i = 0; i < 12; i++) {
    f.FA[i] = pc.wfrob[i].c[i];                   // w^(ip) = gamma^i w^i has coefficients at w^i and w^(i+6 mod 12) only
    f.FB[i] = pc.wfrob[(i + 6) % 12].c[i];
  }
  f.g13 = pc.g13;
  f.g12 = pc.g12;
  f.k18 = Fq::from_u64(18);
  f.k82 = Fq::from_u64(82);
  f.one = Fq::one();
  return f;
}
// every other coefficient of w^(kp) must vanish for the two-term Frobenius above to be exact
inline bool pairing_fast_consts_consistent() {
  const PairingConsts& pc = pairing_consts();
  for (int k = 0; k < 12; k++)
    for (int i = 0; i < 12; i++)
      if (i != k && i != (k + 6) % 12 && !pc.wfrob[k].c[i].is_zero()) return false;
  return true;
}

// lines of the ate loop for a fixed Q, in the order miller_multi consumes them
inline std::vector<LineStep> build_line_table(const G2Affine& Q) {
  const PairingConsts& pc = pairing_consts();
  std::vector<LineStep> tab;
  G2Affine T = Q;
  auto step = [&](const G2Affine* Qa) {
    Fq2 lam;
    if (Qa == nullptr) {
      Fq2 x2 = T.x.sqr();
      lam = (x2.dbl() + x2) * T.y.dbl().inv();
    } else {
      lam = (Qa->y - T.y) * (Qa->x - T.x).inv();
    }
    const Fq2 c = lam * T.x - T.y;
    LineStep s;
    emb(lam, s.a1, s.b1);
    emb(c, s.a3, s.b3);
    tab.push_back(s);
    const Fq2 x3 = lam.sqr() - T.x - (Qa ? Qa->x : T.x);
    const Fq2 y3 = lam * (T.x - x3) - T.y;
    T = {x3, y3};
  };
  for (int i = 63; i >= 0; i--) {
    step(nullptr);
    if ((ATE_LO >> i) & 1) step(&Q);
  }
  auto conj = [](const Fq2& a) { return Fq2{a.c0, a.c1.neg()}; };
  G2Affine Q1{conj(Q.x) * pc.g13, conj(Q.y) * pc.g12};
  G2Affine Q2{conj(Q1.x) * pc.g13, conj(Q1.y) * pc.g12};
  Q2.y = Q2.y.neg();
  step(&Q1);
  step(&Q2);
  return tab;
}
inline F12 f12_from(const Fq12& a) {
  F12 r;
  for (int i = 0; i < 12; i++) r.c[i] = a.c[i];
  return r;
}
inline Fq2 twist_b() { return Fq2{Fq::from_u64(3), Fq::zero()} * Fq2{Fq::from_u64(9), Fq::one()}.inv(); }

}  // namespace spp

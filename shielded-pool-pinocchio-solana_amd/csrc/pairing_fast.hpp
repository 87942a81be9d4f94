// Pairing check for BATCH verification, host + gfx950 (one lane per proof in kernels_verify.hip).
// "a Groth16 verifier on GPU (batched pairings)", the counterpart of `sunspot verify`
// (noir_circuit/prove_linux.sh:86-87, audit_circuit/prove_audit.sh:98-99) and of the on-chain verifier the reference
// deploys (audit_verifier.so, withdraw.rs:19-20) for many proofs against one verifying key.
//
// Same field tower and line convention as pairing.hpp (the single-proof host verifier, which stays the reference
// these routines are tested against): Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), G2 on the twist, l(P) = yP - lambda xP w +
// (lambda xT - yT) w^3.  What differs, because here every multiplication counts:
//   * one shared Miller loop for all pairs of a proof (one squaring of f per bit);
//   * the G2 arguments that belong to the verifying key (gamma, delta, the two Pedersen points) have their line
//     coefficients precomputed once per key (LineStep tables): no point arithmetic and no inversions for them;
//     e(alpha, beta) enters as a precomputed Miller value;
//   * the only G2 point that changes per proof (Bs) is walked in XYZZ coordinates and its lines are used scaled by an
//     Fq2 factor (killed by the final exponentiation): no inversions in the loop;
//   * final exponentiation: f^-1 through the norm to Fq (11 Frobenius maps, 1 Fq inversion), easy part, then the hard
//     part as y^(l0 + l1 p + l2 p^2 + l3 p^3) with l0 = 1+6x+12x^2+12x^3, l1 = 4x+6x^2+12x^3, l2 = 6x+6x^2+12x^3,
//     l3 = -1+4x+6x^2+12x^3 (= 2x(6x^2+3x+1) (p^4-p^2+1)/r, a multiple coprime to r): three exponentiations by the
//     63-bit x instead of a 768-bit one.
#pragma once
#include "bn254.hpp"

#if defined(__HIPCC__)
#define SPP_HDN inline __host__ __device__ __attribute__((noinline))
#else
#define SPP_HDN inline
#endif

namespace spp {

struct F12 {
  Fq c[12];
};
struct LineStep {   // line = yP + (-xP*a1) w + a3 w^3 + (-xP*b1) w^7 + b3 w^9
  Fq a1, b1, a3, b3;
};
struct PairingFastConsts {
  Fq FA[12], FB[12];   // Frobenius: (a^p).c[i] = a.c[i]*FA[i] + a.c[(i+6)%12]*FB[i]
  Fq2 g13, g12;        // xi^((p-1)/3), xi^((p-1)/2): Frobenius on twist coordinates
  Fq k18, k82, one;
};
static constexpr uint64_t BN_X = 4965661367192848881ull;       // curve parameter (63 bits)
static constexpr uint64_t ATE_LO = 0x9d797039be763ba8ull;      // 6x+2 = 2^64 + ATE_LO

SPP_HD F12 f12_one(const PairingFastConsts& pc) {
  F12 r;
  for (int i = 0; i < 12; i++) r.c[i] = Fq::zero();
  r.c[0] = pc.one;
  return r;
}
// t[0..22] -> 12 coefficients using w^12 = 18 w^6 - 82
SPP_HD F12 f12_fold(Fq (&t)[23], const PairingFastConsts& pc) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int k = 22; k >= 12; k--) {
    t[k - 6] = t[k - 6] + t[k] * pc.k18;
    t[k - 12] = t[k - 12] - t[k] * pc.k82;
  }
  F12 r;
  for (int i = 0; i < 12; i++) r.c[i] = t[i];
  return r;
}
SPP_HDN F12 f12_mul(const F12& a, const F12& b, const PairingFastConsts& pc) {
  Fq t[23];
  for (int i = 0; i < 23; i++) t[i] = Fq::zero();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = 0; i < 12; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int j = 0; j < 12; j++) t[i + j] = t[i + j] + a.c[i] * b.c[j];
  }
  return f12_fold(t, pc);
}
// f times the sparse element  l0 + l1 w + l3 w^3 + l6 w^6 + l7 w^7 + l9 w^9
SPP_HDN F12 f12_mul_line(const F12& f, const Fq& l0, const Fq& l1, const Fq& l3, const Fq& l6, const Fq& l7, const Fq& l9,
                         const PairingFastConsts& pc) {
  Fq t[23];
  for (int i = 0; i < 23; i++) t[i] = Fq::zero();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = 0; i < 12; i++) {
    const Fq x = f.c[i];
    t[i] = t[i] + x * l0;
    t[i + 1] = t[i + 1] + x * l1;
    t[i + 3] = t[i + 3] + x * l3;
    t[i + 6] = t[i + 6] + x * l6;
    t[i + 7] = t[i + 7] + x * l7;
    t[i + 9] = t[i + 9] + x * l9;
  }
  return f12_fold(t, pc);
}
SPP_HDN F12 f12_frob(const F12& a, const PairingFastConsts& pc) {
  F12 r;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = 0; i < 12; i++) r.c[i] = a.c[i] * pc.FA[i] + a.c[(i + 6) % 12] * pc.FB[i];
  return r;
}
SPP_HD F12 f12_conj6(const F12& a) {   // a^(p^6): w -> -w
  F12 r = a;
  for (int k = 1; k < 12; k += 2) r.c[k] = a.c[k].neg();
  return r;
}
SPP_HDN F12 f12_pow_x(const F12& y, const PairingFastConsts& pc) {   // y^x, x = BN_X (bit 62 is the top bit)
  F12 r = y;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int b = 61; b >= 0; b--) {
    r = f12_mul(r, r, pc);
    if ((BN_X >> b) & 1) r = f12_mul(r, y, pc);
  }
  return r;
}
// f^((p^12-1)/r * m) == 1 with m = 2x(6x^2+3x+1) coprime to r  <=>  the pairing product is one
SPP_HDN bool final_exp_is_one(const F12& f, const PairingFastConsts& pc) {
  // inverse through the norm: t = f^p * f^(p^2) * ... * f^(p^11), f * t in Fq
  F12 g = f12_frob(f, pc);
  F12 t = g;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = 2; i <= 11; i++) {
    g = f12_frob(g, pc);
    t = f12_mul(t, g, pc);
  }
  const F12 n = f12_mul(f, t, pc);
  const Fq ninv = n.c[0].inv();                 // f = 0 cannot pass: 0^-1 = 0 gives y = 0 below
  F12 finv;
  for (int i = 0; i < 12; i++) finv.c[i] = t.c[i] * ninv;
  // easy part: y = f^((p^6-1)(p^2+1))
  F12 y = f12_mul(f12_conj6(f), finv, pc);
  y = f12_mul(f12_frob(f12_frob(y, pc), pc), y, pc);
  // hard part (y is unitary now: y^-1 = conj6(y))
  const F12 fx = f12_pow_x(y, pc);
  const F12 fx2 = f12_pow_x(fx, pc);
  const F12 fx3 = f12_pow_x(fx2, pc);
  const F12 a2 = f12_mul(fx, fx, pc);                        // fx^2
  const F12 a4 = f12_mul(a2, a2, pc);                        // fx^4
  const F12 b2 = f12_mul(fx2, fx2, pc);
  const F12 b6 = f12_mul(f12_mul(b2, b2, pc), b2, pc);       // fx2^6
  const F12 c2 = f12_mul(fx3, fx3, pc);
  const F12 c4 = f12_mul(c2, c2, pc);
  const F12 c12 = f12_mul(f12_mul(c4, c4, pc), c4, pc);      // fx3^12
  const F12 B = f12_mul(f12_mul(c12, b6, pc), a4, pc);       // y^l1
  const F12 C = f12_mul(B, a2, pc);                          // y^l2
  const F12 A = f12_mul(f12_mul(C, b6, pc), y, pc);          // y^l0
  const F12 D = f12_mul(f12_conj6(y), B, pc);                // y^l3
  F12 r = f12_mul(A, f12_frob(B, pc), pc);
  r = f12_mul(r, f12_frob(f12_frob(C, pc), pc), pc);
  r = f12_mul(r, f12_frob(f12_frob(f12_frob(D, pc), pc), pc), pc);
  bool ok = r.c[0] == pc.one;
  for (int i = 1; i < 12; i++) ok = ok && r.c[i].is_zero();
  return ok;
}

// (a + b u) w^k contributes (a - 9b) at w^k and b at w^(k+6)
SPP_HD void emb(const Fq2& v, Fq& lo, Fq& hi) {
  lo = v.c0 - v.c1.mul_small(9);
  hi = v.c1;
}

// One step of the shared Miller loop for the proof-specific G2 point, T in XYZZ over Fq2 (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2).  Lines are returned scaled by an Fq2 factor:
//   tangent at T :  2 Y ZZZ * yP  -  3 X^2 ZZ * xP w  +  (3 X^3 - 2 Y^2) w^3
//   chord T, Q   :  P' ZZZ * yP   -  R ZZ * xP w      +  (R ZZ xQ - P' ZZZ yQ) w^3,   P' = xQ ZZ - X, R = yQ ZZZ - Y
struct DynLine {
  Fq2 A, Bc, C;   // line = A*yP - Bc*xP w + C w^3
};
SPP_HDN DynLine dyn_double(G2XYZZ& T) {
  const Fq2 X2 = T.X.sqr();
  const Fq2 X2_3 = X2.dbl() + X2;
  const Fq2 Y2 = T.Y.sqr();
  DynLine l;
  l.A = (T.Y * T.ZZZ).dbl();
  l.Bc = X2_3 * T.ZZ;
  l.C = X2_3 * T.X - Y2.dbl();
  T.dbl_inplace();
  return l;
}
SPP_HDN DynLine dyn_add(G2XYZZ& T, const G2Affine& Q) {
  const Fq2 Pp = Q.x * T.ZZ - T.X;
  const Fq2 R = Q.y * T.ZZZ - T.Y;
  DynLine l;
  l.A = Pp * T.ZZZ;
  l.Bc = R * T.ZZ;
  l.C = l.Bc * Q.x - l.A * Q.y;
  T.madd(Q);
  return l;
}
SPP_HDN F12 mul_dyn_line(const F12& f, const DynLine& l, const G1Affine& P, const PairingFastConsts& pc) {
  Fq l0, l6, l1, l7, l3, l9;
  emb(Fq2{l.A.c0 * P.y, l.A.c1 * P.y}, l0, l6);
  emb(Fq2{l.Bc.c0 * P.x, l.Bc.c1 * P.x}.neg(), l1, l7);
  emb(l.C, l3, l9);
  return f12_mul_line(f, l0, l1, l3, l6, l7, l9, pc);
}
SPP_HDN F12 mul_table_line(const F12& f, const LineStep& s, const G1Affine& P, const PairingFastConsts& pc) {
  const Fq nx = P.x.neg();
  return f12_mul_line(f, P.y, nx * s.a1, s.a3, Fq::zero(), nx * s.b1, s.b3, pc);
}
SPP_HD uint32_t miller_steps() {
  uint32_t n = 64 + 2;
  for (int i = 0; i < 64; i++) n += (uint32_t)((ATE_LO >> i) & 1);
  return n;
}

// prod_k e(P_k, Q_k)  (k < nfixed: Q_k given by its line table; optional dynamic pair (Pd, Qd)) times `extra`,
// as the Miller-loop value before the final exponentiation.  Pairs whose P is at infinity contribute 1.
SPP_HDN F12 miller_multi(uint32_t nfixed, const LineStep* const* tables, const G1Affine* Ps, bool has_dyn, const G1Affine& Pd,
                         const G2Affine& Qd, const F12& extra, const PairingFastConsts& pc) {
  F12 f = f12_one(pc);
  const bool dyn = has_dyn && !Pd.is_inf() && !Qd.is_inf();
  G2XYZZ T = G2XYZZ::from_affine(Qd);
  uint32_t idx = 0;
  auto fixed_lines = [&]() {
    for (uint32_t k = 0; k < nfixed; k++)
      if (!Ps[k].is_inf()) f = mul_table_line(f, tables[k][idx], Ps[k], pc);
    idx++;
  };
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = 63; i >= 0; i--) {
    f = f12_mul(f, f, pc);
    if (dyn) f = mul_dyn_line(f, dyn_double(T), Pd, pc);
    fixed_lines();
    if ((ATE_LO >> i) & 1) {
      if (dyn) f = mul_dyn_line(f, dyn_add(T, Qd), Pd, pc);
      fixed_lines();
    }
  }
  if (dyn) {
    auto conj = [](const Fq2& a) { return Fq2{a.c0, a.c1.neg()}; };
    const G2Affine Q1{conj(Qd.x) * pc.g13, conj(Qd.y) * pc.g12};
    G2Affine Q2{conj(Q1.x) * pc.g13, conj(Q1.y) * pc.g12};
    Q2.y = Q2.y.neg();
    f = mul_dyn_line(f, dyn_add(T, Q1), Pd, pc);
    fixed_lines();
    f = mul_dyn_line(f, dyn_add(T, Q2), Pd, pc);
    fixed_lines();
  } else {
    fixed_lines();
    fixed_lines();
  }
  return f12_mul(f, extra, pc);
}

// everything k_verify needs about one verifying key, resident in HBM (built by spp_verify_batch)
struct VerifyKeyDev {
  PairingFastConsts pc;
  const LineStep* tab[4];   // line tables of gamma2, delta2, Pedersen G, Pedersen GSigmaNeg
  F12 e_alpha_beta;         // Miller-loop value of e(-alpha1, beta2)
  Fq2 twist_b;              // 3 / (9 + u)
  const G1Affine* K;        // K[0..nk-1]
  uint32_t nk;
};

// arguments of k_pairing_check (spp_pairing_check): up to 4 pairs; tab[k-1] = line table of Q[k] for k >= 1
struct PairingCheckDev {
  PairingFastConsts pc;
  Fq2 twist_b;
  const LineStep* tab[3];
  G1Affine P[4];
  G2Affine Q[4];
  uint32_t n;
};

// MSB-first double-and-add as a small rolled loop around out-of-line point operations.  Codegen hazard found on gfx950
// (ROCm 7.2, tests/micro/verify_probe.hip): with dbl/madd over Fq2 inlined, this LEAF function grew to ~100 KB, its
// loop back-edges needed long branches, and the branch relaxation scavenged s[30:31] -- the live return address -- for
// its s_getpc/s_setpc sequence: the function "returned" to itself and the kernel never finished.  Keeping every
// function that contains loops small (the heavy bodies are separate leaf functions without back-edges) avoids it.
template <class F>
SPP_HDN void xyzz_dbl_call(XYZZ<F>& a) { a.dbl_inplace(); }
template <class F>
SPP_HDN void xyzz_madd_call(XYZZ<F>& a, const Affine<F>& p) { a.madd(p); }
template <class F>
SPP_HDN XYZZ<F> scalar_mul_rolled(const Affine<F>& p, const uint32_t k[8]) {
  XYZZ<F> acc = XYZZ<F>::infinity();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int w = 7; w >= 0; w--) {
    const uint32_t kw = w == 7 ? k[7] : w == 6 ? k[6] : w == 5 ? k[5] : w == 4 ? k[4] : w == 3 ? k[3] : w == 2 ? k[2] : w == 1 ? k[1] : k[0];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int b = 31; b >= 0; b--) {
      xyzz_dbl_call(acc);
      if ((kw >> b) & 1) xyzz_madd_call(acc, p);
    }
  }
  return acc;
}
// [r]Q == O on the twist (proof-supplied G2 points must lie in the order-r subgroup)
SPP_HDN bool g2_in_subgroup(const G2Affine& Q) {
  uint32_t r[8];
  for (int i = 0; i < 8; i++) r[i] = FrParams::MOD(i);
  return scalar_mul_rolled(Q, r).is_inf();
}

}  // namespace spp

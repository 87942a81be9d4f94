// One Groth16 (+ BSB22 commitment) verification against a prepared key, host + gfx950: the body of k_verify
// (kernels_verify.hip, one lane per proof) and of the host check in tests/host/pairing_check.cpp.
// Same decisions, in the same order, as the host verifier spp_verify (csrc/spp_api.cpp), i.e. `sunspot verify`
// (noir_circuit/prove_linux.sh:86-87) and the byte layout withdraw.rs:13-16,63-90 fixes:
//   1. format: commitment count == 1, witness header; every coordinate < q and every public word < r (canonical encodings); G1 points on the curve, Bs on the twist AND in the order-r subgroup;
//   2. Pedersen proof of knowledge (gnark-crypto pedersen.VerifyingKey.Verify):  e(Cm, GSigmaNeg) * e(PoK, G) == 1;
//   3. challenge = fr.Hash(Cm, "bsb22-commitment");  ksum = K0 + sum pub_i K_i + challenge K_last + Cm;
//   4. e(Ar, Bs) * e(-alpha, beta) * e(-ksum, gamma) * e(-Krs, delta) == 1.
#pragma once
#include "sha256.hpp"
#include "pairing_fast.hpp"

namespace spp {

SPP_HD uint32_t be32_at(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
SPP_HDN bool bytes_all_zero(const uint8_t* b, int n) {
  uint32_t o = 0;
  for (int i = 0; i < n; i++) o |= b[i];
  return o == 0;
}
SPP_HDN Fq fq_from_be(const uint8_t* b) {
  uint8_t t[32];
  for (int i = 0; i < 32; i++) t[i] = b[i];
  return Fq::from_bytes_be(t);
}
SPP_HDN G1Affine g1_from_raw_hd(const uint8_t* b) {
  if (bytes_all_zero(b, 64)) return G1Affine::infinity();
  return {fq_from_be(b), fq_from_be(b + 32)};
}
SPP_HDN G2Affine g2_from_raw_hd(const uint8_t* b) {   // gnark raw: X.A1 | X.A0 | Y.A1 | Y.A0
  if (bytes_all_zero(b, 128)) return G2Affine::infinity();
  G2Affine p;
  p.x.c1 = fq_from_be(b);
  p.x.c0 = fq_from_be(b + 32);
  p.y.c1 = fq_from_be(b + 64);
  p.y.c0 = fq_from_be(b + 96);
  return p;
}
SPP_HDN bool g1_on_curve_hd(const G1Affine& p, const PairingFastConsts& pc) {
  if (p.is_inf()) return true;
  const Fq three = pc.one + pc.one + pc.one;
  return p.y.sqr() == p.x.sqr() * p.x + three;
}
SPP_HDN bool g2_on_curve_hd(const G2Affine& p, const Fq2& b) {
  if (p.is_inf()) return true;
  return p.y.sqr() == p.x.sqr() * p.x + b;
}
SPP_HDN G1XYZZ g1_scalar_mul_fr(const G1Affine& base, const Fr& k) {
  uint32_t lim[8];
  k.to_canonical(lim);
  return scalar_mul_rolled(base, lim);
}

SPP_HDN bool verify_one(const VerifyKeyDev& vk, const uint8_t* proof, const uint8_t* pw) {
  const PairingFastConsts& pc = vk.pc;
  const uint32_t npub = vk.nk - 2;
  if (be32_at(proof + 256) != 1) return false;
  if (be32_at(pw) != npub || be32_at(pw + 4) != 0 || be32_at(pw + 8) != npub) return false;
  // canonical encodings only: coordinates < q, public words < r (never reduced)
  for (int o = 0; o < 388; o += 32) {
    if (o == 256) o = 260;
    if (!be_is_canonical<FqParams>(proof + o)) return false;
  }
  for (uint32_t k = 0; k < npub; k++)
    if (!be_is_canonical<FrParams>(pw + 12 + 32 * k)) return false;
  const G1Affine Ar = g1_from_raw_hd(proof), Krs = g1_from_raw_hd(proof + 192), Cm = g1_from_raw_hd(proof + 260),
                 Pok = g1_from_raw_hd(proof + 324);
  const G2Affine Bs = g2_from_raw_hd(proof + 64);
  if (!g1_on_curve_hd(Ar, pc) || !g1_on_curve_hd(Krs, pc) || !g1_on_curve_hd(Cm, pc) || !g1_on_curve_hd(Pok, pc)) return false;
  if (!g2_on_curve_hd(Bs, vk.twist_b) || !g2_in_subgroup(Bs)) return false;
  {   // 2. proof of knowledge of the commitment
    const LineStep* tabs[2] = {vk.tab[2], vk.tab[3]};
    const G1Affine Ps[2] = {Pok, Cm};   // tab[2] = G, tab[3] = GSigmaNeg: e(PoK, G) * e(Cm, GSigmaNeg)
    const F12 f = miller_multi(2, tabs, Ps, false, G1Affine::infinity(), G2Affine::infinity(), f12_one(pc), pc);
    if (!final_exp_is_one(f, pc)) return false;
  }
  // 3. challenge and the public-input combination
  uint32_t m[16];
  for (int k = 0; k < 16; k++) m[k] = be32_at(proof + 260 + 4 * k);
  const Fr challenge = bsb22_challenge(m);
  G1XYZZ ksum = G1XYZZ::from_affine(vk.K[0]);
  for (uint32_t k = 0; k <= npub; k++) {
    Fr v = challenge;
    if (k < npub) {
      uint8_t t[32];
      for (int b = 0; b < 32; b++) t[b] = pw[12 + 32 * k + b];
      v = Fr::from_bytes_be(t);
    }
    ksum.add(g1_scalar_mul_fr(vk.K[k + 1], v));
  }
  ksum.madd(Cm);
  // 4. the Groth16 equation
  const LineStep* tabs[2] = {vk.tab[0], vk.tab[1]};
  const G1Affine Ps[2] = {ksum.to_affine().neg(), Krs.neg()};
  const F12 f = miller_multi(2, tabs, Ps, true, Ar, Bs, vk.e_alpha_beta, pc);
  return final_exp_is_one(f, pc);
}

}  // namespace spp

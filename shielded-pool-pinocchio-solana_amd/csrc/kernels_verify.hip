// Batched Groth16 (+ BSB22 commitment) verification, one lane per proof -- SURVEY 8f-4: the GPU counterpart of
// `sunspot verify <vk> <proof> <pw>` (noir_circuit/prove_linux.sh:86-87, audit_circuit/prove_audit.sh:98-99) and of the
// checks the deployed verifier makes (withdraw.rs:13-16,63-90: 388-byte proof, 12-byte witness header + 32 B inputs).
// Same decisions, in the same order, as the host verifier spp_verify (csrc/spp_api.cpp), which tests compare it with:
//   1. format: commitment count == 1; G1 points on the curve, Bs on the twist AND in the order-r subgroup;
//   2. Pedersen proof of knowledge:  e(Cm, GSigmaNeg) * e(PoK, G) == 1;
//   3. challenge = fr.Hash(Cm, "bsb22-commitment");  ksum = K0 + sum pub_i K_i + challenge K_last + Cm;
//   4. e(Ar, Bs) * e(-alpha, beta) * e(-ksum, gamma) * e(-Krs, delta) == 1.
// All pairing arithmetic is csrc/pairing_fast.hpp (shared Miller loop, per-key line tables, x-power final exponent).
#include "kernels.hpp"
#include "verify_one.hpp"

namespace spp {

__global__ void __launch_bounds__(64) k_verify(const VerifyKeyDev* __restrict__ vkp, const uint8_t* __restrict__ proofs,
                                               const uint8_t* __restrict__ pws, uint32_t pw_len, uint32_t count, int32_t* __restrict__ ok) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  ok[i] = verify_one(*vkp, proofs + (size_t)i * 388, pws + (size_t)i * pw_len) ? 1 : 0;
}

// prod_k e(P_k, Q_k) == 1 for caller-supplied points, with exactly the device functions k_verify uses (pair 0 through the
// projective-line path a proof's Bs takes, pairs 1.. through host-built line tables like the key-side points), preceded by
// the curve and subgroup checks.  One lane; exists so that key material made by gnark (the reference's .vk files) can be
// put through the device pairing code (spp_pairing_check).
__global__ void __launch_bounds__(64) k_pairing_check(const PairingCheckDev* __restrict__ a, int32_t* __restrict__ ok) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const PairingFastConsts& pc = a->pc;
  bool good = true;
  for (uint32_t k = 0; k < a->n; k++) {
    good = good && g1_on_curve_hd(a->P[k], pc) && !a->Q[k].is_inf() && g2_on_curve_hd(a->Q[k], a->twist_b) && g2_in_subgroup(a->Q[k]);
  }
  int32_t res = 0;
  if (good) {
    const LineStep* tabs[3] = {a->tab[0], a->tab[1], a->tab[2]};
    const F12 f = miller_multi(a->n - 1, tabs, a->P + 1, true, a->P[0], a->Q[0], f12_one(pc), pc);
    res = final_exp_is_one(f, pc) ? 1 : 0;
  }
  *ok = res;
}
void launch_pairing_check(hipStream_t st, const PairingCheckDev* a, int32_t* ok) {
  hipLaunchKernelGGL(k_pairing_check, dim3(1), dim3(64), 0, st, a, ok);
}

void launch_verify(hipStream_t st, const VerifyKeyDev* vk, const uint8_t* proofs, const uint8_t* pws, uint32_t pw_len, uint32_t count,
                   int32_t* ok) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_verify, dim3((count + 63) / 64), dim3(64), 0, st, vk, proofs, pws, pw_len, count, ok);
}

}  // namespace spp

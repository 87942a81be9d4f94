// Batched Groth16 (+ BSB22 commitment) verification, one lane per proof -- SURVEY 8f-4: the GPU counterpart of
// `sunspot verify <vk> <proof> <pw>` (noir_circuit/prove_linux.sh:86-87, audit_circuit/prove_audit.sh:98-99) and of the
// checks the deployed verifier makes (withdraw.rs:13-16,63-90: 388-byte proof, 12-byte witness header + 32 B inputs).
// Same decisions, in the same order, as the host verifier spp_verify (csrc/spp_api.cpp), which tests compare it with:
//   1. format: commitment count == 1; G1 points on the curve, Bs on the twist AND in the order-r subgroup;
//   2. Pedersen proof of knowledge:  e(Cm, G) * e(PoK, GSigmaNeg) == 1;
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

void launch_verify(hipStream_t st, const VerifyKeyDev* vk, const uint8_t* proofs, const uint8_t* pws, uint32_t pw_len, uint32_t count,
                   int32_t* ok) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_verify, dim3((count + 63) / 64), dim3(64), 0, st, vk, proofs, pws, pw_len, count, ok);
}

}  // namespace spp

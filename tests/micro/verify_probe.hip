// Device diagnostics for the batched verifier (test infrastructure): runs the pieces of csrc/verify_one.hpp in separate
// one-wave kernels and prints the time of each, so that a slow or non-terminating piece can be told apart.
// Build: hipcc -O3 --offload-arch=gfx950 -I shielded-pool-pinocchio-solana_amd/csrc tests/micro/verify_probe.hip -o tests/micro/verify_probe
// Run:   verify_probe <vk> <proof> <pw>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <hip/hip_runtime.h>
#include "pairing_fast_host.hpp"
#include "verify_one.hpp"
using namespace spp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); fflush(stdout); return 1; } } while (0)

__global__ void k_f12mul(const PairingFastConsts* pc, F12* io, int reps) {
  F12 a = io[0], b = io[1];
  for (int r = 0; r < reps; r++) a = f12_mul(a, b, *pc);
  if (threadIdx.x == 0) io[2] = a;
}
__global__ void k_frob(const PairingFastConsts* pc, F12* io) { F12 a = f12_frob(io[0], *pc); if (threadIdx.x == 0) io[2] = a; }
__global__ void k_powx(const PairingFastConsts* pc, F12* io) { F12 a = f12_pow_x(io[0], *pc); if (threadIdx.x == 0) io[2] = a; }
__global__ void k_finalexp(const PairingFastConsts* pc, F12* io, int* out) { bool r = final_exp_is_one(io[0], *pc); if (threadIdx.x == 0) out[0] = r; }
__global__ void k_subgroup(const G2Affine* q, int* out) { bool r = g2_in_subgroup(*q); if (threadIdx.x == 0) out[0] = r; }
__global__ void k_miller_fixed(const VerifyKeyDev* vk, const G1Affine* P, F12* io) {
  const LineStep* tabs[2] = {vk->tab[0], vk->tab[0]};
  const G1Affine Ps[2] = {P[0], P[0].neg()};
  F12 f = miller_multi(2, tabs, Ps, false, G1Affine::infinity(), G2Affine::infinity(), f12_one(vk->pc), vk->pc);
  if (threadIdx.x == 0) io[2] = f;
}
__global__ void k_verify1(const VerifyKeyDev* vk, const uint8_t* proof, const uint8_t* pw, int* out) {
  bool r = verify_one(*vk, proof, pw);
  if (threadIdx.x == 0) out[0] = r;
}

static std::vector<uint8_t> slurp(const char* path) {
  std::vector<uint8_t> v; FILE* f = fopen(path, "rb"); if (!f) return v; int c; while ((c = fgetc(f)) != EOF) v.push_back((uint8_t)c); fclose(f); return v;
}
template <class T> static T* up(const T* h, size_t n) { T* d; hipMalloc((void**)&d, n * sizeof(T)); hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice); return d; }
#define TIMED(name, ...) do { auto t0 = std::chrono::steady_clock::now(); __VA_ARGS__; CK(hipDeviceSynchronize()); \
  printf("%-16s %.3f ms\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); fflush(stdout); } while (0)

int main(int argc, char** argv) {
  if (argc != 4) { printf("usage: verify_probe vk proof pw\n"); return 2; }
  std::vector<uint8_t> vk = slurp(argv[1]), proof = slurp(argv[2]), pw = slurp(argv[3]);
  const uint32_t nk = be32_at(vk.data() + 576);
  size_t off = 580;
  std::vector<G1Affine> K(nk);
  for (uint32_t i = 0; i < nk; i++) K[i] = g1_from_raw_hd(vk.data() + off + 64 * (size_t)i);
  off += (size_t)nk * 64 + 12;
  const G1Affine alpha1 = g1_from_raw_hd(vk.data());
  const G2Affine beta2 = g2_from_raw_hd(vk.data() + 128), gamma2 = g2_from_raw_hd(vk.data() + 256), delta2 = g2_from_raw_hd(vk.data() + 448);
  const G2Affine pedG = g2_from_raw_hd(vk.data() + off), pedGS = g2_from_raw_hd(vk.data() + off + 128);
  std::vector<LineStep> t[4] = {build_line_table(gamma2), build_line_table(delta2), build_line_table(pedG), build_line_table(pedGS)};
  VerifyKeyDev h;
  h.pc = make_pairing_fast_consts();
  for (int k = 0; k < 4; k++) h.tab[k] = up(t[k].data(), t[k].size());
  h.e_alpha_beta = f12_from(miller_loop(alpha1.neg(), beta2));
  h.twist_b = twist_b();
  h.K = up(K.data(), K.size());
  h.nk = nk;
  VerifyKeyDev* dvk = up(&h, 1);
  PairingFastConsts* dpc = up(&h.pc, 1);
  F12 io[3];
  io[0] = h.e_alpha_beta; io[1] = h.e_alpha_beta; io[2] = f12_one(h.pc);
  F12* dio = up(io, 3);
  int* dout = up((int*)io, 1);
  G2Affine* dq = up(&gamma2, 1);
  G1Affine* dP = up(&K[0], 1);
  uint8_t* dproof = up(proof.data(), proof.size());
  uint8_t* dpw = up(pw.data(), pw.size());
  int out = -1;
  printf("start\n"); fflush(stdout);
  TIMED("f12_mul x1", hipLaunchKernelGGL(k_f12mul, dim3(1), dim3(64), 0, 0, dpc, dio, 1));
  TIMED("f12_mul x100", hipLaunchKernelGGL(k_f12mul, dim3(1), dim3(64), 0, 0, dpc, dio, 100));
  TIMED("frob", hipLaunchKernelGGL(k_frob, dim3(1), dim3(64), 0, 0, dpc, dio));
  TIMED("pow_x", hipLaunchKernelGGL(k_powx, dim3(1), dim3(64), 0, 0, dpc, dio));
  TIMED("subgroup", hipLaunchKernelGGL(k_subgroup, dim3(1), dim3(64), 0, 0, dq, dout));
  CK(hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost)); printf("  in subgroup: %d\n", out); fflush(stdout);
  TIMED("miller fixed", hipLaunchKernelGGL(k_miller_fixed, dim3(1), dim3(64), 0, 0, dvk, dP, dio));
  F12 one = f12_one(h.pc);
  CK(hipMemcpy(dio, &one, sizeof one, hipMemcpyHostToDevice));
  TIMED("final_exp(1)", hipLaunchKernelGGL(k_finalexp, dim3(1), dim3(64), 0, 0, dpc, dio, dout));
  CK(hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost)); printf("  final_exp(1) is one: %d\n", out); fflush(stdout);
  TIMED("verify_one", hipLaunchKernelGGL(k_verify1, dim3(1), dim3(64), 0, 0, dvk, dproof, dpw, dout));
  CK(hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost)); printf("  verify_one: %d\n", out); fflush(stdout);
  return 0;
}

// Micro-benchmark (test infrastructure): times launch_ntt of the product's NTT kernels on [n][P] x nbatch data with
// HIP events.  Build: hipcc -O3 --offload-arch=gfx950 -I shielded-pool-pinocchio-solana_amd/csrc tests/micro/ntt_bench.hip
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef NTT_SRC
#define NTT_SRC "kernels_ntt.hip"
#endif
#include NTT_SRC
using namespace spp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  uint32_t logn = argc > 1 ? atoi(argv[1]) : 13, P = argc > 2 ? atoi(argv[2]) : 2048, nb = argc > 3 ? atoi(argv[3]) : 3;
  const uint32_t n = 1u << logn;
  size_t total = (size_t)n * P * nb;
  Fr *d, *tw;
  CK(hipMalloc((void**)&d, total * sizeof(Fr)));
  CK(hipMalloc((void**)&tw, (n / 2) * sizeof(Fr)));
  std::vector<Fr> h(n / 2);
  Fr w = Fr::from_u64(5), a = Fr::one();
  for (uint32_t k = 0; k < n / 2; k++) { h[k] = a; a = a * w; }
  CK(hipMemcpy(tw, h.data(), h.size() * sizeof(Fr), hipMemcpyHostToDevice));
  CK(hipMemset(d, 0x11, total * sizeof(Fr)));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int dif = 1; dif >= 0; dif--) {
    launch_ntt(st, d, logn, P, tw, dif, nb, (size_t)n * P);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    const int reps = 5;
    for (int r = 0; r < reps; r++) launch_ntt(st, d, logn, P, tw, dif, nb, (size_t)n * P);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double bf = (double)nb * n / 2 * logn * P;
    printf("%s logn=%u P=%u nbatch=%u: %.3f ms per transform set, %.2f G butterflies/s, %.0f GB/s HBM (2 passes r+w)\n", dif ? "DIF" : "DIT", logn, P, nb,
           ms / reps, bf / (ms / reps * 1e-3) / 1e9, (double)total * 32 * 2 * ((logn + 7) / 8) / (ms / reps * 1e-3) / 1e9);
  }
  return 0;
}

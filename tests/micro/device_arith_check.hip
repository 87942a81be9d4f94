// Diagnostic: the SAME bn254.hpp functions evaluated on the GPU and on the host for random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "bn254.hpp"
using namespace spp;
template <class F>
__global__ void k_ops(const F* a, const F* b, F* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[5 * i + 0] = a[i] * b[i];
  out[5 * i + 1] = a[i].sqr();
  out[5 * i + 2] = a[i] + b[i];
  out[5 * i + 3] = a[i] - b[i];
  out[5 * i + 4] = a[i].neg();
}
static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }
template <class F>
static int check(const char* name) {
  const int n = 4096;
  std::vector<F> a(n), b(n), out(5 * n);
  for (int i = 0; i < n; i++) {
    uint32_t x[8], y[8];
    for (int k = 0; k < 8; k++) { x[k] = rnd(); y[k] = rnd(); }
    if (i < 8) for (int k = 0; k < 8; k++) { x[k] = (i & 1) ? 0xffffffffu : 0; y[k] = (i & 2) ? 0xffffffffu : (i & 4 ? 1 : 0); }
    a[i] = F::from_u256(x);
    b[i] = F::from_u256(y);
  }
  F *da, *db, *dout;
  hipMalloc((void**)&da, sizeof(F) * n); hipMalloc((void**)&db, sizeof(F) * n); hipMalloc((void**)&dout, sizeof(F) * 5 * n);
  hipMemcpy(da, a.data(), sizeof(F) * n, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), sizeof(F) * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_ops<F>, dim3((n + 63) / 64), dim3(64), 0, 0, da, db, dout, n);
  hipMemcpy(out.data(), dout, sizeof(F) * 5 * n, hipMemcpyDeviceToHost);
  int bad[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    F e[5] = {a[i] * b[i], a[i].sqr(), a[i] + b[i], a[i] - b[i], a[i].neg()};
    for (int k = 0; k < 5; k++) if (e[k] != out[5 * i + k]) { if (!bad[k]) printf("%s op %d first mismatch at %d\n", name, k, i); bad[k]++; }
  }
  printf("%s: mul %d sqr %d add %d sub %d neg %d mismatches of %d\n", name, bad[0], bad[1], bad[2], bad[3], bad[4], n);
  return bad[0] + bad[1] + bad[2] + bad[3] + bad[4];
}
int main() { int r = check<Fr>("Fr") + check<Fq>("Fq"); return r ? 1 : 0; }

#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "bn254.hpp"
using namespace spp;
__global__ void k_mul(const Fq* a, const Fq* b, Fq* o, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) o[i] = a[i] * b[i]; }
static uint64_t st = 88172645463325252ull;
static uint32_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); }
int main() {
  const int n = 256;
  std::vector<Fq> a(n), b(n), o(n), o1(n);
  for (int i = 0; i < n; i++) { uint32_t x[8], y[8]; for (int k = 0; k < 8; k++) { x[k] = rnd(); y[k] = rnd(); } a[i] = Fq::from_u256(x); b[i] = Fq::from_u256(y); }
  Fq *da, *db, *dd;
  hipMalloc((void**)&da, sizeof(Fq) * n); hipMalloc((void**)&db, sizeof(Fq) * n); hipMalloc((void**)&dd, sizeof(Fq) * n);
  hipMemcpy(da, a.data(), sizeof(Fq) * n, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), sizeof(Fq) * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mul, dim3(n / 64), dim3(64), 0, 0, da, db, dd, n);
  hipMemcpy(o.data(), dd, sizeof(Fq) * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_mul, dim3(1), dim3(1), 0, 0, da + i, db + i, dd + i, 1);
  hipMemcpy(o1.data(), dd, sizeof(Fq) * n, hipMemcpyDeviceToHost);
  int bad = 0, bad1 = 0;
  for (int i = 0; i < n; i++) {
    Fq e = a[i] * b[i];
    if (e != o[i]) { if (bad < 3) { printf("lane %d: a=", i); for (int k = 7; k >= 0; k--) printf("%08x", a[i].l[k]); printf(" b="); for (int k = 7; k >= 0; k--) printf("%08x", b[i].l[k]);
      printf("\n  dev="); for (int k = 7; k >= 0; k--) printf("%08x", o[i].l[k]); printf("\n host="); for (int k = 7; k >= 0; k--) printf("%08x", e.l[k]); printf("\n"); } bad++; }
    if (e != o1[i]) bad1++;
  }
  printf("64-lane launch: %d bad; single-lane launches: %d bad of %d\n", bad, bad1, n);
  return 0;
}

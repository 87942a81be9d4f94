#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "bn254.hpp"
using namespace spp;
// stage dump: 0..8 a9, 9..17 b9, 18..35 product columns, 36..53 columns after reduction loop k<8
__host__ __device__ void stages(const Fq& a, const Fq& b, uint64_t* o) {
  uint32_t a9[9], b9[9];
  Fq::to9(a.l, a9); Fq::to9(b.l, b9);
  for (int i = 0; i < 9; i++) { o[i] = a9[i]; o[9 + i] = b9[i]; }
  uint64_t c[18];
  for (int k = 0; k < 18; k++) c[k] = 0;
  for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)a9[i] * b9[j];
  for (int k = 0; k < 18; k++) o[18 + k] = c[k];
  for (int k = 0; k < 8; k++) {
    const uint32_t m = ((uint32_t)c[k] * FqParams::INV32) & Fq::M29;
    for (int j = 0; j < 9; j++) c[k + j] += (uint64_t)m * Fq::P9(j);
    c[k + 1] += c[k] >> 29;
  }
  for (int k = 0; k < 18; k++) o[36 + k] = c[k];
  for (int j = 0; j < 9; j++) o[54 + j] = Fq::P9(j);
  {
    const uint32_t m = ((uint32_t)c[8] * FqParams::INV32) & ((1u << 24) - 1u);
    for (int j = 0; j < 9; j++) c[8 + j] += (uint64_t)m * Fq::P9(j);
  }
  for (int k = 0; k < 10; k++) o[64 + k] = c[8 + k];
  const uint32_t lo5 = (uint32_t)(c[8] >> 24) & 31u;
  c[9] += c[8] >> 29;
  uint32_t n[9];
  for (int k = 0; k < 8; k++) { n[k] = (uint32_t)c[9 + k] & Fq::M29; c[10 + k] += c[9 + k] >> 29; }
  n[8] = (uint32_t)c[17];
  o[74] = lo5;
  for (int k = 0; k < 9; k++) o[75 + k] = n[k];
  Fq r = a * b;
  for (int k = 0; k < 8; k++) o[84 + k] = r.l[k];
  Fq r2 = a.sqr();
  for (int k = 0; k < 8; k++) o[92 + k] = r2.l[k];
}
__global__ void k(const Fq* a, const Fq* b, uint64_t* o) { stages(a[0], b[0], o); }
int main() {
  uint32_t x[8] = {0xba394238, 0xf4b3ead1, 0x9748907f, 0x18b10832, 0x6d401c9b, 0x6c310f49, 0x8f4d524d, 0x1e55d413};
  uint32_t y[8] = {0x3d505e75, 0x062fbd2c, 0x63c81655, 0x2bd0e3a3, 0x72412256, 0x5d5a6581, 0x3070406d, 0x28e88da5};
  Fq a, b; for (int i = 0; i < 8; i++) { a.l[i] = x[i]; b.l[i] = y[i]; }
  Fq *da, *db; uint64_t* d;
  hipMalloc((void**)&da, sizeof a); hipMalloc((void**)&db, sizeof b); hipMalloc((void**)&d, 8 * 128);
  hipMemcpy(da, &a, sizeof a, hipMemcpyHostToDevice); hipMemcpy(db, &b, sizeof b, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, da, db, d);
  uint64_t dev[100], host[100];
  hipMemcpy(dev, d, sizeof dev, hipMemcpyDeviceToHost);
  stages(a, b, host);
  for (int i = 0; i < 100; i++) if (dev[i] != host[i]) printf("stage %d: dev %016llx host %016llx\n", i, (unsigned long long)dev[i], (unsigned long long)host[i]);
  printf("done\n");
  return 0;
}

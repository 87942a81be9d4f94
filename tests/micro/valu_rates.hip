// Micro-benchmark (diagnostic only): issue cost of the integer / FP64 VALU instructions a 254-bit Montgomery
// multiplication can be built from, on gfx950. Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define ITER 2000
__global__ void __launch_bounds__(256) k_mad64(uint64_t* out, uint32_t seed) {
  uint64_t a0 = (uint64_t)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  uint32_t x = seed | 1, y = threadIdx.x | 3;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %4, %5, %1\nv_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

__global__ void __launch_bounds__(256) k_mullo(uint64_t* out, uint32_t seed) {
  uint32_t a0 = (uint32_t)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  uint32_t y = threadIdx.x | 3;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      asm volatile("v_mul_lo_u32 %0, %0, %4\nv_mul_lo_u32 %1, %1, %4\nv_mul_lo_u32 %2, %2, %4\nv_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y) );
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

__global__ void __launch_bounds__(256) k_mulhi(uint64_t* out, uint32_t seed) {
  uint32_t a0 = (uint32_t)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  uint32_t y = threadIdx.x | 0x80000003u;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      asm volatile("v_mul_hi_u32 %0, %0, %4\nv_mul_hi_u32 %1, %1, %4\nv_mul_hi_u32 %2, %2, %4\nv_mul_hi_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y) );
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

__global__ void __launch_bounds__(256) k_mad24(uint64_t* out, uint32_t seed) {
  uint32_t a0 = (uint32_t)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  uint32_t x = (seed | 1) & 0xffffff, y = threadIdx.x | 3;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      asm volatile("v_mad_u32_u24 %0, %4, %5, %0\nv_mad_u32_u24 %1, %4, %5, %1\nv_mad_u32_u24 %2, %4, %5, %2\nv_mad_u32_u24 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) );
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

__global__ void __launch_bounds__(256) k_add32(uint64_t* out, uint32_t seed) {
  uint32_t a0 = (uint32_t)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  uint32_t y = threadIdx.x | 3;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      asm volatile("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y) );
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

__global__ void __launch_bounds__(256) k_addc(uint64_t* out, uint32_t seed) {
  uint32_t a0 = (uint32_t)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  uint32_t y = threadIdx.x | 3;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      asm volatile("v_add_co_u32 %0, vcc, %0, %4\nv_addc_co_u32 %1, vcc, %1, %4, vcc\nv_addc_co_u32 %2, vcc, %2, %4, vcc\nv_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

__global__ void __launch_bounds__(256) k_lshladd64(uint64_t* out, uint32_t seed) {
  uint64_t a0 = (uint64_t)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  uint64_t y = threadIdx.x | 3;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      asm volatile("v_lshl_add_u64 %0, %0, 0, %4\nv_lshl_add_u64 %1, %1, 0, %4\nv_lshl_add_u64 %2, %2, 0, %4\nv_lshl_add_u64 %3, %3, 0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y) );
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

__global__ void __launch_bounds__(256) k_fma64(uint64_t* out, uint32_t seed) {
  double a0 = (double)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  double x = 1.0000001, y = 1e-9;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      asm volatile("v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) );
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)(a0 + a1 + a2 + a3);
}

__global__ void __launch_bounds__(256) k_alignbit(uint64_t* out, uint32_t seed) {
  uint32_t a0 = (uint32_t)(seed + threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  uint32_t y = threadIdx.x | 3;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      asm volatile("v_alignbit_b32 %0, %0, %4, 13\nv_alignbit_b32 %1, %1, %4, 13\nv_alignbit_b32 %2, %2, %4, 13\nv_alignbit_b32 %3, %3, %4, 13" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y) );
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

typedef void (*kern_t)(uint64_t*, uint32_t);
static void run(const char* name, kern_t kern, int blocks_per_cu) {
  uint64_t* out;
  int blocks = 256 * blocks_per_cu;
  hipMalloc((void**)&out, sizeof(uint64_t) * 256 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)ITER * REP * blocks_per_cu;   // one wave per SIMD per block
  double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
  printf("%-14s waves/SIMD=%d  %8.3f ms  %6.2f cycles per wave-instruction per SIMD (2.4 GHz nominal)\n", name, blocks_per_cu, ms, cyc);
  hipFree(out);
}
int main() {
  struct { const char* n; kern_t k; } ks[] = {{"mad_u64_u32", k_mad64}, {"mul_lo_u32", k_mullo}, {"mul_hi_u32", k_mulhi},
    {"mad_u32_u24", k_mad24}, {"add_u32", k_add32}, {"addc chain", k_addc}, {"lshl_add_u64", k_lshladd64}, {"fma_f64", k_fma64},
    {"alignbit_b32", k_alignbit}};
  for (int w : {1, 2, 4})
    for (auto& k : ks) run(k.n, k.k, w);
  return 0;
}

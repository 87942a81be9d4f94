// Calibration of the HBM counters on THIS path's access pattern (MI355X_MICROARCH.md, "HBM": "Other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute").
//
// The table walks (k_msm_flat) and the bucket kernel (k_pip_segments) read one randomly placed 64-byte affine point per
// addition: four 16-byte loads by ONE lane.  profiles/summarize_pmc.py doubles FETCH_SIZE as the guide prescribes for wide
// coalesced streams (a 128-byte request tallied at 64); whether a lone 64-byte gather is one 128-byte request (doubling right:
// half of every line is wasted) or one 64-byte request (doubling wrong) is what this measures, by time and under
// `rocprofv3 --pmc FETCH_SIZE`:
//   k_gather<64>    each lane reads 64 B at a random 64-byte-aligned place of a 4 GiB table
//   k_gather<128>   each lane reads 128 B at a random 128-byte-aligned place (a whole line)
//   k_gather<32>    each lane reads 32 B at a random 32-byte-aligned place
//   k_stream        16 B per lane, coalesced (the guide's calibrated case: FETCH_SIZE = half the bytes)
// Every kernel reads `n` places; the known byte counts are n * width.  Build: hipcc -O3 --offload-arch=gfx950 gather64.hip -o gather64
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

// four independent places in flight per lane (the loads of a group are issued before any is used): as much memory-level
// parallelism as the CU tracks.  If this exceeds ~49 G places/s, a 64-byte gather cannot be moving a whole 128-byte line
// (49 G x 128 B = the 6.3 TB/s the guide gives as achievable).
template <int WIDTH>
__global__ void __launch_bounds__(256) k_gather4(const uint4* __restrict__ table, uint64_t slots, uint32_t per_lane, uint32_t* __restrict__ sink) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = {0, 0, 0, 0};
  for (uint32_t k = 0; k < per_lane; k += 4) {
    uint4 v[4][WIDTH / 16];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t slot = mix(g * 0x9e3779b97f4a7c15ull + k + u) % slots;
      const uint4* p = table + slot * (WIDTH / 16);
#pragma unroll
      for (int w = 0; w < WIDTH / 16; w++) v[u][w] = p[w];
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int w = 0; w < WIDTH / 16; w++) { acc.x ^= v[u][w].x; acc.y ^= v[u][w].y; acc.z ^= v[u][w].z; acc.w ^= v[u][w].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int WIDTH>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, uint64_t slots, uint32_t per_lane, uint32_t* __restrict__ sink) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = {0, 0, 0, 0};
  for (uint32_t k = 0; k < per_lane; k++) {
    const uint64_t slot = mix(g * 0x9e3779b97f4a7c15ull + k) % slots;         // uniformly random, no two lanes of a wave adjacent
    const uint4* p = table + slot * (WIDTH / 16);
#pragma unroll
    for (int w = 0; w < WIDTH / 16; w++) {
      const uint4 v = p[w];
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;           // never true for the pattern below: keeps the loads
}

// The discriminating case: 64 B at a random 128-byte-aligned place, then -- at an address that DEPENDS on what came back, so
// strictly later -- the other 64 B of the same line.  One fabric request per place (FETCH_SIZE = 64 B per place) means the first
// request brought the whole line: a lone 64-byte gather moves 128 B.  Two requests (128 B per place) mean the L2 fills 64-byte halves.
__global__ void __launch_bounds__(256) k_gather_halves(const uint4* __restrict__ table, uint64_t lines, uint32_t per_lane, uint32_t* __restrict__ sink) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = {0, 0, 0, 0};
  for (uint32_t k = 0; k < per_lane; k++) {
    const uint64_t line = mix(g * 0x9e3779b97f4a7c15ull + k) % lines;
    const uint4* p = table + line * 8;
    uint4 v[4];
#pragma unroll
    for (int w = 0; w < 4; w++) v[w] = p[w];
    const uint32_t zero = (~(v[0].x & v[1].x & v[2].x & v[3].x)) & 1u;       // every table word is odd: always 0, unknown to the compiler
    const uint4* q = p + 4 + zero * 8;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint4 u = q[w];
      acc.x ^= v[w].x ^ u.x; acc.y ^= v[w].y ^ u.y; acc.z ^= v[w].z ^ u.z; acc.w ^= v[w].w ^ u.w;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ table, uint64_t n16, uint32_t* __restrict__ sink) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint4 acc = {0, 0, 0, 0};
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const uint4 v = table[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

__global__ void k_fill(uint4* t, uint64_t n16) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const uint32_t v = (uint32_t)mix(i) | 1u;                                  // odd, and so are its odd multiples
    t[i] = uint4{v, v * 3u, v * 5u, v * 7u};
  }
}

template <int WIDTH, bool FOUR = false>
static void run(const uint4* table, uint64_t bytes, uint32_t* sink, const char* name) {
  const uint32_t per_lane = 16, blocks = 256 * 64;                           // 2^26 places
  const uint64_t places = (uint64_t)blocks * 256 * per_lane;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto kern = FOUR ? k_gather4<WIDTH> : k_gather<WIDTH>;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, table, bytes / WIDTH, per_lane, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, table, bytes / WIDTH, per_lane, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-14s places %llu  bytes asked %.3f GB  %.3f ms  %.2f G places/s  %.0f GB/s asked  (%.0f GB/s if every place moves a 128-byte line)\n", name,
         (unsigned long long)places, places * (double)WIDTH / 1e9, ms, places / ms / 1e6, places * (double)WIDTH / ms / 1e6,
         places * 128.0 / ms / 1e6);
}

int main() {
  const uint64_t bytes = 4ull << 30;                                           // far beyond the 256 MiB Infinity Cache
  uint4* table; uint32_t* sink;
  CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(sink, 0, 4));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, table, bytes / 16);
  CHECK(hipDeviceSynchronize());
  run<32>(table, bytes, sink, "k_gather<32>");
  run<64>(table, bytes, sink, "k_gather<64>");
  run<128>(table, bytes, sink, "k_gather<128>");
  run<32, true>(table, bytes, sink, "k_gather4<32>");
  run<64, true>(table, bytes, sink, "k_gather4<64>");
  run<128, true>(table, bytes, sink, "k_gather4<128>");
  hipLaunchKernelGGL(k_gather_halves, dim3(256 * 64), dim3(256), 0, 0, table, bytes / 128, 16u, sink);
  CHECK(hipDeviceSynchronize());
  printf("k_gather_halves ran: 67108864 places, 64 B + the other 64 B of the same line afterwards (read its FETCH_SIZE under rocprofv3)\n");
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, table, bytes / 16, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, table, bytes / 16, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-14s bytes %.3f GB  %.3f ms  %.0f GB/s\n", "k_stream", bytes / 1e9, ms, bytes / ms / 1e6);
  uint32_t h = 0;
  CHECK(hipMemcpy(&h, sink, 4, hipMemcpyDeviceToHost));
  printf("sink %u\n", h);
  return 0;
}

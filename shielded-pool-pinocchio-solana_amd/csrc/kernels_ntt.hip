// Radix-2 NTT over BN254 Fr for the R1CS -> QAP step (gnark computeH: 3 inverse, 3 coset-forward, 1
// coset-inverse transform of size n = 2^13..2^15 per proof), batched over the proofs of a batch.
//
// Layout [n][P] (P = proofs, fastest).  A transform is done in ceil(log n / 8) passes; each pass keeps a
// group of up to 256 butterfly-connected elements x T adjacent columns in LDS (64 KiB), runs up to 8
// radix-2 stages there, and touches HBM once for read and once for write: 64 B per element per pass,
// always as runs of >= 256 contiguous bytes (adjacent columns = adjacent proofs / adjacent low indices).
// DIF (natural -> bit-reversed) and DIT (bit-reversed -> natural) forms are both provided so the whole
// computeH pipeline needs no bit-reversal pass: coefficient-side tables are simply stored bit-reversed.
#include "kernels.hpp"

namespace spp {

static constexpr uint32_t NTT_TILE_ELEMS = 1024;  // 32 KiB of Fr in LDS: 4-5 blocks per CU hide the load/store phases of each other
static constexpr uint32_t NTT_THREADS = 256;

template <bool DIF>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_pass(Fr* __restrict__ data, uint32_t logn, uint32_t P, const Fr* __restrict__ tw,
                                                          uint32_t s_lo, uint32_t s_hi, size_t batch_stride, const Fr* __restrict__ post) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  Fr* sh = reinterpret_cast<Fr*>(smem_raw);
  data += (size_t)blockIdx.y * batch_stride;
  const uint32_t n = 1u << logn;
  const uint32_t lg = s_hi - s_lo;
  const uint32_t G = 1u << lg;
  const uint32_t T = NTT_TILE_ELEMS / G;
  // element (j, col): index = bh*G*stride + j*stride + base_low, column q = base_low*P + p within [0, Cq)
  const uint32_t stride = DIF ? (n >> s_hi) : (1u << s_lo);
  const uint64_t Cq = (uint64_t)stride * P;
  const uint64_t ncols = ((uint64_t)n >> lg) * P;
  const uint64_t col0 = (uint64_t)blockIdx.x * T;
  const bool col_major = Cq >= T;  // adjacent tile columns are adjacent in memory

  if (col_major && T <= NTT_THREADS) {
    // ---- fast path: a lane keeps ONE column for the whole pass, so the (64-bit) column -> address / twiddle-base
    // arithmetic is done once per lane instead of once per element and butterfly ----
    const uint32_t tc = threadIdx.x & (T - 1), jr = threadIdx.x / T, jstep = NTT_THREADS / T;
    const uint64_t col = col0 + tc;
    const bool valid = col < ncols;
    const uint64_t bh = valid ? col / Cq : 0, q = valid ? col % Cq : 0;
    const uint32_t base_low = (uint32_t)(q / P);
    Fr* ptr = data + bh * G * Cq + q;
    if (valid)
      for (uint32_t j = jr; j < G; j += jstep) sh[j * T + tc] = ptr[(uint64_t)j * Cq];
    __syncthreads();
    for (uint32_t t = 0; t < lg; t++) {
      const uint32_t half_l = DIF ? (G >> (t + 1)) : (1u << t);
      const uint32_t mult_log = logn - 1 - (31 - __builtin_clz(half_l)) - (31 - __builtin_clz(stride));
      if (valid)
        for (uint32_t jj = jr; jj < G / 2; jj += jstep) {
          const uint32_t lo = jj & (half_l - 1);
          const uint32_t j = ((jj - lo) << 1) + lo;
          const uint32_t e = (lo * stride + base_low) << mult_log;
          Fr u = sh[j * T + tc];
          Fr v = sh[(j + half_l) * T + tc];
          if (DIF) {
            sh[j * T + tc] = u + v;
            Fr d = u - v;
            sh[(j + half_l) * T + tc] = e ? d * tw[e] : d;
          } else {
            if (e) v = v * tw[e];
            sh[j * T + tc] = u + v;
            sh[(j + half_l) * T + tc] = u - v;
          }
        }
      __syncthreads();
    }
    if (valid) {
      if (post) {   // fused row scaling (coset shift): element row = bh*G*stride + j*stride + base_low
        const uint32_t row0 = (uint32_t)(bh * G) * stride + base_low;
        for (uint32_t j = jr; j < G; j += jstep) ptr[(uint64_t)j * Cq] = sh[j * T + tc] * post[row0 + j * stride];
      } else {
        for (uint32_t j = jr; j < G; j += jstep) ptr[(uint64_t)j * Cq] = sh[j * T + tc];
      }
    }
    return;
  }

  // ---- generic path (tiny transforms, or fewer contiguous columns than the tile is wide) ----
  // ---- load tile ----
  for (uint32_t idx = threadIdx.x; idx < G * T; idx += NTT_THREADS) {
    uint32_t j, tc;
    if (col_major) { tc = idx % T; j = idx / T; } else { j = idx % G; tc = idx / G; }
    uint64_t col = col0 + tc;
    if (col < ncols) {
      uint64_t bh = col / Cq, q = col % Cq;
      sh[j * T + tc] = data[bh * G * Cq + (uint64_t)j * Cq + q];
    }
  }
  __syncthreads();

  // ---- butterflies ----
  for (uint32_t t = 0; t < lg; t++) {
    const uint32_t half_l = DIF ? (G >> (t + 1)) : (1u << t);
    // twiddle exponent = ((j mod half_l)*stride + base_low) * mult ; mult = n / (2*half_l*stride)
    const uint32_t mult_log = logn - 1 - (31 - __builtin_clz(half_l)) - (31 - __builtin_clz(stride));
    for (uint32_t idx = threadIdx.x; idx < (G / 2) * T; idx += NTT_THREADS) {
      uint32_t tc = idx % T, jj = idx / T;
      uint64_t col = col0 + tc;
      if (col >= ncols) continue;
      uint32_t j = (jj / half_l) * 2 * half_l + (jj % half_l);
      uint32_t base_low = (uint32_t)((col % Cq) / P);
      uint32_t e = ((jj % half_l) * stride + base_low) << mult_log;
      Fr u = sh[j * T + tc];
      Fr v = sh[(j + half_l) * T + tc];
      if (DIF) {
        sh[j * T + tc] = u + v;
        Fr d = u - v;
        sh[(j + half_l) * T + tc] = e ? d * tw[e] : d;
      } else {
        if (e) v = v * tw[e];
        sh[j * T + tc] = u + v;
        sh[(j + half_l) * T + tc] = u - v;
      }
    }
    __syncthreads();
  }

  // ---- store tile ----
  for (uint32_t idx = threadIdx.x; idx < G * T; idx += NTT_THREADS) {
    uint32_t j, tc;
    if (col_major) { tc = idx % T; j = idx / T; } else { j = idx % G; tc = idx / G; }
    uint64_t col = col0 + tc;
    if (col < ncols) {
      uint64_t bh = col / Cq, q = col % Cq;
      Fr v = sh[j * T + tc];
      if (post) v = v * post[(uint32_t)(bh * G) * stride + j * stride + (uint32_t)(q / P)];
      data[bh * G * Cq + (uint64_t)j * Cq + q] = v;
    }
  }
}

// post (optional): table of n row factors applied to the OUTPUT rows (data[i][p] *= post[i]) while the last pass stores them --
// the coset shifts of computeH, which used to be separate full read + write passes over the arrays (k_scale_rows)
void launch_ntt(hipStream_t st, Fr* data, uint32_t logn, uint32_t P, const Fr* tw, bool dif, uint32_t nbatch, size_t batch_stride,
                const Fr* post) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_ntt_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NTT_TILE_ELEMS * sizeof(Fr)));
    (void)hipFuncSetAttribute((const void*)k_ntt_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NTT_TILE_ELEMS * sizeof(Fr)));
    attr_set = true;
  }
  uint32_t npass = (logn + 7) / 8;
  uint32_t s = 0;
  for (uint32_t ps = 0; ps < npass; ps++) {
    uint32_t lg = (logn - s + (npass - ps) - 1) / (npass - ps);
    uint32_t G = 1u << lg, T = NTT_TILE_ELEMS / G;
    uint64_t ncols = ((uint64_t)1 << (logn - lg)) * P;
    dim3 grid((uint32_t)((ncols + T - 1) / T), nbatch);
    size_t shmem = (size_t)NTT_TILE_ELEMS * sizeof(Fr);
    if (dif)
      hipLaunchKernelGGL(k_ntt_pass<true>, grid, dim3(NTT_THREADS), shmem, st, data, logn, P, tw, s, s + lg, batch_stride,
                         ps + 1 == npass ? post : (const Fr*)nullptr);
    else
      hipLaunchKernelGGL(k_ntt_pass<false>, grid, dim3(NTT_THREADS), shmem, st, data, logn, P, tw, s, s + lg, batch_stride,
                         ps + 1 == npass ? post : (const Fr*)nullptr);
    s += lg;
  }
}

// data[b][i][p] *= table[i]
__global__ void __launch_bounds__(256) k_scale_rows(Fr* __restrict__ data, const Fr* __restrict__ table, uint32_t n, uint32_t P,
                                                    size_t batch_stride) {
  data += (size_t)blockIdx.y * batch_stride;
  uint64_t total = (uint64_t)n * P;
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t i = (uint32_t)(g / P);
    data[g] = data[g] * table[i];
  }
}
void launch_scale_rows(hipStream_t st, Fr* data, const Fr* table, uint32_t n, uint32_t P, uint32_t nbatch, size_t batch_stride) {
  uint64_t total = (uint64_t)n * P;
  uint32_t blocks = (uint32_t)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_scale_rows, dim3(blocks, nbatch), dim3(256), 0, st, data, table, n, P, batch_stride);
}

// a <- (a*b - c) * zinv over [n][P]  (evaluations on the coset; Z is the constant g^n - 1 there)
__global__ void __launch_bounds__(256) k_qap_pointwise(Fr* __restrict__ abc, uint64_t total, Fr zinv) {
  Fr* a = abc;
  const Fr* b = abc + total;
  const Fr* c = abc + 2 * total;
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x)
    a[g] = (a[g] * b[g] - c[g]) * zinv;
}
// a <- a * b over [n][P]: the values of A(X) B(X) on the coset zeta * H (product form of computeH, spp_api.cpp)
__global__ void __launch_bounds__(256) k_qap_product(Fr* __restrict__ abc, uint64_t total) {
  Fr* a = abc;
  const Fr* b = abc + total;
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) a[g] = a[g] * b[g];
}
void launch_qap_product(hipStream_t st, Fr* abc, uint32_t n, uint32_t P) {
  uint64_t total = (uint64_t)n * P;
  uint32_t blocks = (uint32_t)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_qap_product, dim3(blocks), dim3(256), 0, st, abc, total);
}
void launch_qap_pointwise(hipStream_t st, Fr* abc, uint32_t n, uint32_t P, Fr zinv) {
  uint64_t total = (uint64_t)n * P;
  uint32_t blocks = (uint32_t)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_qap_pointwise, dim3(blocks), dim3(256), 0, st, abc, total, zinv);
}

}  // namespace spp

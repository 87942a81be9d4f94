// Fixed-base multi-scalar multiplication for batched Groth16 proving on gfx950.
//
// Replaces the Pippenger MSMs of gnark's Prove: Ar / Bs1 / Krs over pk.G1.{A,B,K,Z},
// Bs over pk.G2.B, and the BSB22 Pedersen commitment + proof of knowledge. In the reference these run
// inside `sunspot prove` (client/proof.helper.ts:64) on CPU threads, one proof at a time.
//
// MI355X design: the proving key is constant across a batch and there are 288 GB of HBM, so every base
// carries a precomputed table of its window multiples  T[i][j][d] = (d+1) * 2^(c*j) * Base_i  (affine,
// d < 2^(c-1), signed digits).  An MSM then needs no buckets, no sorting and no bucket reduction: each
// lane owns one proof of the batch and folds  sum_i sum_j  +-T[i][j][|digit|]  into a private XYZZ
// accumulator with mixed additions (8M+2S).  All 64 lanes of a wave walk the SAME (i, j) sequence, so
// scalar loads are one coalesced 2 KiB row, table gathers hit one 2^(c-1)*64 B segment, control flow is
// uniform, and windows in which every proof has a zero digit (bits, bytes, small signed noise -- most of
// the audit witness) are skipped for the whole wave.  Work is split over S slices of the base range to
// fill 256 CUs; the S partial sums of every proof are then folded pairwise (k_msm_fold, log2 S launches).
#include <hip/hip_ext.h>
#include "kernels.hpp"
#include <algorithm>
#include <cstdlib>
#include "f29.hpp"

namespace spp {

uint32_t msm_windows(uint32_t c) { return (254 + c - 1) / c; }

// Window chunks per base for small batches (1 = a lane takes whole bases): when 4 bases per slice cannot give ~64K lanes, the
// windows of a base are split over up to 8 lanes (>= 4 windows each); msm_slices(N * Q', P) lanes then share the items.
uint32_t msm_window_chunks(uint32_t N, uint32_t P, uint32_t c) {
  const uint32_t Wn = msm_windows(c);
  const uint64_t target = 65536;
  uint32_t Q = 1;
  while ((uint64_t)((N + 3) / 4) * Q * P < target && Q < 8 && (Wn + 2 * Q - 1) / (2 * Q) >= 4) Q *= 2;
  return Q;
}
uint32_t msm_slices_split(uint32_t N, uint32_t P, uint32_t Q) {
  if (Q <= 1) return msm_slices(N, P);
  const uint64_t items = (uint64_t)N * Q;
  uint64_t S = (65536 + P - 1) / P;
  if (S > items) S = items;
  if (S == 0) S = 1;
  return (uint32_t)S;
}

// enough (slice, proof) lanes to fill 256 CUs x 4 SIMDs x ~4 waves, but at least 4 bases per slice
uint32_t msm_slices(uint32_t N, uint32_t P) {
  static const uint32_t waves_per_simd = [] {   // SPP_MSM_WAVES (experiment): lanes launched = 256 CUs x 4 SIMDs x this x 64
    const char* e = getenv("SPP_MSM_WAVES");
    const int v = e ? atoi(e) : 4;
    return (uint32_t)(v >= 1 && v <= 8 ? v : 4);
  }();
  const uint32_t target_lanes = 256u * 4u * waves_per_simd * 64u;
  uint32_t S = (target_lanes + P - 1) / P;
  uint32_t maxS = (N + 3) / 4;
  if (maxS == 0) maxS = 1;
  if (S > maxS) S = maxS;
  if (S == 0) S = 1;
  return S;
}

// ----------------------------------------------------------------------------------------------------
// table construction: one lane per (base, window) row.
// Table layout: rows are grouped in blocks of 64; entry d of row `row` lives at ((row/64)*E + d)*64 + row%64, so the
// 64 lanes of a wave that build 64 consecutive rows write 64 consecutive points (coalesced 4 KiB per step), and so do
// the XYZZ temporaries.  The MSM kernels gather single entries at random d anyway, so they lose nothing.
// ----------------------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(64) k_build_table(const Affine<F>* __restrict__ bases, uint32_t N, uint32_t c, uint32_t Wn,
                                                    uint32_t row0, uint32_t nrows, Affine<F>* __restrict__ table,
                                                    XYZZ<F>* __restrict__ tmp, F* __restrict__ tmp_pre) {
  const uint32_t rl = blockIdx.x * blockDim.x + threadIdx.x;   // row within this launch (row0 is a multiple of 64)
  if (rl >= nrows) return;
  const uint32_t row = row0 + rl;
  const uint32_t E = 1u << (c - 1);
  Affine<F>* out = table + ((size_t)(row >> 6) * E) * 64 + (row & 63);   // entry d at out[d * 64]
  XYZZ<F>* t = tmp + rl;                                                  // entry d at t[d * nrows]
  F* pre = tmp_pre + rl;
  if (row >= N * Wn) {   // padding rows of the last block
    for (uint32_t d = 0; d < E; d++) out[(size_t)d * 64] = Affine<F>::infinity();
    return;
  }
  const uint32_t i = row / Wn, j = row % Wn;
  Affine<F> base = bases[i];
  if (base.is_inf()) {
    for (uint32_t d = 0; d < E; d++) out[(size_t)d * 64] = Affine<F>::infinity();
    return;
  }
  XYZZ<F> b = XYZZ<F>::from_affine(base);
  for (uint32_t k = 0; k < c * j; k++) b.dbl_inplace();
  Affine<F> bj = b.to_affine();
  XYZZ<F> acc = XYZZ<F>::from_affine(bj);
  F prod = F::one();
  for (uint32_t d = 0; d < E; d++) {
    t[(size_t)d * nrows] = acc;
    pre[(size_t)d * nrows] = prod;
    prod = prod * (acc.ZZ * acc.ZZZ);
    acc.madd(bj);
  }
  F inv = prod.inv();
  for (uint32_t d = E; d-- > 0;) {
    XYZZ<F> q = t[(size_t)d * nrows];
    F I = inv * pre[(size_t)d * nrows];
    inv = inv * (q.ZZ * q.ZZZ);
    F izz = I * q.ZZZ;
    F izzz = I * q.ZZ;
    out[(size_t)d * 64] = {q.X * izz, q.Y * izzz};
  }
}

// number of table elements (points) for N bases at window c, including the padding of the last 64-row block
size_t msm_table_elems(uint32_t N, uint32_t c) {
  size_t rows = (size_t)N * msm_windows(c);
  return ((rows + 63) / 64) * 64 * ((size_t)1 << (c - 1));
}

template <class F>
void launch_build_table(hipStream_t st, const Affine<F>* bases, uint32_t N, uint32_t c, uint32_t row0, uint32_t nrows, Affine<F>* table,
                        XYZZ<F>* tmp, F* tmp_pre) {
  if (nrows == 0) return;
  hipLaunchKernelGGL(k_build_table<F>, dim3((nrows + 63) / 64), dim3(64), 0, st, bases, N, c, msm_windows(c), row0, nrows, table, tmp,
                     tmp_pre);
}
template void launch_build_table<Fq>(hipStream_t, const Affine<Fq>*, uint32_t, uint32_t, uint32_t, uint32_t, Affine<Fq>*, XYZZ<Fq>*, Fq*);
template void launch_build_table<Fq2>(hipStream_t, const Affine<Fq2>*, uint32_t, uint32_t, uint32_t, uint32_t, Affine<Fq2>*, XYZZ<Fq2>*,
                                      Fq2*);

// ----------------------------------------------------------------------------------------------------
// signed-window recoding helpers (scalar in canonical limbs, magnitude < 2^253 after sign folding)
// ----------------------------------------------------------------------------------------------------
struct Recoder {
  uint32_t l[8];
  uint32_t carry;
  bool neg;
  __device__ __forceinline__ void init(const Fr& s) {
    uint32_t cl[8];
    s.to_canonical(cl);
    neg = canonical_gt_half<FrParams>(cl);
    if (neg) {
      canonical_negate<FrParams>(cl, l);
    } else {
      SPP_UNROLL for (int k = 0; k < 8; k++) l[k] = cl[k];
    }
    carry = 0;
  }
  __device__ __forceinline__ bool rest_is_zero() const {
    uint32_t o = carry;
    SPP_UNROLL for (int k = 0; k < 8; k++) o |= l[k];
    return o == 0;
  }
  // next window: returns magnitude (0..2^(c-1)) and sign (true = subtract)
  __device__ __forceinline__ uint32_t next(uint32_t c, bool& sgn) {
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    uint32_t d = (l[0] & mask) + carry;
    SPP_UNROLL for (int k = 0; k < 7; k++) l[k] = (l[k] >> c) | (l[k + 1] << (32 - c));
    l[7] >>= c;
    if (d > half) {
      d = (1u << c) - d;
      carry = 1;
      sgn = !neg;
    } else {
      carry = 0;
      sgn = neg;
    }
    return d;
  }
};

// ----------------------------------------------------------------------------------------------------
// MSM accumulate: lane g -> (slice = g / P, proof p = g % P)
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool words_all_zero(const Fq& v) {
  uint32_t o = 0;
  SPP_UNROLL for (int i = 0; i < 8; i++) o |= v.l[i];
  return o == 0;
}
// accumulator used by the table walk: G1 and G2 run on the unsaturated 9x29-bit form (f29.hpp); the generic
// template (saturated Fp words) is kept for other coordinate fields
template <class F>
struct MsmAcc {
  XYZZ<F> a;
  __device__ __forceinline__ void init() { a = XYZZ<F>::infinity(); }
  __device__ __forceinline__ void madd(Affine<F> e, bool sgn) {
    if (sgn) e.y = e.y.neg();
    a.madd(e);
  }
  __device__ __forceinline__ XYZZ<F> result() const { return a; }
};
template <>
struct MsmAcc<Fq> {
  XYZZ29<FqParams> a;
  __device__ __forceinline__ void init() { a = XYZZ29<FqParams>::infinity(); }
  __device__ __forceinline__ void madd(const Affine<Fq>& e, bool sgn) {
    if (words_all_zero(e.x) && words_all_zero(e.y)) return;   // table row of an infinity base (spp_msm_g1 callers)
    a.madd(e, sgn);
  }
  __device__ __forceinline__ XYZZ<Fq> result() const { return a.to_xyzz(); }
};

template <>
struct MsmAcc<Fq2> {
  XYZZ29G2 a;
  __device__ __forceinline__ void init() { a = XYZZ29G2::infinity(); }
  __device__ __forceinline__ void madd(const Affine<Fq2>& e, bool sgn) {
    if (words_all_zero(e.x.c0) && words_all_zero(e.x.c1) && words_all_zero(e.y.c0) && words_all_zero(e.y.c1)) return;
    a.madd(e, sgn);
  }
  __device__ __forceinline__ XYZZ<Fq2> result() const { return a.to_xyzz(); }
};

template <class F>
__global__ void __launch_bounds__(256) k_msm_fixed(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ rows,
                                                   const Fr* __restrict__ scalars, XYZZ<F>* __restrict__ partial, uint32_t N,
                                                   uint32_t P, uint32_t c, uint32_t Wn, uint32_t S) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= S * P) return;
  const uint32_t p = g % P, slice = g / P;
  const uint32_t E = 1u << (c - 1);
  // slice s takes bases s, s+S, s+2S, ...: neighbouring wires have similar scalar sizes (runs of bits, runs of hash
  // states), so a strided split gives every slice the same mix and the launch no tail of heavy slices
  MsmAcc<F> acc;
  acc.init();
  for (uint32_t i = slice; i < N; i += S) {
    Fr s = scalars[(size_t)rows[i] * P + p];
    if (s.is_zero()) continue;
    Recoder rc;
    rc.init(s);
    uint32_t row = i * Wn;
#pragma unroll 1
    for (uint32_t j = 0; j < Wn; j++, row++) {
      if (rc.rest_is_zero()) break;
      bool sgn;
      uint32_t d = rc.next(c, sgn);
      if (d != 0) {
        acc.madd(table[((size_t)(row >> 6) * E + (d - 1)) * 64 + (row & 63)], sgn);
      }
    }
  }
  partial[(size_t)slice * P + p] = acc.result();
}

// Small batches (a single proof is the drop-in generateProof case): with one slice per 4 bases a lane walks 4 x Wn windows one
// after the other -- 128 dependent additions, 0.7 ms per set for one withdraw proof on 32 of the chip's 1024 SIMDs.  Here the work
// item is (base, chunk of Wq windows): Q = ceil(Wn / Wq) items per base, lane g -> (slice, proof), slice walks items slice,
// slice + S, ...  The recoder still runs from window 0 (the signed digits carry upwards), a few shifts per skipped window.
template <class F>
__global__ void __launch_bounds__(256) k_msm_fixed_split(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ rows,
                                                         const Fr* __restrict__ scalars, XYZZ<F>* __restrict__ partial, uint32_t N,
                                                         uint32_t P, uint32_t c, uint32_t Wn, uint32_t S, uint32_t Q, uint32_t Wq) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= S * P) return;
  const uint32_t p = g % P, slice = g / P;
  const uint32_t E = 1u << (c - 1);
  MsmAcc<F> acc;
  acc.init();
  const uint32_t items = N * Q;
  for (uint32_t t = slice; t < items; t += S) {
    const uint32_t i = t % N, q = t / N;   // chunk-major: the lanes of a wave share q (same skip length, same add phase)
    Fr s = scalars[(size_t)rows[i] * P + p];
    if (s.is_zero()) continue;
    Recoder rc;
    rc.init(s);
    const uint32_t j0 = q * Wq, j1 = j0 + Wq < Wn ? j0 + Wq : Wn;
    // skip phase apart from the add phase: the lanes of a wave hold different chunks q, and an addition inside a loop
    // whose trip count differs per lane would be executed once per distinct j (a few live lanes each time)
    bool sgn;
#pragma unroll 1
    for (uint32_t j = 0; j < j0; j++) (void)rc.next(c, sgn);
#pragma unroll 1
    for (uint32_t j = j0; j < j1; j++) {
      if (rc.rest_is_zero()) break;
      uint32_t d = rc.next(c, sgn);
      if (d != 0) {
        const uint32_t row = i * Wn + j;
        acc.madd(table[((size_t)(row >> 6) * E + (d - 1)) * 64 + (row & 63)], sgn);
      }
    }
  }
  partial[(size_t)slice * P + p] = acc.result();
}

// fold the S partial sums of every proof: pairwise, one launch per level, every lane busy -- lane (s, p) with s < S_cur - half
// adds partial[s + half][p] into partial[s][p] (half = ceil(S_cur / 2)); log2(S) launches of S*P/2, S*P/4, ... lanes.  (The
// first version used one 64-lane block per proof with an LDS tree: most lanes idle, six dependent additions behind barriers;
// it cost 1.9 ms per G1 set and 10.7 ms for the G2 set on a 4096-proof batch, 12 % of a step.)
template <class F>
__global__ void __launch_bounds__(64) k_msm_fold(XYZZ<F>* __restrict__ partial, XYZZ<F>* __restrict__ out, uint32_t P, uint32_t half,
                                                 uint32_t S_cur) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t pairs = S_cur - half;
  if (g >= half * P) return;
  const uint32_t s = g / P, p = g % P;
  XYZZ<F> a = partial[(size_t)s * P + p];
  if (s < pairs) a.add(partial[(size_t)(s + half) * P + p]);
  if (half == 1) out[p] = a;                       // last level: the result leaves the scratch array
  else if (s < pairs) partial[(size_t)s * P + p] = a;
}
template <class F>
__global__ void __launch_bounds__(256) k_msm_fill_inf(XYZZ<F>* __restrict__ out, uint32_t P) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < P) out[g] = XYZZ<F>::infinity();
}

// ev_start / ev_stop (optional): receive the dispatch's own start and stop timestamps (hipExtLaunchKernelGGL), i.e. the
// kernel's duration as a profiler reports it -- an event pair recorded around the launch would also count the time the
// launch waits for kernels of the other proving stream.
template <class F>
void launch_msm_accumulate(hipStream_t st, const Affine<F>* table, const uint32_t* rows, const Fr* scalars, XYZZ<F>* partial, uint32_t N,
                           uint32_t P, uint32_t c, uint32_t S, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t Q) {
  if (N == 0 || S == 0) {
    if (ev_start) hipEventRecord(ev_start, st);
    if (ev_stop) hipEventRecord(ev_stop, st);
    return;
  }
  uint64_t lanes = (uint64_t)S * P;
  if (Q > 1) {
    const uint32_t Wn = msm_windows(c);
    hipExtLaunchKernelGGL(k_msm_fixed_split<F>, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, st, ev_start, ev_stop, 0, table, rows,
                          scalars, partial, N, P, c, Wn, S, Q, (Wn + Q - 1) / Q);
    return;
  }
  hipExtLaunchKernelGGL(k_msm_fixed<F>, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, st, ev_start, ev_stop, 0, table, rows, scalars,
                        partial, N, P, c, msm_windows(c), S);
}
template <class F>
void launch_msm_reduce(hipStream_t st, XYZZ<F>* partial, XYZZ<F>* out, uint32_t P, uint32_t S) {
  if (P == 0) return;
  if (S == 0) {
    hipLaunchKernelGGL(k_msm_fill_inf<F>, dim3((P + 255) / 256), dim3(256), 0, st, out, P);
    return;
  }
  uint32_t cur = S;
  do {
    const uint32_t half = (cur + 1) / 2;
    const uint64_t lanes = (uint64_t)half * P;
    hipLaunchKernelGGL(k_msm_fold<F>, dim3((uint32_t)((lanes + 63) / 64)), dim3(64), 0, st, partial, out, P, half, cur);
    cur = half;
  } while (cur > 1);
}
// the same fold for up to MSM_FOLD_SETS sets at once (blockIdx.y = set): the five G1 sums of a proof are independent, and one
// launch per level for all of them instead of one per level and set takes 64 of the 80 ~9 us launches off a single proof
template <class F>
__global__ void __launch_bounds__(64) k_msm_fold_multi(MsmFoldSets<F> fs, uint32_t P) {
  const uint32_t set = blockIdx.y;
  const uint32_t half = fs.half[set], S_cur = fs.cur[set];
  if (half == 0) return;   // this set is already folded
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t pairs = S_cur - half;
  if (g >= half * P) return;
  XYZZ<F>* __restrict__ partial = fs.partial[set];
  const uint32_t s = g / P, p = g % P;
  XYZZ<F> a = partial[(size_t)s * P + p];
  if (s < pairs) a.add(partial[(size_t)(s + half) * P + p]);
  if (half == 1) fs.out[set][p] = a;
  else if (s < pairs) partial[(size_t)s * P + p] = a;
}
template <class F>
void launch_msm_reduce_multi(hipStream_t st, MsmFoldSets<F> fs, uint32_t nsets, const uint32_t* S, uint32_t P) {
  if (P == 0 || nsets == 0) return;
  uint32_t cur[MSM_FOLD_SETS];
  bool done[MSM_FOLD_SETS];
  for (uint32_t i = 0; i < nsets; i++) {
    cur[i] = S[i];
    done[i] = false;
    if (S[i] == 0) {
      hipLaunchKernelGGL(k_msm_fill_inf<F>, dim3((P + 255) / 256), dim3(256), 0, st, fs.out[i], P);
      done[i] = true;
    }
  }
  for (;;) {
    uint64_t lanes = 0;
    for (uint32_t i = 0; i < nsets; i++) {
      fs.half[i] = done[i] ? 0 : (cur[i] + 1) / 2;
      fs.cur[i] = cur[i];
      lanes = std::max<uint64_t>(lanes, (uint64_t)fs.half[i] * P);
    }
    if (lanes == 0) break;
    hipLaunchKernelGGL(k_msm_fold_multi<F>, dim3((uint32_t)((lanes + 63) / 64), nsets), dim3(64), 0, st, fs, P);
    for (uint32_t i = 0; i < nsets; i++)
      if (!done[i]) {
        cur[i] = fs.half[i];
        if (cur[i] == 1) done[i] = true;
      }
  }
}
template void launch_msm_reduce_multi<Fq>(hipStream_t, MsmFoldSets<Fq>, uint32_t, const uint32_t*, uint32_t);
template void launch_msm_accumulate<Fq>(hipStream_t, const Affine<Fq>*, const uint32_t*, const Fr*, XYZZ<Fq>*, uint32_t, uint32_t, uint32_t,
                                        uint32_t, hipEvent_t, hipEvent_t, uint32_t);
template void launch_msm_accumulate<Fq2>(hipStream_t, const Affine<Fq2>*, const uint32_t*, const Fr*, XYZZ<Fq2>*, uint32_t, uint32_t,
                                         uint32_t, uint32_t, hipEvent_t, hipEvent_t, uint32_t);
template void launch_msm_reduce<Fq>(hipStream_t, XYZZ<Fq>*, XYZZ<Fq>*, uint32_t, uint32_t);
template void launch_msm_reduce<Fq2>(hipStream_t, XYZZ<Fq2>*, XYZZ<Fq2>*, uint32_t, uint32_t);

// ----------------------------------------------------------------------------------------------------
// setup: out[i] = scalars[i] * G using the window table of the single base G
// ----------------------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(64) k_fixed_base_mul(const Affine<F>* __restrict__ gen_table, uint32_t c, uint32_t Wn,
                                                       const Fr* __restrict__ scalars, uint32_t n, Affine<F>* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t E = 1u << (c - 1);
  Fr s = scalars[i];
  XYZZ<F> acc = XYZZ<F>::infinity();
  if (!s.is_zero()) {
    Recoder rc;
    rc.init(s);
#pragma unroll 1
    for (uint32_t j = 0; j < Wn; j++) {
      bool sgn;
      uint32_t d = rc.next(c, sgn);
      if (d != 0) {
        Affine<F> e = gen_table[((size_t)(j >> 6) * E + (d - 1)) * 64 + (j & 63)];
        if (sgn) e.y = e.y.neg();
        acc.madd(e);
      }
    }
  }
  out[i] = acc.to_affine();
}
template <class F>
void launch_fixed_base_mul(hipStream_t st, const Affine<F>* gen_table, uint32_t c, const Fr* scalars, uint32_t n, Affine<F>* out,
                           XYZZ<F>* /*tmp*/) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_fixed_base_mul<F>, dim3((n + 63) / 64), dim3(64), 0, st, gen_table, c, msm_windows(c), scalars, n, out);
}
template void launch_fixed_base_mul<Fq>(hipStream_t, const Affine<Fq>*, uint32_t, const Fr*, uint32_t, Affine<Fq>*, XYZZ<Fq>*);
template void launch_fixed_base_mul<Fq2>(hipStream_t, const Affine<Fq2>*, uint32_t, const Fr*, uint32_t, Affine<Fq2>*, XYZZ<Fq2>*);

}  // namespace spp

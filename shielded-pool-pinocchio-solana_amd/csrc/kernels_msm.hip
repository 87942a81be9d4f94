// Fixed-base multi-scalar multiplication for batched Groth16 proving on gfx950.
//
// Replaces the Pippenger MSMs of gnark's Prove: Ar / Bs1 / Krs over pk.G1.{A,B,K,Z},
// Bs over pk.G2.B, and the BSB22 Pedersen commitment + proof of knowledge. In the reference these run
// inside `sunspot prove` (client/proof.helper.ts:64) on CPU threads, one proof at a time.
//
// MI355X design: the proving key is constant across a batch and there are 288 GB of HBM, so every base
// carries a precomputed table of its multiples (affine, signed c-bit digits: 2^(c-1) entries per row).  An MSM
// then needs no buckets, no sorting and no bucket reduction: each lane owns one proof of the batch and folds
// +-T[base][|digit|] into a private XYZZ accumulator with mixed additions (8M+2S).  The scalars are recoded once
// into int16 digit planes (k_msm_digits); all 64 lanes of a wave walk the SAME (base, window) sequence, so digit
// loads are one coalesced row, table gathers hit one 2^(c-1)*64 B segment, control flow is uniform, and windows in
// which every proof has a zero digit (bits, bytes, small signed noise -- most of the audit witness) are skipped for
// the whole wave.  Under the HBM budget a base keeps ONE row (16-bit windows, 2 MB per G1 base) and the 16 windows
// of a scalar become 16 passes over the same table whose sums are put together by Horner (k_msm_horner); with
// explicit small windows every window has its own row and there is a single pass.  Work is split over passes and
// slices of the base range to fill 256 CUs; the slice sums of every (pass, proof) are folded pairwise.
// This file is compiled TWICE: as it is (every non-template function and the G1 instantiations, with the scheduler strategy
// max-ilp: the G1 walk is 2 % faster for it, 247 registers, still two waves per SIMD) and through kernels_msm_g2.hip with
// SPP_MSM_TU_G2 defined (the G2 instantiations alone, default scheduler: max-ilp makes the 512-register G2 walk spill and 4 % slower).
#include <hip/hip_ext.h>
#include "kernels.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include "f29.hpp"

#ifndef SPP_MSM_WALK_BLOCK
#define SPP_MSM_WALK_BLOCK 64
#endif
static constexpr unsigned MSM_FOLD_COOP_MAX_BATCH = 8;            // batches up to this size fold their slice sums with 8 lanes per output
static constexpr unsigned MSM_WALK_BLOCK = SPP_MSM_WALK_BLOCK;   // lanes per workgroup of the flat table walk
#ifndef SPP_G1_GATHER_PIPELINE
#define SPP_G1_GATHER_PIPELINE 0   // experiment: the one-deep gather pipeline of the G2 walk for G1 as well
#endif

namespace spp {

#ifndef SPP_MSM_TU_G2
uint32_t msm_windows(uint32_t c) { return (254 + c - 1) / c; }

// Lane layout of one launch (see kernels.hpp, MsmPlan).  Big batches: the chip holds 1024 SIMDs x `occ` waves of this kernel at
// a time (occ = 2 for G1 at ~200 VGPRs, 1 for G2); the waves of a launch take about the same time each, so a launch of w waves
// runs ceil(w / capacity) rounds and the last, partly filled round costs a whole one.  Sg is therefore searched around
// SPP_MSM_WAVES (default 4) rounds for the value that fills its last round best (17 passes x 8 slices x 32 waves were 2.125
// rounds: the 15-bit sets ran 12 % slower per addition than the 16-bit ones until this was done).  At least 4 bases per slice.
// Small batches (a single proof is the drop-in generateProof case): when that cannot give ~64K lanes the table windows of a base
// are shared by up to 8 lanes (Q chunks of >= 4 windows) and a slice may be a single (base, chunk) item.
MsmPlan msm_plan(uint32_t N, uint32_t P, uint32_t c, uint32_t Wt, uint32_t occ) {
  static const uint32_t rounds = [] {
    const char* e = getenv("SPP_MSM_WAVES");
    const int v = e ? atoi(e) : 4;
    return (uint32_t)(v >= 1 && v <= 16 ? v : 4);
  }();
  MsmPlan pl{};
  pl.W = msm_windows(c);
  if (Wt == 0 || Wt > pl.W) Wt = pl.W;
  pl.Wt = Wt;
  pl.R = (pl.W + Wt - 1) / Wt;
  pl.Pp = P >= 64 ? (P + 63) / 64 * 64 : P;
  pl.Q = 1;
  const uint64_t lanes_per_slice = (uint64_t)(pl.Pp ? pl.Pp : 1) * pl.R;
  while ((uint64_t)((N + 3) / 4) * pl.Q * lanes_per_slice < 65536 && pl.Q < 8 && (Wt + 2 * pl.Q - 1) / (2 * pl.Q) >= 4) pl.Q *= 2;
  pl.Wq = (Wt + pl.Q - 1) / pl.Q;
  uint64_t S, maxS;
  if (pl.Q > 1) {
    S = (65536 + lanes_per_slice - 1) / lanes_per_slice;
    maxS = (uint64_t)N * pl.Q;
  } else {
    maxS = std::max<uint64_t>((N + 3) / 4, 1);
    const double cap = 1024.0 * (occ ? occ : 1);                       // resident waves
    const double wps = (double)lanes_per_slice / 64.0;                  // waves per slice (all passes)
    // small batches (P <= 256): TWO rounds.  Round 3's first guess was the opposite -- three times as many rounds, so that the short
    // kernels of the other batches in flight find free SIMDs sooner -- but every slice ends in a 128-byte partial sum per lane that
    // the folds read again: with the radix-8 folds and four batches in flight, 128-proof audit batches measure 25.0 ms per step
    // at 12 rounds, 24.5 at 4, 23.6 at 2, 24.1 at 1 (one box, profiles/rehearsal_probe.py; SPP_MSM_WAVES_SMALL overrides)
    static const uint32_t rounds_small = [] {
      const char* e = getenv("SPP_MSM_WAVES_SMALL");
      const int v = e ? atoi(e) : 2;
      return (uint32_t)(v >= 1 && v <= 64 ? v : 2);
    }();
    const uint32_t rnd = P <= 256 ? rounds_small : rounds;
    const uint64_t S0 = std::max<uint64_t>(1, (uint64_t)(rnd * cap / wps + 0.5));
    uint64_t lo = std::max<uint64_t>(1, S0 - S0 / 4), hi = S0 + S0 / 2;
    lo = std::min(lo, maxS);
    hi = std::min(hi, maxS);
    S = lo;
    double best = -1;
    for (uint64_t s = lo; s <= hi; s++) {
      const double r = s * wps / cap, eff = r / std::ceil(r - 1e-9);
      if (eff > best + 1e-6) { best = eff; S = s; }
    }
  }
  if (maxS == 0) maxS = 1;
  if (S > maxS) S = maxS;
  if (S == 0) S = 1;
  pl.Sg = (uint32_t)S;
  return pl;
}

#endif  // SPP_MSM_TU_G2
// ----------------------------------------------------------------------------------------------------
// table construction: one lane per (base, window) row.
// Table layout: rows are grouped in blocks of 64; entry d of row `row` lives at ((row/64)*E + d)*64 + row%64, so the
// 64 lanes of a wave that build 64 consecutive rows write 64 consecutive points (coalesced 4 KiB per step), and so do
// the XYZZ temporaries.  The MSM kernels gather single entries at random d anyway, so they lose nothing.
// ----------------------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(64) k_build_table(const Affine<F>* __restrict__ bases, uint32_t N, uint32_t c, uint32_t Wn,
                                                    uint32_t step_bits, uint32_t row0, uint32_t nrows, Affine<F>* __restrict__ table,
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
  for (uint32_t k = 0; k < step_bits * j; k++) b.dbl_inplace();
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

#ifndef SPP_MSM_TU_G2
// number of table elements (points) for N bases with Wt window rows each, including the padding of the last 64-row block
size_t msm_table_elems(uint32_t N, uint32_t c, uint32_t Wt) {
  if (Wt == 0 || Wt > msm_windows(c)) Wt = msm_windows(c);
  size_t rows = (size_t)N * Wt;
  return ((rows + 63) / 64) * 64 * ((size_t)1 << (c - 1));
}

#endif  // SPP_MSM_TU_G2
template <class F>
void launch_build_table(hipStream_t st, const Affine<F>* bases, uint32_t N, uint32_t c, uint32_t Wt, uint32_t row0, uint32_t nrows,
                        Affine<F>* table, XYZZ<F>* tmp, F* tmp_pre) {
  if (nrows == 0) return;
  const uint32_t W = msm_windows(c);
  if (Wt == 0 || Wt > W) Wt = W;
  const uint32_t R = (W + Wt - 1) / Wt;   // row m of a base holds the multiples of 2^(c*R*m) * Base (pass rho takes windows rho + R*m)
  hipLaunchKernelGGL(k_build_table<F>, dim3((nrows + 63) / 64), dim3(64), 0, st, bases, N, c, Wt, c * R, row0, nrows, table, tmp, tmp_pre);
}
#ifndef SPP_MSM_TU_G2
template void launch_build_table<Fq>(hipStream_t, const Affine<Fq>*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, Affine<Fq>*,
                                     XYZZ<Fq>*, Fq*);
#endif
#ifdef SPP_MSM_TU_G2
template void launch_build_table<Fq2>(hipStream_t, const Affine<Fq2>*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, Affine<Fq2>*,
                                      XYZZ<Fq2>*, Fq2*);
#endif

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
// Digit planes.  One lane per (base, proof): the scalar is brought to canonical form ONCE, folded to its magnitude
// (scalars above (r-1)/2 become their negatives: the small signed noise of the audit witness stays small), recoded into
// W = ceil(254/c) signed c-bit digits and stored as int16 planes  dig[j][i][p]  (p fastest, Pp per row).  Digits lie in
// [-2^(c-1), 2^(c-1) - 1] -- for a folded (negated) scalar the recoding keeps +2^(c-1) and carries above it, so that the
// negated digit is -2^(c-1) -- which is what lets c = 16 fit int16.  The accumulate kernels then read 2 bytes per (window,
// base, proof), coalesced, with no recoder state in registers and no carry chain between windows: any window can be
// processed by any lane, which is what the window passes below need.
// ----------------------------------------------------------------------------------------------------
#ifndef SPP_MSM_TU_G2
__global__ void __launch_bounds__(256) k_msm_digits(const uint32_t* __restrict__ rows, const Fr* __restrict__ scalars,
                                                    int16_t* __restrict__ dig, uint32_t N, uint32_t P, uint32_t Pp, uint32_t c, uint32_t W) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t p = (uint32_t)(g % Pp), i = (uint32_t)(g / Pp);
  if (i >= N || p >= P) return;
  const Fr s = scalars[(size_t)rows[i] * P + p];
  uint32_t l[8];
  s.to_canonical(l);
  const bool neg = canonical_gt_half<FrParams>(l);
  if (neg) {
    uint32_t t[8];
    canonical_negate<FrParams>(l, t);
    SPP_UNROLL for (int k = 0; k < 8; k++) l[k] = t[k];
  }
  const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
  uint32_t carry = 0;
  int16_t* out = dig + (size_t)i * Pp + p;
  const size_t plane = (size_t)N * Pp;
#pragma unroll 1
  for (uint32_t j = 0; j < W; j++) {
    const uint32_t d = (l[0] & mask) + carry;
    SPP_UNROLL for (int k = 0; k < 7; k++) l[k] = (l[k] >> c) | (l[k + 1] << (32 - c));
    l[7] >>= c;
    const bool over = neg ? d > half : d >= half;
    int v = over ? (int)d - (int)(1u << c) : (int)d;
    carry = over ? 1u : 0u;
    if (neg) v = -v;
    out[(size_t)j * plane] = (int16_t)v;
  }
}
void launch_msm_digits(hipStream_t st, const uint32_t* rows, const Fr* scalars, int16_t* dig, uint32_t N, uint32_t P, uint32_t c) {
  if (N == 0 || P == 0) return;
  const uint32_t Pp = P >= 64 ? (P + 63) / 64 * 64 : P;
  const uint64_t lanes = (uint64_t)N * Pp;
  hipLaunchKernelGGL(k_msm_digits, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, st, rows, scalars, dig, N, P, Pp, c, msm_windows(c));
}
size_t msm_digit_elems(uint32_t N, uint32_t P, uint32_t c) {
  const size_t Pp = P >= 64 ? (size_t)(P + 63) / 64 * 64 : P;
  return (size_t)msm_windows(c) * N * Pp;
}

#endif  // SPP_MSM_TU_G2
// ----------------------------------------------------------------------------------------------------
// MSM accumulate over the digit planes.
//
// Window passes.  A base carries Wt table rows; row m holds the multiples (d+1) * 2^(c*R*m) * Base with R = ceil(W / Wt).
// Pass rho (0 <= rho < R) adds up  T[i][m][digit_{rho + R*m}]  over every base and row: sum_rho.  The MSM is
// sum_rho 2^(c*rho) * sum_rho, put together per proof by k_msm_horner (c doublings per pass, once per proof and set, not per
// base).  Wt = W (R = 1) is the classic layout, every window its own row -- small tables (8-bit windows) for the one-proof
// latency path, no Horner step.  Wt = 1 (R = W) is the throughput layout chosen under the HBM budget: ONE row of 2^(c-1)
// multiples per base, so for the same bytes the window is log2(W) bits wider than with a row per window -- 16-bit windows
// (16 additions per full-size scalar) where the classic layout affords 11-12 bits (22-24 additions).
//
// Lane g -> (t = g / Pp, p = g % Pp), t -> (pass rho = t / Sg, slice t % Sg); Pp = P rounded up to 64 so that a wave never
// straddles two slices (batches below 64 proofs keep Pp = P: there every lane is its own (slice, proof) anyway).  All
// lanes of a wave walk the same (base, window) sequence: digit loads are one coalesced 128 B row, a (base, window) in which
// every proof of the wave has a zero digit (bits, bytes, small signed noise -- most of the audit witness above window 0)
// is skipped for the whole wave.
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


// throughput layout (Wt = 1): one table row per base, slice sl takes bases sl, sl + Sg, ... (neighbouring wires have similar
// scalar sizes -- runs of bits, runs of hash states -- so a strided split gives every slice the same mix).  The digits of
// four bases are fetched ahead of their additions (2 B each, packed into one register pair).
// One wave per workgroup: a 256-lane workgroup needs FOUR free wave slots of a CU at once, and with 2 slots per SIMD and waves of
// unequal length (passes over sparse windows are shorter) a finished wave's slot waited for three more -- 1.79 resident waves per SIMD
// on average where 2 fit.  (The second launch-bound keeps the G1 walk within 256 registers = two waves per SIMD.)
template <class F>
__global__ void __launch_bounds__(MSM_WALK_BLOCK, sizeof(F) > sizeof(Fq) ? 1 : 2) k_msm_flat(const Affine<F>* __restrict__ table, const int16_t* __restrict__ dig,
                                                  XYZZ<F>* __restrict__ partial, uint32_t N, uint32_t P, uint32_t Pp, uint32_t c,
                                                  uint32_t R, uint32_t Sg) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t p = g % Pp, t = g / Pp;
  if (t >= R * Sg || p >= P) return;
  const uint32_t rho = t / Sg, sl = t % Sg;
  const uint32_t E = 1u << (c - 1);
  const int16_t* __restrict__ dg = dig + ((size_t)rho * N) * Pp + p;
  MsmAcc<F> acc;
  acc.init();
  if constexpr (sizeof(F) > sizeof(Fq) || SPP_G1_GATHER_PIPELINE) {
    // G2: one wave per SIMD (512 registers), nothing else to run while a gather is in flight -- the counters of the first
    // version showed 59 % VALU issue.  One-deep software pipeline: the entry of the next non-zero digit is requested BEFORE the
    // pending addition is computed.  A last pass over the loop body (flush) retires the pending addition, so that the ~40 KB
    // of a G2 mixed addition are instantiated once.
    int d_next = 0;
    Affine<F> e_next;
    for (uint32_t i0 = sl;; i0 += 4 * Sg) {
      const bool last = i0 >= N;
      uint64_t pack = 0;
      if (!last) {
        SPP_UNROLL for (uint32_t k = 0; k < 4; k++) {
          const uint32_t i = i0 + k * Sg;
          const uint32_t d = i < N ? (uint32_t)(uint16_t)dg[(size_t)i * Pp] : 0u;
          pack |= (uint64_t)d << (16 * k);
        }
      }
#pragma unroll 1
      for (uint32_t k = 0; k < 4; k++) {
        const int d = (int)(int16_t)(uint16_t)(pack >> (16 * k));
        const bool flush = last && k == 0;
        if (d != 0 || flush) {
          Affine<F> e;
          if (d != 0) {
            const uint32_t i = i0 + k * Sg;
            const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
            e = table[((size_t)(i >> 6) * E + (mag - 1)) * 64 + (i & 63)];
          }
          if (d_next != 0) acc.madd(e_next, d_next < 0);
          d_next = d;
          if (d != 0) e_next = e;
        }
      }
      if (last) break;
    }
  } else {
    for (uint32_t i0 = sl; i0 < N; i0 += 4 * Sg) {
      uint64_t pack = 0;
      SPP_UNROLL for (uint32_t k = 0; k < 4; k++) {
        const uint32_t i = i0 + k * Sg;
        const uint32_t d = i < N ? (uint32_t)(uint16_t)dg[(size_t)i * Pp] : 0u;
        pack |= (uint64_t)d << (16 * k);
      }
      if (pack == 0) continue;
#pragma unroll 1
      for (uint32_t k = 0; k < 4; k++) {
        const int d = (int)(int16_t)(uint16_t)(pack >> (16 * k));
        if (d != 0) {
          const uint32_t i = i0 + k * Sg;
          const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
          acc.madd(table[((size_t)(i >> 6) * E + (mag - 1)) * 64 + (i & 63)], d < 0);
        }
      }
    }
  }
  partial[(size_t)t * P + p] = acc.result();
}

// general layout (Wt rows per base; Q > 1: the rows of a base are shared by Q lanes, items = (base, chunk of Wq rows) in
// chunk-major order so that the lanes of a wave share the chunk)
template <class F>
__global__ void __launch_bounds__(256) k_msm_rows(const Affine<F>* __restrict__ table, const int16_t* __restrict__ dig,
                                                  XYZZ<F>* __restrict__ partial, uint32_t N, uint32_t P, uint32_t Pp, uint32_t c,
                                                  uint32_t Wt, uint32_t R, uint32_t W, uint32_t Sg, uint32_t Q, uint32_t Wq) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t p = g % Pp, t = g / Pp;
  if (t >= R * Sg || p >= P) return;
  const uint32_t rho = t / Sg, sl = t % Sg;
  const uint32_t E = 1u << (c - 1);
  const size_t plane = (size_t)N * Pp;
  MsmAcc<F> acc;
  acc.init();
  const uint32_t items = N * Q;
  for (uint32_t it = sl; it < items; it += Sg) {
    uint32_t i = it, q = 0;
    if (Q > 1) {
      i = it % N;
      q = it / N;
    }
    const uint32_t m0 = q * Wq, m1 = m0 + Wq < Wt ? m0 + Wq : Wt;
    const int16_t* __restrict__ dg = dig + (size_t)i * Pp + p;
#pragma unroll 1
    for (uint32_t m = m0; m < m1; m++) {
      const uint32_t j = rho + R * m;
      if (j >= W) break;
      const int d = dg[(size_t)j * plane];
      if (d != 0) {
        const uint32_t row = i * Wt + m;
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        acc.madd(table[((size_t)(row >> 6) * E + (mag - 1)) * 64 + (row & 63)], d < 0);
      }
    }
  }
  partial[(size_t)t * P + p] = acc.result();
}

// ev_start / ev_stop (optional): receive the dispatch's own start and stop timestamps (hipExtLaunchKernelGGL), i.e. the
// kernel's duration as a profiler reports it -- an event pair recorded around the launch would also count the time the
// launch waits for kernels of the other proving stream.
template <class F>
void launch_msm_accumulate(hipStream_t st, const Affine<F>* table, const int16_t* dig, XYZZ<F>* partial, uint32_t N, uint32_t P, uint32_t c,
                           const MsmPlan& pl, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (N == 0 || P == 0) {
    if (ev_start) hipEventRecord(ev_start, st);
    if (ev_stop) hipEventRecord(ev_stop, st);
    return;
  }
  const uint64_t lanes = (uint64_t)pl.R * pl.Sg * pl.Pp;
  const dim3 grid((uint32_t)((lanes + 255) / 256));
  if (pl.Wt == 1)
    hipExtLaunchKernelGGL(k_msm_flat<F>, dim3((uint32_t)((lanes + MSM_WALK_BLOCK - 1) / MSM_WALK_BLOCK)), dim3(MSM_WALK_BLOCK), 0, st, ev_start, ev_stop, 0, table, dig, partial, N, P, pl.Pp, c, pl.R, pl.Sg);
  else
    hipExtLaunchKernelGGL(k_msm_rows<F>, grid, dim3(256), 0, st, ev_start, ev_stop, 0, table, dig, partial, N, P, pl.Pp, c, pl.Wt, pl.R,
                          pl.W, pl.Sg, pl.Q, pl.Wq);
}

// Fold the Sg slice sums of every (set, pass, proof): radix 8, one launch per level, every lane busy -- lane (s, p) with
// s < next = ceil(S_cur / 8) adds partial[s + next * m][p], m = 1 .. 7, into partial[s][p].  (Pairwise levels were 8 dependent
// launches for the 192 slices of a 128-proof batch; behind a chip full of MSM waves every launch waits for free SIMDs, the
// G2 fold with its 256 registers longest: 0.7 ms per level in the trace of pipelined 128-proof batches.)  blockIdx.y = set,
// blockIdx.z = pass; up to MSM_FOLD_SETS sets share the launches (the five G1 sums of a proof are independent).  A set with
// one pass leaves the fold in out[p]; with R > 1 passes the pass sums stay in partial[rho * Sg * P + p] for k_msm_horner.
template <class T>
__device__ __forceinline__ T msm_shfl_xor(const T& v, int mask) {
  static_assert(sizeof(T) % 4 == 0, "word-sized");
  T r;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
  uint32_t* d = reinterpret_cast<uint32_t*>(&r);
  SPP_UNROLL for (uint32_t i = 0; i < sizeof(T) / 4; i++) d[i] = (uint32_t)__shfl_xor((int)s[i], mask, 64);
  return r;
}
// coop (a handful of proofs, the generateProof latency path): the 8 terms of an output sit on 8 neighbouring lanes and meet in a
// 3-step shuffle tree instead of one lane adding 7 of them in sequence -- the fold of a single proof's G2 sum is 5 levels deep:
// 35 dependent G2 additions (0.9 ms, what k_assemble waited for) become 15.
template <class F>
__global__ void __launch_bounds__(64) k_msm_fold_multi(MsmFoldSets<F> fs, uint32_t P, uint32_t coop) {
  const uint32_t set = blockIdx.y, rho = blockIdx.z;
  const uint32_t next = fs.half[set], S_cur = fs.cur[set];
  if (next == 0 || rho >= fs.R[set]) return;   // this set is already folded / has fewer passes (whole workgroups)
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ<F>* __restrict__ partial = fs.partial[set] + (size_t)rho * fs.Sg[set] * P;
  if (coop) {
    static_assert(MSM_FOLD_RADIX == 8, "three shuffle steps");
    const uint32_t o = g >> 3, m = g & 7;
    const bool live = o < next * P;
    const uint32_t s = live ? o / P : 0, p = live ? o % P : 0, t = s + next * m;
    XYZZ<F> a = XYZZ<F>::infinity();
    if (live && t < S_cur) a = partial[(size_t)t * P + p];
    SPP_UNROLL for (int d = 4; d >= 1; d >>= 1) {               // every lane of the wave reaches the shuffles
      const XYZZ<F> b = msm_shfl_xor(a, d);
      a.add(b);
    }
    if (live && m == 0) {
      if (next == 1 && fs.R[set] == 1) fs.out[set][p] = a;
      else partial[(size_t)s * P + p] = a;
    }
    return;
  }
  if (g >= next * P) return;
  const uint32_t s = g / P, p = g % P;
  XYZZ<F> a = partial[(size_t)s * P + p];
#pragma unroll 1
  for (uint32_t t = s + next; t < S_cur; t += next) a.add(partial[(size_t)t * P + p]);
  if (next == 1 && fs.R[set] == 1) fs.out[set][p] = a;   // last level of a one-pass set: the result leaves the scratch array
  else partial[(size_t)s * P + p] = a;
}
// out[p] = sum_rho 2^(c*rho) * pass_sum[rho][p]  (Horner from the top pass down; one lane per (set, proof)).  The c doublings
// between two passes run in Jacobian coordinates (a = 0: 2M + 5S, "dbl-2009-l", against 6M + 3S for an XYZZ doubling):
// (X, Y, ZZ, ZZZ) -> (X*ZZ, Y*ZZZ, Z = ZZ) and back with ZZ' = Z^2, ZZZ' = Z^3 -- two products each way per pass.  The
// chain of c * (R - 1) = 240 doublings is what a single proof waits for here (G2: 3.7 -> 2.7 ms).
template <class F>
__global__ void __launch_bounds__(64) k_msm_horner(MsmFoldSets<F> fs, uint32_t P) {
  const uint32_t set = blockIdx.y;
  const uint32_t R = fs.R[set], c = fs.c[set];
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (R <= 1 || p >= P) return;
  const size_t stride = (size_t)fs.Sg[set] * P;
  const XYZZ<F>* __restrict__ sums = fs.partial[set];
  XYZZ<F> acc = sums[(size_t)(R - 1) * stride + p];
#pragma unroll 1
  for (uint32_t rho = R - 1; rho-- > 0;) {
    if (!acc.is_inf()) {
      F X = acc.X * acc.ZZ, Y = acc.Y * acc.ZZZ, Z = acc.ZZ;
#pragma unroll 1
      for (uint32_t k = 0; k < c; k++) {
        const F A = X.sqr(), B = Y.sqr(), C = B.sqr();
        const F t = (X + B).sqr() - A - C;
        const F D = t.dbl();
        const F E = A.dbl() + A;
        const F X3 = E.sqr() - D.dbl();
        const F C8 = C.dbl().dbl().dbl();
        const F Z3 = (Y * Z).dbl();
        Y = E * (D - X3) - C8;
        X = X3;
        Z = Z3;
      }
      const F zz = Z.sqr();
      acc.X = X;
      acc.Y = Y;
      acc.ZZ = zz;
      acc.ZZZ = zz * Z;
    }
    acc.add(sums[(size_t)rho * stride + p]);
  }
  fs.out[set][p] = acc;
}
template <class F>
__global__ void __launch_bounds__(256) k_msm_fill_inf(XYZZ<F>* __restrict__ out, uint32_t P) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < P) out[g] = XYZZ<F>::infinity();
}

// fs: partial / out / Sg / R / c filled by the caller per set (Sg = 0: empty set, out = infinity)
template <class F>
void launch_msm_reduce_multi(hipStream_t st, MsmFoldSets<F> fs, uint32_t nsets, uint32_t P) {
  if (P == 0 || nsets == 0) return;
  uint32_t cur[MSM_FOLD_SETS];
  bool done[MSM_FOLD_SETS];
  uint32_t maxR = 1;
  bool horner = false;
  for (uint32_t i = 0; i < nsets; i++) {
    cur[i] = fs.Sg[i];
    done[i] = false;
    if (fs.Sg[i] == 0) {
      hipLaunchKernelGGL(k_msm_fill_inf<F>, dim3((P + 255) / 256), dim3(256), 0, st, fs.out[i], P);
      done[i] = true;
      fs.R[i] = 1;
      continue;
    }
    if (fs.R[i] > 1) horner = true;
    if (fs.R[i] > 1 && fs.Sg[i] == 1) done[i] = true;   // nothing to fold: the pass sums are already in place
    maxR = std::max(maxR, fs.R[i]);
  }
  for (;;) {
    uint64_t lanes = 0;
    for (uint32_t i = 0; i < nsets; i++) {
      fs.half[i] = done[i] ? 0 : (cur[i] + MSM_FOLD_RADIX - 1) / MSM_FOLD_RADIX;
      fs.cur[i] = cur[i];
      lanes = std::max<uint64_t>(lanes, (uint64_t)fs.half[i] * P);
    }
    if (lanes == 0) break;
    const uint32_t coop = P <= MSM_FOLD_COOP_MAX_BATCH ? 1u : 0u;
    if (coop) lanes *= MSM_FOLD_RADIX;
    hipLaunchKernelGGL(k_msm_fold_multi<F>, dim3((uint32_t)((lanes + 63) / 64), nsets, maxR), dim3(64), 0, st, fs, P, coop);
    for (uint32_t i = 0; i < nsets; i++)
      if (!done[i]) {
        cur[i] = fs.half[i];
        if (cur[i] == 1) done[i] = true;
      }
  }
  if (horner) {
    for (uint32_t i = 0; i < nsets; i++)
      if (fs.Sg[i] == 0) fs.R[i] = 1;
    hipLaunchKernelGGL(k_msm_horner<F>, dim3((P + 63) / 64, nsets), dim3(64), 0, st, fs, P);
  }
}
template <class F>
void launch_msm_reduce(hipStream_t st, XYZZ<F>* partial, XYZZ<F>* out, uint32_t P, const MsmPlan& pl, uint32_t c, bool empty) {
  MsmFoldSets<F> fs{};
  fs.partial[0] = partial;
  fs.out[0] = out;
  fs.Sg[0] = empty ? 0 : pl.Sg;
  fs.R[0] = pl.R;
  fs.c[0] = c;
  launch_msm_reduce_multi<F>(st, fs, 1, P);
}
#ifndef SPP_MSM_TU_G2
template void launch_msm_reduce_multi<Fq>(hipStream_t, MsmFoldSets<Fq>, uint32_t, uint32_t);
#endif
#ifdef SPP_MSM_TU_G2
template void launch_msm_reduce_multi<Fq2>(hipStream_t, MsmFoldSets<Fq2>, uint32_t, uint32_t);
#endif
#ifndef SPP_MSM_TU_G2
template void launch_msm_accumulate<Fq>(hipStream_t, const Affine<Fq>*, const int16_t*, XYZZ<Fq>*, uint32_t, uint32_t, uint32_t, const MsmPlan&,
                                        hipEvent_t, hipEvent_t);
#endif
#ifdef SPP_MSM_TU_G2
template void launch_msm_accumulate<Fq2>(hipStream_t, const Affine<Fq2>*, const int16_t*, XYZZ<Fq2>*, uint32_t, uint32_t, uint32_t,
                                         const MsmPlan&, hipEvent_t, hipEvent_t);
#endif
#ifndef SPP_MSM_TU_G2
template void launch_msm_reduce<Fq>(hipStream_t, XYZZ<Fq>*, XYZZ<Fq>*, uint32_t, const MsmPlan&, uint32_t, bool);
#endif
#ifdef SPP_MSM_TU_G2
template void launch_msm_reduce<Fq2>(hipStream_t, XYZZ<Fq2>*, XYZZ<Fq2>*, uint32_t, const MsmPlan&, uint32_t, bool);
#endif

// ----------------------------------------------------------------------------------------------------
// The H bases in the evaluation basis (load time).  gnark's computeH ends with an inverse coset transform that turns the
// values of h on the coset g*H into coefficients, because pk.G1.Z is a coefficient basis (Z_j = [tau^j t(tau) / delta]).
// sum_j h_j Z_j = sum_i h(g w^i) Z'_i  with  Z'_i = sum_j (g^-j / n) w^(-ij) Z_j : a DFT "in the exponent" of the n - 1 points
// (scaled, padded with the point at infinity), done ONCE when the circuit is loaded -- n + (n/2) log2 n scalar multiplications,
// ~60 ms for n = 2^15 -- and every proof saves its seventh transform (the scalars of the Z walk are the values the pointwise
// kernel leaves, natural order).  The group element is the same, so are the proof bytes.
// ----------------------------------------------------------------------------------------------------
#ifndef SPP_MSM_TU_G2
__device__ __forceinline__ XYZZ<Fq> g1_scalar_mul(const XYZZ<Fq>& pt, const Fr& k) {
  uint32_t c[8];
  k.to_canonical(c);
  XYZZ<Fq> acc = XYZZ<Fq>::infinity();
  if (pt.is_inf()) return acc;
#pragma unroll 1
  for (int w = 7; w >= 0; w--) {
    const uint32_t word = w == 0 ? c[0] : w == 1 ? c[1] : w == 2 ? c[2] : w == 3 ? c[3] : w == 4 ? c[4] : w == 5 ? c[5] : w == 6 ? c[6] : c[7];
#pragma unroll 1
    for (int bit = 31; bit >= 0; bit--) {
      acc.dbl_inplace();
      if ((word >> bit) & 1) acc.add(pt);
    }
  }
  return acc;
}
// x[j] = scale[j] * affine[j] (j < n_pts), infinity above
__global__ void __launch_bounds__(64) k_g1_dft_load(const G1Affine* __restrict__ pts, uint32_t n_pts, const Fr* __restrict__ scale,
                                                    G1XYZZ* __restrict__ x, uint32_t n) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  x[j] = j < n_pts ? g1_scalar_mul(G1XYZZ::from_affine(pts[j]), scale[j]) : G1XYZZ::infinity();
}
// one decimation-in-frequency stage (natural in, bit-reversed out after log2 n stages): block length len, twiddles tw[k] = w^k
__global__ void __launch_bounds__(64) k_g1_dft_stage(G1XYZZ* __restrict__ x, uint32_t n, uint32_t len, const Fr* __restrict__ tw) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n / 2) return;
  const uint32_t half = len / 2, j = g % half, b = (g / half) * len;
  G1XYZZ u = x[b + j], v = x[b + j + half];
  G1XYZZ d = u;
  d.add(v.neg());
  u.add(v);
  x[b + j] = u;
  const uint32_t e = j * (n / len);
  x[b + j + half] = e ? g1_scalar_mul(d, tw[e]) : d;
}
__global__ void __launch_bounds__(64) k_g1_to_affine(const G1XYZZ* __restrict__ x, G1Affine* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[i].to_affine();
}
// term t: coeff[t] * base[row[t]]; then segment s = sum of terms [seg[s], seg[s+1]) as an affine point (the column sums
// X_wire = sum_i C[i][wire] * W_i of the product form of computeH, spp_api.cpp)
__global__ void __launch_bounds__(64) k_g1_terms(const G1Affine* __restrict__ base, const uint32_t* __restrict__ row, const Fr* __restrict__ coeff,
                                                 uint32_t nterms, G1XYZZ* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nterms) return;
  const G1XYZZ p = G1XYZZ::from_affine(base[row[t]]);
  const Fr c = coeff[t];
  if (c == Fr::one()) out[t] = p;
  else if (c == Fr::one().neg()) out[t] = p.neg();
  else out[t] = g1_scalar_mul(p, c);
}
__global__ void __launch_bounds__(64) k_g1_segsum(const G1XYZZ* __restrict__ terms, const uint32_t* __restrict__ seg, uint32_t nseg,
                                                  G1Affine* __restrict__ out) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  G1XYZZ acc = G1XYZZ::infinity();
  for (uint32_t t = seg[s]; t < seg[s + 1]; t++) acc.add(terms[t]);
  out[s] = acc.to_affine();
}
void launch_g1_column_sums(hipStream_t st, const G1Affine* base, const uint32_t* row, const Fr* coeff, uint32_t nterms, const uint32_t* seg,
                           uint32_t nseg, G1XYZZ* work, G1Affine* out) {
  if (nterms) hipLaunchKernelGGL(k_g1_terms, dim3((nterms + 63) / 64), dim3(64), 0, st, base, row, coeff, nterms, work);
  if (nseg) hipLaunchKernelGGL(k_g1_segsum, dim3((nseg + 63) / 64), dim3(64), 0, st, work, seg, nseg, out);
}
// out[bitrev(i)] = Z'_i for i < n = 2^logn (the caller undoes the bit reversal); scale[j] = g^-j / n, tw_inv[k] = w^-k (k < n/2)
void launch_g1_eval_basis(hipStream_t st, const G1Affine* pts, uint32_t n_pts, uint32_t logn, const Fr* scale, const Fr* tw_inv, G1XYZZ* work,
                          G1Affine* out) {
  const uint32_t n = 1u << logn;
  hipLaunchKernelGGL(k_g1_dft_load, dim3((n + 63) / 64), dim3(64), 0, st, pts, n_pts, scale, work, n);
  for (uint32_t len = n; len >= 2; len >>= 1)
    hipLaunchKernelGGL(k_g1_dft_stage, dim3((n / 2 + 63) / 64), dim3(64), 0, st, work, n, len, tw_inv);
  hipLaunchKernelGGL(k_g1_to_affine, dim3((n + 63) / 64), dim3(64), 0, st, work, out, n);
}

#endif  // SPP_MSM_TU_G2
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
#ifndef SPP_MSM_TU_G2
template void launch_fixed_base_mul<Fq>(hipStream_t, const Affine<Fq>*, uint32_t, const Fr*, uint32_t, Affine<Fq>*, XYZZ<Fq>*);
#endif
#ifdef SPP_MSM_TU_G2
template void launch_fixed_base_mul<Fq2>(hipStream_t, const Affine<Fq2>*, uint32_t, const Fr*, uint32_t, Affine<Fq2>*, XYZZ<Fq2>*);
#endif

}  // namespace spp

// General-base G1 multi-scalar multiplication (Pippenger bucket method) for large N -- BASELINE.json configs[4]
// (2^24 points) and any MSM whose bases are not a resident proving key.  SURVEY 8a a6: "digit extract -> bucket
// sort -> bucket accumulate -> bucket reduce -> window combine".
//
//   0. k_pip_digits   lane per point: canonical scalar -> 16 signed 16-bit digits (fold sign applied), stored window-major
//                     as int16 [16][n] -- the later passes read 2 B per point and window instead of a 32 B scalar
//   1. k_pip_count    one 1024-lane workgroup per (window, tile of 2^20 points): histogram over the 2^15 buckets in LDS
//                     (128 KB of the CU's 160 KB; ds_add_u32), written out once per tile -- no global atomics
//   2. k_pip_totals   lane per (window, bucket): bucket totals + exclusive prefix over the tiles;  k_pip_scan: exclusive
//                     prefix over the buckets of a window                                                       (LDS scan)
//   3. k_pip_scatter  same workgroups as 1: LDS cursors start at offs[bucket] + prefix[tile][bucket]; a point's slot comes
//                     from an LDS atomic, (index | sign) goes to its bucket's segment -- again no global atomics
//   4. k_pip_segments lane per 256-entry segment of a window's sorted list: gathers its points (64 B each) and folds them
//                     with mixed additions per bucket; k_pip_fixup joins the buckets that span segments
//   5. k_pip_chunks   lane per 64-bucket chunk: running sums  S = sum B_b,  T = sum (b_local+1) B_b
//   6. k_pip_windows  64-lane block per window:  sum_chunks (T_c + 64 c S_c)  with an LDS tree
//   host: Horner over the 16 window sums.
// HBM traffic is dominated by step 4: every base is gathered once per window (64 B x N x 16), on top of the
// algorithmic 96 B x N; lanes own equal-size segments of the sorted lists, so the load is balanced whatever the scalar
// distribution (uniform: N / 2^15 points per bucket; witness-like: most points in a few hundred buckets of window 0).
#include "kernels.hpp"
#include "f29.hpp"

namespace spp {

static constexpr uint32_t PIP_C = 16, PIP_W = 16, PIP_B = 1u << (PIP_C - 1);   // 2^15 buckets per window
static constexpr uint32_t PIP_CHUNK = 64, PIP_NCHUNK = PIP_B / PIP_CHUNK;

// Signed 16-bit window digits of a canonical scalar, fold sign applied: point i contributes dig[j] * 2^(16 j) * P_i with
// dig[j] in [-2^15, 2^15 - 1] (int16).  The scalar is folded to |s| <= (r-1)/2 and recoded from the bottom; a window value of
// exactly 2^15 may stay (+2^15) or carry (-2^15): it carries when the fold sign is + and stays when it is - , so that the
// stored digit (fold sign applied) is -2^15 in both cases and always fits.  Bucket = |digit| - 1 in [0, 2^15).
static constexpr uint32_t PIP_TILE_LOG = 20, PIP_TILE = 1u << PIP_TILE_LOG;   // points per counting / scattering workgroup
__global__ void __launch_bounds__(256) k_pip_digits(const Fr* __restrict__ scalars, uint32_t n, int16_t* __restrict__ digits) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t cl[8], l[8];
  scalars[i].to_canonical(cl);
  const bool neg = canonical_gt_half<FrParams>(cl);
  if (neg) canonical_negate<FrParams>(cl, l);
  else { SPP_UNROLL for (int k = 0; k < 8; k++) l[k] = cl[k]; }
  const int32_t keep_max = neg ? (int32_t)PIP_B : (int32_t)PIP_B - 1;         // largest window value that does not carry
  uint32_t carry = 0;
  SPP_UNROLL for (int j = 0; j < (int)PIP_W; j++) {
    const uint32_t word = (j & 1) ? (l[j >> 1] >> 16) : (l[j >> 1] & 0xffffu);
    int32_t d = (int32_t)(word + carry);
    if (d > keep_max) { d -= 65536; carry = 1; } else carry = 0;              // the top window never carries: |s| < 2^253
    digits[(size_t)j * n + i] = (int16_t)(neg ? -d : d);
  }
}
__device__ __forceinline__ void pip_decode(int16_t v, uint32_t& bucket, uint32_t& sgn) {
  const int32_t d = v;
  sgn = d < 0 ? 0x80000000u : 0u;
  bucket = (uint32_t)(d < 0 ? -d : d) - 1;
}

__global__ void __launch_bounds__(1024) k_pip_count(const int16_t* __restrict__ digits, uint32_t n, uint32_t ntiles,
                                                    uint32_t* __restrict__ hist_tile) {
  __shared__ uint32_t lh[PIP_B];                              // 128 KB
  const uint32_t j = blockIdx.x / ntiles, tile = blockIdx.x % ntiles, t = threadIdx.x;
  for (uint32_t b = t; b < PIP_B; b += 1024) lh[b] = 0;
  __syncthreads();
  const int16_t* dg = digits + (size_t)j * n;
  const uint32_t lo = tile << PIP_TILE_LOG, hi = min(n, lo + PIP_TILE);
  for (uint32_t i = lo + t; i < hi; i += 1024) {
    const int16_t v = dg[i];
    if (v != 0) {
      uint32_t b, sg;
      pip_decode(v, b, sg);
      atomicAdd(&lh[b], 1u);
    }
  }
  __syncthreads();
  uint32_t* out = hist_tile + ((size_t)j * ntiles + tile) * PIP_B;
  for (uint32_t b = t; b < PIP_B; b += 1024) out[b] = lh[b];
}
// lane per (window, bucket): hist = total over the tiles; hist_tile[tile] becomes the exclusive prefix over the tiles
__global__ void __launch_bounds__(256) k_pip_totals(uint32_t ntiles, uint32_t* __restrict__ hist_tile, uint32_t* __restrict__ hist) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= PIP_W * PIP_B) return;
  const uint32_t j = g / PIP_B, b = g % PIP_B;
  uint32_t run = 0;
  for (uint32_t tile = 0; tile < ntiles; tile++) {
    uint32_t* p = hist_tile + ((size_t)j * ntiles + tile) * PIP_B + b;
    const uint32_t v = *p;
    *p = run;
    run += v;
  }
  hist[g] = run;
}

// exclusive scan of one window's 2^15 counts (1024 lanes x 32 counts each)
__global__ void __launch_bounds__(1024) k_pip_scan(const uint32_t* __restrict__ hist, uint32_t* __restrict__ offs) {
  __shared__ uint32_t part[1024];
  const uint32_t j = blockIdx.x, t = threadIdx.x;
  const uint32_t* h = hist + j * PIP_B + t * 32;
  uint32_t local[32], sum = 0;
  SPP_UNROLL for (int k = 0; k < 32; k++) { local[k] = sum; sum += h[k]; }
  part[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  const uint32_t base = t ? part[t - 1] : 0;
  uint32_t* o = offs + j * PIP_B + t * 32;
  SPP_UNROLL for (int k = 0; k < 32; k++) o[k] = base + local[k];
}

__global__ void __launch_bounds__(1024) k_pip_scatter(const int16_t* __restrict__ digits, uint32_t n, uint32_t ntiles,
                                                      const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist_tile,
                                                      uint32_t* __restrict__ sorted) {
  __shared__ uint32_t cur[PIP_B];                             // 128 KB: next free slot of every bucket for THIS tile
  const uint32_t j = blockIdx.x / ntiles, tile = blockIdx.x % ntiles, t = threadIdx.x;
  const uint32_t* pre = hist_tile + ((size_t)j * ntiles + tile) * PIP_B;
  for (uint32_t b = t; b < PIP_B; b += 1024) cur[b] = offs[j * PIP_B + b] + pre[b];
  __syncthreads();
  const int16_t* dg = digits + (size_t)j * n;
  uint32_t* out = sorted + (size_t)j * n;
  const uint32_t lo = tile << PIP_TILE_LOG, hi = min(n, lo + PIP_TILE);
  for (uint32_t i = lo + t; i < hi; i += 1024) {
    const int16_t v = dg[i];
    if (v != 0) {
      uint32_t b, sg;
      pip_decode(v, b, sg);
      out[atomicAdd(&cur[b], 1u)] = i | sg;
    }
  }
}

// Bucket accumulation, robust to skewed digit distributions (real witnesses are mostly small: one window then holds
// tens of thousands of points in a few hundred buckets).  Lanes own fixed-size SEGMENTS of a window's sorted point list
// instead of buckets: lane (j, s) folds entries [s*SEG, (s+1)*SEG) with mixed additions, restarting its accumulator at
// every bucket boundary.  A bucket that lies inside one segment is written directly; a bucket that spans segments
// gets one partial per segment it touches -- `tail[s]` from the segment where it starts, `head[s]` from every later
// one -- and k_pip_fixup adds them up (a handful of additions for uniform scalars, cnt/SEG for a heavy bucket).
static constexpr uint32_t PIP_SEG = 256;
__device__ __forceinline__ uint32_t pip_bucket_of(const uint32_t* __restrict__ offs, uint32_t pos) {
  // largest b with offs[b] <= pos (offs is non-decreasing; empty buckets repeat the same offset)
  uint32_t lo = 0, hi = PIP_B - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (offs[mid] <= pos) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// bucket accumulator per coordinate field: the unsaturated 9x29-bit forms of f29.hpp (G1: XYZZ29, G2: XYZZ29G2)
template <class F> struct PipAcc;
template <> struct PipAcc<Fq> { using type = XYZZ29<FqParams>; };
template <> struct PipAcc<Fq2> { using type = XYZZ29G2; };

template <class F>
__global__ void __launch_bounds__(256) k_pip_segments(const Affine<F>* __restrict__ bases, uint32_t n, uint32_t nseg,
                                                      const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist,
                                                      const uint32_t* __restrict__ sorted, XYZZ<F>* __restrict__ buckets,
                                                      XYZZ<F>* __restrict__ head, XYZZ<F>* __restrict__ tail) {
  using Acc = typename PipAcc<F>::type;
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= PIP_W * nseg) return;
  const uint32_t j = g / nseg, sg = g % nseg;
  const uint32_t* o = offs + j * PIP_B;
  const uint32_t* h = hist + j * PIP_B;
  const uint32_t total = o[PIP_B - 1] + h[PIP_B - 1];     // points with a non-zero digit in this window
  uint32_t pos = sg * PIP_SEG;
  if (pos >= total) return;
  const uint32_t end = min(pos + PIP_SEG, total);
  const uint32_t* seg = sorted + (size_t)j * n;
  // one flat loop of exactly (end - pos) additions per lane: the 64 lanes of a wave stay in lockstep whatever their
  // bucket boundaries are; only the (rare) flush at a boundary diverges
  uint32_t b = pip_bucket_of(o, pos);                      // non-empty: pos lies in [o[b], o[b] + h[b])
  uint32_t bend = o[b] + h[b];
  bool from_before = pos > o[b];
  Acc acc = Acc::infinity();     // unsaturated 9x29-bit accumulator (f29.hpp), as in k_msm_fixed
  auto flush = [&](bool continues_after) {
    const XYZZ<F> r = acc.to_xyzz();
    if (from_before) head[g] = r;
    else if (continues_after) tail[g] = r;
    else buckets[(size_t)j * PIP_B + b] = r;
  };
  // software pipeline: the index and the (randomly placed, 64 B) base of entry k + 1 are requested before the ~2.3 K
  // instructions of the addition of entry k, so the HBM round trip of the gather hides behind arithmetic of the same lane
  uint32_t e = seg[pos];
  Affine<F> p = bases[e & 0x7fffffffu];
  for (uint32_t k = pos; k < end; k++) {
    const uint32_t e_cur = e;
    const Affine<F> p_cur = p;
    if (k + 1 < end) {
      e = seg[k + 1];
      p = bases[e & 0x7fffffffu];
    }
    if (k == bend) {                                       // next non-empty bucket starts here
      flush(false);
      acc = Acc::infinity();
      from_before = false;
      b++;
      while (h[b] == 0) b++;
      bend = o[b] + h[b];
    }
    if (p_cur.is_inf()) continue;
    acc.madd(p_cur, (e_cur & 0x80000000u) != 0);
  }
  flush(bend > end);
}
// lane per (window, bucket): empty buckets -> infinity; buckets spanning several segments -> tail[s0] + head[s0+1..s1]
template <class F>
__global__ void __launch_bounds__(256) k_pip_fixup(uint32_t nseg, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist,
                                                   XYZZ<F>* __restrict__ buckets, const XYZZ<F>* __restrict__ head,
                                                   const XYZZ<F>* __restrict__ tail) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= PIP_W * PIP_B) return;
  const uint32_t j = g / PIP_B, cnt = hist[g];
  if (cnt == 0) {
    buckets[g] = XYZZ<F>::infinity();
    return;
  }
  const uint32_t s0 = offs[g] / PIP_SEG, s1 = (offs[g] + cnt - 1) / PIP_SEG;
  if (s0 == s1) return;                                    // written directly by its segment lane
  XYZZ<F> acc = tail[(size_t)j * nseg + s0];
  for (uint32_t s = s0 + 1; s <= s1; s++) acc.add(head[(size_t)j * nseg + s]);
  buckets[g] = acc;
}

// chunk c of window j: S = sum_b B_b, T = sum_b (b_local + 1) B_b  (running-sum trick from the top bucket down)
template <class F>
__global__ void __launch_bounds__(64) k_pip_chunks(const XYZZ<F>* __restrict__ buckets, XYZZ<F>* __restrict__ S, XYZZ<F>* __restrict__ T) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= PIP_W * PIP_NCHUNK) return;
  const XYZZ<F>* b = buckets + (size_t)g * PIP_CHUNK;
  XYZZ<F> run = XYZZ<F>::infinity(), tot = XYZZ<F>::infinity();
  for (int k = PIP_CHUNK - 1; k >= 0; k--) {
    run.add(b[k]);
    tot.add(run);
  }
  S[g] = run;
  T[g] = tot;
}

// window sum = sum_c ( T_c + (64 c) * S_c ); one 64-lane block per window, 8 chunks per lane
template <class F>
__global__ void __launch_bounds__(64) k_pip_windows(const XYZZ<F>* __restrict__ S, const XYZZ<F>* __restrict__ T, XYZZ<F>* __restrict__ out) {
  __shared__ XYZZ<F> sh[64];
  const uint32_t j = blockIdx.x, t = threadIdx.x;
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (uint32_t c = t; c < PIP_NCHUNK; c += 64) {
    acc.add(T[j * PIP_NCHUNK + c]);
    // (64 c) * S_c by double-and-add on the 15-bit multiplier
    XYZZ<F> s = S[j * PIP_NCHUNK + c], m = XYZZ<F>::infinity();
    uint32_t k = PIP_CHUNK * c;
    for (int bit = 14; bit >= 0; bit--) {
      m.dbl_inplace();
      if ((k >> bit) & 1) m.add(s);
    }
    acc.add(m);
  }
  sh[t] = acc;
  __syncthreads();
  for (uint32_t w = 32; w > 0; w >>= 1) {
    if (t < w) { XYZZ<F> a = sh[t]; a.add(sh[t + w]); sh[t] = a; }
    __syncthreads();
  }
  if (t == 0) out[j] = sh[0];
}

// workspace layout (u32 words unless noted): hist[W*B] | offs[W*B] | hist_tile[W*ntiles*B] | sorted[W*n] | digits (int16 [W][n],
// padded to whole words) ; then XYZZ: buckets[W*B] | S | T | out[W] | head[W*nseg] | tail[W*nseg]
static uint32_t pip_nseg(uint32_t n) { return (n + PIP_SEG - 1) / PIP_SEG + 1; }
static uint32_t pip_ntiles(uint32_t n) { return n ? (n + PIP_TILE - 1) / PIP_TILE : 1; }
static size_t pip_words(uint32_t n) {
  return (size_t)2 * PIP_W * PIP_B + (size_t)PIP_W * pip_ntiles(n) * PIP_B + (size_t)PIP_W * n + ((size_t)PIP_W * n + 1) / 2;
}
template <class F>
static size_t pip_ws_bytes(uint32_t n) {
  size_t pts = (size_t)PIP_W * PIP_B + 2 * (size_t)PIP_W * PIP_NCHUNK + PIP_W + 2 * (size_t)PIP_W * pip_nseg(n);
  return ((pip_words(n) * 4 + 255) / 256) * 256 + pts * sizeof(XYZZ<F>);
}
size_t pippenger_workspace_bytes(uint32_t n) { return pip_ws_bytes<Fq>(n); }
size_t pippenger_workspace_bytes_g2(uint32_t n) { return pip_ws_bytes<Fq2>(n); }
uint32_t pippenger_windows() { return PIP_W; }

// window sums land in out_windows[16] (device); ev0/ev1 (optional) bracket the bucket-accumulation kernel
template <class F>
static void launch_pippenger(hipStream_t st, const Affine<F>* bases, const Fr* scalars, uint32_t n, void* workspace, XYZZ<F>** out_windows,
                             hipEvent_t ev0, hipEvent_t ev1) {
  const uint32_t ntiles = pip_ntiles(n);
  uint32_t* hist = (uint32_t*)workspace;
  uint32_t* offs = hist + PIP_W * PIP_B;
  uint32_t* hist_tile = offs + PIP_W * PIP_B;
  uint32_t* sorted = hist_tile + (size_t)PIP_W * ntiles * PIP_B;
  int16_t* digits = (int16_t*)(sorted + (size_t)PIP_W * n);
  XYZZ<F>* buckets = (XYZZ<F>*)((char*)workspace + ((pip_words(n) * 4 + 255) / 256) * 256);
  XYZZ<F>* S = buckets + (size_t)PIP_W * PIP_B;
  XYZZ<F>* T = S + (size_t)PIP_W * PIP_NCHUNK;
  XYZZ<F>* out = T + (size_t)PIP_W * PIP_NCHUNK;
  const uint32_t nseg = pip_nseg(n);
  XYZZ<F>* head = out + PIP_W;
  XYZZ<F>* tail = head + (size_t)PIP_W * nseg;
  if (n) hipLaunchKernelGGL(k_pip_digits, dim3((n + 255) / 256), dim3(256), 0, st, scalars, n, digits);
  hipLaunchKernelGGL(k_pip_count, dim3(PIP_W * ntiles), dim3(1024), 0, st, digits, n, ntiles, hist_tile);
  hipLaunchKernelGGL(k_pip_totals, dim3(PIP_W * PIP_B / 256), dim3(256), 0, st, ntiles, hist_tile, hist);
  hipLaunchKernelGGL(k_pip_scan, dim3(PIP_W), dim3(1024), 0, st, hist, offs);
  hipLaunchKernelGGL(k_pip_scatter, dim3(PIP_W * ntiles), dim3(1024), 0, st, digits, n, ntiles, offs, hist_tile, sorted);
  if (ev0) hipEventRecord(ev0, st);
  hipLaunchKernelGGL(k_pip_segments<F>, dim3((PIP_W * nseg + 255) / 256), dim3(256), 0, st, bases, n, nseg, offs, hist, sorted, buckets,
                     head, tail);
  if (ev1) hipEventRecord(ev1, st);
  hipLaunchKernelGGL(k_pip_fixup<F>, dim3(PIP_W * PIP_B / 256), dim3(256), 0, st, nseg, offs, hist, buckets, head, tail);
  hipLaunchKernelGGL(k_pip_chunks<F>, dim3(PIP_W * PIP_NCHUNK / 64), dim3(64), 0, st, buckets, S, T);
  hipLaunchKernelGGL(k_pip_windows<F>, dim3(PIP_W), dim3(64), 0, st, S, T, out);
  *out_windows = out;
}
void launch_pippenger_g1(hipStream_t st, const G1Affine* bases, const Fr* scalars, uint32_t n, void* workspace, G1XYZZ** out_windows,
                         hipEvent_t ev0, hipEvent_t ev1) {
  launch_pippenger<Fq>(st, bases, scalars, n, workspace, out_windows, ev0, ev1);
}
// the same over G2 (the variant VERDICT r1 item 8 asked for): only the point kernels differ
void launch_pippenger_g2(hipStream_t st, const G2Affine* bases, const Fr* scalars, uint32_t n, void* workspace, G2XYZZ** out_windows,
                         hipEvent_t ev0, hipEvent_t ev1) {
  launch_pippenger<Fq2>(st, bases, scalars, n, workspace, out_windows, ev0, ev1);
}

}  // namespace spp

// General-base G1 multi-scalar multiplication (Pippenger bucket method) for large N -- BASELINE.json configs[4]
// (2^24 points) and any MSM whose bases are not a resident proving key.  SURVEY 8a a6: "digit extract -> bucket
// sort -> bucket accumulate -> bucket reduce -> window combine".
//
//   1. k_pip_count    lane per point: 16 signed 16-bit digits -> histogram[window][bucket]            (u32 atomics)
//   2. k_pip_scan     block per window: exclusive prefix sum of the 2^15 bucket counts                 (LDS scan)
//   3. k_pip_scatter  lane per point: (index | sign) into its bucket's segment                        (u32 atomics)
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

// signed 16-bit window digits of a canonical scalar folded to |s| <= (r-1)/2; returns the fold sign
__device__ __forceinline__ bool pip_digits(const Fr& s, int32_t (&dig)[PIP_W]) {
  uint32_t cl[8], l[8];
  s.to_canonical(cl);
  const bool neg = canonical_gt_half<FrParams>(cl);
  if (neg) canonical_negate<FrParams>(cl, l);
  else { SPP_UNROLL for (int k = 0; k < 8; k++) l[k] = cl[k]; }
  uint32_t carry = 0;
  SPP_UNROLL for (int j = 0; j < (int)PIP_W; j++) {
    const uint32_t word = (j & 1) ? (l[j >> 1] >> 16) : (l[j >> 1] & 0xffffu);
    uint32_t d = word + carry;
    if (d > PIP_B) { dig[j] = (int32_t)d - 65536; carry = 1; } else { dig[j] = (int32_t)d; carry = 0; }
  }
  return neg;
}

__global__ void __launch_bounds__(256) k_pip_count(const Fr* __restrict__ scalars, uint32_t n, uint32_t* __restrict__ hist) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t dig[PIP_W];
  pip_digits(scalars[i], dig);
  SPP_UNROLL for (int j = 0; j < (int)PIP_W; j++) {
    const int32_t d = dig[j];
    if (d != 0) atomicAdd(&hist[j * PIP_B + (uint32_t)((d < 0 ? -d : d) - 1)], 1u);
  }
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

__global__ void __launch_bounds__(256) k_pip_scatter(const Fr* __restrict__ scalars, uint32_t n, const uint32_t* __restrict__ offs,
                                                     uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t dig[PIP_W];
  const bool neg = pip_digits(scalars[i], dig);
  SPP_UNROLL for (int j = 0; j < (int)PIP_W; j++) {
    const int32_t d = dig[j];
    if (d == 0) continue;
    const uint32_t b = (uint32_t)((d < 0 ? -d : d) - 1);
    const uint32_t pos = atomicAdd(&cursor[j * PIP_B + b], 1u);
    const uint32_t sgn = ((d < 0) != neg) ? 0x80000000u : 0u;
    sorted[(size_t)j * n + offs[j * PIP_B + b] + pos] = i | sgn;
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
__global__ void __launch_bounds__(256) k_pip_segments(const G1Affine* __restrict__ bases, uint32_t n, uint32_t nseg,
                                                      const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist,
                                                      const uint32_t* __restrict__ sorted, G1XYZZ* __restrict__ buckets,
                                                      G1XYZZ* __restrict__ head, G1XYZZ* __restrict__ tail) {
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
  XYZZ29<FqParams> acc = XYZZ29<FqParams>::infinity();     // unsaturated 9x29-bit accumulator (f29.hpp), as in k_msm_fixed
  auto flush = [&](bool continues_after) {
    const G1XYZZ r = acc.to_xyzz();
    if (from_before) head[g] = r;
    else if (continues_after) tail[g] = r;
    else buckets[(size_t)j * PIP_B + b] = r;
  };
  for (uint32_t k = pos; k < end; k++) {
    if (k == bend) {                                       // next non-empty bucket starts here
      flush(false);
      acc = XYZZ29<FqParams>::infinity();
      from_before = false;
      b++;
      while (h[b] == 0) b++;
      bend = o[b] + h[b];
    }
    const uint32_t e = seg[k];
    const G1Affine p = bases[e & 0x7fffffffu];
    if (p.is_inf()) continue;
    acc.madd(p, (e & 0x80000000u) != 0);
  }
  flush(bend > end);
}
// lane per (window, bucket): empty buckets -> infinity; buckets spanning several segments -> tail[s0] + head[s0+1..s1]
__global__ void __launch_bounds__(256) k_pip_fixup(uint32_t nseg, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist,
                                                   G1XYZZ* __restrict__ buckets, const G1XYZZ* __restrict__ head,
                                                   const G1XYZZ* __restrict__ tail) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= PIP_W * PIP_B) return;
  const uint32_t j = g / PIP_B, cnt = hist[g];
  if (cnt == 0) {
    buckets[g] = G1XYZZ::infinity();
    return;
  }
  const uint32_t s0 = offs[g] / PIP_SEG, s1 = (offs[g] + cnt - 1) / PIP_SEG;
  if (s0 == s1) return;                                    // written directly by its segment lane
  G1XYZZ acc = tail[(size_t)j * nseg + s0];
  for (uint32_t s = s0 + 1; s <= s1; s++) acc.add(head[(size_t)j * nseg + s]);
  buckets[g] = acc;
}

// chunk c of window j: S = sum_b B_b, T = sum_b (b_local + 1) B_b  (running-sum trick from the top bucket down)
__global__ void __launch_bounds__(64) k_pip_chunks(const G1XYZZ* __restrict__ buckets, G1XYZZ* __restrict__ S, G1XYZZ* __restrict__ T) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= PIP_W * PIP_NCHUNK) return;
  const G1XYZZ* b = buckets + (size_t)g * PIP_CHUNK;
  G1XYZZ run = G1XYZZ::infinity(), tot = G1XYZZ::infinity();
  for (int k = PIP_CHUNK - 1; k >= 0; k--) {
    run.add(b[k]);
    tot.add(run);
  }
  S[g] = run;
  T[g] = tot;
}

// window sum = sum_c ( T_c + (64 c) * S_c ); one 64-lane block per window, 8 chunks per lane
__global__ void __launch_bounds__(64) k_pip_windows(const G1XYZZ* __restrict__ S, const G1XYZZ* __restrict__ T, G1XYZZ* __restrict__ out) {
  __shared__ G1XYZZ sh[64];
  const uint32_t j = blockIdx.x, t = threadIdx.x;
  G1XYZZ acc = G1XYZZ::infinity();
  for (uint32_t c = t; c < PIP_NCHUNK; c += 64) {
    acc.add(T[j * PIP_NCHUNK + c]);
    // (64 c) * S_c by double-and-add on the 15-bit multiplier
    G1XYZZ s = S[j * PIP_NCHUNK + c], m = G1XYZZ::infinity();
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
    if (t < w) { G1XYZZ a = sh[t]; a.add(sh[t + w]); sh[t] = a; }
    __syncthreads();
  }
  if (t == 0) out[j] = sh[0];
}

// workspace layout (u32 words unless noted): hist[W*B] | offs[W*B] | cursor[W*B] | sorted[W*n] ; then XYZZ:
// buckets[W*B] | S | T | out[W] | head[W*nseg] | tail[W*nseg]
static uint32_t pip_nseg(uint32_t n) { return (n + PIP_SEG - 1) / PIP_SEG + 1; }
size_t pippenger_workspace_bytes(uint32_t n) {
  size_t words = (size_t)3 * PIP_W * PIP_B + (size_t)PIP_W * n;
  size_t pts = (size_t)PIP_W * PIP_B + 2 * (size_t)PIP_W * PIP_NCHUNK + PIP_W + 2 * (size_t)PIP_W * pip_nseg(n);
  return ((words * 4 + 255) / 256) * 256 + pts * sizeof(G1XYZZ);
}
uint32_t pippenger_windows() { return PIP_W; }

// window sums land in out_windows[16] (device); ev0/ev1 (optional) bracket the bucket-accumulation kernel
void launch_pippenger_g1(hipStream_t st, const G1Affine* bases, const Fr* scalars, uint32_t n, void* workspace, G1XYZZ** out_windows,
                         hipEvent_t ev0, hipEvent_t ev1) {
  uint32_t* hist = (uint32_t*)workspace;
  uint32_t* offs = hist + PIP_W * PIP_B;
  uint32_t* cursor = offs + PIP_W * PIP_B;
  uint32_t* sorted = cursor + PIP_W * PIP_B;
  size_t words = (size_t)3 * PIP_W * PIP_B + (size_t)PIP_W * n;
  G1XYZZ* buckets = (G1XYZZ*)((char*)workspace + ((words * 4 + 255) / 256) * 256);
  G1XYZZ* S = buckets + (size_t)PIP_W * PIP_B;
  G1XYZZ* T = S + (size_t)PIP_W * PIP_NCHUNK;
  G1XYZZ* out = T + (size_t)PIP_W * PIP_NCHUNK;
  const uint32_t nseg = pip_nseg(n);
  G1XYZZ* head = out + PIP_W;
  G1XYZZ* tail = head + (size_t)PIP_W * nseg;
  (void)hipMemsetAsync(hist, 0, sizeof(uint32_t) * 3 * PIP_W * PIP_B, st);   // hist, offs, cursor
  if (n) hipLaunchKernelGGL(k_pip_count, dim3((n + 255) / 256), dim3(256), 0, st, scalars, n, hist);
  hipLaunchKernelGGL(k_pip_scan, dim3(PIP_W), dim3(1024), 0, st, hist, offs);
  if (n) hipLaunchKernelGGL(k_pip_scatter, dim3((n + 255) / 256), dim3(256), 0, st, scalars, n, offs, cursor, sorted);
  if (ev0) hipEventRecord(ev0, st);
  hipLaunchKernelGGL(k_pip_segments, dim3((PIP_W * nseg + 255) / 256), dim3(256), 0, st, bases, n, nseg, offs, hist, sorted, buckets, head,
                     tail);
  if (ev1) hipEventRecord(ev1, st);
  hipLaunchKernelGGL(k_pip_fixup, dim3(PIP_W * PIP_B / 256), dim3(256), 0, st, nseg, offs, hist, buckets, head, tail);
  hipLaunchKernelGGL(k_pip_chunks, dim3(PIP_W * PIP_NCHUNK / 64), dim3(64), 0, st, buckets, S, T);
  hipLaunchKernelGGL(k_pip_windows, dim3(PIP_W), dim3(64), 0, st, S, T, out);
  *out_windows = out;
}

}  // namespace spp

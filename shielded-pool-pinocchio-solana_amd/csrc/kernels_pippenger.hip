// General-base G1 multi-scalar multiplication (Pippenger bucket method) for large N -- BASELINE.json configs[4]
// (2^24 points) and any MSM whose bases are not a resident proving key.  SURVEY 8a a6: "digit extract -> bucket
// sort -> bucket accumulate -> bucket reduce -> window combine".
//
//   0. k_pip_digits   lane per point: canonical scalar -> 16 signed 16-bit digits (fold sign applied), stored window-major
//                     as int16 [16][n] -- the later passes read 2 B per point and window instead of a 32 B scalar
//   1. k_pip_count    one 1024-lane workgroup per (window, tile of 2^20 points): histogram over the 2^15 buckets in LDS
//                     (128 KB of the CU's 160 KB; ds_add_u32), written out once per tile -- no global atomics
//   2. k_pip_totals   lane per (window, bucket): bucket totals;  k_pip_scan: exclusive prefix over the buckets of a window
//   3. the bucket sort, TWO LEVELS (round 3).  One pass with 2^15 destinations wrote every 4-byte entry
//      to a line of its own (6.8 GB of HBM writes for 1.07 GB of payload, 6.8 ms): a 2^20-point tile puts 32 entries into each
//      bucket, spread over the whole pass.  Now each pass has few destinations per workgroup, all written between two barriers:
//      k_pip_part1    workgroup per (window, 16 K points): ranks within the 256 COARSE bins (bucket >> 7) from LDS atomics,
//                     one global atomic per bin and workgroup claims the run, the entries (index | sign | 7 fine key bits) are
//                     put in bin order in LDS and leave as runs of ~64 entries (256 B), consecutive lanes on consecutive words
//      k_pip_part2    workgroup per (window, 16 K entries of that list): the same by bucket inside the coarse bins the
//                     piece touches (normally one or two: <= 256 counters; a skewed window may put many small bins into
//                     one piece, taken 64 bins at a time), runs of ~64 entries into the final list
//   4. k_pip_segments lane per 256-entry segment of a window's sorted list: gathers its points (64 B each) and folds them
//                     with mixed additions per bucket; k_pip_fixup joins the buckets that span segments
//   5. k_pip_chunks   lane per 16-bucket chunk: running sums  S = sum B_b,  T = sum (b_local+1) B_b          (32 additions deep)
//   6. k_pip_bits     sum_c c S_c over the 2048 chunks of a window by the BITS of c: workgroup per (window, bit k) tree-adds
//                     the S_c with bit k set (plus one workgroup for sum_c T_c) -- 12 additions deep instead of the 128 + 8 x 15
//                     of the round-2 running sums;  k_pip_windows: lane per window, T + 16 sum_k 2^k U_k by Horner
//   host: Horner over the 16 window sums.
// HBM traffic is dominated by step 4: every base is gathered once per window (64 B x N x 16), on top of the
// algorithmic 96 B x N; lanes own equal-size segments of the sorted lists, so the load is balanced whatever the scalar
// distribution (uniform: N / 2^15 points per bucket; witness-like: most points in a few hundred buckets of window 0).
#include "kernels.hpp"
#include "f29.hpp"
#include <cstdlib>

namespace spp {

static constexpr uint32_t PIP_C = 16, PIP_W = 16, PIP_B = 1u << (PIP_C - 1);   // 2^15 buckets per window
static constexpr uint32_t PIP_CHUNK = 16, PIP_NCHUNK = PIP_B / PIP_CHUNK, PIP_CHUNK_BITS = 11;   // 2048 chunks of 16 buckets
static constexpr uint32_t PIP_SLOTS = PIP_CHUNK_BITS + 1;     // per window: sum T_c, then sum of the S_c with bit k of c set
static constexpr uint32_t PIP_FINE_BITS = 7, PIP_FINE = 1u << PIP_FINE_BITS, PIP_COARSE = PIP_B >> PIP_FINE_BITS;
static constexpr uint32_t PIP_PART = 16384;                   // entries per sorting workgroup: 1024 lanes x 16

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
// lane per (window, bucket): hist = total over the tiles
__global__ void __launch_bounds__(256) k_pip_totals(uint32_t ntiles, const uint32_t* __restrict__ hist_tile, uint32_t* __restrict__ hist) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= PIP_W * PIP_B) return;
  const uint32_t j = g / PIP_B, b = g % PIP_B;
  uint32_t run = 0;
  for (uint32_t tile = 0; tile < ntiles; tile++) run += hist_tile[((size_t)j * ntiles + tile) * PIP_B + b];
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

// next free slot of every coarse bin (level 1) and of every bucket (level 2), window-relative
__global__ void __launch_bounds__(256) k_pip_cursors(const uint32_t* __restrict__ offs, uint32_t* __restrict__ cur1, uint32_t* __restrict__ cur2) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= PIP_W * PIP_B) return;
  const uint32_t o = offs[g];
  cur2[g] = o;
  if ((g & (PIP_FINE - 1)) == 0) cur1[g >> PIP_FINE_BITS] = o;
}

// Level 1: workgroup per (window, piece of PIP_PART points).  Phase 1 ranks every entry inside its coarse bin (LDS atomic, the
// returned count is the rank); phase 2 turns the counts into local offsets (scan) and claims cnt[bin] slots of the bin's global
// range with ONE atomic per bin; phase 3 puts the entries in bin order into an LDS staging buffer; phase 4 copies the buffer
// out with consecutive lanes on consecutive addresses -- a run of ~64 entries is two full-line stores instead of 64 four-byte
// ones (the unstaged version left 2.3 GB of HBM writes for 1.07 GB of entries: the L2 did not always see a run complete).
// PACKED (n <= 2^24): the 7 key bits travel in bits 24..30 of the entry, no byte array.
__device__ __forceinline__ uint32_t pip_wave_incl_scan(uint32_t v, uint32_t lane) {
  SPP_UNROLL for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
// largest i in [0, count) with pre[i] <= s   (pre = non-decreasing exclusive prefix; empty slots repeat the next start)
__device__ __forceinline__ uint32_t pip_slot_of(const uint32_t* pre, uint32_t count, uint32_t s) {
  uint32_t a = 0, b = count - 1;
  while (a < b) {
    const uint32_t mid = (a + b + 1) >> 1;
    if (pre[mid] <= s) a = mid; else b = mid - 1;
  }
  return a;
}
template <bool PACKED>
__global__ void __launch_bounds__(1024) k_pip_part1(const int16_t* __restrict__ digits, uint32_t n, uint32_t nparts,
                                                    uint32_t* __restrict__ cur1, uint32_t* __restrict__ l1_idx, uint8_t* __restrict__ l1_key) {
  __shared__ uint32_t cnt[PIP_COARSE];                        // counts, then the local exclusive prefix
  __shared__ uint32_t gb[PIP_COARSE];                         // where the bin's run starts in the window's list
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t stage[PIP_PART];                        // 64 KB
  __shared__ uint8_t stagek[PACKED ? 4 : PIP_PART];
  const uint32_t j = blockIdx.x / nparts, part = blockIdx.x % nparts, t = threadIdx.x, lane = t & 63;
  if (t < PIP_COARSE) cnt[t] = 0;
  __syncthreads();
  const int16_t* dg = digits + (size_t)j * n;
  const uint32_t lo = part * PIP_PART;
  uint32_t ent[PIP_PART / 1024], kr[PIP_PART / 1024];         // entry; bucket << 16 | rank (rank < 2^14)
  SPP_UNROLL for (uint32_t k = 0; k < PIP_PART / 1024; k++) {
    const uint32_t i = lo + k * 1024 + t;
    const int16_t v = i < n ? dg[i] : (int16_t)0;
    kr[k] = 0xffffffffu;
    if (v != 0) {
      uint32_t b, sg;
      pip_decode(v, b, sg);
      ent[k] = i | sg;
      kr[k] = (b << 16) | atomicAdd(&cnt[b >> PIP_FINE_BITS], 1u);
    }
  }
  __syncthreads();
  uint32_t c = 0, incl = 0;
  if (t < PIP_COARSE) {                                       // waves 0..3, whole
    c = cnt[t];
    incl = pip_wave_incl_scan(c, lane);
    if (lane == 63) wsum[t >> 6] = incl;
  }
  __syncthreads();
  if (t < PIP_COARSE) {
    uint32_t base = 0;
    for (uint32_t w = 0; w < (t >> 6); w++) base += wsum[w];
    cnt[t] = base + incl - c;
    gb[t] = c ? atomicAdd(&cur1[j * PIP_COARSE + t], c) : 0u;
  }
  __syncthreads();
  SPP_UNROLL for (uint32_t k = 0; k < PIP_PART / 1024; k++) {
    if (kr[k] != 0xffffffffu) {
      const uint32_t b = kr[k] >> 16, s = cnt[b >> PIP_FINE_BITS] + (kr[k] & 0xffffu);
      if constexpr (PACKED) stage[s] = ent[k] | ((b & (PIP_FINE - 1)) << 24);
      else {
        stage[s] = ent[k];
        stagek[s] = (uint8_t)(b & (PIP_FINE - 1));
      }
    }
  }
  __syncthreads();
  const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  uint32_t* oi = l1_idx + (size_t)j * n;
  uint8_t* ok = l1_key + (size_t)j * n;
  for (uint32_t s = t; s < total; s += 1024) {
    const uint32_t bin = pip_slot_of(cnt, PIP_COARSE, s), dest = gb[bin] + (s - cnt[bin]);
    oi[dest] = stage[s];
    if constexpr (!PACKED) ok[dest] = stagek[s];
  }
}

// Level 2: workgroup per (window, piece of PIP_PART2 entries of the level-1 list).  The piece lies in coarse bins c_lo..c_hi
// (one or two for uniform scalars; a skewed window may put many small bins into one piece: the piece is then taken in sub-pieces
// of at most 16 coarse bins = PIP_R2 counters); counter (c - c0) * 128 + fine key, the same four phases, cursor per bucket.
static constexpr uint32_t PIP_PART2 = 8192, PIP_R2 = 2048, PIP_R2_PER = PIP_R2 / 1024;    // 49 KB of LDS: two workgroups per CU
template <bool PACKED>
__global__ void __launch_bounds__(1024) k_pip_part2(uint32_t n, uint32_t nparts, const uint32_t* __restrict__ offs,
                                                    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ l1_idx,
                                                    const uint8_t* __restrict__ l1_key, uint32_t* __restrict__ cur2,
                                                    uint32_t* __restrict__ sorted) {
  __shared__ uint32_t cnt[PIP_R2];                            // counts, then the local exclusive prefix
  __shared__ uint32_t gb[PIP_R2];                             // where the bucket's run starts in the window's list
  __shared__ uint32_t stage[PIP_PART2];                       // 32 KB
  __shared__ uint32_t cstart[PIP_COARSE + 1];
  __shared__ uint32_t wsum[16];
  const uint32_t j = blockIdx.x / nparts, part = blockIdx.x % nparts, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t* o = offs + j * PIP_B;
  const uint32_t total = o[PIP_B - 1] + hist[j * PIP_B + PIP_B - 1];
  const uint32_t lo = part * PIP_PART2;
  if (lo >= total) return;                                    // the whole workgroup
  const uint32_t hi = min(lo + PIP_PART2, total);
  if (t < PIP_COARSE) cstart[t] = o[t << PIP_FINE_BITS];
  if (t == PIP_COARSE) cstart[PIP_COARSE] = total;
  __syncthreads();
  const uint32_t c_lo = pip_slot_of(cstart, PIP_COARSE, lo), c_hi = pip_slot_of(cstart, PIP_COARSE, hi - 1);
  const uint32_t* li = l1_idx + (size_t)j * n;
  const uint8_t* lk = l1_key + (size_t)j * n;
  uint32_t* out = sorted + (size_t)j * n;
  for (uint32_t c0 = c_lo; c0 <= c_hi; c0 += PIP_R2 / PIP_FINE) {
    const uint32_t c1 = min(c0 + PIP_R2 / PIP_FINE - 1, c_hi);
    const uint32_t sub_lo = max(lo, cstart[c0]), sub_hi = min(hi, cstart[c1 + 1]);
    if (sub_lo >= sub_hi) continue;                           // block-uniform
    const uint32_t range = (c1 - c0 + 1) << PIP_FINE_BITS;
    for (uint32_t b = t; b < range; b += 1024) cnt[b] = 0;
    __syncthreads();
    uint32_t ent[PIP_PART2 / 1024], kr[PIP_PART2 / 1024];       // entry; counter << 16 | rank
    SPP_UNROLL for (uint32_t k = 0; k < PIP_PART2 / 1024; k++) {
      const uint32_t pos = sub_lo + k * 1024 + t;
      kr[k] = 0xffffffffu;
      if (pos < sub_hi) {
        const uint32_t c = c0 == c1 ? c0 : pip_slot_of(cstart, PIP_COARSE, pos);
        uint32_t en = li[pos], fine;
        if constexpr (PACKED) {
          fine = (en >> 24) & (PIP_FINE - 1);
          en &= 0x80ffffffu;
        } else fine = lk[pos];
        const uint32_t kk = ((c - c0) << PIP_FINE_BITS) + fine;
        ent[k] = en;
        kr[k] = (kk << 16) | atomicAdd(&cnt[kk], 1u);
      }
    }
    __syncthreads();
    uint32_t loc[PIP_R2_PER], sum = 0;                        // scan: PIP_R2_PER consecutive counters per lane
    SPP_UNROLL for (uint32_t q = 0; q < PIP_R2_PER; q++) {
      const uint32_t idx = t * PIP_R2_PER + q;
      loc[q] = idx < range ? cnt[idx] : 0u;
      sum += loc[q];
    }
    const uint32_t incl = pip_wave_incl_scan(sum, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < wave; w++) run += wsum[w];
    SPP_UNROLL for (uint32_t q = 0; q < PIP_R2_PER; q++) {
      const uint32_t idx = t * PIP_R2_PER + q;
      if (idx < range) {
        cnt[idx] = run;
        gb[idx] = loc[q] ? atomicAdd(&cur2[j * PIP_B + (c0 << PIP_FINE_BITS) + idx], loc[q]) : 0u;
        run += loc[q];
      }
    }
    __syncthreads();
    SPP_UNROLL for (uint32_t k = 0; k < PIP_PART2 / 1024; k++)
      if (kr[k] != 0xffffffffu) stage[cnt[kr[k] >> 16] + (kr[k] & 0xffffu)] = ent[k];
    __syncthreads();
    for (uint32_t s = t; s < sub_hi - sub_lo; s += 1024) {
      const uint32_t kk = pip_slot_of(cnt, range, s);
      out[gb[kk] + (s - cnt[kk])] = stage[s];
    }
    __syncthreads();                                          // the next sub-piece reuses cnt / gb / stage
  }
}

// Bucket accumulation, robust to skewed digit distributions (real witnesses are mostly small: one window then holds
// tens of thousands of points in a few hundred buckets).  Lanes own fixed-size SEGMENTS of a window's sorted point list
// instead of buckets: lane (j, s) folds entries [s*SEG, (s+1)*SEG) with mixed additions, restarting its accumulator at
// every bucket boundary.  A bucket that lies inside one segment is written directly; a bucket that spans segments
// gets one partial per segment it touches -- `tail[s]` from the segment where it starts, `head[s]` from every later
// one -- and k_pip_fixup adds them up (a handful of additions for uniform scalars, cnt/SEG for a heavy bucket).
static constexpr uint32_t PIP_SEG = 256, PIP_IDX = 16;
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
  // software pipeline: the (randomly placed, 64 B) base of entry k + 1 is requested before the ~2.3 K
  // instructions of the addition of entry k, so the HBM round trip of the gather hides behind arithmetic of the same lane.
  // The lane's list entries come through LDS sixteen at a time (64 B, one sector, fetched once): a lane reading one 4-byte
  // entry per addition found its sector evicted between two reads -- the gathers sweep the L2 in tens of microseconds -- and
  // the list cost 4.5x its size in fabric requests (profiles/round3_pippenger_pmc_hbm.json).
  __shared__ uint32_t sidx[PIP_IDX][256];
  const uint32_t tid = threadIdx.x;
  auto refill = [&](uint32_t k0) {                         // entries k0 .. k0 + 15 of this lane's segment, column tid
    uint32_t v[PIP_IDX];
    SPP_UNROLL for (uint32_t q = 0; q < PIP_IDX; q++) v[q] = k0 + q < end ? seg[k0 + q] : 0u;
    SPP_UNROLL for (uint32_t q = 0; q < PIP_IDX; q++) sidx[q][tid] = v[q];
  };
  refill(pos);
  uint32_t e = sidx[0][tid];
  Affine<F> p = bases[e & 0x7fffffffu];
  for (uint32_t k = pos; k < end; k++) {
    const uint32_t e_cur = e;
    const Affine<F> p_cur = p;
    if (k + 1 < end) {
      const uint32_t r = (k + 1 - pos) & (PIP_IDX - 1);
      if (r == 0) refill(k + 1);
      e = sidx[r][tid];
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
// lane per (window, bucket): empty buckets -> infinity; buckets spanning several segments -> tail[s0] + head[s0+1..s1].
// A bucket that spans more than PIP_FIX_SEQ segments (byte-sized witness values: tens of thousands of points in each of 255
// buckets of window 0) is summed by a whole wave of k_pip_fixup_heavy -- strided partial sums, then a shuffle tree -- instead of
// one lane adding hundreds of partials in sequence.
static constexpr uint32_t PIP_FIX_SEQ = 12;
template <class T>
__device__ __forceinline__ T pip_shfl_down(const T& v, int delta) {
  static_assert(sizeof(T) % 4 == 0, "word-sized");
  T r;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
  uint32_t* d = reinterpret_cast<uint32_t*>(&r);
  SPP_UNROLL for (uint32_t i = 0; i < sizeof(T) / 4; i++) d[i] = (uint32_t)__shfl_down((int)s[i], delta, 64);
  return r;
}
template <class F>
__global__ void __launch_bounds__(256) k_pip_fixup(uint32_t nseg, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist,
                                                   XYZZ<F>* __restrict__ buckets, const XYZZ<F>* __restrict__ head,
                                                   const XYZZ<F>* __restrict__ tail) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;       // the grid is exactly PIP_W * PIP_B lanes
  const uint32_t j = g / PIP_B, cnt = hist[g];
  uint32_t s0 = 0, s1 = 0;
  if (cnt == 0) buckets[g] = XYZZ<F>::infinity();
  else {
    s0 = offs[g] / PIP_SEG;
    s1 = (offs[g] + cnt - 1) / PIP_SEG;
  }
  if (s1 != s0 && s1 - s0 <= PIP_FIX_SEQ) {                      // (s0 == s1: written directly by its segment lane)
    XYZZ<F> acc = tail[(size_t)j * nseg + s0];
    for (uint32_t s = s0 + 1; s <= s1; s++) acc.add(head[(size_t)j * nseg + s]);
    buckets[g] = acc;
  }
}
// wave per (window, bucket): the heavy buckets (all other waves leave at once; heavy buckets are neighbours -- the byte values of
// window 0 -- so a lane-per-bucket kernel would serialise 64 of them in every wave it does not leave idle)
template <class F>
__global__ void __launch_bounds__(256) k_pip_fixup_heavy(uint32_t nseg, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ hist,
                                                         XYZZ<F>* __restrict__ buckets, const XYZZ<F>* __restrict__ head,
                                                         const XYZZ<F>* __restrict__ tail) {
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // the grid is exactly PIP_W * PIP_B waves
  const uint32_t j = g / PIP_B, cnt = hist[g];
  if (cnt == 0) return;
  const uint32_t s0 = offs[g] / PIP_SEG, s1 = (offs[g] + cnt - 1) / PIP_SEG;
  if (s1 - s0 <= PIP_FIX_SEQ) return;
  const XYZZ<F>* hd = head + (size_t)j * nseg;
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (uint32_t s = s0 + 1 + lane; s <= s1; s += 64) acc.add(hd[s]);
  for (int d = 32; d > 0; d >>= 1) {
    const XYZZ<F> o = pip_shfl_down(acc, d);
    if ((int)lane < d) acc.add(o);
  }
  if (lane == 0) {
    XYZZ<F> r = tail[(size_t)j * nseg + s0];
    r.add(acc);
    buckets[g] = r;
  }
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

// window sum = sum_c ( T_c + 16 c S_c ),  sum_c c S_c = sum_k 2^k U_k with U_k = sum of the S_c whose chunk number has bit k set.
// Workgroup (slot, window): slot 0 adds up the T_c, slot 1 + k the S_c with bit k set -- strided partial sums, then an LDS tree.
template <class F>
__global__ void __launch_bounds__(128) k_pip_bits(const XYZZ<F>* __restrict__ S, const XYZZ<F>* __restrict__ T, XYZZ<F>* __restrict__ U) {
  __shared__ XYZZ<F> sh[128];
  const uint32_t slot = blockIdx.x, j = blockIdx.y, t = threadIdx.x;
  XYZZ<F> acc = XYZZ<F>::infinity();
  if (slot == 0) {
    for (uint32_t c = t; c < PIP_NCHUNK; c += 128) acc.add(T[j * PIP_NCHUNK + c]);
  } else {
    const uint32_t k = slot - 1;
    // the chunk numbers with bit k set, enumerated densely: m in [0, NCHUNK / 2) -> insert a 1 at bit k
    for (uint32_t m = t; m < PIP_NCHUNK / 2; m += 128) {
      const uint32_t c = ((m >> k) << (k + 1)) | (1u << k) | (m & ((1u << k) - 1));
      acc.add(S[j * PIP_NCHUNK + c]);
    }
  }
  sh[t] = acc;
  __syncthreads();
  for (uint32_t w = 64; w > 0; w >>= 1) {
    if (t < w) { XYZZ<F> a = sh[t]; a.add(sh[t + w]); sh[t] = a; }
    __syncthreads();
  }
  if (t == 0) U[j * PIP_SLOTS + slot] = sh[0];
}
// lane per window: T + 16 (U_0 + 2 U_1 + ... + 2^10 U_10)
template <class F>
__global__ void __launch_bounds__(64) k_pip_windows(const XYZZ<F>* __restrict__ U, XYZZ<F>* __restrict__ out) {
  const uint32_t j = threadIdx.x;
  if (j >= PIP_W) return;
  const XYZZ<F>* u = U + j * PIP_SLOTS;
  XYZZ<F> r = u[PIP_CHUNK_BITS];
  for (int k = (int)PIP_CHUNK_BITS - 1; k >= 1; k--) {
    r.dbl_inplace();
    r.add(u[k]);
  }
  for (uint32_t m = PIP_CHUNK; m > 1; m >>= 1) r.dbl_inplace();
  r.add(u[0]);
  out[j] = r;
}

// workspace layout (u32 words unless noted): hist[W*B] | offs[W*B] | cur2[W*B] | cur1[W*COARSE] | hist_tile[W*ntiles*B] | sorted[W*n] |
// l1_idx[W*n] | digits (int16 [W][n], padded to whole words) | l1_key (u8 [W][n], padded) ; then XYZZ: buckets[W*B] | S | T | U[W*SLOTS] |
// out[W] | head[W*nseg] | tail[W*nseg]
static uint32_t pip_nseg(uint32_t n) { return (n + PIP_SEG - 1) / PIP_SEG + 1; }
static uint32_t pip_ntiles(uint32_t n) { return n ? (n + PIP_TILE - 1) / PIP_TILE : 1; }
static size_t pip_words(uint32_t n) {
  return (size_t)3 * PIP_W * PIP_B + (size_t)PIP_W * PIP_COARSE + (size_t)PIP_W * pip_ntiles(n) * PIP_B + (size_t)2 * PIP_W * n +
         ((size_t)PIP_W * n + 1) / 2 + ((size_t)PIP_W * n + 3) / 4;
}
template <class F>
static size_t pip_ws_bytes(uint32_t n) {
  size_t pts = (size_t)PIP_W * PIP_B + 2 * (size_t)PIP_W * PIP_NCHUNK + (size_t)PIP_W * PIP_SLOTS + PIP_W + 2 * (size_t)PIP_W * pip_nseg(n);
  return ((pip_words(n) * 4 + 255) / 256) * 256 + pts * sizeof(XYZZ<F>);
}
size_t pippenger_workspace_bytes(uint32_t n) { return pip_ws_bytes<Fq>(n); }
size_t pippenger_workspace_bytes_g2(uint32_t n) { return pip_ws_bytes<Fq2>(n); }
uint32_t pippenger_windows() { return PIP_W; }

// window sums land in out_windows[16] (device); ev0/ev1 (optional) bracket the bucket-accumulation kernel
template <class F>
static void launch_pippenger(hipStream_t st, const Affine<F>* bases, const Fr* scalars, uint32_t n, void* workspace, XYZZ<F>** out_windows,
                             hipEvent_t ev0, hipEvent_t ev1) {
  const uint32_t ntiles = pip_ntiles(n), nparts = n ? (n + PIP_PART - 1) / PIP_PART : 1, nparts2 = n ? (n + PIP_PART2 - 1) / PIP_PART2 : 1;
  uint32_t* hist = (uint32_t*)workspace;
  uint32_t* offs = hist + PIP_W * PIP_B;
  uint32_t* cur2 = offs + PIP_W * PIP_B;
  uint32_t* cur1 = cur2 + PIP_W * PIP_B;
  uint32_t* hist_tile = cur1 + PIP_W * PIP_COARSE;
  uint32_t* sorted = hist_tile + (size_t)PIP_W * ntiles * PIP_B;
  uint32_t* l1_idx = sorted + (size_t)PIP_W * n;
  int16_t* digits = (int16_t*)(l1_idx + (size_t)PIP_W * n);
  uint8_t* l1_key = (uint8_t*)((uint32_t*)digits + ((size_t)PIP_W * n + 1) / 2);
  XYZZ<F>* buckets = (XYZZ<F>*)((char*)workspace + ((pip_words(n) * 4 + 255) / 256) * 256);
  XYZZ<F>* S = buckets + (size_t)PIP_W * PIP_B;
  XYZZ<F>* T = S + (size_t)PIP_W * PIP_NCHUNK;
  XYZZ<F>* U = T + (size_t)PIP_W * PIP_NCHUNK;
  XYZZ<F>* out = U + (size_t)PIP_W * PIP_SLOTS;
  const uint32_t nseg = pip_nseg(n);
  XYZZ<F>* head = out + PIP_W;
  XYZZ<F>* tail = head + (size_t)PIP_W * nseg;
  if (n) hipLaunchKernelGGL(k_pip_digits, dim3((n + 255) / 256), dim3(256), 0, st, scalars, n, digits);
  hipLaunchKernelGGL(k_pip_count, dim3(PIP_W * ntiles), dim3(1024), 0, st, digits, n, ntiles, hist_tile);
  hipLaunchKernelGGL(k_pip_totals, dim3(PIP_W * PIP_B / 256), dim3(256), 0, st, ntiles, hist_tile, hist);
  hipLaunchKernelGGL(k_pip_scan, dim3(PIP_W), dim3(1024), 0, st, hist, offs);
  hipLaunchKernelGGL(k_pip_cursors, dim3(PIP_W * PIP_B / 256), dim3(256), 0, st, offs, cur1, cur2);
  if (n <= (1u << 24) && !getenv("SPP_PIP_UNPACKED")) {     // (the switch lets the tests run the large-n form on small inputs)
    hipLaunchKernelGGL(k_pip_part1<true>, dim3(PIP_W * nparts), dim3(1024), 0, st, digits, n, nparts, cur1, l1_idx, l1_key);
    hipLaunchKernelGGL(k_pip_part2<true>, dim3(PIP_W * nparts2), dim3(1024), 0, st, n, nparts2, offs, hist, l1_idx, l1_key, cur2, sorted);
  } else {
    hipLaunchKernelGGL(k_pip_part1<false>, dim3(PIP_W * nparts), dim3(1024), 0, st, digits, n, nparts, cur1, l1_idx, l1_key);
    hipLaunchKernelGGL(k_pip_part2<false>, dim3(PIP_W * nparts2), dim3(1024), 0, st, n, nparts2, offs, hist, l1_idx, l1_key, cur2, sorted);
  }
  if (ev0) hipEventRecord(ev0, st);
  hipLaunchKernelGGL(k_pip_segments<F>, dim3((PIP_W * nseg + 255) / 256), dim3(256), 0, st, bases, n, nseg, offs, hist, sorted, buckets,
                     head, tail);
  if (ev1) hipEventRecord(ev1, st);
  hipLaunchKernelGGL(k_pip_fixup<F>, dim3(PIP_W * PIP_B / 256), dim3(256), 0, st, nseg, offs, hist, buckets, head, tail);
  hipLaunchKernelGGL(k_pip_fixup_heavy<F>, dim3(PIP_W * PIP_B / 4), dim3(256), 0, st, nseg, offs, hist, buckets, head, tail);
  hipLaunchKernelGGL(k_pip_chunks<F>, dim3(PIP_W * PIP_NCHUNK / 64), dim3(64), 0, st, buckets, S, T);
  hipLaunchKernelGGL(k_pip_bits<F>, dim3(PIP_SLOTS, PIP_W), dim3(128), 0, st, S, T, U);
  hipLaunchKernelGGL(k_pip_windows<F>, dim3(1), dim3(64), 0, st, U, out);
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

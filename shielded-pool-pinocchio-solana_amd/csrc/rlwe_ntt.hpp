// Exact negacyclic products for the RLWE witness kernel through a 1024-point NTT held in LDS.
//
// Replaces the O(n^2) loops of the reference -- negacyclic_mul_mod_q (scripts/generate_audit.py:45-54),
// negacyclic_matrix_row_mod_q (:57-66) and the quotient loops (:539-554) -- by  NTT(r) . NTT(pk)  in TWO prime fields:
//   P0 = q = 167 772 161 = 5 * 2^25 + 1   (the RLWE modulus itself: the residue IS the ciphertext coefficient)
//   P1 =         786 433 = 3 * 2^18 + 1
// and a CRT lift.  The integer  S_i = <row_i(pk), r>  satisfies |S_i| <= 1024 * (q-1) * 128, i.e. S = s0 + q*t with
// |t| <= 2^17 + 1 < P1/2: the CRT digit t, centred mod P1, is exact, and the signed quotient witness
//   k = floor((S + e + Delta*m) / q) = t + floor((s0 + e + Delta*m) / q)          (generate_audit.py:236-243)
// follows without any wide division.
//
// One wavefront per polynomial.  Lane t holds the 16 coefficients  t + 64*j ; the transform is 1024 = 16 x 16 x 4:
//   pass 1: 16-point DFT over the stride-64 coefficients in registers, twiddle w^(t*k1), store to LDS;
//   pass 2: 16-point DFT over n2, twiddle w^(16*n3*k2), store to LDS;
//   pass 3: four 4-point DFTs; the results land in the SAME lane layout in natural order (no bit reversal anywhere),
// so forward and inverse are one routine with w or w^-1, and pointwise products / global loads are coalesced.
// Arithmetic: SIGNED 32-bit Montgomery, lazily reduced: a product of any int32 with a constant |c| < p comes out in (-p, p)
// (3 instructions: v_mad_i64_i32, v_mul_lo_u32, v_mad_i64_i32), additions and subtractions are single instructions and
// values are simply allowed to grow between products; the one place where field 0 would pass 2^31 (the all-sums path of a
// 16-point DFT reaches 16p) takes two extra reductions.  tests/host/rlwe_ntt_check.cpp re-derives those bounds and runs the
// identical per-lane phases on the host, lane by lane, against the schoolbook definition.
// Data stay in the plain domain, constants carry the factor 2^32.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RN_HD __host__ __device__ __forceinline__
#else
#define RN_HD inline
#endif

namespace spp {

static constexpr uint32_t RN_N = 1024, RN_SLOTS = 64;
static constexpr int32_t RN_P[2] = {167772161, 786433};
static constexpr uint32_t RN_S1 = 68, RN_S2 = 264;             // padded LDS strides of the two exchange layouts
static constexpr uint32_t RN_LDS_WORDS = 16 * RN_S1;           // 1088 words (>= 4 * RN_S2 = 1056)

// per-field constants and tables, built on the host (rn_build_tables) and resident in HBM / L2
struct RnField {
  int32_t p;
  uint32_t pinv_neg;               // -p^-1 mod 2^32
  int32_t one;                     // 2^32 mod p: multiplying by it reduces any int32 into (-p, p)
  int32_t w16[2][8];               // (w^64)^e * 2^32 mod p, e < 8, for direction 0 (forward) / 1 (inverse)
  int32_t crt;                     // field 1 only: q^-1 * 2^32 mod P1
};
struct RnTables {
  RnField f[2];
  // all entries carry the Montgomery factor 2^32 and lie in [0, p):
  const int32_t* w[2][2];          // [field][direction][1024]: w^e  /  w^-e
  const int32_t* psi[2];           // [field][1024]: psi^j                       (twist before the forward transform)
  const int32_t* ipsi[2];          // [field][1024]: psi^-j                      (untwist; 1/1024 is folded into the pk transform)
};

// a * b * 2^-32 mod p in (-p, p) for ANY int32 a and |b| < p
RN_HD int32_t rn_mul(int32_t a, int32_t b, const RnField& f) {
  const int64_t t = (int64_t)a * b;
  const int32_t m = (int32_t)((uint32_t)t * f.pinv_neg);
  return (int32_t)((t + (int64_t)m * f.p) >> 32);
}
RN_HD int32_t rn_canon(int32_t a, const RnField& f) { return a + ((a >> 31) & f.p); }      // (-p, p) -> [0, p)

// in-register 16-point DFT, natural order in and out, twiddles w16[e] = (w^64)^e.  Inputs |x| < p.  Growth (units of p): the
// sums double per stage, every twiddle product resets to 1; before the last stage x[0] and x[1] stand at 8 and would leave at
// 16 (field 0: 2^31 / q = 12.8), so with REDUCE they are multiplied by one first.  Outputs: |X| <= 9p (REDUCE) / 16p.
template <bool REDUCE>
RN_HD void rn_dft16(int32_t (&x)[16], const int32_t (&w16)[8], const RnField& f) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int32_t a = x[i], b = x[i + 8];
    x[i] = a + b;
    x[i + 8] = i == 0 ? a - b : rn_mul(a - b, w16[i], f);
  }
#pragma unroll
  for (int blk = 0; blk < 16; blk += 8) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int32_t a = x[blk + i], b = x[blk + i + 4];
      x[blk + i] = a + b;
      x[blk + i + 4] = i == 0 ? a - b : rn_mul(a - b, w16[2 * i], f);
    }
  }
#pragma unroll
  for (int blk = 0; blk < 16; blk += 4) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int32_t a = x[blk + i], b = x[blk + i + 2];
      x[blk + i] = a + b;
      x[blk + i + 2] = i == 0 ? a - b : rn_mul(a - b, w16[4], f);
    }
  }
  if (REDUCE) {
    x[0] = rn_mul(x[0], f.one, f);
    x[1] = rn_mul(x[1], f.one, f);
  }
#pragma unroll
  for (int blk = 0; blk < 16; blk += 2) {
    const int32_t a = x[blk], b = x[blk + 1];
    x[blk] = a + b;
    x[blk + 1] = a - b;
  }
  // bit-reversed -> natural (register renaming: every index is a compile-time constant)
  int32_t y[16];
#pragma unroll
  for (int k = 0; k < 16; k++) y[k] = x[((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3)];
#pragma unroll
  for (int k = 0; k < 16; k++) x[k] = y[k];
}

// ---- the three passes of one 1024-point transform; x = the lane's 16 values (coefficient lane + 64*j), in place ----
// Everything written to LDS is a fresh product, i.e. in (-p, p).
// phase A (lane t): pass 1 + twiddle, write exchange layout 1.  The twiddles w^(lane*k1) are running powers of w^lane: one
// coalesced read + 14 extra multiplications.  Measured alternatives on MI355X (2^16 instances, sustained): 15 gathers from the
// w table per transform -- the vector-memory path of a CU is shared by its four SIMDs; a [k1][lane] table read as 15 coalesced
// rows -- 0.65 ms against 0.52 ms for the running product, although it executes 7 % fewer VALU instructions.
template <bool REDUCE>
RN_HD void rn_pass1(uint32_t lane, int32_t (&x)[16], int32_t* lds, const RnField& f, const int32_t* w, int dir) {
  rn_dft16<REDUCE>(x, f.w16[dir], f);
  lds[lane] = rn_mul(x[0], f.one, f);
  const int32_t wl = w[lane];
  int32_t tw = wl;
#pragma unroll
  for (uint32_t k1 = 1; k1 < 16; k1++) {
    lds[k1 * RN_S1 + lane] = rn_mul(x[k1], tw, f);
    if (k1 < 15) tw = rn_mul(tw, wl, f);
  }
}
// phase B (lane u = 4*k1 + n3), first half: read layout 1   -- barrier before AND after (the writes of pass 2 reuse the buffer)
RN_HD void rn_pass2_read(uint32_t lane, int32_t (&x)[16], const int32_t* lds) {
  const uint32_t k1 = lane >> 2, n3 = lane & 3;
#pragma unroll
  for (uint32_t n2 = 0; n2 < 16; n2++) x[n2] = lds[k1 * RN_S1 + 4 * n2 + n3];
}
// phase B, second half: pass 2 + twiddle, write exchange layout 2
template <bool REDUCE>
RN_HD void rn_pass2(uint32_t lane, int32_t (&x)[16], int32_t* lds, const RnField& f, const int32_t* w, int dir) {
  const uint32_t k1 = lane >> 2, n3 = lane & 3;
  rn_dft16<REDUCE>(x, f.w16[dir], f);
#pragma unroll
  for (uint32_t k2 = 0; k2 < 16; k2++)    // k2 == 0 or n3 == 0: w^0 = `one`, the product is then just the reduction
    lds[n3 * RN_S2 + k1 + 16 * k2] = rn_mul(x[k2], k2 == 0 ? f.one : w[(16 * n3 * k2) & 1023], f);
}
// phase C (lane v): four 4-point DFTs; x[i + 4*k3] = X[v + 64*(i + 4*k3)], |X| <= 4p
RN_HD void rn_pass3(uint32_t lane, int32_t (&x)[16], const int32_t* lds, const RnField& f, int dir) {
  const int32_t w4 = f.w16[dir][4];
#pragma unroll
  for (uint32_t i = 0; i < 4; i++) {
    const uint32_t m = lane + 64 * i;
    const int32_t y0 = lds[m], y1 = lds[RN_S2 + m], y2 = lds[2 * RN_S2 + m], y3 = lds[3 * RN_S2 + m];
    const int32_t s0 = y0 + y2, d0 = y0 - y2, s1 = y1 + y3, d1 = rn_mul(y1 - y3, w4, f);
    x[i] = s0 + s1;
    x[i + 4] = d0 + d1;
    x[i + 8] = s0 - s1;
    x[i + 12] = d0 - d1;
  }
}

// ---- pruned inverse transform for the message slots: only coefficients 0..63 of b*r are ever used (c0 / k0 have 64 slots,
// generate_audit.py:539-545), i.e. element j = 0 of every lane.  Pass 3 then needs m = lane < 64 only, so pass 2 needs its
// outputs k2 = 0..3 only: X[k] = sum_b w16^(b k) * (4-point DFT over a of x[4a + b])[k].
template <bool REDUCE>
RN_HD void rn_dft16_first4(int32_t (&x)[16], const int32_t (&w16)[8], const RnField& f) {
  int32_t Y[4][4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int32_t u0 = x[b], u1 = x[4 + b], u2 = x[8 + b], u3 = x[12 + b];
    const int32_t s0 = u0 + u2, d0 = u0 - u2, s1 = u1 + u3, d1 = rn_mul(u1 - u3, w16[4], f);
    Y[b][0] = s0 + s1;      // |.| <= 4p
    Y[b][1] = d0 + d1;      // 3p
    Y[b][2] = s0 - s1;      // 4p
    Y[b][3] = d0 - d1;      // 3p
  }
  // k = 0: 16p would pass 2^31 in field 0 -> two partial sums, reduced
  x[0] = REDUCE ? rn_mul(Y[0][0] + Y[1][0], f.one, f) + rn_mul(Y[2][0] + Y[3][0], f.one, f) : (Y[0][0] + Y[1][0]) + (Y[2][0] + Y[3][0]);
  x[1] = Y[0][1] + rn_mul(Y[1][1], w16[1], f) + rn_mul(Y[2][1], w16[2], f) + rn_mul(Y[3][1], w16[3], f);
  x[2] = Y[0][2] + rn_mul(Y[1][2], w16[2], f) + rn_mul(Y[2][2], w16[4], f) + rn_mul(Y[3][2], w16[6], f);
  x[3] = Y[0][3] + rn_mul(Y[1][3], w16[3], f) + rn_mul(Y[2][3], w16[6], f) - rn_mul(Y[3][3], w16[1], f);   // w16^9 = -w16
}
template <bool REDUCE>
RN_HD void rn_pass2_first64(uint32_t lane, int32_t (&x)[16], int32_t* lds, const RnField& f, const int32_t* w, int dir) {
  const uint32_t k1 = lane >> 2, n3 = lane & 3;
  rn_dft16_first4<REDUCE>(x, f.w16[dir], f);
#pragma unroll
  for (uint32_t k2 = 0; k2 < 4; k2++)
    lds[n3 * RN_S2 + k1 + 16 * k2] = rn_mul(x[k2], k2 == 0 ? f.one : w[(16 * n3 * k2) & 1023], f);
}
RN_HD int32_t rn_pass3_first64(uint32_t lane, const int32_t* lds) {   // X[lane], |X| <= 4p
  return (lds[lane] + lds[2 * RN_S2 + lane]) + (lds[RN_S2 + lane] + lds[3 * RN_S2 + lane]);
}

// CRT digit: residues s0 in [0,q), s1 in [0,P1) of the integer S = s0 + q*t with |t| < P1/2  ->  t
RN_HD int32_t rn_crt_digit(int32_t s0, int32_t s1, const RnField& f1) {
  int32_t t = rn_mul(s1 - s0, f1.crt, f1);                  // (s1 - s0) / q mod P1, in (-P1, P1)
  const int32_t half = f1.p >> 1;
  if (t > half) t -= f1.p;
  if (t < -half) t += f1.p;
  return t;
}
// floor-division bookkeeping of compute_quotient_and_remainder (generate_audit.py:236-243): v = S + add with S = s0 + q*t
RN_HD void rn_quot_rem(int32_t s0, int32_t t, int32_t add, int32_t& k, uint32_t& rem) {
  const int32_t q = RN_P[0];
  int32_t w = s0 + add;                                     // in (-2^8, 2q): add in [-128, 127 + 655360*255]
  int32_t adj = 0;
  if (w >= q) { w -= q; adj = 1; }
  if (w < 0) { w += q; adj = -1; }
  k = t + adj;
  rem = (uint32_t)w;
}

// ---- wrap correction --------------------------------------------------------------------------------------------------
// The reference's matrix rows hold (q - a) mod q, not -a, in the wrapped positions (negacyclic_matrix_row_mod_q,
// generate_audit.py:57-66), so its INTEGER inner product is  S_ref = S_negacyclic + q * C_i  with
//   C_i = sum_{j > i, a[1024 + i - j] != 0} r_j  =  (sum_{j > i} r_j)  -  sum_{zeros z of a, z > i} r[1024 + i - z].
// Same remainder, quotient shifted by C_i.  The suffix sums come from a prefix scan through LDS (natural order, lane t owns
// the contiguous chunk 16t .. 16t+15); zeros of a public key are rare (probability 1/q each) and handled from a list.
// pre: 1024 + 64 ints of LDS, coefficient i at RN_PAD(i) = i + i/16 (chunk stride 17: the per-chunk walks of the 64 lanes hit
// 32 different banks), tot: 64 ints.  Phases (barrier between each): scatter r -> chunk scan -> offsets -> gather.
RN_HD uint32_t RN_PAD(uint32_t i) { return i + (i >> 4); }
RN_HD void rn_scan_scatter(uint32_t lane, const int32_t (&r)[16], int32_t* pre) {
#pragma unroll
  for (int j = 0; j < 16; j++) pre[RN_PAD(lane + 64 * j)] = r[j];
}
RN_HD void rn_scan_chunk(uint32_t lane, int32_t* pre, int32_t* tot) {
  int32_t acc = 0;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    acc += pre[17 * lane + c];
    pre[17 * lane + c] = acc;          // inclusive prefix inside the chunk
  }
  tot[lane] = acc;
}
// returns (offset of this lane's chunk, total): call after the barrier that follows rn_scan_chunk
RN_HD void rn_scan_offsets(uint32_t lane, const int32_t* tot, int32_t& offset, int32_t& total) {
  int32_t off = 0, all = 0;
  for (uint32_t l = 0; l < 64; l++) {
    const int32_t v = tot[l];
    all += v;
    if (l < lane) off += v;
  }
  offset = off;
  total = all;
}
// suffix (exclusive) sums for this lane's coefficients i = lane + 64 j; the chunk of coefficient i is lane i/16, whose offset
// the caller supplies through chunk_offset (gathered from LDS): suffix = total - (offset[i/16] + inclusive prefix in chunk)
RN_HD void rn_scan_gather(uint32_t lane, int32_t total, const int32_t* pre, const int32_t* offs, int32_t (&suffix)[16]) {
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint32_t i = lane + 64 * j;
    suffix[j] = total - (offs[i >> 4] + pre[RN_PAD(i)]);
  }
}
// zero-list correction for coefficient i: subtract r[1024 + i - z] for every zero position z > i (rbytes: r in natural order)
RN_HD int32_t rn_zero_correction(uint32_t i, const uint16_t* zeros, uint32_t nzeros, const int8_t* rbytes) {
  int32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (uint32_t k = 0; k < nzeros; k++) {
    const uint32_t z = zeros[k];
    if (z > i) c += rbytes[1024 + i - z];
  }
  return c;
}

// ---- host: constant tables ----
inline uint32_t rn_powmod(uint64_t b, uint64_t e, uint64_t p) {
  uint64_t r = 1, x = b % p;
  while (e) {
    if (e & 1) r = r * x % p;
    x = x * x % p;
    e >>= 1;
  }
  return (uint32_t)r;
}
inline int32_t rn_to_mont(uint32_t v, uint32_t p) { return (int32_t)((((uint64_t)v) << 32) % p); }
// host arrays: w[f][dir][1024], psi[f][1024], ipsi[f][1024] (Montgomery form, in [0,p)); scalar members in f[]
struct RnHostTables {
  int32_t w[2][2][1024], psi[2][1024], ipsi[2][1024];
  RnField f[2];
  int32_t pk_scale[2];   // (1/1024) * 2^64 mod p: rn_mul(NTT(pk), pk_scale) = NTT(pk)/1024 in Montgomery form
};
inline void rn_build_tables(RnHostTables& h) {
  for (int k = 0; k < 2; k++) {
    const uint32_t p = (uint32_t)RN_P[k];
    RnField& f = h.f[k];
    f.p = (int32_t)p;
    uint32_t inv = 1;                                   // Newton: p^-1 mod 2^32
    for (int i = 0; i < 5; i++) inv *= 2 - p * inv;
    f.pinv_neg = 0u - inv;
    f.one = rn_to_mont(1, p);
    // a generator of the multiplicative group: p - 1 = 5 * 2^25 (q) resp. 3 * 2^18 (P1)
    const uint32_t odd = k == 0 ? 5 : 3;
    uint32_t g = 2;
    while (rn_powmod(g, (p - 1) / 2, p) == 1 || rn_powmod(g, (p - 1) / odd, p) == 1) g++;
    const uint32_t psi = rn_powmod(g, (p - 1) / 2048, p), ipsi = rn_powmod(psi, p - 2, p);
    const uint32_t w = (uint32_t)((uint64_t)psi * psi % p), iw = (uint32_t)((uint64_t)ipsi * ipsi % p);
    uint64_t a = 1, b = 1, c = 1, d = 1;
    for (int e = 0; e < 1024; e++) {
      h.w[k][0][e] = rn_to_mont((uint32_t)a, p);
      h.w[k][1][e] = rn_to_mont((uint32_t)b, p);
      h.psi[k][e] = rn_to_mont((uint32_t)c, p);
      h.ipsi[k][e] = rn_to_mont((uint32_t)d, p);
      a = a * w % p; b = b * iw % p; c = c * psi % p; d = d * ipsi % p;
    }
    for (int e = 0; e < 8; e++) {
      f.w16[0][e] = h.w[k][0][64 * e];
      f.w16[1][e] = h.w[k][1][64 * e];
    }

    f.crt = k == 1 ? rn_to_mont(rn_powmod((uint32_t)RN_P[0], p - 2, p), p) : 0;
    const uint32_t ninv = rn_powmod(1024, p - 2, p);
    h.pk_scale[k] = rn_to_mont((uint32_t)rn_to_mont(ninv, p), p);
  }
}

}  // namespace spp

// Exact negacyclic products for the RLWE witness kernel through a 1024-point NTT held in LDS.
//
// Replaces the O(n^2) loops of the reference -- negacyclic_mul_mod_q (scripts/generate_audit.py:45-54),
// negacyclic_matrix_row_mod_q (:57-66) and the quotient loops (:539-554) -- by  NTT(r) . NTT(pk)  in TWO prime fields:
//   P0 = q = 167 772 161 = 5 * 2^25 + 1   (the RLWE modulus itself: the residue IS the ciphertext coefficient)
//   P1 =     469 762 049 = 7 * 2^26 + 1
// and a CRT lift.  The integer  S_i = <row_i(pk), r>  satisfies |S_i| <= 1024 * (q-1) * 128 < 2^45 << P0*P1/2 ~ 2^55.1, so
// the lift is exact and the signed quotient  k = floor((S + e + Delta*m) / q)  (generate_audit.py:236-243) follows from the
// CRT digit without any wide division:  S = s0 + q*t  (t centred mod P1)  =>  k = t + floor((s0 + e + Delta*m) / q).
//
// One wavefront per polynomial.  Lane t holds the 16 coefficients  t + 64*j ; the transform is 1024 = 16 x 16 x 4:
//   pass 1: 16-point DFT over the stride-64 coefficients in registers, twiddle w^(t*k1), store to LDS;
//   pass 2: 16-point DFT over n2, twiddle w^(16*n3*k2), store to LDS;
//   pass 3: four 4-point DFTs; the results land in the SAME lane layout in natural order (no bit reversal anywhere),
// so forward and inverse are one routine with w or w^-1, and pointwise products / global loads are coalesced.
// Arithmetic: 32-bit Montgomery with lazy reduction (values in [0, 2p); p < 2^30 so 4p fits a word): a butterfly is
// 3 (add) + 2 (sub) + 3 (multiply) instructions.  Data stay in the plain domain, constants carry the factor 2^32.
// The per-lane phases are plain functions of (lane, registers, LDS array) so that tests/host/rlwe_ntt_check.cpp runs the
// identical code on the host, lane by lane, against the schoolbook definition.
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
static constexpr uint32_t RN_P[2] = {167772161u, 469762049u};
static constexpr uint32_t RN_S1 = 68, RN_S2 = 264;             // padded LDS strides of the two exchange layouts
static constexpr uint32_t RN_LDS_WORDS = 16 * RN_S1;           // 1088 words per field (>= 4 * RN_S2 = 1056)

// per-field constants and tables, built on the host (rn_build_tables) and resident in HBM / L2
struct RnField {
  uint32_t p, two_p, pinv_neg;     // -p^-1 mod 2^32
  uint32_t w16[2][8];              // (w^64)^e * 2^32 mod p, e < 8, for direction 0 (forward) / 1 (inverse)
  uint32_t crt;                    // field 1 only: q^-1 * 2^32 mod P1
};
struct RnTables {
  RnField f[2];
  // all entries carry the Montgomery factor 2^32:
  const uint32_t* w[2][2];         // [field][direction][1024]: w^e  /  w^-e
  const uint32_t* psi[2];          // [field][1024]: psi^j                       (twist before the forward transform)
  const uint32_t* ipsi[2];         // [field][1024]: psi^-j                      (untwist; 1/1024 is folded into the pk transform)
};

RN_HD uint32_t rn_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
// a < 4p, b < p  ->  a*b*2^-32 mod p in [0, 2p)
RN_HD uint32_t rn_mul(uint32_t a, uint32_t b, const RnField& f) {
  const uint64_t t = (uint64_t)a * b;
  const uint32_t m = (uint32_t)t * f.pinv_neg;
  return (uint32_t)((t + (uint64_t)m * f.p) >> 32);
}
RN_HD uint32_t rn_red2p(uint32_t s, const RnField& f) { return rn_min(s, s - f.two_p); }      // [0,4p) -> [0,2p)
RN_HD uint32_t rn_add(uint32_t a, uint32_t b, const RnField& f) { return rn_red2p(a + b, f); }
RN_HD uint32_t rn_subraw(uint32_t a, uint32_t b, const RnField& f) { return a - b + f.two_p; }  // (0,4p)
RN_HD uint32_t rn_canon(uint32_t a, const RnField& f) {                                         // [0,2p) -> [0,p)
  return rn_min(a, a - f.p);
}

// in-register 16-point DFT, natural order in and out, twiddles w16[e] = (w^64)^e
RN_HD void rn_dft16(uint32_t (&x)[16], const uint32_t (&w16)[8], const RnField& f) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t a = x[i], b = x[i + 8];
    x[i] = rn_add(a, b, f);
    const uint32_t d = rn_subraw(a, b, f);
    x[i + 8] = i == 0 ? rn_red2p(d, f) : rn_mul(d, w16[i], f);
  }
#pragma unroll
  for (int blk = 0; blk < 16; blk += 8) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t a = x[blk + i], b = x[blk + i + 4];
      x[blk + i] = rn_add(a, b, f);
      const uint32_t d = rn_subraw(a, b, f);
      x[blk + i + 4] = i == 0 ? rn_red2p(d, f) : rn_mul(d, w16[2 * i], f);
    }
  }
#pragma unroll
  for (int blk = 0; blk < 16; blk += 4) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const uint32_t a = x[blk + i], b = x[blk + i + 2];
      x[blk + i] = rn_add(a, b, f);
      const uint32_t d = rn_subraw(a, b, f);
      x[blk + i + 2] = i == 0 ? rn_red2p(d, f) : rn_mul(d, w16[4], f);
    }
  }
#pragma unroll
  for (int blk = 0; blk < 16; blk += 2) {
    const uint32_t a = x[blk], b = x[blk + 1];
    x[blk] = rn_add(a, b, f);
    x[blk + 1] = rn_red2p(rn_subraw(a, b, f), f);
  }
  // bit-reversed -> natural (register renaming: every index is a compile-time constant)
  uint32_t y[16];
#pragma unroll
  for (int k = 0; k < 16; k++) y[k] = x[((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3)];
#pragma unroll
  for (int k = 0; k < 16; k++) x[k] = y[k];
}

// ---- the three passes of one 1024-point transform; x = the lane's 16 values (coefficient lane + 64*j), in place ----
// phase A (lane t): pass 1 + twiddle, write exchange layout 1
RN_HD void rn_pass1(uint32_t lane, uint32_t (&x)[16], uint32_t* lds, const RnField& f, const uint32_t* w, int dir) {
  rn_dft16(x, f.w16[dir], f);
  lds[lane] = x[0];
#pragma unroll
  for (uint32_t k1 = 1; k1 < 16; k1++) lds[k1 * RN_S1 + lane] = rn_mul(x[k1], w[(lane * k1) & 1023], f);
}
// phase B (lane u = 4*k1 + n3), first half: read layout 1   -- barrier before AND after (the writes of B2 reuse the buffer)
RN_HD void rn_pass2_read(uint32_t lane, uint32_t (&x)[16], const uint32_t* lds) {
  const uint32_t k1 = lane >> 2, n3 = lane & 3;
#pragma unroll
  for (uint32_t n2 = 0; n2 < 16; n2++) x[n2] = lds[k1 * RN_S1 + 4 * n2 + n3];
}
// phase B, second half: pass 2 + twiddle, write exchange layout 2
RN_HD void rn_pass2(uint32_t lane, uint32_t (&x)[16], uint32_t* lds, const RnField& f, const uint32_t* w, int dir) {
  const uint32_t k1 = lane >> 2, n3 = lane & 3;
  rn_dft16(x, f.w16[dir], f);
#pragma unroll
  for (uint32_t k2 = 0; k2 < 16; k2++) {
    const uint32_t v = (k2 == 0) ? x[0] : rn_mul(x[k2], w[(16 * n3 * k2) & 1023], f);   // n3 == 0: w^0 = 2^32 mod p, a plain multiply by one
    lds[n3 * RN_S2 + k1 + 16 * k2] = v;
  }
}
// phase C (lane v): four 4-point DFTs; x[i + 4*k3] = X[v + 64*(i + 4*k3)]
RN_HD void rn_pass3(uint32_t lane, uint32_t (&x)[16], const uint32_t* lds, const RnField& f, int dir) {
  const uint32_t w4 = f.w16[dir][4];
#pragma unroll
  for (uint32_t i = 0; i < 4; i++) {
    const uint32_t m = lane + 64 * i;
    const uint32_t y0 = lds[m], y1 = lds[RN_S2 + m], y2 = lds[2 * RN_S2 + m], y3 = lds[3 * RN_S2 + m];
    const uint32_t s0 = rn_add(y0, y2, f), d0 = rn_red2p(rn_subraw(y0, y2, f), f);
    const uint32_t s1 = rn_add(y1, y3, f), d1 = rn_mul(rn_subraw(y1, y3, f), w4, f);
    x[i] = rn_add(s0, s1, f);
    x[i + 4] = rn_add(d0, d1, f);
    x[i + 8] = rn_red2p(rn_subraw(s0, s1, f), f);
    x[i + 12] = rn_red2p(rn_subraw(d0, d1, f), f);
  }
}

// CRT digit: residues s0 in [0,q), s1 in [0,P1) of the integer S, |S| < q*P1/2  ->  t with S = s0 + q*t (t signed)
RN_HD int32_t rn_crt_digit(uint32_t s0, uint32_t s1, const RnField& f1) {
  const uint32_t d = s1 + f1.p - s0;                        // (0, 2 P1): s0 < q < P1
  const uint32_t t = rn_canon(rn_mul(d, f1.crt, f1), f1);   // (s1 - s0) / q mod P1
  return t > (f1.p >> 1) ? (int32_t)t - (int32_t)f1.p : (int32_t)t;
}
// floor-division bookkeeping of compute_quotient_and_remainder (generate_audit.py:236-243): v = S + add with S = s0 + q*t
RN_HD void rn_quot_rem(uint32_t s0, int32_t t, int32_t add, int32_t& k, uint32_t& rem) {
  const int32_t q = (int32_t)RN_P[0];
  int32_t w = (int32_t)s0 + add;                            // in (-2^8, 2q): add in [-128, 127 + 655360*255]
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
// pre: 1024 + 64 ints of LDS.  Phases (barrier between each): scatter r -> chunk scan -> offsets -> gather.
RN_HD void rn_scan_scatter(uint32_t lane, const int32_t (&r)[16], int32_t* pre) {
#pragma unroll
  for (int j = 0; j < 16; j++) pre[lane + 64 * j] = r[j];
}
RN_HD void rn_scan_chunk(uint32_t lane, int32_t* pre) {
  int32_t acc = 0;
  for (int c = 0; c < 16; c++) {
    acc += pre[16 * lane + c];
    pre[16 * lane + c] = acc;          // inclusive prefix inside the chunk
  }
  pre[1024 + lane] = acc;
}
// returns (offset of this lane's chunk, total): call after the barrier that follows rn_scan_chunk
RN_HD void rn_scan_offsets(uint32_t lane, const int32_t* pre, int32_t& offset, int32_t& total) {
  int32_t off = 0, tot = 0;
  for (uint32_t l = 0; l < 64; l++) {
    const int32_t v = pre[1024 + l];
    tot += v;
    if (l < lane) off += v;
  }
  offset = off;
  total = tot;
}
RN_HD void rn_scan_apply(uint32_t lane, int32_t offset, int32_t* pre) {
  for (int c = 0; c < 16; c++) pre[16 * lane + c] += offset;
}
// suffix (exclusive) sums for this lane's coefficients i = lane + 64 j, after the barrier that follows rn_scan_apply
RN_HD void rn_scan_gather(uint32_t lane, int32_t total, const int32_t* pre, int32_t (&suffix)[16]) {
#pragma unroll
  for (int j = 0; j < 16; j++) suffix[j] = total - pre[lane + 64 * j];
}
// zero-list correction for coefficient i: subtract r[1024 + i - z] for every zero position z > i (rbytes: r in natural order)
RN_HD int32_t rn_zero_correction(uint32_t i, const uint16_t* zeros, uint32_t nzeros, const int8_t* rbytes) {
  int32_t c = 0;
  for (uint32_t k = 0; k < nzeros; k++) {
    const uint32_t z = zeros[k];
    if (z > i) c += rbytes[1024 + i - z];
  }
  return c;
}

// ---- host: constant tables ----
inline uint32_t rn_powmod(uint32_t b, uint64_t e, uint32_t p) {
  uint64_t r = 1, x = b;
  while (e) {
    if (e & 1) r = r * x % p;
    x = x * x % p;
    e >>= 1;
  }
  return (uint32_t)r;
}
inline uint32_t rn_to_mont(uint32_t v, uint32_t p) { return (uint32_t)((((uint64_t)v) << 32) % p); }
// host arrays: w[f][dir][1024], psi[f][1024], ipsi[f][1024] (Montgomery form); fills the scalar members of tb.f[]
struct RnHostTables {
  uint32_t w[2][2][1024], psi[2][1024], ipsi[2][1024];
  RnField f[2];
  uint32_t pk_scale[2];   // (1/1024) * 2^64 mod p: rn_mul(NTT(pk), pk_scale) = NTT(pk)/1024 in Montgomery form
};
inline void rn_build_tables(RnHostTables& h) {
  for (int k = 0; k < 2; k++) {
    const uint32_t p = RN_P[k];
    RnField& f = h.f[k];
    f.p = p;
    f.two_p = 2 * p;
    uint32_t inv = 1;                                   // Newton: p^-1 mod 2^32
    for (int i = 0; i < 5; i++) inv *= 2 - p * inv;
    f.pinv_neg = 0u - inv;
    const uint32_t psi = rn_powmod(3, (p - 1) / 2048, p), ipsi = rn_powmod(psi, p - 2, p);
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
    f.crt = k == 1 ? rn_to_mont(rn_powmod(RN_P[0], p - 2, p), p) : 0;
    const uint32_t ninv = rn_powmod(1024, p - 2, p);
    h.pk_scale[k] = rn_to_mont(rn_to_mont(ninv, p), p);
  }
}

}  // namespace spp

// SHA-256 and RFC 9380 expand_message_xmd, host + device (the BSB22 commitment challenge is hashed on the
// GPU so that a batch never round-trips to the host between the two solver phases).
// Replaces gnark-crypto fr.Hash as used by groth16 Prove/Verify with DST "bsb22-commitment" (the DST
// strings are visible in the reference's audit_circuit/target/audit_verifier.so).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "bn254.hpp"

namespace spp {

struct Sha256 {
  uint32_t h[8];
  SPP_HD void init() {
    h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au;
    h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
  }
  static SPP_HD uint32_t K(int i) {
    constexpr uint32_t k[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u,
        0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu,
        0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u,
        0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u,
        0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
        0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    return k[i];
  }
  static SPP_HD uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  // one 64-byte block given as 16 big-endian words
  SPP_HD void compress(const uint32_t blk[16]) {
    uint32_t w[16];
    for (int i = 0; i < 16; i++) w[i] = blk[i];
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    SPP_UNROLL for (int i = 0; i < 64; i++) {
      uint32_t wi;
      if (i < 16) {
        wi = w[i];
      } else {
        uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
        uint32_t s0 = ror(w15, 7) ^ ror(w15, 18) ^ (w15 >> 3);
        uint32_t s1 = ror(w2, 17) ^ ror(w2, 19) ^ (w2 >> 10);
        wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        w[i & 15] = wi;
      }
      uint32_t S1 = ror(e, 6) ^ ror(e, 11) ^ ror(e, 25);
      uint32_t ch = (e & f) ^ (~e & g);
      uint32_t t1 = hh + S1 + ch + K(i) + wi;
      uint32_t S0 = ror(a, 2) ^ ror(a, 13) ^ ror(a, 22);
      uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
      uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
};

// generic byte-oriented hashing (host side: setup seed expansion)
inline void sha256_bytes(const uint8_t* msg, size_t len, uint8_t out[32]) {
  Sha256 s;
  s.init();
  uint32_t blk[16];
  size_t full = len / 64;
  for (size_t b = 0; b < full; b++) {
    for (int i = 0; i < 16; i++) {
      const uint8_t* q = msg + 64 * b + 4 * i;
      blk[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    }
    s.compress(blk);
  }
  uint8_t tail[128];
  for (int i = 0; i < 128; i++) tail[i] = 0;
  size_t rem = len - 64 * full;
  for (size_t i = 0; i < rem; i++) tail[i] = msg[64 * full + i];
  tail[rem] = 0x80;
  size_t tl = rem < 56 ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
  for (size_t b = 0; b < tl / 64; b++) {
    for (int i = 0; i < 16; i++) {
      const uint8_t* q = tail + 64 * b + 4 * i;
      blk[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    }
    s.compress(blk);
  }
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(s.h[i] >> 24);
    out[4 * i + 1] = (uint8_t)(s.h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(s.h[i] >> 8);
    out[4 * i + 3] = (uint8_t)s.h[i];
  }
}

// RFC 9380 expand_message_xmd (SHA-256), host side
inline void expand_message_xmd(const uint8_t* msg, size_t mlen, const uint8_t* dst, size_t dlen, uint8_t* out, size_t outlen) {
  size_t ell = (outlen + 31) / 32;
  size_t plen = 64 + mlen + 3 + dlen + 1;
  uint8_t* buf = new uint8_t[plen];
  for (size_t i = 0; i < plen; i++) buf[i] = 0;
  for (size_t i = 0; i < mlen; i++) buf[64 + i] = msg[i];
  buf[64 + mlen] = (uint8_t)(outlen >> 8);
  buf[64 + mlen + 1] = (uint8_t)outlen;
  for (size_t i = 0; i < dlen; i++) buf[64 + mlen + 3 + i] = dst[i];
  buf[plen - 1] = (uint8_t)dlen;
  uint8_t b0[32], bi[32], blk[32 + 1 + 256];
  sha256_bytes(buf, plen, b0);
  delete[] buf;
  size_t off = 0;
  for (size_t i = 1; i <= ell; i++) {
    for (int j = 0; j < 32; j++) blk[j] = (i == 1) ? b0[j] : (uint8_t)(b0[j] ^ bi[j]);
    blk[32] = (uint8_t)i;
    for (size_t j = 0; j < dlen; j++) blk[33 + j] = dst[j];
    blk[33 + dlen] = (uint8_t)dlen;
    sha256_bytes(blk, 34 + dlen, bi);
    size_t take = outlen - off < 32 ? outlen - off : 32;
    for (size_t j = 0; j < take; j++) out[off + j] = bi[j];
    off += take;
  }
}

// 48 big-endian bytes (as 12 big-endian words, most significant first) -> Fr (value mod r)
SPP_HD Fr fr_from_wide48(const uint32_t w[12]) {
  uint32_t hi[8], lo[8];
  // hi = top 16 bytes = words 0..3 ; lo = words 4..11
  for (int i = 0; i < 8; i++) hi[i] = 0;
  hi[3] = w[0]; hi[2] = w[1]; hi[1] = w[2]; hi[0] = w[3];
  for (int i = 0; i < 8; i++) lo[i] = w[11 - i];
  Fr l = Fr::from_u256(lo);
  Fr hm = Fr::from_canonical(hi) * Fr::r2();  // hi * 2^256
  return l + hm;
}

// fr.Hash(msg, DST) for a 64-byte message and a 16-byte domain-separation tag: expand_message_xmd(SHA-256), 48 output bytes -> Fr.
// Blocks are laid out by hand (message 16 big-endian words, DST 16 B).  Two tags are in use, one per protocol value, so that the
// two never come out of the same random-oracle domain:
//   DstCommitment "bsb22-commitment"  the commitment challenge fr.Hash(Cm.x || Cm.y) (gnark's tag; the string is in the reference's
//                                     audit_circuit/target/audit_verifier.so)
//   DstMask       "spp-commit-mask1"  the commitment's hiding mask, derived from the proof's blinding factors r || s (OP_MASK)
struct DstCommitment { static SPP_HD constexpr uint32_t b(int i) { constexpr char d[17] = "bsb22-commitment"; return (uint32_t)(uint8_t)d[i]; } };
struct DstMask { static SPP_HD constexpr uint32_t b(int i) { constexpr char d[17] = "spp-commit-mask1"; return (uint32_t)(uint8_t)d[i]; } };
template <class D>
SPP_HD Fr hash64_to_fr(const uint32_t m[16]) {
  constexpr uint32_t w0 = 0x00300000u | D::b(0);
  constexpr uint32_t w1 = (D::b(1) << 24) | (D::b(2) << 16) | (D::b(3) << 8) | D::b(4);
  constexpr uint32_t w2 = (D::b(5) << 24) | (D::b(6) << 16) | (D::b(7) << 8) | D::b(8);
  constexpr uint32_t w3 = (D::b(9) << 24) | (D::b(10) << 16) | (D::b(11) << 8) | D::b(12);
  constexpr uint32_t w4 = (D::b(13) << 24) | (D::b(14) << 16) | (D::b(15) << 8) | 0x10u;
  constexpr uint32_t x0 = (D::b(0) << 16) | (D::b(1) << 8) | D::b(2);            // low 24 bits of the word that starts with the block counter
  constexpr uint32_t x1 = (D::b(3) << 24) | (D::b(4) << 16) | (D::b(5) << 8) | D::b(6);
  constexpr uint32_t x2 = (D::b(7) << 24) | (D::b(8) << 16) | (D::b(9) << 8) | D::b(10);
  constexpr uint32_t x3 = (D::b(11) << 24) | (D::b(12) << 16) | (D::b(13) << 8) | D::b(14);
  constexpr uint32_t x4 = (D::b(15) << 24) | 0x00108000u;
  uint32_t blk[16];
  // b0 = H(Z_pad || msg || I2OSP(48,2) || 0 || DST || len(DST))
  Sha256 s0;
  s0.init();
  for (int i = 0; i < 16; i++) blk[i] = 0;
  s0.compress(blk);
  s0.compress(m);
  blk[0] = w0; blk[1] = w1; blk[2] = w2; blk[3] = w3; blk[4] = w4; blk[5] = 0x80000000u;
  for (int i = 6; i < 15; i++) blk[i] = 0;
  blk[15] = 0x4a0u;
  s0.compress(blk);
  // b1 = H(b0 || 1 || DST')
  Sha256 s1;
  s1.init();
  for (int i = 0; i < 8; i++) blk[i] = s0.h[i];
  blk[8] = 0x01000000u | x0; blk[9] = x1; blk[10] = x2; blk[11] = x3; blk[12] = x4;
  blk[13] = 0; blk[14] = 0; blk[15] = 0x190u;
  s1.compress(blk);
  // b2 = H((b0 ^ b1) || 2 || DST')
  Sha256 s2;
  s2.init();
  for (int i = 0; i < 8; i++) blk[i] = s0.h[i] ^ s1.h[i];
  blk[8] = 0x02000000u | x0;
  s2.compress(blk);
  uint32_t wide[12];
  for (int i = 0; i < 8; i++) wide[i] = s1.h[i];
  for (int i = 0; i < 4; i++) wide[8 + i] = s2.h[i];
  return fr_from_wide48(wide);
}
SPP_HD Fr bsb22_challenge(const uint32_t m[16]) { return hash64_to_fr<DstCommitment>(m); }
SPP_HD Fr commitment_mask(const uint32_t m[16]) { return hash64_to_fr<DstMask>(m); }

}  // namespace spp

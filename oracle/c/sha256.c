/* ORACLE (test infrastructure only) */
#include "sha256.h"
#include <stdlib.h>
#include <string.h>
static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void compress(uint32_t h[8], const uint8_t blk[64]) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = hh + S1 + ch + K[i] + w[i];
    uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
void orc_sha256(const uint8_t* msg, size_t len, uint8_t out[32]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t full = len / 64;
  for (size_t i = 0; i < full; i++) compress(h, msg + 64 * i);
  uint8_t tail[128] = {0};
  size_t rem = len - 64 * full;
  memcpy(tail, msg + 64 * full, rem);
  tail[rem] = 0x80;
  size_t tl = (rem < 56) ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
  compress(h, tail);
  if (tl == 128) compress(h, tail + 64);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
  }
}
void orc_expand_xmd(const uint8_t* msg, size_t mlen, const uint8_t* dst, size_t dlen, uint8_t* out, size_t outlen) {
  size_t ell = (outlen + 31) / 32;
  size_t plen = 64 + mlen + 2 + 1 + dlen + 1;
  uint8_t* buf = (uint8_t*)calloc(plen, 1);
  memcpy(buf + 64, msg, mlen);
  buf[64 + mlen] = (uint8_t)(outlen >> 8);
  buf[64 + mlen + 1] = (uint8_t)outlen;
  buf[64 + mlen + 2] = 0;
  memcpy(buf + 64 + mlen + 3, dst, dlen);
  buf[plen - 1] = (uint8_t)dlen;
  uint8_t b0[32], bi[32];
  orc_sha256(buf, plen, b0);
  free(buf);
  uint8_t blk[32 + 1 + 256];
  size_t off = 0;
  for (size_t i = 1; i <= ell; i++) {
    for (int j = 0; j < 32; j++) blk[j] = (i == 1) ? b0[j] : (uint8_t)(b0[j] ^ bi[j]);
    blk[32] = (uint8_t)i;
    memcpy(blk + 33, dst, dlen);
    blk[33 + dlen] = (uint8_t)dlen;
    orc_sha256(blk, 34 + dlen, bi);
    size_t take = outlen - off < 32 ? outlen - off : 32;
    memcpy(out + off, bi, take);
    off += take;
  }
}
void orc_hash_to_fr(const uint8_t* msg, size_t mlen, const char* dst, fe* out, int count) {
  uint8_t* u = (uint8_t*)malloc((size_t)count * 48);
  orc_expand_xmd(msg, mlen, (const uint8_t*)dst, strlen(dst), u, (size_t)count * 48);
  for (int i = 0; i < count; i++) fe_from_wide_be(&out[i], u + 48 * i, 48, &FR);
  free(u);
}

/* ORACLE (test infrastructure only) -- see field.h */
#include "field.h"
#include <stdlib.h>

field_t FR, FQ;
g1a G1_GEN;
g2a G2_GEN;

static void field_setup(field_t* F, const uint64_t p[4]) {
  memcpy(F->p, p, 32);
  /* inv = -p^-1 mod 2^64 (Newton) */
  uint64_t x = 1;
  for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;
  F->inv = (uint64_t)0 - x;
  /* one = 2^256 mod p by 256 doublings of 1 (plain modular) */
  uint64_t t[4] = {1, 0, 0, 0};
  fe r1, r2, r3;
  for (int k = 0; k < 768; k++) {
    uint64_t c = 0, d[4];
    for (int i = 0; i < 4; i++) {
      d[i] = (t[i] << 1) | c;
      c = t[i] >> 63;
    }
    memcpy(t, d, 32);
    if (c || raw_geq(t, p)) raw_sub(t, t, p);
    if (k == 255) memcpy(r1.l, t, 32);
    if (k == 511) memcpy(r2.l, t, 32);
    if (k == 767) memcpy(r3.l, t, 32);
  }
  F->one = r1; F->r2 = r2; F->r3 = r3;
  uint64_t two[4] = {2, 0, 0, 0};
  raw_sub(F->pm2, p, two);
}

void fe_pow(fe* r, const fe* a, const uint64_t e[4], const field_t* F) {
  fe acc = F->one, base = *a;
  for (int i = 0; i < 256; i++) {
    if ((e[i / 64] >> (i % 64)) & 1) fe_mul(&acc, &acc, &base, F);
    fe_sqr(&base, &base, F);
  }
  *r = acc;
}
void fe_inv(fe* r, const fe* a, const field_t* F) { fe_pow(r, a, F->pm2, F); }
void fe_from_raw(fe* r, const uint64_t v[4], const field_t* F) {
  fe t;
  memcpy(t.l, v, 32);
  while (raw_geq(t.l, F->p)) raw_sub(t.l, t.l, F->p);
  fe_mul(r, &t, &F->r2, F);
}
void fe_from_u64(fe* r, uint64_t v, const field_t* F) {
  uint64_t t[4] = {v, 0, 0, 0};
  fe_from_raw(r, t, F);
}
void fe_to_raw(uint64_t v[4], const fe* a, const field_t* F) {
  fe one = {{1, 0, 0, 0}}, t;
  fe_mul(&t, a, &one, F);
  memcpy(v, t.l, 32);
}
void fe_from_be(fe* r, const uint8_t b[32], const field_t* F) {
  uint64_t v[4];
  for (int i = 0; i < 4; i++) {
    uint64_t w = 0;
    for (int j = 0; j < 8; j++) w = (w << 8) | b[8 * i + j];
    v[3 - i] = w;
  }
  fe_from_raw(r, v, F);
}
void fe_to_be(uint8_t b[32], const fe* a, const field_t* F) {
  uint64_t v[4];
  fe_to_raw(v, a, F);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(v[3 - i] >> (56 - 8 * j));
}
void fe_from_wide_be(fe* r, const uint8_t* b, int len, const field_t* F) {
  /* value = hi * 2^256 + lo, len <= 48 */
  uint8_t buf[64] = {0};
  memcpy(buf + 64 - len, b, (size_t)len);
  fe hi, lo, t;
  uint64_t v[4];
  for (int i = 0; i < 4; i++) {
    uint64_t w = 0;
    for (int j = 0; j < 8; j++) w = (w << 8) | buf[8 * i + j];
    v[3 - i] = w;
  }
  memcpy(hi.l, v, 32);            /* hi < 2^128 < p, raw */
  fe_from_be(&lo, buf + 32, F);   /* lo in Montgomery form */
  fe_mul(&t, &hi, &F->r3, F);     /* hi * 2^512/2^256... = hi*R^2 -> Montgomery form of hi*2^256 */
  fe_add(r, &t, &lo, F);
}

/* ---- Fq2 ---- */
void fe2_add(fe2* r, const fe2* a, const fe2* b) { fe_add(&r->c0, &a->c0, &b->c0, &FQ); fe_add(&r->c1, &a->c1, &b->c1, &FQ); }
void fe2_sub(fe2* r, const fe2* a, const fe2* b) { fe_sub(&r->c0, &a->c0, &b->c0, &FQ); fe_sub(&r->c1, &a->c1, &b->c1, &FQ); }
void fe2_neg(fe2* r, const fe2* a) { fe_neg(&r->c0, &a->c0, &FQ); fe_neg(&r->c1, &a->c1, &FQ); }
void fe2_mul(fe2* r, const fe2* a, const fe2* b) {
  fe t0, t1, t2, t3;
  fe_mul(&t0, &a->c0, &b->c0, &FQ);
  fe_mul(&t1, &a->c1, &b->c1, &FQ);
  fe_mul(&t2, &a->c0, &b->c1, &FQ);
  fe_mul(&t3, &a->c1, &b->c0, &FQ);
  fe_sub(&r->c0, &t0, &t1, &FQ);
  fe_add(&r->c1, &t2, &t3, &FQ);
}
void fe2_sqr(fe2* r, const fe2* a) { fe2_mul(r, a, a); }
void fe2_inv(fe2* r, const fe2* a) {
  fe n, t, d;
  fe_sqr(&n, &a->c0, &FQ);
  fe_sqr(&t, &a->c1, &FQ);
  fe_add(&n, &n, &t, &FQ);
  fe_inv(&d, &n, &FQ);
  fe_mul(&r->c0, &a->c0, &d, &FQ);
  fe_mul(&t, &a->c1, &d, &FQ);
  fe_neg(&r->c1, &t, &FQ);
}
static void fe2_one(fe2* r) { r->c0 = FQ.one; memset(&r->c1, 0, sizeof(fe)); }
static void feq_one(fe* r) { *r = FQ.one; }

/* ---- group law instantiations ---- */
#define FT fe
#define AT g1a
#define JT g1j
#define PFX(n) g1##n
#define FADD(r, a, b) fe_add(r, a, b, &FQ)
#define FSUB(r, a, b) fe_sub(r, a, b, &FQ)
#define FMUL(r, a, b) fe_mul(r, a, b, &FQ)
#define FSQR(r, a) fe_sqr(r, a, &FQ)
#define FINV(r, a) fe_inv(r, a, &FQ)
#define FISZERO(a) fe_is_zero(a)
#define FONE(r) feq_one(r)
#include "curve_tmpl.h"
#undef FT
#undef AT
#undef JT
#undef PFX
#undef FADD
#undef FSUB
#undef FMUL
#undef FSQR
#undef FINV
#undef FISZERO
#undef FONE

#define FT fe2
#define AT g2a
#define JT g2j
#define PFX(n) g2##n
#define FADD(r, a, b) fe2_add(r, a, b)
#define FSUB(r, a, b) fe2_sub(r, a, b)
#define FMUL(r, a, b) fe2_mul(r, a, b)
#define FSQR(r, a) fe2_sqr(r, a)
#define FINV(r, a) fe2_inv(r, a)
#define FISZERO(a) fe2_is_zero(a)
#define FONE(r) fe2_one(r)
#include "curve_tmpl.h"

void g1a_neg(g1a* r, const g1a* p) { *r = *p; if (!p->inf) fe_neg(&r->y, &p->y, &FQ); }

void g1a_to_bytes(uint8_t out[64], const g1a* p) {
  if (p->inf) { memset(out, 0, 64); return; }
  fe_to_be(out, &p->x, &FQ);
  fe_to_be(out + 32, &p->y, &FQ);
}
void g1a_from_bytes(g1a* p, const uint8_t in[64]) {
  int z = 1;
  for (int i = 0; i < 64; i++) if (in[i]) z = 0;
  memset(p, 0, sizeof *p);
  if (z) { p->inf = 1; return; }
  fe_from_be(&p->x, in, &FQ);
  fe_from_be(&p->y, in + 32, &FQ);
}
void g2a_to_bytes(uint8_t out[128], const g2a* p) {
  if (p->inf) { memset(out, 0, 128); return; }
  fe_to_be(out, &p->x.c1, &FQ);
  fe_to_be(out + 32, &p->x.c0, &FQ);
  fe_to_be(out + 64, &p->y.c1, &FQ);
  fe_to_be(out + 96, &p->y.c0, &FQ);
}
void g2a_from_bytes(g2a* p, const uint8_t in[128]) {
  int z = 1;
  for (int i = 0; i < 128; i++) if (in[i]) z = 0;
  memset(p, 0, sizeof *p);
  if (z) { p->inf = 1; return; }
  fe_from_be(&p->x.c1, in, &FQ);
  fe_from_be(&p->x.c0, in + 32, &FQ);
  fe_from_be(&p->y.c1, in + 64, &FQ);
  fe_from_be(&p->y.c0, in + 96, &FQ);
}

static void dec_to_fe(fe* r, const char* dec, const field_t* F) {
  fe acc, ten, d;
  memset(&acc, 0, sizeof acc);
  fe_from_u64(&ten, 10, F);
  for (const char* c = dec; *c; c++) {
    fe_mul(&acc, &acc, &ten, F);
    fe_from_u64(&d, (uint64_t)(*c - '0'), F);
    fe_add(&acc, &acc, &d, F);
  }
  *r = acc;
}

void orc_fields_init(void) {
  static int done = 0;
  if (done) return;
  const uint64_t r[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  const uint64_t q[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  field_setup(&FR, r);
  field_setup(&FQ, q);
  memset(&G1_GEN, 0, sizeof G1_GEN);
  fe_from_u64(&G1_GEN.x, 1, &FQ);
  fe_from_u64(&G1_GEN.y, 2, &FQ);
  memset(&G2_GEN, 0, sizeof G2_GEN);
  dec_to_fe(&G2_GEN.x.c0, "10857046999023057135944570762232829481370756359578518086990519993285655852781", &FQ);
  dec_to_fe(&G2_GEN.x.c1, "11559732032986387107991004021392285783925812861821192530917403151452391805634", &FQ);
  dec_to_fe(&G2_GEN.y.c0, "8495653923123431417604973247489272438418190587263600148770280649306958101930", &FQ);
  dec_to_fe(&G2_GEN.y.c1, "4082367875863433681332203403145435568316851327593401208105741076214120093531", &FQ);
  done = 1;
}

/* ORACLE (test infrastructure only) -- BN254 Fr/Fq on 4 x 64-bit Montgomery limbs (unsigned __int128),
 * Fq2, G1/G2 in Jacobian coordinates. Plain C restatement of the arithmetic that the reference's
 * external prover (`sunspot prove`, client/proof.helper.ts:64; gnark 0.14.0 groth16/bn254, not present
 * under /root/reference) performs on the CPU. Deliberately different from the product's device code
 * (csrc/bn254.hpp: 8 x 32-bit limbs, XYZZ coordinates): two independent implementations must agree.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg link this. */
#ifndef ORC_FIELD_H
#define ORC_FIELD_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;            /* Montgomery form, < p */
typedef struct {
  uint64_t p[4];      /* modulus */
  uint64_t inv;       /* -p^-1 mod 2^64 */
  fe one, r2, r3;
  uint64_t pm2[4];    /* p-2 */
} field_t;

extern field_t FR, FQ;
void orc_fields_init(void);

static inline int fe_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe* a, const fe* b) {
  return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int raw_geq(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
static inline void raw_sub(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - b[i] - br;
    r[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
static inline void fe_add(fe* r, const fe* a, const fe* b, const field_t* F) {
  u128 c = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; i++) {
    c += (u128)a->l[i] + b->l[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  if (raw_geq(t, F->p)) raw_sub(r->l, t, F->p); else memcpy(r->l, t, 32);
}
static inline void fe_sub(fe* r, const fe* a, const fe* b, const field_t* F) {
  u128 br = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->l[i] - b->l[i] - br;
    t[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)t[i] + F->p[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  memcpy(r->l, t, 32);
}
static inline void fe_neg(fe* r, const fe* a, const field_t* F) {
  if (fe_is_zero(a)) { *r = *a; return; }
  raw_sub(r->l, F->p, a->l);
}
static inline void fe_dbl(fe* r, const fe* a, const field_t* F) { fe_add(r, a, a, F); }

/* Montgomery product: operand scanning with separate reduction (SOS). */
static inline void fe_mul(fe* r, const fe* a, const fe* b, const field_t* F) {
  uint64_t t[9] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a->l[j] * b->l[i] + t[i + j];
      t[i + j] = (uint64_t)c;
      c >>= 64;
    }
    t[i + 4] = (uint64_t)c;
  }
  uint64_t extra = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t m = t[i] * F->inv;
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)m * F->p[j] + t[i + j];
      t[i + j] = (uint64_t)c;
      c >>= 64;
    }
    for (int j = i + 4; j < 8 && c; j++) {
      c += t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    extra += (uint64_t)c;
  }
  (void)extra; /* p < 2^254 and inputs < p: result < 2p, no overflow past t[7] */
  if (raw_geq(t + 4, F->p)) raw_sub(r->l, t + 4, F->p); else memcpy(r->l, t + 4, 32);
}
static inline void fe_sqr(fe* r, const fe* a, const field_t* F) { fe_mul(r, a, a, F); }

void fe_inv(fe* r, const fe* a, const field_t* F);
void fe_pow(fe* r, const fe* a, const uint64_t e[4], const field_t* F);
void fe_from_u64(fe* r, uint64_t v, const field_t* F);
void fe_from_raw(fe* r, const uint64_t v[4], const field_t* F);      /* v arbitrary 256-bit, reduced */
void fe_to_raw(uint64_t v[4], const fe* a, const field_t* F);        /* canonical little-endian limbs */
void fe_from_be(fe* r, const uint8_t b[32], const field_t* F);
void fe_to_be(uint8_t b[32], const fe* a, const field_t* F);
void fe_from_wide_be(fe* r, const uint8_t* b, int len, const field_t* F);   /* len <= 48 bytes, big-endian */

/* Fq2 */
typedef struct { fe c0, c1; } fe2;
void fe2_add(fe2* r, const fe2* a, const fe2* b);
void fe2_sub(fe2* r, const fe2* a, const fe2* b);
void fe2_neg(fe2* r, const fe2* a);
void fe2_mul(fe2* r, const fe2* a, const fe2* b);
void fe2_sqr(fe2* r, const fe2* a);
void fe2_inv(fe2* r, const fe2* a);
static inline int fe2_is_zero(const fe2* a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }

/* G1: affine (inf flag) and Jacobian */
typedef struct { fe x, y; int inf; } g1a;
typedef struct { fe X, Y, Z; } g1j;      /* Z == 0 <=> infinity */
void g1j_set_inf(g1j* r);
void g1j_from_affine(g1j* r, const g1a* p);
void g1j_dbl(g1j* r, const g1j* p);
void g1j_add_affine(g1j* r, const g1j* p, const g1a* q);
void g1j_add(g1j* r, const g1j* p, const g1j* q);
void g1j_to_affine(g1a* r, const g1j* p);
void g1a_neg(g1a* r, const g1a* p);
void g1_mul(g1j* r, const g1a* p, const uint64_t k[4]);
void g1a_to_bytes(uint8_t out[64], const g1a* p);
void g1a_from_bytes(g1a* p, const uint8_t in[64]);
void g1_batch_to_affine(g1a* out, const g1j* in, size_t n);

typedef struct { fe2 x, y; int inf; } g2a;
typedef struct { fe2 X, Y, Z; } g2j;
void g2j_set_inf(g2j* r);
void g2j_from_affine(g2j* r, const g2a* p);
void g2j_dbl(g2j* r, const g2j* p);
void g2j_add_affine(g2j* r, const g2j* p, const g2a* q);
void g2j_add(g2j* r, const g2j* p, const g2j* q);
void g2j_to_affine(g2a* r, const g2j* p);
void g2_mul(g2j* r, const g2a* p, const uint64_t k[4]);
void g2a_to_bytes(uint8_t out[128], const g2a* p);
void g2a_from_bytes(g2a* p, const uint8_t in[128]);
void g2_batch_to_affine(g2a* out, const g2j* in, size_t n);

extern g1a G1_GEN;
extern g2a G2_GEN;
#endif

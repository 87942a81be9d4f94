/* ORACLE (test infrastructure only) -- RLWE audit witness generation, plain C restatement of
 * scripts/generate_audit.py:45-66 (negacyclic product / matrix rows), :236-243 (quotient+remainder),
 * :507-554 (encrypt + quotient witnesses), :154-163 (pack_values). Exact 64-bit integer arithmetic:
 * |<row_i, r>| < 1024 * 2^28 * 3 < 2^40. Used as checker and as the CPU baseline for the RLWE kernel. */
#include <stdint.h>
#include <string.h>
#define RLWE_N 1024
#define RLWE_Q 167772161LL
#define RLWE_DELTA 655360LL
#define RLWE_SLOTS 64

/* floor division and non-negative remainder, as Python's // and % (generate_audit.py:241-242) */
static inline void floordiv(int64_t v, int64_t q, int64_t* k, int64_t* r) {
  int64_t rem = v % q;
  if (rem < 0) rem += q;
  *r = rem;
  *k = (v - rem) / q;
}

/* pk_a, pk_b: [1024] in [0,q); r, e2: [1024] signed; e1, msg: [64].
 * out: c0[64], c1[1024] in [0,q); k0[64], k1[1024] signed. */
void orc_rlwe_witness(const uint32_t* pk_a, const uint32_t* pk_b, const int32_t* r, const int32_t* e1, const int32_t* e2,
                      const uint32_t* msg, uint32_t* c0, uint32_t* c1, int64_t* k0, int64_t* k1) {
  for (int i = 0; i < RLWE_N; i++) {
    int64_t acc_a = 0, acc_b = 0;
    for (int j = 0; j < RLWE_N; j++) {
      int idx = i - j;
      /* row_i[j] = poly[i-j] mod q, or (-poly[i-j+n]) mod q  (generate_audit.py:57-66) */
      int64_t ca = idx >= 0 ? (int64_t)pk_a[idx] : (pk_a[idx + RLWE_N] ? RLWE_Q - (int64_t)pk_a[idx + RLWE_N] : 0);
      acc_a += ca * r[j];
      if (i < RLWE_SLOTS) {
        int64_t cb = idx >= 0 ? (int64_t)pk_b[idx] : (pk_b[idx + RLWE_N] ? RLWE_Q - (int64_t)pk_b[idx + RLWE_N] : 0);
        acc_b += cb * r[j];
      }
    }
    int64_t k, rem;
    floordiv(acc_a + e2[i], RLWE_Q, &k, &rem);
    c1[i] = (uint32_t)rem;
    k1[i] = k;
    if (i < RLWE_SLOTS) {
      floordiv(acc_b + e1[i] + RLWE_DELTA * (int64_t)msg[i], RLWE_Q, &k, &rem);
      c0[i] = (uint32_t)rem;
      k0[i] = k;
    }
  }
}

/* c = a*b mod (X^n+1, q), inputs in [0,q) (generate_audit.py:45-54) */
void orc_negacyclic_mul_mod_q(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  for (int k = 0; k < RLWE_N; k++) {
    unsigned __int128 pos = 0, neg = 0;
    for (int i = 0; i <= k; i++) pos += (uint64_t)a[i] * b[k - i];
    for (int i = k + 1; i < RLWE_N; i++) neg += (uint64_t)a[i] * b[k - i + RLWE_N];
    int64_t p = (int64_t)(pos % RLWE_Q), m = (int64_t)(neg % RLWE_Q);
    int64_t v = p - m;
    if (v < 0) v += RLWE_Q;
    out[k] = (uint32_t)v;
  }
}

/* pack 7 x 32-bit values per 32-byte big-endian field element (generate_audit.py:154-163) */
void orc_pack_values(const uint32_t* vals, int n, uint8_t* out_be) {
  int nf = (n + 6) / 7;
  memset(out_be, 0, (size_t)nf * 32);
  for (int i = 0; i < n; i++) {
    int f = i / 7, j = i % 7;
    uint8_t* o = out_be + 32 * f;
    /* value occupies bits [32j, 32j+32): big-endian bytes 31-4j-3 .. 31-4j */
    for (int b = 0; b < 4; b++) o[31 - 4 * j - b] = (uint8_t)(vals[i] >> (8 * b));
  }
}

/* ORACLE (test infrastructure only) -- CPU restatement of the Groth16(+BSB22 commitment) prover path.
 *
 * What it restates: the pipeline inside `sunspot prove` (reference call sites client/proof.helper.ts:64,
 * noir_circuit/prove_linux.sh:83, audit_circuit/prove_audit.sh:95, scripts/generate_audit.py:680), i.e.
 * gnark 0.14.0 groth16/bn254 Prove (third-party, pinned by the GnarkVersion string inside
 * noir_circuit/target/shielded_pool_verifier.ccs; source absent from /root/reference): witness solving,
 * Pedersen commitment + hash-to-field challenge, computeH with 7 radix-2 NTTs, 4 G1 MSMs + 1 G2 MSM,
 * proof assembly in gnark's raw layout (388 B, withdraw.rs:13). The R1CS itself is read from the SPPC
 * container (data). Proof BYTES are "parity unpinned" against Sunspot (no pk / .proof fixtures exist in
 * the reference); they are the reference for the HIP path under the same pk and (r,s).
 *
 * Used as (1) parity checker for libspp's GPU proofs, (2) bench.py's cpu_baseline ("port"), OpenMP.
 */
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "field.h"
#include "sha256.h"

/* ------------------------------------------------------------------------------------------------ */
/* circuit container                                                                                 */
/* ------------------------------------------------------------------------------------------------ */
enum { OP_END, OP_SOLVE_C, OP_SOLVE_A, OP_BATCH_DIV, OP_BITS, OP_LIMBS8, OP_COUNT8, OP_POSEIDON, OP_POSEIDON2, OP_COMMIT, OP_GRUMPKIN, OP_INV_H, OP_MASK };

typedef struct { uint32_t rows, nnz; uint32_t* rowptr; uint32_t* wire; uint32_t* coeff; } sparse_t;
typedef struct {
  uint32_t id, n_public, n_secret, n_wires, n_constraints, domain_log, challenge_wire, n_coeffs, n_committed, n_prog, n_aux;
  fe* coeffs;
  fe* aux;
  sparse_t A, B, C, H;
  uint32_t* committed;
  uint32_t* prog;
} circuit_t;

static uint32_t rd32(const uint8_t** p) {
  uint32_t v = (uint32_t)(*p)[0] | ((uint32_t)(*p)[1] << 8) | ((uint32_t)(*p)[2] << 16) | ((uint32_t)(*p)[3] << 24);
  *p += 4;
  return v;
}
static void rd_sparse(const uint8_t** p, sparse_t* m) {
  m->rows = rd32(p);
  m->nnz = rd32(p);
  m->rowptr = (uint32_t*)malloc(4 * (size_t)(m->rows + 1));
  for (uint32_t i = 0; i <= m->rows; i++) m->rowptr[i] = rd32(p);
  m->wire = (uint32_t*)malloc(4 * (size_t)m->nnz + 4);
  m->coeff = (uint32_t*)malloc(4 * (size_t)m->nnz + 4);
  for (uint32_t i = 0; i < m->nnz; i++) {
    m->wire[i] = rd32(p);
    m->coeff[i] = rd32(p);
  }
}
static uint8_t* read_file(const char* path, size_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* buf = (uint8_t*)malloc((size_t)sz + 1);
  if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
  fclose(f);
  *len = (size_t)sz;
  return buf;
}
static circuit_t* circuit_load(const char* path) {
  size_t len;
  uint8_t* buf = read_file(path, &len);
  if (!buf) return NULL;
  const uint8_t* p = buf;
  if (rd32(&p) != 0x43505053u || rd32(&p) != 2) { free(buf); return NULL; }
  circuit_t* c = (circuit_t*)calloc(1, sizeof *c);
  c->id = rd32(&p); c->n_public = rd32(&p); c->n_secret = rd32(&p); c->n_wires = rd32(&p);
  c->n_constraints = rd32(&p); c->domain_log = rd32(&p); c->challenge_wire = rd32(&p);
  c->n_coeffs = rd32(&p); c->n_committed = rd32(&p); c->n_prog = rd32(&p); c->n_aux = rd32(&p);
  c->coeffs = (fe*)malloc(sizeof(fe) * c->n_coeffs);
  for (uint32_t i = 0; i < c->n_coeffs; i++) {
    uint64_t v[4];
    for (int j = 0; j < 4; j++) { uint64_t lo = rd32(&p); uint64_t hi = rd32(&p); v[j] = lo | (hi << 32); }
    fe_from_raw(&c->coeffs[i], v, &FR);
  }
  rd_sparse(&p, &c->A); rd_sparse(&p, &c->B); rd_sparse(&p, &c->C); rd_sparse(&p, &c->H);
  c->committed = (uint32_t*)malloc(4 * (size_t)c->n_committed + 4);
  for (uint32_t i = 0; i < c->n_committed; i++) c->committed[i] = rd32(&p);
  c->prog = (uint32_t*)malloc(4 * (size_t)c->n_prog + 4);
  for (uint32_t i = 0; i < c->n_prog; i++) c->prog[i] = rd32(&p);
  c->aux = (fe*)malloc(sizeof(fe) * (size_t)c->n_aux + 1);
  for (uint32_t i = 0; i < c->n_aux; i++) {
    uint64_t v[4];
    for (int j = 0; j < 4; j++) { uint64_t lo = rd32(&p); uint64_t hi = rd32(&p); v[j] = lo | (hi << 32); }
    fe_from_raw(&c->aux[i], v, &FR);
  }
  free(buf);
  return c;
}

static void row_dot(fe* out, const circuit_t* c, const sparse_t* m, uint32_t k, const fe* w) {
  fe acc, t;
  memset(&acc, 0, sizeof acc);
  for (uint32_t i = m->rowptr[k]; i < m->rowptr[k + 1]; i++) {
    fe_mul(&t, &c->coeffs[m->coeff[i]], &w[m->wire[i]], &FR);
    fe_add(&acc, &acc, &t, &FR);
  }
  *out = acc;
}

/* ------------------------------------------------------------------------------------------------ */
/* Poseidon / Poseidon2 parameters via Grain LFSR and native permutations           */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { uint8_t s[80]; } grain_t;
static int grain_step(grain_t* g) {
  int nb = g->s[0] ^ g->s[13] ^ g->s[23] ^ g->s[38] ^ g->s[51] ^ g->s[62];
  memmove(g->s, g->s + 1, 79);
  g->s[79] = (uint8_t)nb;
  return nb;
}
static void grain_init(grain_t* g, int t, int rf, int rp) {
  int k = 0;
  int vals[6] = {1, 0, 254, t, rf, rp}, widths[6] = {2, 4, 12, 12, 10, 10};
  for (int f = 0; f < 6; f++)
    for (int i = widths[f] - 1; i >= 0; i--) g->s[k++] = (uint8_t)((vals[f] >> i) & 1);
  for (int i = 0; i < 30; i++) g->s[k++] = 1;
  for (int i = 0; i < 160; i++) grain_step(g);
}
static int grain_bit(grain_t* g) {
  for (;;) {
    int a = grain_step(g), b = grain_step(g);
    if (a) return b;
  }
}
static void grain_sample(grain_t* g, uint64_t v[4]) {
  memset(v, 0, 32);
  for (int i = 253; i >= 0; i--)
    if (grain_bit(g)) v[i / 64] |= 1ull << (i % 64);
}
static void grain_field_rej(grain_t* g, fe* out) {
  uint64_t v[4];
  do grain_sample(g, v); while (raw_geq(v, FR.p));
  fe_from_raw(out, v, &FR);
}
static void grain_field_mod(grain_t* g, fe* out) {
  uint64_t v[4];
  grain_sample(g, v);
  fe_from_raw(out, v, &FR);
}

typedef struct { int t, rf, rp; fe* rc; fe mds[5][5]; } pparams_t;
static pparams_t PP[6];
static fe P2_RC[88], P2_MU[4];
static int params_ready = 0;

static void params_init(void) {
  if (params_ready) return;
  for (int t = 3; t <= 5; t += 2) {
    pparams_t* p = &PP[t];
    p->t = t; p->rf = 8; p->rp = (t == 3) ? 57 : 60;
    grain_t g;
    grain_init(&g, t, p->rf, p->rp);
    int nrc = (p->rf + p->rp) * t;
    p->rc = (fe*)malloc(sizeof(fe) * (size_t)nrc);
    for (int i = 0; i < nrc; i++) grain_field_rej(&g, &p->rc[i]);
    fe xy[10];
    for (;;) {
      for (int i = 0; i < 2 * t; i++) grain_field_mod(&g, &xy[i]);
      int dup = 0;
      for (int i = 0; i < 2 * t; i++)
        for (int j = i + 1; j < 2 * t; j++)
          if (fe_eq(&xy[i], &xy[j])) dup = 1;
      if (!dup) break;
    }
    for (int i = 0; i < t; i++)
      for (int j = 0; j < t; j++) {
        fe s;
        fe_add(&s, &xy[i], &xy[t + j], &FR);
        fe_inv(&p->mds[i][j], &s, &FR);
      }
  }
  grain_t g;
  grain_init(&g, 4, 8, 56);
  for (int i = 0; i < 88; i++) grain_field_rej(&g, &P2_RC[i]);
  for (int cand = 0; cand < 5; cand++) {
    fe d[4];
    for (int i = 0; i < 4; i++) grain_field_mod(&g, &d[i]);
    if (cand == 4)
      for (int i = 0; i < 4; i++) fe_sub(&P2_MU[i], &d[i], &FR.one, &FR);
  }
  params_ready = 1;
}

static void sbox_emit(fe* x, fe** out) {
  fe x2, x3, x4, x5;
  fe_sqr(&x2, x, &FR);
  fe_mul(&x3, &x2, x, &FR);
  fe_mul(&x4, &x3, x, &FR);
  fe_mul(&x5, &x4, x, &FR);
  if (out && *out) { (*out)[0] = x2; (*out)[1] = x3; (*out)[2] = x4; (*out)[3] = x5; *out += 4; }
  *x = x5;
}
/* in-place permutation; if out != NULL, x^2,x^3,x^4,x^5 of every S-box are written consecutively */
static void poseidon_permute(fe* s, int t, fe* out) {
  const pparams_t* p = &PP[t];
  fe* o = out;
  for (int r = 0; r < p->rf + p->rp; r++) {
    for (int i = 0; i < t; i++) fe_add(&s[i], &s[i], &p->rc[r * t + i], &FR);
    int full = r < p->rf / 2 || r >= p->rf / 2 + p->rp;
    if (full) for (int i = 0; i < t; i++) sbox_emit(&s[i], &o);
    else sbox_emit(&s[0], &o);
    fe n[5];
    for (int i = 0; i < t; i++) {
      memset(&n[i], 0, sizeof(fe));
      for (int j = 0; j < t; j++) {
        fe m;
        fe_mul(&m, &p->mds[i][j], &s[j], &FR);
        fe_add(&n[i], &n[i], &m, &FR);
      }
    }
    memcpy(s, n, sizeof(fe) * (size_t)t);
  }
}
static void p2_external(fe s[4]) {
  static const int ME[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
  fe n[4];
  for (int i = 0; i < 4; i++) {
    memset(&n[i], 0, sizeof(fe));
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < ME[i][j]; k++) fe_add(&n[i], &n[i], &s[j], &FR);
  }
  memcpy(s, n, sizeof n);
}
static void poseidon2_permute(fe s[4], fe* out) {
  fe* o = out;
  int k = 0;
  p2_external(s);
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 4; i++) { fe_add(&s[i], &s[i], &P2_RC[k + i], &FR); sbox_emit(&s[i], &o); }
    k += 4;
    p2_external(s);
  }
  for (int r = 0; r < 56; r++) {
    fe_add(&s[0], &s[0], &P2_RC[k], &FR);
    sbox_emit(&s[0], &o);
    k++;
    fe tot = s[0];
    for (int i = 1; i < 4; i++) fe_add(&tot, &tot, &s[i], &FR);
    for (int i = 0; i < 4; i++) { fe m; fe_mul(&m, &P2_MU[i], &s[i], &FR); fe_add(&s[i], &m, &tot, &FR); }
  }
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 4; i++) { fe_add(&s[i], &s[i], &P2_RC[k + i], &FR); sbox_emit(&s[i], &o); }
    k += 4;
    p2_external(s);
  }
}

/* exported: hash of n inputs (n = 2 or 4), 32-byte big-endian in/out */
void orc_poseidon_hash(const uint8_t* in, int n, uint8_t out[32]) {
  orc_fields_init(); params_init();
  fe s[5];
  memset(s, 0, sizeof s);
  for (int i = 0; i < n; i++) fe_from_be(&s[i + 1], in + 32 * i, &FR);
  poseidon_permute(s, n + 1, NULL);
  fe_to_be(out, &s[0], &FR);
}
void orc_poseidon2_permute(const uint8_t in[128], uint8_t out[128]) {
  orc_fields_init(); params_init();
  fe s[4];
  for (int i = 0; i < 4; i++) fe_from_be(&s[i], in + 32 * i, &FR);
  poseidon2_permute(s, NULL);
  for (int i = 0; i < 4; i++) fe_to_be(out + 32 * i, &s[i], &FR);
}

/* ------------------------------------------------------------------------------------------------ */
/* witness solver (interprets the SPPC program)                                                       */
/* ------------------------------------------------------------------------------------------------ */
typedef void (*challenge_fn)(void* ctx, const fe* w, fe* out);

static void solve_div_range(const circuit_t* c, fe* w, uint32_t k0, uint32_t n) {
  /* Montgomery batch inversion over the denominators */
  fe* den = (fe*)malloc(sizeof(fe) * n);
  fe* pre = (fe*)malloc(sizeof(fe) * n);
  fe acc = FR.one;
  for (uint32_t i = 0; i < n; i++) {
    row_dot(&den[i], c, &c->B, k0 + i, w);
    pre[i] = acc;
    if (!fe_is_zero(&den[i])) fe_mul(&acc, &acc, &den[i], &FR);
  }
  fe ia;
  fe_inv(&ia, &acc, &FR);
  for (uint32_t i = n; i-- > 0;) {
    uint32_t out = c->A.wire[c->A.rowptr[k0 + i]];
    if (fe_is_zero(&den[i])) { memset(&w[out], 0, sizeof(fe)); continue; }
    fe di, num;
    fe_mul(&di, &ia, &pre[i], &FR);
    fe_mul(&ia, &ia, &den[i], &FR);
    row_dot(&num, c, &c->C, k0 + i, w);
    fe_mul(&w[out], &num, &di, &FR);
  }
  free(den);
  free(pre);
}

/* rs64: the proof's blinding factors r || s (32 B big-endian each): OP_MASK derives the commitment's random mask from them */
static int solve(const circuit_t* c, fe* w, challenge_fn chal, void* chal_ctx, const uint8_t rs64[64]) {
  const uint32_t* pr = c->prog;
  uint32_t pc = 0;
  for (;;) {
    uint32_t op = pr[pc];
    if (op == OP_END) break;
    switch (op) {
      case OP_SOLVE_C: {
        uint32_t k = pr[pc + 1];
        pc += 2;
        fe a, b, ab, rest, t;
        row_dot(&a, c, &c->A, k, w);
        row_dot(&b, c, &c->B, k, w);
        fe_mul(&ab, &a, &b, &FR);
        memset(&rest, 0, sizeof rest);
        uint32_t e = c->C.rowptr[k + 1] - 1;
        for (uint32_t i = c->C.rowptr[k]; i < e; i++) {
          fe_mul(&t, &c->coeffs[c->C.coeff[i]], &w[c->C.wire[i]], &FR);
          fe_add(&rest, &rest, &t, &FR);
        }
        fe_sub(&w[c->C.wire[e]], &ab, &rest, &FR);
        break;
      }
      case OP_SOLVE_A: {
        uint32_t k = pr[pc + 1];
        pc += 2;
        solve_div_range(c, w, k, 1);
        break;
      }
      case OP_BATCH_DIV: {
        solve_div_range(c, w, pr[pc + 1], pr[pc + 2]);
        pc += 3;
        break;
      }
      case OP_BITS: {
        uint32_t h = pr[pc + 1], nb = pr[pc + 2], out0 = pr[pc + 3];
        pc += 4;
        fe v;
        uint64_t raw[4];
        row_dot(&v, c, &c->H, h, w);
        fe_to_raw(raw, &v, &FR);
        for (uint32_t i = 0; i < nb; i++) fe_from_u64(&w[out0 + i], (raw[i / 64] >> (i % 64)) & 1, &FR);
        break;
      }
      case OP_INV_H: {   /* unconstrained inverse hint: 1 / <H_h,w>, 0 for 0 */
        uint32_t h = pr[pc + 1], out = pr[pc + 2];
        pc += 3;
        fe v;
        row_dot(&v, c, &c->H, h, w);
        if (fe_is_zero(&v)) memset(&w[out], 0, sizeof(fe));
        else fe_inv(&w[out], &v, &FR);
        break;
      }
      case OP_LIMBS8: {
        uint32_t h = pr[pc + 1], n = pr[pc + 2], out0 = pr[pc + 3];
        pc += 4;
        fe v;
        uint64_t raw[4];
        row_dot(&v, c, &c->H, h, w);
        fe_to_raw(raw, &v, &FR);
        for (uint32_t i = 0; i < n; i++) fe_from_u64(&w[out0 + i], (raw[i / 8] >> (8 * (i % 8))) & 0xFF, &FR);
        break;
      }
      case OP_COUNT8: {
        uint32_t h0 = pr[pc + 1], n = pr[pc + 2], out0 = pr[pc + 3];
        pc += 4;
        uint64_t cnt[256] = {0};
        for (uint32_t i = 0; i < n; i++) {
          fe v;
          uint64_t raw[4];
          row_dot(&v, c, &c->H, h0 + i, w);
          fe_to_raw(raw, &v, &FR);
          if (raw[0] < 256 && !(raw[1] | raw[2] | raw[3])) cnt[raw[0]]++;
        }
        for (int j = 0; j < 256; j++) fe_from_u64(&w[out0 + j], cnt[j], &FR);
        break;
      }
      case OP_POSEIDON: {
        uint32_t t = pr[pc + 1], h0 = pr[pc + 2], out0 = pr[pc + 3];
        pc += 4;
        fe s[5];
        for (uint32_t i = 0; i < t; i++) row_dot(&s[i], c, &c->H, h0 + i, w);
        poseidon_permute(s, (int)t, &w[out0]);
        break;
      }
      case OP_POSEIDON2: {
        uint32_t h0 = pr[pc + 1], out0 = pr[pc + 2];
        pc += 3;
        fe s[4];
        for (uint32_t i = 0; i < 4; i++) row_dot(&s[i], c, &c->H, h0 + i, w);
        poseidon2_permute(s, &w[out0]);
        break;
      }
      case OP_GRUMPKIN: {
        /* slopes of the affine ladder over Grumpkin (y^2 = x^3 - 17 over Fr): acc_0 = O, acc += T_j[digit_j], then + N */
        uint32_t bit0 = pr[pc + 1], nbits = pr[pc + 2], aux_off = pr[pc + 3], nl = pr[pc + 4];
        const uint32_t* lw = &pr[pc + 5];
        pc += 5 + nl;
        const fe* aux = c->aux + aux_off;
        fe ax = aux[0], ay = aux[1];
        for (uint32_t j = 0; j < nl; j++) {
          fe sx, sy;
          if (j < 64) {
            uint32_t d = 0;
            for (uint32_t k = 0; k < 4; k++) {
              uint32_t bi = 4 * j + k;
              if (bi < nbits && !fe_is_zero(&w[bit0 + bi])) d |= 1u << k;
            }
            sx = aux[4 + (j * 16 + d) * 2];
            sy = aux[4 + (j * 16 + d) * 2 + 1];
          } else {
            sx = aux[2];
            sy = aux[3];
          }
          fe dx, dy, lam, t, x3, y3;
          fe_sub(&dx, &sx, &ax, &FR);
          fe_sub(&dy, &sy, &ay, &FR);
          if (fe_is_zero(&dx)) { memset(&lam, 0, sizeof lam); } else { fe_inv(&t, &dx, &FR); fe_mul(&lam, &dy, &t, &FR); }
          w[lw[j]] = lam;
          fe_sqr(&x3, &lam, &FR); fe_sub(&x3, &x3, &ax, &FR); fe_sub(&x3, &x3, &sx, &FR);
          fe_sub(&t, &ax, &x3, &FR); fe_mul(&y3, &lam, &t, &FR); fe_sub(&y3, &y3, &ay, &FR);
          ax = x3; ay = y3;
        }
        break;
      }
      case OP_COMMIT: {
        pc += 1;
        chal(chal_ctx, w, &w[c->challenge_wire]);
        break;
      }
      case OP_MASK: {   /* fr.Hash(r || s): the hiding mask gnark's api.Commit adds (hints.Randomize) */
        uint32_t out = pr[pc + 1];
        pc += 2;
        orc_hash_to_fr(rs64, 64, "spp-commit-mask1", &w[out], 1);   /* a tag of its own: not the challenge's domain */
        break;
      }
      default:
        return -1;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* NTT over Fr                                                                                        */
/* ------------------------------------------------------------------------------------------------ */
static void fr_root_of_unity(fe* out, uint32_t logn) {
  /* 5^((r-1)/2^logn) */
  uint64_t e[4], one[4] = {1, 0, 0, 0};
  raw_sub(e, FR.p, one);
  for (uint32_t i = 0; i < logn; i++) {
    for (int j = 0; j < 3; j++) e[j] = (e[j] >> 1) | (e[j + 1] << 63);
    e[3] >>= 1;
  }
  fe g;
  fe_from_u64(&g, 5, &FR);
  fe_pow(out, &g, e, &FR);
}
static void bitrev_permute(fe* a, uint32_t logn) {
  uint32_t n = 1u << logn;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t j = 0;
    for (uint32_t b = 0; b < logn; b++) j |= ((i >> b) & 1) << (logn - 1 - b);
    if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
  }
}
/* in-place, natural order in and out; inverse includes the 1/n scaling */
static void ntt(fe* a, uint32_t logn, int inverse) {
  uint32_t n = 1u << logn;
  fe w;
  fr_root_of_unity(&w, logn);
  if (inverse) fe_inv(&w, &w, &FR);
  bitrev_permute(a, logn);
  fe* tw = (fe*)malloc(sizeof(fe) * (n / 2 + 1));
  tw[0] = FR.one;
  for (uint32_t i = 1; i < n / 2; i++) fe_mul(&tw[i], &tw[i - 1], &w, &FR);
  for (uint32_t len = 2; len <= n; len <<= 1) {
    uint32_t half = len / 2, step = n / len;
    for (uint32_t i = 0; i < n; i += len)
      for (uint32_t j = 0; j < half; j++) {
        fe u = a[i + j], v;
        fe_mul(&v, &a[i + j + half], &tw[j * step], &FR);
        fe_add(&a[i + j], &u, &v, &FR);
        fe_sub(&a[i + j + half], &u, &v, &FR);
      }
  }
  free(tw);
  if (inverse) {
    fe ninv, nn;
    fe_from_u64(&nn, n, &FR);
    fe_inv(&ninv, &nn, &FR);
    for (uint32_t i = 0; i < n; i++) fe_mul(&a[i], &a[i], &ninv, &FR);
  }
}
void orc_ntt(uint8_t* data_be, uint32_t logn, int inverse) {
  orc_fields_init();
  uint32_t n = 1u << logn;
  fe* a = (fe*)malloc(sizeof(fe) * n);
  for (uint32_t i = 0; i < n; i++) fe_from_be(&a[i], data_be + 32 * (size_t)i, &FR);
  ntt(a, logn, inverse);
  for (uint32_t i = 0; i < n; i++) fe_to_be(data_be + 32 * (size_t)i, &a[i], &FR);
  free(a);
}

/* h = (a*b - c)/Z as coefficients, via the coset g*H, g = 5. a,b,c: evaluations on H (length n). */
static void compute_h(fe* a, fe* b, fe* c, uint32_t logn) {
  uint32_t n = 1u << logn;
  fe* v[3] = {a, b, c};
  fe g, gi;
  fe_from_u64(&g, 5, &FR);
  fe_inv(&gi, &g, &FR);
#pragma omp parallel for
  for (int k = 0; k < 3; k++) {
    ntt(v[k], logn, 1);
    fe p = FR.one;
    for (uint32_t i = 0; i < n; i++) {
      fe_mul(&v[k][i], &v[k][i], &p, &FR);
      fe_mul(&p, &p, &g, &FR);
    }
    ntt(v[k], logn, 0);
  }
  /* Z(g*w^i) = g^n - 1 */
  uint64_t e[4] = {n, 0, 0, 0};
  fe zn, zi;
  fe_pow(&zn, &g, e, &FR);
  fe_sub(&zn, &zn, &FR.one, &FR);
  fe_inv(&zi, &zn, &FR);
  for (uint32_t i = 0; i < n; i++) {
    fe t;
    fe_mul(&t, &a[i], &b[i], &FR);
    fe_sub(&t, &t, &c[i], &FR);
    fe_mul(&a[i], &t, &zi, &FR);
  }
  ntt(a, logn, 1);
  fe p = FR.one;
  for (uint32_t i = 0; i < n; i++) {
    fe_mul(&a[i], &a[i], &p, &FR);
    fe_mul(&p, &p, &gi, &FR);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* Pippenger MSM (G1 and G2), scalars canonical 4x64                                                   */
/* ------------------------------------------------------------------------------------------------ */
static int msm_window(size_t n) {
  int c = 3;
  while ((1ull << (c + 4)) < n && c < 16) c++;
  return c;
}
static inline uint32_t get_digit(const uint64_t s[4], int pos, int c) {
  int w = pos / 64, o = pos % 64;
  uint64_t v = s[w] >> o;
  if (o + c > 64 && w < 3) v |= s[w + 1] << (64 - o);
  return (uint32_t)(v & ((1ull << c) - 1));
}
#define MSM_IMPL(NAME, AT, JT, SETINF, ADDAFF, ADD, DBL)                                          \
  static void NAME(JT* out, const AT* bases, const uint64_t* scalars, size_t n) {                  \
    SETINF(out);                                                                                   \
    if (n == 0) return;                                                                            \
    int c = msm_window(n);                                                                         \
    int nw = (254 + c - 1) / c;                                                                    \
    JT* wsum = (JT*)malloc(sizeof(JT) * (size_t)nw);                                               \
    _Pragma("omp parallel for schedule(dynamic)") for (int w = 0; w < nw; w++) {                   \
      size_t nb = (1ull << c) - 1;                                                                 \
      JT* bk = (JT*)malloc(sizeof(JT) * nb);                                                       \
      for (size_t i = 0; i < nb; i++) SETINF(&bk[i]);                                              \
      for (size_t i = 0; i < n; i++) {                                                             \
        uint32_t d = get_digit(scalars + 4 * i, w * c, c);                                         \
        if (d) ADDAFF(&bk[d - 1], &bk[d - 1], &bases[i]);                                          \
      }                                                                                            \
      JT run, acc;                                                                                 \
      SETINF(&run);                                                                                \
      SETINF(&acc);                                                                                \
      for (size_t i = nb; i-- > 0;) {                                                              \
        ADD(&run, &run, &bk[i]);                                                                   \
        ADD(&acc, &acc, &run);                                                                     \
      }                                                                                            \
      wsum[w] = acc;                                                                               \
      free(bk);                                                                                    \
    }                                                                                              \
    JT r;                                                                                          \
    SETINF(&r);                                                                                    \
    for (int w = nw - 1; w >= 0; w--) {                                                            \
      for (int k = 0; k < c; k++) DBL(&r, &r);                                                     \
      ADD(&r, &r, &wsum[w]);                                                                       \
    }                                                                                              \
    *out = r;                                                                                      \
    free(wsum);                                                                                    \
  }
MSM_IMPL(msm_g1, g1a, g1j, g1j_set_inf, g1j_add_affine, g1j_add, g1j_dbl)
MSM_IMPL(msm_g2, g2a, g2j, g2j_set_inf, g2j_add_affine, g2j_add, g2j_dbl)

/* exported for parity tests of spp_msm_g1: bases 64 B BE each, scalars 32 B BE each, out 64 B */
void orc_msm_g1(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[64]) {
  orc_fields_init();
  g1a* b = (g1a*)malloc(sizeof(g1a) * (n ? n : 1));
  uint64_t* s = (uint64_t*)malloc(32 * (n ? n : 1));
  for (size_t i = 0; i < n; i++) {
    g1a_from_bytes(&b[i], bases + 64 * i);
    fe t;
    fe_from_be(&t, scalars + 32 * i, &FR);
    fe_to_raw(s + 4 * i, &t, &FR);
  }
  g1j r;
  msm_g1(&r, b, s, n);
  g1a ra;
  g1j_to_affine(&ra, &r);
  g1a_to_bytes(out, &ra);
  free(b);
  free(s);
}

void orc_msm_g2(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[128]) {
  orc_fields_init();
  g2a* b = (g2a*)malloc(sizeof(g2a) * (n ? n : 1));
  uint64_t* s = (uint64_t*)malloc(32 * (n ? n : 1));
  for (size_t i = 0; i < n; i++) {
    g2a_from_bytes(&b[i], bases + 128 * i);
    fe t;
    fe_from_be(&t, scalars + 32 * i, &FR);
    fe_to_raw(s + 4 * i, &t, &FR);
  }
  g2j r;
  msm_g2(&r, b, s, n);
  g2a ra;
  g2j_to_affine(&ra, &r);
  g2a_to_bytes(out, &ra);
  free(b);
  free(s);
}

/* ------------------------------------------------------------------------------------------------ */
/* fixed-base tables for setup                                                                        */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { g1a* t; } fb1_t;   /* 32 windows x 255 entries */
typedef struct { g2a* t; } fb2_t;
static void fb1_build(fb1_t* f, const g1a* base) {
  g1j* tmp = (g1j*)malloc(sizeof(g1j) * 32 * 255);
  g1j cur;
  g1j_from_affine(&cur, base);
  for (int w = 0; w < 32; w++) {
    g1a ca;
    g1j_to_affine(&ca, &cur);
    g1j run = cur;
    for (int d = 0; d < 255; d++) {
      tmp[w * 255 + d] = run;
      g1j_add_affine(&run, &run, &ca);
    }
    cur = run; /* 256 * cur */
  }
  f->t = (g1a*)malloc(sizeof(g1a) * 32 * 255);
  g1_batch_to_affine(f->t, tmp, 32 * 255);
  free(tmp);
}
static void fb1_mul(g1j* out, const fb1_t* f, const fe* k) {
  uint64_t raw[4];
  fe_to_raw(raw, k, &FR);
  g1j_set_inf(out);
  for (int w = 0; w < 32; w++) {
    uint32_t d = (uint32_t)((raw[w / 8] >> (8 * (w % 8))) & 0xFF);
    if (d) g1j_add_affine(out, out, &f->t[w * 255 + d - 1]);
  }
}
static void fb2_build(fb2_t* f, const g2a* base) {
  g2j* tmp = (g2j*)malloc(sizeof(g2j) * 32 * 255);
  g2j cur;
  g2j_from_affine(&cur, base);
  for (int w = 0; w < 32; w++) {
    g2a ca;
    g2j_to_affine(&ca, &cur);
    g2j run = cur;
    for (int d = 0; d < 255; d++) {
      tmp[w * 255 + d] = run;
      g2j_add_affine(&run, &run, &ca);
    }
    cur = run;
  }
  f->t = (g2a*)malloc(sizeof(g2a) * 32 * 255);
  g2_batch_to_affine(f->t, tmp, 32 * 255);
  free(tmp);
}
static void fb2_mul(g2j* out, const fb2_t* f, const fe* k) {
  uint64_t raw[4];
  fe_to_raw(raw, k, &FR);
  g2j_set_inf(out);
  for (int w = 0; w < 32; w++) {
    uint32_t d = (uint32_t)((raw[w / 8] >> (8 * (w % 8))) & 0xFF);
    if (d) g2j_add_affine(out, out, &f->t[w * 255 + d - 1]);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* proving key container ("SPPK") and setup                                                           */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { uint32_t n; uint32_t* wire; g1a* pt; } sec1_t;
typedef struct { uint32_t n; uint32_t* wire; g2a* pt; } sec2_t;
typedef struct {
  uint32_t circuit_id, n_wires, domain_log, n_public, challenge_wire;
  g1a alpha1, beta1, delta1;
  g2a beta2, delta2;
  sec1_t A, B1, K, Z, CB, CS;
  sec2_t B2;
} pk_t;

static void wr32(FILE* f, uint32_t v) {
  uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
  fwrite(b, 1, 4, f);
}
static void wr32be(FILE* f, uint32_t v) {
  uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
  fwrite(b, 1, 4, f);
}
static void wr_g1(FILE* f, const g1a* p) { uint8_t b[64]; g1a_to_bytes(b, p); fwrite(b, 1, 64, f); }
static void wr_g2(FILE* f, const g2a* p) { uint8_t b[128]; g2a_to_bytes(b, p); fwrite(b, 1, 128, f); }
static void wr_sec1(FILE* f, const sec1_t* s, int with_wires) {
  wr32(f, s->n);
  if (with_wires) for (uint32_t i = 0; i < s->n; i++) wr32(f, s->wire[i]);
  for (uint32_t i = 0; i < s->n; i++) wr_g1(f, &s->pt[i]);
}
static void rd_sec1(const uint8_t** p, sec1_t* s, int with_wires) {
  s->n = rd32(p);
  s->wire = (uint32_t*)malloc(4 * (size_t)s->n + 4);
  s->pt = (g1a*)malloc(sizeof(g1a) * (size_t)s->n + 1);
  if (with_wires) for (uint32_t i = 0; i < s->n; i++) s->wire[i] = rd32(p);
  for (uint32_t i = 0; i < s->n; i++) { g1a_from_bytes(&s->pt[i], *p); *p += 64; }
}

static pk_t* pk_load(const char* path) {
  size_t len;
  uint8_t* buf = read_file(path, &len);
  if (!buf) return NULL;
  const uint8_t* p = buf;
  if (rd32(&p) != 0x4b505053u || rd32(&p) != 1) { free(buf); return NULL; }
  pk_t* k = (pk_t*)calloc(1, sizeof *k);
  k->circuit_id = rd32(&p); k->n_wires = rd32(&p); k->domain_log = rd32(&p); k->n_public = rd32(&p); k->challenge_wire = rd32(&p);
  g1a_from_bytes(&k->alpha1, p); p += 64;
  g1a_from_bytes(&k->beta1, p); p += 64;
  g1a_from_bytes(&k->delta1, p); p += 64;
  g2a_from_bytes(&k->beta2, p); p += 128;
  g2a_from_bytes(&k->delta2, p); p += 128;
  rd_sec1(&p, &k->A, 1);
  rd_sec1(&p, &k->B1, 1);
  k->B2.n = rd32(&p);
  k->B2.wire = (uint32_t*)malloc(4 * (size_t)k->B2.n + 4);
  k->B2.pt = (g2a*)malloc(sizeof(g2a) * (size_t)k->B2.n + 1);
  for (uint32_t i = 0; i < k->B2.n; i++) k->B2.wire[i] = rd32(&p);
  for (uint32_t i = 0; i < k->B2.n; i++) { g2a_from_bytes(&k->B2.pt[i], p); p += 128; }
  rd_sec1(&p, &k->K, 1);
  rd_sec1(&p, &k->Z, 0);
  rd_sec1(&p, &k->CB, 1);
  rd_sec1(&p, &k->CS, 1);
  free(buf);
  return k;
}

/* Deterministic trusted setup from a 32-byte seed (toxic waste = hash_to_fr(seed)). Writes pk + vk. */
int orc_setup(const char* circuit_path, const uint8_t seed[32], const char* pk_path, const char* vk_path) {
  orc_fields_init();
  circuit_t* c = circuit_load(circuit_path);
  if (!c) return -1;
  fe tox[7];
  orc_hash_to_fr(seed, 32, "spp-groth16-setup-v1", tox, 7);
  fe tau = tox[0], alpha = tox[1], beta = tox[2], gamma = tox[3], delta = tox[4], sigma = tox[5], rho = tox[6];
  uint32_t logn = c->domain_log, n = 1u << logn, W = c->n_wires;

  /* Lagrange basis at tau: L_k = (tau^n - 1)/n * w^k / (tau - w^k) */
  fe omega, tn, zt, ninv, nn;
  fr_root_of_unity(&omega, logn);
  uint64_t e[4] = {n, 0, 0, 0};
  fe_pow(&tn, &tau, e, &FR);
  fe_sub(&zt, &tn, &FR.one, &FR);
  fe_from_u64(&nn, n, &FR);
  fe_inv(&ninv, &nn, &FR);
  fe* L = (fe*)malloc(sizeof(fe) * n);
  fe* den = (fe*)malloc(sizeof(fe) * n);
  fe* pre = (fe*)malloc(sizeof(fe) * n);
  fe wk = FR.one, acc = FR.one;
  for (uint32_t k = 0; k < n; k++) {
    fe_sub(&den[k], &tau, &wk, &FR);
    pre[k] = acc;
    fe_mul(&acc, &acc, &den[k], &FR);
    L[k] = wk;
    fe_mul(&wk, &wk, &omega, &FR);
  }
  fe ia, scale;
  fe_inv(&ia, &acc, &FR);
  fe_mul(&scale, &zt, &ninv, &FR);
  for (uint32_t k = n; k-- > 0;) {
    fe di;
    fe_mul(&di, &ia, &pre[k], &FR);
    fe_mul(&ia, &ia, &den[k], &FR);
    fe_mul(&L[k], &L[k], &di, &FR);
    fe_mul(&L[k], &L[k], &scale, &FR);
  }
  free(den);
  free(pre);

  /* per-wire polynomial evaluations */
  fe* aw = (fe*)calloc(W, sizeof(fe));
  fe* bw = (fe*)calloc(W, sizeof(fe));
  fe* cw = (fe*)calloc(W, sizeof(fe));
  const sparse_t* M[3] = {&c->A, &c->B, &c->C};
  fe* O[3] = {aw, bw, cw};
  for (int m = 0; m < 3; m++)
    for (uint32_t k = 0; k < c->n_constraints; k++)
      for (uint32_t i = M[m]->rowptr[k]; i < M[m]->rowptr[k + 1]; i++) {
        fe t;
        fe_mul(&t, &c->coeffs[M[m]->coeff[i]], &L[k], &FR);
        fe_add(&O[m][M[m]->wire[i]], &O[m][M[m]->wire[i]], &t, &FR);
      }
  free(L);

  /* wire classes */
  uint8_t* cls = (uint8_t*)calloc(W, 1); /* 0 private, 1 public/challenge, 2 committed */
  for (uint32_t j = 0; j < c->n_public; j++) cls[j] = 1;
  cls[c->challenge_wire] = 1;
  for (uint32_t i = 0; i < c->n_committed; i++) cls[c->committed[i]] = 2;

  fe gi, di;
  fe_inv(&gi, &gamma, &FR);
  fe_inv(&di, &delta, &FR);

  fb1_t T1;
  fb2_t T2;
  fb1_build(&T1, &G1_GEN);
  fb2_build(&T2, &G2_GEN);

  /* scalars -> points */
  g1j* pA = (g1j*)malloc(sizeof(g1j) * W);
  g1j* pB1 = (g1j*)malloc(sizeof(g1j) * W);
  g2j* pB2 = (g2j*)malloc(sizeof(g2j) * W);
  g1j* pK = (g1j*)malloc(sizeof(g1j) * W);
  g1j* pS = (g1j*)malloc(sizeof(g1j) * W);   /* sigma * K (committed only) */
  g1j* pZ = (g1j*)malloc(sizeof(g1j) * n);
#pragma omp parallel for schedule(dynamic, 64)
  for (uint32_t j = 0; j < W; j++) {
    fb1_mul(&pA[j], &T1, &aw[j]);
    fb1_mul(&pB1[j], &T1, &bw[j]);
    fb2_mul(&pB2[j], &T2, &bw[j]);
    fe k, t;
    fe_mul(&k, &beta, &aw[j], &FR);
    fe_mul(&t, &alpha, &bw[j], &FR);
    fe_add(&k, &k, &t, &FR);
    fe_add(&k, &k, &cw[j], &FR);
    fe_mul(&k, &k, cls[j] ? &gi : &di, &FR);
    fb1_mul(&pK[j], &T1, &k);
    if (cls[j] == 2) {
      fe_mul(&k, &k, &sigma, &FR);
      fb1_mul(&pS[j], &T1, &k);
    } else {
      g1j_set_inf(&pS[j]);
    }
  }
  fe zd;
  fe_mul(&zd, &zt, &di, &FR);
  fe* zs = (fe*)malloc(sizeof(fe) * n);
  fe pw = zd;
  for (uint32_t i = 0; i + 1 < n; i++) { zs[i] = pw; fe_mul(&pw, &pw, &tau, &FR); }
#pragma omp parallel for schedule(dynamic, 64)
  for (uint32_t i = 0; i < n - 1; i++) fb1_mul(&pZ[i], &T1, &zs[i]);
  free(zs);

  g1a* aA = (g1a*)malloc(sizeof(g1a) * W);
  g1a* aB1 = (g1a*)malloc(sizeof(g1a) * W);
  g2a* aB2 = (g2a*)malloc(sizeof(g2a) * W);
  g1a* aK = (g1a*)malloc(sizeof(g1a) * W);
  g1a* aS = (g1a*)malloc(sizeof(g1a) * W);
  g1a* aZ = (g1a*)malloc(sizeof(g1a) * n);
  g1_batch_to_affine(aA, pA, W);
  g1_batch_to_affine(aB1, pB1, W);
  g2_batch_to_affine(aB2, pB2, W);
  g1_batch_to_affine(aK, pK, W);
  g1_batch_to_affine(aS, pS, W);
  g1_batch_to_affine(aZ, pZ, n - 1);

  g1j t1;
  g2j t2;
  g1a alpha1, beta1, delta1;
  g2a beta2, gamma2, delta2, pedG, pedGS;
  fb1_mul(&t1, &T1, &alpha); g1j_to_affine(&alpha1, &t1);
  fb1_mul(&t1, &T1, &beta); g1j_to_affine(&beta1, &t1);
  fb1_mul(&t1, &T1, &delta); g1j_to_affine(&delta1, &t1);
  fb2_mul(&t2, &T2, &beta); g2j_to_affine(&beta2, &t2);
  fb2_mul(&t2, &T2, &gamma); g2j_to_affine(&gamma2, &t2);
  fb2_mul(&t2, &T2, &delta); g2j_to_affine(&delta2, &t2);
  fb2_mul(&t2, &T2, &rho); g2j_to_affine(&pedG, &t2);
  fe nrs;
  fe_mul(&nrs, &rho, &sigma, &FR);   /* GSigmaNeg = -sigma * G (gnark-crypto pedersen.Setup) */
  fe_neg(&nrs, &nrs, &FR);
  fb2_mul(&t2, &T2, &nrs); g2j_to_affine(&pedGS, &t2);

  /* ---- write pk ---- */
  FILE* f = fopen(pk_path, "wb");
  if (!f) return -2;
  wr32(f, 0x4b505053u); wr32(f, 1);
  wr32(f, c->id); wr32(f, W); wr32(f, logn); wr32(f, c->n_public); wr32(f, c->challenge_wire);
  wr_g1(f, &alpha1); wr_g1(f, &beta1); wr_g1(f, &delta1); wr_g2(f, &beta2); wr_g2(f, &delta2);
  uint32_t cnt;
  /* A */
  cnt = 0; for (uint32_t j = 0; j < W; j++) cnt += !aA[j].inf;
  wr32(f, cnt);
  for (uint32_t j = 0; j < W; j++) if (!aA[j].inf) wr32(f, j);
  for (uint32_t j = 0; j < W; j++) if (!aA[j].inf) wr_g1(f, &aA[j]);
  /* B1 */
  cnt = 0; for (uint32_t j = 0; j < W; j++) cnt += !aB1[j].inf;
  wr32(f, cnt);
  for (uint32_t j = 0; j < W; j++) if (!aB1[j].inf) wr32(f, j);
  for (uint32_t j = 0; j < W; j++) if (!aB1[j].inf) wr_g1(f, &aB1[j]);
  /* B2 */
  wr32(f, cnt);
  for (uint32_t j = 0; j < W; j++) if (!aB2[j].inf) wr32(f, j);
  for (uint32_t j = 0; j < W; j++) if (!aB2[j].inf) wr_g2(f, &aB2[j]);
  /* K: private, non-committed, non-challenge */
  cnt = 0; for (uint32_t j = 0; j < W; j++) cnt += (cls[j] == 0 && !aK[j].inf);
  wr32(f, cnt);
  for (uint32_t j = 0; j < W; j++) if (cls[j] == 0 && !aK[j].inf) wr32(f, j);
  for (uint32_t j = 0; j < W; j++) if (cls[j] == 0 && !aK[j].inf) wr_g1(f, &aK[j]);
  /* Z */
  wr32(f, n - 1);
  for (uint32_t i = 0; i + 1 < n; i++) wr_g1(f, &aZ[i]);
  /* commitment basis and basis^sigma (all committed wires, in list order) */
  wr32(f, c->n_committed);
  for (uint32_t i = 0; i < c->n_committed; i++) wr32(f, c->committed[i]);
  for (uint32_t i = 0; i < c->n_committed; i++) wr_g1(f, &aK[c->committed[i]]);
  wr32(f, c->n_committed);
  for (uint32_t i = 0; i < c->n_committed; i++) wr32(f, c->committed[i]);
  for (uint32_t i = 0; i < c->n_committed; i++) wr_g1(f, &aS[c->committed[i]]);
  fclose(f);

  /* ---- write vk (gnark raw layout) ---- */
  f = fopen(vk_path, "wb");
  if (!f) return -3;
  wr_g1(f, &alpha1); wr_g1(f, &beta1); wr_g2(f, &beta2); wr_g2(f, &gamma2); wr_g1(f, &delta1); wr_g2(f, &delta2);
  wr32be(f, c->n_public + 1);
  for (uint32_t j = 0; j < c->n_public; j++) wr_g1(f, &aK[j]);
  wr_g1(f, &aK[c->challenge_wire]);
  wr32be(f, 1); wr32be(f, 0); wr32be(f, 1);
  wr_g2(f, &pedG); wr_g2(f, &pedGS);
  fclose(f);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* prover                                                                                             */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { circuit_t* c; pk_t* pk; } orc_ctx;

void* orc_load(const char* circuit_path, const char* pk_path) {
  orc_fields_init();
  params_init();
  orc_ctx* x = (orc_ctx*)calloc(1, sizeof *x);
  x->c = circuit_load(circuit_path);
  x->pk = pk_load(pk_path);
  if (!x->c || !x->pk) { free(x); return NULL; }
  return x;
}
uint32_t orc_n_inputs(void* ctx) { orc_ctx* x = (orc_ctx*)ctx; return x->c->n_public - 1 + x->c->n_secret; }
uint32_t orc_n_public(void* ctx) { return ((orc_ctx*)ctx)->c->n_public - 1; }
uint32_t orc_n_wires(void* ctx) { return ((orc_ctx*)ctx)->c->n_wires; }
uint32_t orc_n_constraints(void* ctx) { return ((orc_ctx*)ctx)->c->n_constraints; }

typedef struct { orc_ctx* x; g1a commitment; } chal_ctx_t;
static void gather_scalars(uint64_t* s, const fe* w, const uint32_t* wires, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) fe_to_raw(s + 4 * (size_t)i, &w[wires[i]], &FR);
}
static void challenge_cb(void* vctx, const fe* w, fe* out) {
  chal_ctx_t* cc = (chal_ctx_t*)vctx;
  pk_t* pk = cc->x->pk;
  uint64_t* s = (uint64_t*)malloc(32 * (size_t)pk->CB.n + 32);
  gather_scalars(s, w, pk->CB.wire, pk->CB.n);
  g1j r;
  msm_g1(&r, pk->CB.pt, s, pk->CB.n);
  free(s);
  g1j_to_affine(&cc->commitment, &r);
  uint8_t bytes[64];
  g1a_to_bytes(bytes, &cc->commitment);
  orc_hash_to_fr(bytes, 64, "bsb22-commitment", out, 1);
}

/* inputs: n_inputs x 32 B big-endian (public without the constant, then secret);
 * r32,s32: blinding scalars, 32 B big-endian each (reduced mod r).
 * wires_out (optional): n_wires x 32 B big-endian full witness, for solver parity tests.
 * returns 0 ok, 1 unsatisfied, <0 error. */
int orc_prove(void* ctx, const uint8_t* inputs, const uint8_t r32[32], const uint8_t s32[32], uint8_t proof[388],
              uint8_t* pw, uint8_t* wires_out) {
  orc_ctx* x = (orc_ctx*)ctx;
  circuit_t* c = x->c;
  pk_t* pk = x->pk;
  uint32_t W = c->n_wires, logn = c->domain_log, n = 1u << logn, nin = c->n_public - 1 + c->n_secret;
  fe* w = (fe*)calloc(W, sizeof(fe));
  w[0] = FR.one;
  for (uint32_t i = 0; i < nin; i++) fe_from_be(&w[1 + i], inputs + 32 * (size_t)i, &FR);
  chal_ctx_t cc;
  cc.x = x;
  uint8_t rs64[64];
  {   /* canonical r || s, as the device derives the mask from the reduced blinding factors */
    fe rr_, ss_;
    fe_from_be(&rr_, r32, &FR);
    fe_from_be(&ss_, s32, &FR);
    fe_to_be(rs64, &rr_, &FR);
    fe_to_be(rs64 + 32, &ss_, &FR);
  }
  if (solve(c, w, challenge_cb, &cc, rs64) != 0) { free(w); return -1; }
  if (wires_out) for (uint32_t i = 0; i < W; i++) fe_to_be(wires_out + 32 * (size_t)i, &w[i], &FR);

  /* a, b, c evaluations + satisfaction check */
  fe* a = (fe*)calloc(n, sizeof(fe));
  fe* b = (fe*)calloc(n, sizeof(fe));
  fe* cv = (fe*)calloc(n, sizeof(fe));
  int unsat = 0;
#pragma omp parallel for schedule(static)
  for (uint32_t k = 0; k < c->n_constraints; k++) {
    row_dot(&a[k], c, &c->A, k, w);
    row_dot(&b[k], c, &c->B, k, w);
    row_dot(&cv[k], c, &c->C, k, w);
    fe t;
    fe_mul(&t, &a[k], &b[k], &FR);
    if (!fe_eq(&t, &cv[k])) {
#pragma omp atomic write
      unsat = 1;
    }
  }
  if (unsat) { free(w); free(a); free(b); free(cv); return 1; }
  compute_h(a, b, cv, logn); /* a now holds h coefficients */

  fe r, s, rs;
  fe_from_be(&r, r32, &FR);
  fe_from_be(&s, s32, &FR);
  fe_mul(&rs, &r, &s, &FR);
  uint64_t rr[4], sr[4], rsr[4];
  fe_to_raw(rr, &r, &FR); fe_to_raw(sr, &s, &FR); fe_to_raw(rsr, &rs, &FR);

  uint64_t* sc = (uint64_t*)malloc(32 * (size_t)(W > n ? W : n) + 32);
  g1j mA, mB1, mK, mZ, mPok, t;
  g2j mB2, t2;
  gather_scalars(sc, w, pk->A.wire, pk->A.n);   msm_g1(&mA, pk->A.pt, sc, pk->A.n);
  gather_scalars(sc, w, pk->B1.wire, pk->B1.n); msm_g1(&mB1, pk->B1.pt, sc, pk->B1.n);
  gather_scalars(sc, w, pk->B2.wire, pk->B2.n); msm_g2(&mB2, pk->B2.pt, sc, pk->B2.n);
  gather_scalars(sc, w, pk->K.wire, pk->K.n);   msm_g1(&mK, pk->K.pt, sc, pk->K.n);
  for (uint32_t i = 0; i + 1 < n; i++) fe_to_raw(sc + 4 * (size_t)i, &a[i], &FR);
  msm_g1(&mZ, pk->Z.pt, sc, n - 1);
  gather_scalars(sc, w, pk->CS.wire, pk->CS.n); msm_g1(&mPok, pk->CS.pt, sc, pk->CS.n);
  free(sc);

  /* Ar = alpha + A + r*delta */
  g1j Ar, Bs1, Krs;
  g2j Bs;
  g1_mul(&t, &pk->delta1, rr);
  g1j_add_affine(&Ar, &mA, &pk->alpha1);
  g1j_add(&Ar, &Ar, &t);
  /* Bs1 = beta + B1 + s*delta ; Bs = beta2 + B2 + s*delta2 */
  g1_mul(&t, &pk->delta1, sr);
  g1j_add_affine(&Bs1, &mB1, &pk->beta1);
  g1j_add(&Bs1, &Bs1, &t);
  g2_mul(&t2, &pk->delta2, sr);
  g2j_add_affine(&Bs, &mB2, &pk->beta2);
  g2j_add(&Bs, &Bs, &t2);
  /* Krs = K + Z + s*Ar + r*Bs1 - rs*delta */
  g1a ArA, Bs1A, nd;
  g1j_to_affine(&ArA, &Ar);
  g1j_to_affine(&Bs1A, &Bs1);
  g1j_add(&Krs, &mK, &mZ);
  g1_mul(&t, &ArA, sr);  g1j_add(&Krs, &Krs, &t);
  g1_mul(&t, &Bs1A, rr); g1j_add(&Krs, &Krs, &t);
  g1a_neg(&nd, &pk->delta1);
  g1_mul(&t, &nd, rsr);  g1j_add(&Krs, &Krs, &t);

  g1a KrsA, PokA;
  g2a BsA;
  g1j_to_affine(&KrsA, &Krs);
  g1j_to_affine(&PokA, &mPok);
  g2j_to_affine(&BsA, &Bs);
  g1a_to_bytes(proof, &ArA);
  g2a_to_bytes(proof + 64, &BsA);
  g1a_to_bytes(proof + 192, &KrsA);
  proof[256] = 0; proof[257] = 0; proof[258] = 0; proof[259] = 1;
  g1a_to_bytes(proof + 260, &cc.commitment);
  g1a_to_bytes(proof + 324, &PokA);

  if (pw) {
    uint32_t np = c->n_public - 1;
    uint8_t hdr[12] = {(uint8_t)(np >> 24), (uint8_t)(np >> 16), (uint8_t)(np >> 8), (uint8_t)np, 0, 0, 0, 0,
                       (uint8_t)(np >> 24), (uint8_t)(np >> 16), (uint8_t)(np >> 8), (uint8_t)np};
    memcpy(pw, hdr, 12);
    for (uint32_t i = 0; i < np; i++) fe_to_be(pw + 12 + 32 * (size_t)i, &w[1 + i], &FR);
  }
  free(w); free(a); free(b); free(cv);
  return 0;
}

/* Solver + satisfaction check only (no proof), `count` input rows in parallel: first_unsat[i] = -1 when row i satisfies every
 * constraint, else the index of the first unsatisfied one (-2: the solver itself failed).  Used by the soundness sweeps in
 * tests/test_circuit_soundness.py (every input +-1 must be refused). */
int orc_check_many(void* ctx, int count, const uint8_t* inputs, int32_t* first_unsat) {
  orc_ctx* x = (orc_ctx*)ctx;
  circuit_t* c = x->c;
  uint32_t W = c->n_wires, nin = c->n_public - 1 + c->n_secret;
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < count; i++) {
    fe* w = (fe*)calloc(W, sizeof(fe));
    w[0] = FR.one;
    for (uint32_t k = 0; k < nin; k++) fe_from_be(&w[1 + k], inputs + ((size_t)i * nin + k) * 32, &FR);
    chal_ctx_t cc;
    cc.x = x;
    int32_t res = -1;
    static const uint8_t zero_rs[64] = {0};
    if (solve(c, w, challenge_cb, &cc, zero_rs) != 0) {
      res = -2;
    } else {
      for (uint32_t k = 0; k < c->n_constraints; k++) {
        fe a, b, cv, t;
        row_dot(&a, c, &c->A, k, w);
        row_dot(&b, c, &c->B, k, w);
        row_dot(&cv, c, &c->C, k, w);
        fe_mul(&t, &a, &b, &FR);
        if (!fe_eq(&t, &cv)) { res = (int32_t)k; break; }
      }
    }
    first_unsat[i] = res;
    free(w);
  }
  return 0;
}

/* Throughput mode for the CPU baseline: `count` independent proofs, one per OpenMP thread (the inner
 * parallel regions of orc_prove then run single-threaded: nested parallelism is off by default). */
int orc_prove_many(void* ctx, int count, const uint8_t* inputs, const uint8_t* rs, uint8_t* proofs, uint8_t* pws) {
  orc_ctx* x = (orc_ctx*)ctx;
  uint32_t nin = x->c->n_public - 1 + x->c->n_secret, np = x->c->n_public - 1;
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < count; i++) {
    int rc = orc_prove(ctx, inputs + (size_t)i * nin * 32, rs + (size_t)i * 64, rs + (size_t)i * 64 + 32, proofs + (size_t)i * 388,
                       pws + (size_t)i * (12 + 32 * np), NULL);
    if (rc) {
#pragma omp atomic write
      bad = rc;
    }
  }
  return bad;
}

void orc_set_threads(int n) { omp_set_num_threads(n); }
int orc_max_threads(void) { return omp_get_max_threads(); }

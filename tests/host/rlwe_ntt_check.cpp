// Host check of csrc/rlwe_ntt.hpp: the per-lane phases of the LDS NTT (exactly the functions k_rlwe_witness runs on the GPU)
// executed lane by lane on the host, against the schoolbook definition of the reference's negacyclic products and quotients
// (scripts/generate_audit.py:45-66,236-243).   g++ -O2 -std=c++17 -I <csrc> rlwe_ntt_check.cpp && ./a.out
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rlwe_ntt.hpp"
using namespace spp;

static RnHostTables T;
struct Wave {
  int32_t x[2][64][16];
  int32_t lds[2][RN_LDS_WORDS];
};
static long long max_abs[2];   // largest |value| a lane ever holds after a pass, per field (overflow margin report)
static void track(int k, const int32_t (&x)[16]) { for (int j = 0; j < 16; j++) { long long a = x[j] < 0 ? -(long long)x[j] : x[j]; if (a > max_abs[k]) max_abs[k] = a; } }
static void ntt2(Wave& w, int dir) {   // both fields; the loops over `lane` stand for the 64 lanes between two barriers
  for (int k = 0; k < 2; k++) {
    for (uint32_t l = 0; l < 64; l++) { if (k == 0) rn_pass1<true>(l, w.x[k][l], w.lds[k], T.f[k], T.w[k][dir], dir); else rn_pass1<false>(l, w.x[k][l], w.lds[k], T.f[k], T.w[k][dir], dir); track(k, w.x[k][l]); }
    for (uint32_t l = 0; l < 64; l++) rn_pass2_read(l, w.x[k][l], w.lds[k]);
    for (uint32_t l = 0; l < 64; l++) { if (k == 0) rn_pass2<true>(l, w.x[k][l], w.lds[k], T.f[k], T.w[k][dir], dir); else rn_pass2<false>(l, w.x[k][l], w.lds[k], T.f[k], T.w[k][dir], dir); track(k, w.x[k][l]); }
    for (uint32_t l = 0; l < 64; l++) { rn_pass3(l, w.x[k][l], w.lds[k], T.f[k], dir); track(k, w.x[k][l]); }
  }
}
static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

int main() {
  rn_build_tables(T);
  const long long Q = RN_P[0], DELTA = 655360;
  int bad = 0;
  for (int trial = 0; trial < 6; trial++) {
    std::vector<uint32_t> a(1024);
    std::vector<int> r(1024), e(1024), m(1024);
    for (int j = 0; j < 1024; j++) {
      a[j] = trial == 1 ? (uint32_t)(Q - 1) : rnd() % (uint32_t)Q;
      r[j] = trial == 1 ? ((j & 1) ? 127 : -128) : trial == 2 ? -128 : (int)(rnd() % 7) - 3;
      e[j] = trial == 1 ? 127 : trial == 2 ? -128 : (int)(rnd() % 7) - 3;
      m[j] = trial == 1 ? 255 : (trial == 2 ? 0 : (int)(rnd() % 256));
    }
    if (trial == 3) for (int j = 0; j < 1024; j++) r[j] = 0;
    if (trial == 4) { for (int j = 0; j < 1024; j++) a[j] = 0; a[0] = 1; }   // identity: S = r
    // ---- public-key transform (k_rlwe_pk_ntt): ahat[k][i] = NTT(a psi^j)[i] / 1024, Montgomery form, canonical ----
    static Wave w;
    std::vector<int32_t> ahat[2] = {std::vector<int32_t>(1024), std::vector<int32_t>(1024)};
    for (int k = 0; k < 2; k++)
      for (uint32_t l = 0; l < 64; l++)
        for (int j = 0; j < 16; j++) w.x[k][l][j] = rn_mul((int32_t)a[l + 64 * j], T.psi[k][l + 64 * j], T.f[k]);
    ntt2(w, 0);
    for (int k = 0; k < 2; k++)
      for (uint32_t l = 0; l < 64; l++)
        for (int j = 0; j < 16; j++) ahat[k][l + 64 * j] = rn_mul(w.x[k][l][j], T.pk_scale[k], T.f[k]);
    // ---- instance: forward transform of r, product, inverse, untwist, CRT ----
    for (int k = 0; k < 2; k++)
      for (uint32_t l = 0; l < 64; l++)
        for (int j = 0; j < 16; j++) {
          w.x[k][l][j] = rn_mul(r[l + 64 * j], T.psi[k][l + 64 * j], T.f[k]);
        }
    ntt2(w, 0);
    for (int k = 0; k < 2; k++)
      for (uint32_t l = 0; l < 64; l++)
        for (int j = 0; j < 16; j++) w.x[k][l][j] = rn_mul(w.x[k][l][j], ahat[k][l + 64 * j], T.f[k]);
    // pruned inverse (message slots): coefficient l of every lane l must equal element 0 of the full inverse transform
    static Wave wp;
    wp = w;
    for (int k = 0; k < 2; k++) {
      for (uint32_t l = 0; l < 64; l++) { if (k == 0) rn_pass1<true>(l, wp.x[k][l], wp.lds[k], T.f[k], T.w[k][1], 1); else rn_pass1<false>(l, wp.x[k][l], wp.lds[k], T.f[k], T.w[k][1], 1); }
      for (uint32_t l = 0; l < 64; l++) rn_pass2_read(l, wp.x[k][l], wp.lds[k]);
      for (uint32_t l = 0; l < 64; l++) { if (k == 0) rn_pass2_first64<true>(l, wp.x[k][l], wp.lds[k], T.f[k], T.w[k][1], 1); else rn_pass2_first64<false>(l, wp.x[k][l], wp.lds[k], T.f[k], T.w[k][1], 1); }
      for (uint32_t l = 0; l < 64; l++) wp.x[k][l][0] = rn_pass3_first64(l, wp.lds[k]);
    }
    ntt2(w, 1);
    for (int k = 0; k < 2; k++)
      for (uint32_t l = 0; l < 64; l++) {
        const long long d = ((long long)wp.x[k][l][0] - w.x[k][l][0]) % RN_P[k];
        if (d != 0) { if (bad < 5) printf("pruned inverse differs: field %d lane %u\n", k, l); bad++; }
      }
    // wrap correction C_i (suffix sums of r through the LDS scan phases + the zero list of a)
    static int32_t pre[1024 + 64];
    static int32_t suffix[64][16];
    int8_t rbytes[1024];
    std::vector<uint16_t> zeros;
    for (int j = 0; j < 1024; j++) { rbytes[j] = (int8_t)r[j]; if (a[j] == 0) zeros.push_back((uint16_t)j); }
    for (uint32_t l = 0; l < 64; l++) { int32_t rr[16]; for (int j = 0; j < 16; j++) rr[j] = r[l + 64 * j]; rn_scan_scatter(l, rr, pre); }
    static int32_t tot[64];
    for (uint32_t l = 0; l < 64; l++) rn_scan_chunk(l, pre, tot);
    int32_t offs[64], total[64];
    for (uint32_t l = 0; l < 64; l++) rn_scan_offsets(l, tot, offs[l], total[l]);
    for (uint32_t l = 0; l < 64; l++) rn_scan_gather(l, total[l], pre, offs, suffix[l]);
    for (uint32_t l = 0; l < 64; l++)
      for (int j = 0; j < 16; j++) {
        const uint32_t i = l + 64 * j;
        const int32_t s0 = rn_canon(rn_mul(w.x[0][l][j], T.ipsi[0][i], T.f[0]), T.f[0]);
        const int32_t s1 = rn_canon(rn_mul(w.x[1][l][j], T.ipsi[1][i], T.f[1]), T.f[1]);
        const int32_t t = rn_crt_digit(s0, s1, T.f[1]);
        const int32_t add = e[i] + (int32_t)(DELTA * m[i]);
        int32_t k;
        uint32_t rem;
        rn_quot_rem(s0, t, add, k, rem);
        k += suffix[l][j] - rn_zero_correction(i, zeros.data(), (uint32_t)zeros.size(), rbytes);
        // schoolbook: S = sum_j A2[(i - j) mod 2048] r_j with A2 = [a, (q - a) mod q]  (negacyclic_matrix_row_mod_q)
        long long S = 0;
        for (int jj = 0; jj < 1024; jj++) {
          const int idx = ((int)i - jj) & 2047;
          const long long av = idx < 1024 ? a[idx] : (a[idx - 1024] ? Q - a[idx - 1024] : 0);
          S += av * r[jj];
        }
        const long long v = S + add;
        long long kq = v / Q, rr = v % Q;
        if (rr < 0) { rr += Q; kq -= 1; }
        if (kq != k || rr != rem) {
          if (bad < 5) printf("MISMATCH trial %d i %u: k %d vs %lld, rem %u vs %lld\n", trial, i, k, kq, rem, rr);
          bad++;
        }
      }
  }
  // every table constant lies in [0, p) (the multiplier bound of rn_mul)
  for (int k = 0; k < 2; k++)
    for (int e = 0; e < 1024; e++)
      for (int32_t v : {T.w[k][0][e], T.w[k][1][e], T.psi[k][e], T.ipsi[k][e]})
        if (v < 0 || v >= RN_P[k]) bad++;
  // growth bound of the lazy signed arithmetic, derived, not observed: interval propagation through rn_dft16 with |inputs| < p
  // (units of p): sums double per stage, twiddle products reset to 1, the REDUCE step resets x[0], x[1] before the last stage
  for (int reduce = 0; reduce < 2; reduce++) {
    double b[16];
    for (int i = 0; i < 16; i++) b[i] = 1;
    double worst = 1;
    auto bf = [&](int i, int j, bool trivial) { const double s = b[i] + b[j]; b[i] = s; b[j] = trivial ? s : 1; if (s > worst) worst = s; };
    for (int i = 0; i < 8; i++) bf(i, i + 8, i == 0);
    for (int blk = 0; blk < 16; blk += 8) for (int i = 0; i < 4; i++) bf(blk + i, blk + i + 4, i == 0);
    for (int blk = 0; blk < 16; blk += 4) for (int i = 0; i < 2; i++) bf(blk + i, blk + i + 2, i == 0);
    if (reduce) b[0] = b[1] = 1;
    for (int blk = 0; blk < 16; blk += 2) bf(blk, blk + 1, true);
    const int k = reduce ? 0 : 1;
    if (worst * RN_P[k] >= 2147483648.0) { printf("dft16 bound %g p overflows field %d\n", worst, k); bad++; }
    else printf("dft16 growth bound field %d: %g p of %.1f p\n", k, worst, 2147483648.0 / RN_P[k]);
  }
  for (int k = 0; k < 2; k++) if (max_abs[k] >= 2147483648ll) bad++;
  printf("largest |value| seen after a pass: field0 %.2f p, field1 %.2f p\n", (double)max_abs[0] / RN_P[0], (double)max_abs[1] / RN_P[1]);
  if (bad) { printf("FAIL %d\n", bad); return 1; }
  printf("OK rlwe_ntt: 6 polynomials x 1024 coefficients, quotients and remainders equal the schoolbook values\n");
  return 0;
}

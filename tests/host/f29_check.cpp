// Test-only host build of csrc/f29.hpp: the unsaturated 9x29-bit arithmetic against Fp (bn254.hpp, itself checked
// against the oracle by tests/test_host_cpu.py) on random and extremal-limb inputs, and XYZZ29::madd chains against
// XYZZ<Fq>::madd including the doubling / cancellation / negated-entry paths.  Prints "OK <n checks>" or "FAIL ...".
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include "f29.hpp"
using namespace spp;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd32() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 16);
}
template <class F> static F rnd_field() {
  uint32_t w[8];
  for (int i = 0; i < 8; i++) w[i] = rnd32();
  return F::from_u256(w);
}
static int checks = 0;
#define CHECK(c, msg) do { checks++; if (!(c)) { printf("FAIL %s (line %d)\n", msg, __LINE__); exit(1); } } while (0)

// words whose 29-bit limbs are all ones below a top limb `top` (value < 2p needs top small enough)
template <class F> static F maxlimb_words(uint32_t top) {
  F r;
  uint32_t l[9];
  for (int i = 0; i < 8; i++) l[i] = (1u << 29) - 1u;
  l[8] = top;
  for (int k = 0; k < 8; k++) {
    const int bit = 32 * k, i = bit / 29, o = bit % 29;
    uint32_t v = l[i] >> o;
    if (i + 1 < 9) v |= l[i + 1] << (29 - o);
    if (i + 2 < 9 && 58 - o < 32) v |= l[i + 2] << (58 - o);
    r.l[k] = v;
  }
  return r;
}

template <class Pm> static void field_tests(const char* name) {
  using B = Fp<Pm>;
  using F = F29<Pm>;
  for (int it = 0; it < 2000; it++) {
    B a = rnd_field<B>(), b = rnd_field<B>(), c = rnd_field<B>(), d = rnd_field<B>();
    F a9 = F::from_fp(a), b9 = F::from_fp(b), c9 = F::from_fp(c), d9 = F::from_fp(d);
    CHECK(a9.to_fp() == a, "roundtrip");
    CHECK((a9 * b9).to_fp() == a * b, "mul");
    CHECK(a9.sqr().to_fp() == a.sqr(), "sqr");
    CHECK(F::mul2(a9, b9, c9, d9).to_fp() == a * b + c * d, "mul2");
    CHECK(F::template sub_norm<Pm::SUBC_2P_1>(a9, b9).to_fp() == a - b, "sub_norm");
    CHECK((F::template sub_lazy<Pm::SUBC_6P_1>(a9, b9) * c9).to_fp() == (a - b) * c, "sub_lazy");
    CHECK(F::template sub3_norm<Pm::SUBC_4P_3>(a9, b9, c9).to_fp() == a - b - c - c, "sub3_norm");
    CHECK((F::template neg_lazy<Pm::SUBC_2P_1>(a9) * b9).to_fp() == a.neg() * b, "neg_lazy");
    CHECK(add_norm(a9, b9).to_fp() == a + b, "add_norm");
    CHECK((add_lazy(a9, b9) * c9).to_fp() == (a + b) * c, "add_lazy");
    F z = F::template sub_norm<Pm::SUBC_6P_1>(a9, a9);
    CHECK(z.template is_zero_mod_p<7>(), "zero");
    F nz = F::template sub_norm<Pm::SUBC_6P_1>(a9, b9);
    CHECK(nz.template is_zero_mod_p<7>() == (a == b), "nonzero");
  }
  // extremal limbs: top limb of a value just below 2p is ~6.3M; use words with all-ones low limbs
  const uint32_t top2p = (uint32_t)(((uint64_t)Pm::TWOP(7) << 32 | Pm::TWOP(6)) >> 40) - 1;   // bits 232.. of 2p, minus 1
  B m = maxlimb_words<B>(top2p);                 // raw words, value < 2p: a legal "almost Montgomery" element
  F m9 = F::from_words(m.l);                     // same integer, limbs all ones
  for (int l = 0; l < 9; l++) CHECK(m9.l[l] == (l < 8 ? (1u << 29) - 1u : top2p), "maxlimb layout");
  // interpret the integer m as an R'-domain element: to_fp gives m*R/R'; compare products through Fp
  B mR = m9.to_fp();
  F three = add_lazy(add_lazy(m9, m9), m9);      // limbs 3*(2^29-1): the largest operand madd feeds (T)
  F two = add_lazy(m9, m9);
  CHECK((three * m9).to_fp() == (mR + mR + mR) * mR, "3x1 extremal");
  CHECK(F::mul2(m9, three, two, m9).to_fp() == mR * (mR + mR + mR) + (mR + mR) * mR, "mul2 extremal (1x3 + 2x1)");
  CHECK(m9.sqr().to_fp() == mR.sqr(), "sqr extremal");
  CHECK((two * m9).to_fp() == (mR + mR) * mR, "2x1 extremal");
  printf("%s field ok\n", name);
}

static void curve_tests() {
  using X29 = XYZZ29<FqParams>;
  // points k*G for a few k, affine, through the Fp code
  G1Affine G{Fq::one(), Fq::one().dbl()};
  std::vector<G1Affine> pts;
  for (int i = 0; i < 24; i++) {
    uint32_t k[8];
    for (int j = 0; j < 8; j++) k[j] = rnd32();
    k[7] &= 0x0fffffff;
    pts.push_back(scalar_mul(G, k).to_affine());
  }
  for (int trial = 0; trial < 50; trial++) {
    G1XYZZ ref = G1XYZZ::infinity();
    X29 acc = X29::infinity();
    for (int s = 0; s < 40; s++) {
      int idx = rnd32() % pts.size();
      bool neg = rnd32() & 1;
      int mode = rnd32() % 16;
      G1Affine e = pts[idx];
      if (mode == 0 && !ref.is_inf()) { e = ref.to_affine(); neg = false; }        // doubling path
      if (mode == 1 && !ref.is_inf()) { e = ref.to_affine(); neg = true; }         // cancellation -> infinity
      G1Affine en = e;
      if (neg) en.y = en.y.neg();
      ref.madd(en);
      acc.madd(e, neg);
      CHECK(acc.inf == ref.is_inf(), "inf flag");
      if (!ref.is_inf()) {
        G1Affine a = ref.to_affine(), b = acc.to_xyzz().to_affine();
        CHECK(a.x == b.x && a.y == b.y, "madd chain");
      }
    }
  }
  printf("curve ok\n");
}

static Fq fq_hex(const char* h) {
  uint32_t c[8];
  for (int i = 0; i < 8; i++) {
    char buf[9];
    memcpy(buf, h + 8 * i, 8);
    buf[8] = 0;
    c[7 - i] = (uint32_t)strtoul(buf, nullptr, 16);
  }
  return Fq::from_canonical(c);
}

static void fq2_tests() {
  using E = F29x2;
  for (int it = 0; it < 1000; it++) {
    Fq2 a{rnd_field<Fq>(), rnd_field<Fq>()}, b{rnd_field<Fq>(), rnd_field<Fq>()};
    E a9 = E::from_fp(a), b9 = E::from_fp(b);
    CHECK(E::mul<FqParams::SUBC_2P_1>(a9, b9).to_fp() == a * b, "fq2 mul");
    CHECK(a9.sqr<FqParams::SUBC_2P_1>().to_fp() == a.sqr(), "fq2 sqr");
  }
  printf("fq2 field ok\n");
}

static void g2_tests() {
  // BN254 G2 generator (EIP-197), x = x0 + x1 u, y = y0 + y1 u
  G2Affine G{{fq_hex("1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed"),
              fq_hex("198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2")},
             {fq_hex("12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa"),
              fq_hex("090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b")}};
  std::vector<G2Affine> pts;
  for (int i = 0; i < 12; i++) {
    uint32_t k[8];
    for (int j = 0; j < 8; j++) k[j] = rnd32();
    k[7] &= 0x0fffffff;
    pts.push_back(scalar_mul(G, k).to_affine());
  }
  for (int trial = 0; trial < 20; trial++) {
    G2XYZZ ref = G2XYZZ::infinity();
    XYZZ29G2 acc = XYZZ29G2::infinity();
    for (int s = 0; s < 40; s++) {
      int idx = rnd32() % pts.size();
      bool neg = rnd32() & 1;
      int mode = rnd32() % 16;
      G2Affine e = pts[idx];
      if (mode == 0 && !ref.is_inf()) { e = ref.to_affine(); neg = false; }
      if (mode == 1 && !ref.is_inf()) { e = ref.to_affine(); neg = true; }
      G2Affine en = e;
      if (neg) en.y = en.y.neg();
      ref.madd(en);
      acc.madd(e, neg);
      CHECK(acc.inf == ref.is_inf(), "g2 inf flag");
      if (!ref.is_inf()) {
        G2Affine a = ref.to_affine(), b = acc.to_xyzz().to_affine();
        CHECK(a.x == b.x && a.y == b.y, "g2 madd chain");
      }
    }
  }
  printf("g2 ok\n");
}

int main() {
  fq2_tests();
  g2_tests();
  field_tests<FqParams>("fq");
  field_tests<FrParams>("fr");
  curve_tests();
  printf("OK %d\n", checks);
  return 0;
}

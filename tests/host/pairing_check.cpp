// Test-only host build: the batched-verifier pairing path (csrc/pairing_fast.hpp: shared Miller loop with line tables,
// projective lines for the per-proof G2 point, x-power final exponentiation) against the single-proof host pairing
// (csrc/pairing.hpp, itself checked against the Python oracle's pairing by tests/test_host_cpu.py) on products that
// are one by bilinearity and on products that are not.  Prints "OK <n>" or "FAIL ...".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "pairing_fast_host.hpp"
#include "verify_one.hpp"
#include <vector>
#include <string>
using namespace spp;

static uint64_t rng_state = 0x2545F4914F6CDD1Dull;
static uint32_t rnd32() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 16);
}
static int checks = 0;
#define CHECK(c, msg) do { checks++; if (!(c)) { printf("FAIL %s (line %d)\n", msg, __LINE__); return 1; } } while (0)

static Fq fq_hex(const char* h) {
  uint32_t c[8];
  for (int i = 0; i < 8; i++) { char b[9]; memcpy(b, h + 8 * i, 8); b[8] = 0; c[7 - i] = (uint32_t)strtoul(b, nullptr, 16); }
  return Fq::from_canonical(c);
}
static void small_scalar(uint32_t k[8]) { for (int j = 0; j < 8; j++) k[j] = j < 2 ? rnd32() : 0; }
static Fr fr_of(const uint32_t k[8]) { return Fr::from_canonical(k); }

static std::vector<uint8_t> slurp(const char* path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path, "rb");
  if (!f) return v;
  int c;
  while ((c = fgetc(f)) != EOF) v.push_back((uint8_t)c);
  fclose(f);
  return v;
}
static G1Affine g1_raw(const uint8_t* b) { return g1_from_raw_hd(b); }

// pairing_check <vk> <proof> <pw> : verify_one on the host against a key prepared like spp_verify_batch does
static int verify_files(const char* vkp, const char* prp, const char* pwp) {
  std::vector<uint8_t> vk = slurp(vkp), proof = slurp(prp), pw = slurp(pwp);
  if (vk.size() < 580 || proof.size() != 388 || pw.size() < 12) { printf("VERIFY bad-files\n"); return 2; }
  const uint32_t nk = be32_at(vk.data() + 576);
  size_t off = 580;
  std::vector<G1Affine> K(nk);
  for (uint32_t i = 0; i < nk; i++) K[i] = g1_raw(vk.data() + off + 64 * (size_t)i);
  off += (size_t)nk * 64 + 12;
  const G1Affine alpha1 = g1_raw(vk.data());
  const G2Affine beta2 = g2_from_raw_hd(vk.data() + 128), gamma2 = g2_from_raw_hd(vk.data() + 256), delta2 = g2_from_raw_hd(vk.data() + 448);
  const G2Affine pedG = g2_from_raw_hd(vk.data() + off), pedGS = g2_from_raw_hd(vk.data() + off + 128);
  std::vector<LineStep> t0 = build_line_table(gamma2), t1 = build_line_table(delta2), t2 = build_line_table(pedG), t3 = build_line_table(pedGS);
  VerifyKeyDev h;
  h.pc = make_pairing_fast_consts();
  h.tab[0] = t0.data(); h.tab[1] = t1.data(); h.tab[2] = t2.data(); h.tab[3] = t3.data();
  h.e_alpha_beta = f12_from(miller_loop(alpha1.neg(), beta2));
  h.twist_b = twist_b();
  h.K = K.data();
  h.nk = nk;
  printf("VERIFY %d\n", verify_one(h, proof.data(), pw.data()) ? 1 : 0);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 4) return verify_files(argv[1], argv[2], argv[3]);
  CHECK(pairing_fast_consts_consistent(), "Frobenius sparsity");
  const PairingFastConsts pc = make_pairing_fast_consts();
  G1Affine G1{Fq::one(), Fq::one().dbl()};
  G2Affine G2{{fq_hex("1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed"),
               fq_hex("198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2")},
              {fq_hex("12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa"),
               fq_hex("090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b")}};
  CHECK(g2_in_subgroup(G2), "generator in subgroup");
  {   // Frobenius and f12 arithmetic: fast == host
    Fq12 a = Fq12::zero(), b = Fq12::zero();
    for (int i = 0; i < 12; i++) { uint32_t w[8]; for (auto& x : w) x = rnd32(); a.c[i] = Fq::from_u256(w); for (auto& x : w) x = rnd32(); b.c[i] = Fq::from_u256(w); }
    F12 fa = f12_from(a), fb = f12_from(b);
    Fq12 ab = f12_mul(a, b);
    F12 fab = f12_mul(fa, fb, pc);
    for (int i = 0; i < 12; i++) CHECK(ab.c[i] == fab.c[i], "f12_mul");
    Fq12 fr = f12_frobenius(a);
    F12 ffr = f12_frob(fa, pc);
    for (int i = 0; i < 12; i++) CHECK(fr.c[i] == ffr.c[i], "frobenius");
  }
  for (int trial = 0; trial < 4; trial++) {
    // e(aP, bQ) * e(-abP, Q) * e(cP, Q) * e(-P, cQ) == 1
    uint32_t a[8], b[8], c[8], ab[8];
    small_scalar(a); small_scalar(b); small_scalar(c);
    Fr abf = fr_of(a) * fr_of(b);
    abf.to_canonical(ab);
    G1Affine aP = scalar_mul(G1, a).to_affine(), abP = scalar_mul(G1, ab).to_affine(), cP = scalar_mul(G1, c).to_affine();
    G2Affine bQ = scalar_mul(G2, b).to_affine(), cQ = scalar_mul(G2, c).to_affine();
    std::vector<std::pair<G1Affine, G2Affine>> good = {{aP, bQ}, {abP.neg(), G2}, {cP, G2}, {G1.neg(), cQ}};
    CHECK(pairing_product_is_one(good), "host: product is one");
    // fast path: bQ dynamic, two others through line tables, the last folded in as `extra`
    std::vector<LineStep> t1 = build_line_table(G2), t2 = build_line_table(cQ);
    CHECK(t1.size() == miller_steps(), "table length");
    const LineStep* tabs[2] = {t1.data(), t2.data()};
    G1Affine Ps[2] = {abP.neg(), G1.neg()};
    F12 extra = f12_from(miller_loop(cP, G2));
    F12 f = miller_multi(2, tabs, Ps, true, aP, bQ, extra, pc);
    CHECK(final_exp_is_one(f, pc), "fast: product is one");
    // wrong products must fail in both
    std::vector<std::pair<G1Affine, G2Affine>> bad = good;
    bad[0].first = cP;
    CHECK(!pairing_product_is_one(bad), "host: product is not one");
    F12 fb = miller_multi(2, tabs, Ps, true, cP, bQ, extra, pc);
    CHECK(!final_exp_is_one(fb, pc), "fast: product is not one");
    // tables only (no dynamic pair): e(aP, Q) * e(-aP, Q) == 1, and a pair at infinity contributes nothing
    const LineStep* tabs2[2] = {t1.data(), t1.data()};
    G1Affine Ps2[2] = {aP, aP.neg()};
    CHECK(final_exp_is_one(miller_multi(2, tabs2, Ps2, false, G1Affine::infinity(), G2Affine::infinity(), f12_one(pc), pc), pc), "fixed only");
    G1Affine Ps3[2] = {aP, G1Affine::infinity()};
    CHECK(!final_exp_is_one(miller_multi(2, tabs2, Ps3, false, G1Affine::infinity(), G2Affine::infinity(), f12_one(pc), pc), pc), "fixed only, not one");
  }
  printf("OK %d\n", checks);
  return 0;
}

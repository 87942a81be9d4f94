// Test-only host build of the product's field/curve header (same source the HIP kernels compile):
// reads "op hex..." lines on stdin, prints results; tests/test_host_arith.py compares with the oracle.
#include <cstdio>
#include <cstring>
#include <string>
#include <iostream>
#include <sstream>
#include <vector>
#include "bn254.hpp"
using namespace spp;

static void parse_hex(const std::string& h, uint32_t out[8]) {
  std::string s = h;
  if (s.rfind("0x", 0) == 0) s = s.substr(2);
  while (s.size() < 64) s = "0" + s;
  for (int i = 0; i < 8; i++) out[7 - i] = (uint32_t)strtoul(s.substr(8 * i, 8).c_str(), nullptr, 16);
}
template <class F> static F rd(std::istringstream& is) {
  std::string h; is >> h; uint32_t c[8]; parse_hex(h, c); return F::from_canonical(c);
}
template <class F> static std::string hx(const F& a) {
  uint32_t c[8]; a.to_canonical(c); char buf[80];
  snprintf(buf, sizeof buf, "%08x%08x%08x%08x%08x%08x%08x%08x", c[7], c[6], c[5], c[4], c[3], c[2], c[1], c[0]);
  return buf;
}
static Fq2 rd2(std::istringstream& is) { Fq a = rd<Fq>(is); Fq b = rd<Fq>(is); return {a, b}; }
static std::string hx2(const Fq2& a) { return hx(a.c0) + " " + hx(a.c1); }

template <class F> static void field_ops(const std::string& op, std::istringstream& is) {
  if (op == "mul") { F a = rd<F>(is), b = rd<F>(is); std::cout << hx(a * b) << "\n"; }
  else if (op == "add") { F a = rd<F>(is), b = rd<F>(is); std::cout << hx(a + b) << "\n"; }
  else if (op == "sub") { F a = rd<F>(is), b = rd<F>(is); std::cout << hx(a - b) << "\n"; }
  else if (op == "neg") { F a = rd<F>(is); std::cout << hx(a.neg()) << "\n"; }
  else if (op == "inv") { F a = rd<F>(is); std::cout << hx(a.inv()) << "\n"; }
  else if (op == "invf") { F a = rd<F>(is); std::cout << hx(a.inv_fermat()) << "\n"; }
  else if (op == "invnc") {   // the non-canonical word (value + p, still < 2p) of the same element
    F a = rd<F>(is), m;
    for (int i = 0; i < 8; i++) m.l[i] = F::zero().l[i];
    uint32_t c = 0;
    F w = a;
    for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)a.l[i] + F::modulus_word(i) + c; w.l[i] = (uint32_t)t; c = (uint32_t)(t >> 32); }
    if (!F::geq_mod(a.l) ) std::cout << hx(w.inv()) << "\n"; else std::cout << hx(a.inv()) << "\n";
  }
  else if (op == "u256") { std::string h; is >> h; uint32_t c[8]; parse_hex(h, c); std::cout << hx(F::from_u256(c)) << "\n"; }
  else if (op == "small") { F a = rd<F>(is); unsigned k; is >> k; std::cout << hx(a.mul_small(k)) << "\n"; }
}

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream is(line);
    std::string kind, op; is >> kind;
    if (kind == "fr") { is >> op; field_ops<Fr>(op, is); }
    else if (kind == "fq") { is >> op; field_ops<Fq>(op, is); }
    else if (kind == "fq2") {
      is >> op;
      if (op == "mul") { Fq2 a = rd2(is), b = rd2(is); std::cout << hx2(a * b) << "\n"; }
      else if (op == "sqr") { Fq2 a = rd2(is); std::cout << hx2(a.sqr()) << "\n"; }
      else if (op == "inv") { Fq2 a = rd2(is); std::cout << hx2(a.inv()) << "\n"; }
    } else if (kind == "g1") {
      is >> op;
      if (op == "mul") {   // point x y scalar
        G1Affine p{rd<Fq>(is), rd<Fq>(is)}; std::string h; is >> h; uint32_t k[8]; parse_hex(h, k);
        G1Affine r = scalar_mul(p, k).to_affine(); std::cout << hx(r.x) << " " << hx(r.y) << "\n";
      } else if (op == "add") {  // (p*k1 as xyzz) + (q xyzz from affine)  and madd
        G1Affine p{rd<Fq>(is), rd<Fq>(is)}; G1Affine q{rd<Fq>(is), rd<Fq>(is)};
        G1XYZZ a = G1XYZZ::from_affine(p); a.dbl_inplace(); a.madd(p);  // 3p with nontrivial ZZ
        G1XYZZ b = G1XYZZ::from_affine(q); b.dbl_inplace();             // 2q
        G1XYZZ c = a; c.add(b);                                          // 3p+2q
        G1XYZZ d = a; d.madd(q);                                         // 3p+q
        G1Affine rc = c.to_affine(), rd_ = d.to_affine();
        std::cout << hx(rc.x) << " " << hx(rc.y) << " " << hx(rd_.x) << " " << hx(rd_.y) << "\n";
      }
    } else if (kind == "g2") {
      is >> op;
      if (op == "mul") {
        G2Affine p{rd2(is), rd2(is)}; std::string h; is >> h; uint32_t k[8]; parse_hex(h, k);
        G2Affine r = scalar_mul(p, k).to_affine(); std::cout << hx2(r.x) << " " << hx2(r.y) << "\n";
      }
    } else if (kind == "gk") {
      is >> op;
      if (op == "mul") {
        GkAffine p{rd<Fr>(is), rd<Fr>(is)}; std::string h; is >> h; uint32_t k[8]; parse_hex(h, k);
        GkAffine r = scalar_mul(p, k).to_affine(); std::cout << hx(r.x) << " " << hx(r.y) << "\n";
      }
    }
  }
  return 0;
}

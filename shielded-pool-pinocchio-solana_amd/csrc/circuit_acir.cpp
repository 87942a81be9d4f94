// ACIR -> R1CS: the `sunspot compile <acir>` step of the reference's pipeline (noir_circuit/prove_linux.sh:66-70,
// client/proof.helper.ts:58-64) for circuits compiled by nargo, with the opcode set the reference's withdraw circuit uses
// (noir_circuit/target/shielded_pool_verifier.json: 6 148 AssertZero, 25 RANGE, 1 MultiScalarMul over Grumpkin's generator,
// 6 Brillig calls).  The result is an ordinary SPPC circuit: the same prover, setup, verifier and byte
// formats as the two hand-written circuits -- but the constraint system is derived mechanically from the REFERENCE'S OWN
// compiled program, so "does this R1CS state the reference's statement" is not a question of reading circuit.cpp.
//
// Input: the flat blob written by spp/acir.py:to_blob (the bincode decoding stays in Python):
//   u32 magic "ACR1" | u32 n_public | u32 n_secret | u32 n_ops | ops
//   expr   = u32 n_mul, n_mul x (F q, u32 a, u32 b), u32 n_lin, n_lin x (F q, u32 w), F c        F = 32 B little-endian canonical
//   op 0 ASSERT_ZERO  expr
//   op 1 RANGE        u32 witness, u32 bits
//   op 2 MSM_GRUMPKIN_G  u32 lo, u32 hi, u32 out_x, u32 out_y, u32 out_inf        (scalar = lo + 2^128 hi, base = the generator)
//   op 3 HINT_DIVMOD_POW2  expr, u32 log2(divisor), u32 out_q, u32 out_r
//   op 4 HINT_INVERSE      expr, u32 out
//   op 5 HINT_BITS         expr, u32 n, n x u32 out      (little-endian radix-2 digits)
// Witnesses 0 .. n_public-1 are the public inputs, the next n_secret the private ones (nargo's ABI order).
//
// Lowering.  Every ACIR witness is tracked as a linear combination of R1CS wires.
//   AssertZero with every witness known: one row (q a)(b) = -(linear part); purely linear: (linear)(1) = 0.
//   AssertZero with ONE unknown witness u (ACVM's solving rule): u appears linearly  -> u := (-q/q_u) a b - rest/q_u, one row,
//     or no row at all when there is no product (u is just a linear combination); u inside the product -> u := -rest / (q b + q_u),
//     one row u * den = num.  Extra products of a wide opcode become helper wires t = a b first.
//   RANGE n: n = 1 -> v v = v; otherwise ceil(n/8) byte limbs through the 8-bit lookup argument (+ one scaled lookup of the
//     top limb when n is not a multiple of 8) and the recomposition row.
//   Brillig calls are UNCONSTRAINED hints, exactly as in ACIR: fresh wires filled by the solver (bits / bytes / inverse),
//     bound only by the AssertZero and RANGE opcodes that follow them.
//   MultiScalarMul(G; lo, hi): the fixed-base Grumpkin ladder of circuit.cpp over the 256 bits of (lo, hi).
// Builder::finish() adds the lookup argument with its BSB22 commitment, so the 388-byte proof layout is kept.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include "circuit.hpp"

namespace spp {
namespace {

struct BlobReader {
  const uint8_t* p;
  size_t n, o = 0;
  bool ok = true;
  uint32_t u32() {
    if (o + 4 > n) { ok = false; return 0; }
    uint32_t v;
    memcpy(&v, p + o, 4);
    o += 4;
    return v;
  }
  Fr field() {
    if (o + 32 > n) { ok = false; return Fr::zero(); }
    uint32_t l[8];
    memcpy(l, p + o, 32);
    o += 32;
    // canonical check: < r
    for (int i = 7; i >= 0; i--) {
      if (l[i] < FrParams::MOD(i)) break;
      if (l[i] > FrParams::MOD(i) || i == 0) { ok = false; return Fr::zero(); }
    }
    return Fr::from_canonical(l);
  }
};

struct MulTerm { Fr q; uint32_t a, b; };
struct LinTerm { Fr q; uint32_t w; };
struct Expr {
  std::vector<MulTerm> mul;
  std::vector<LinTerm> lin;
  Fr c;
};
static Expr read_expr(BlobReader& r) {
  Expr e;
  uint32_t nm = r.u32();
  for (uint32_t i = 0; i < nm && r.ok; i++) { MulTerm t; t.q = r.field(); t.a = r.u32(); t.b = r.u32(); e.mul.push_back(t); }
  uint32_t nl = r.u32();
  for (uint32_t i = 0; i < nl && r.ok; i++) { LinTerm t; t.q = r.field(); t.w = r.u32(); e.lin.push_back(t); }
  e.c = r.field();
  return e;
}

struct Compiler {
  Builder b;
  std::map<uint32_t, LC> wmap;     // ACIR witness -> linear combination of R1CS wires
  std::string err;
  explicit Compiler(uint32_t id) : b(id) {}

  bool known(uint32_t w) const { return wmap.count(w) != 0; }
  const LC& val(uint32_t w) { return wmap[w]; }
  // a linear combination that has grown long is pinned to one wire (one row) so that later rows stay short
  size_t compact_above = 40;
  LC compact(const LC& v) {
    if (v.t.size() <= compact_above) return v;
    return b.mul(v, LC::constant(Fr::one()), true, false);
  }
  void define(uint32_t w, const LC& v) { wmap[w] = compact(v); }
  // Which factor of a product goes to the B side: a wire in B costs a G1 AND a G2 table walk in every proof (the G2 one three
  // times as expensive), a wire in A one G1 walk -- so the factor that brings fewer NEW wires into B goes there.
  std::set<uint32_t> b_wires;
  size_t new_in_b(const LC& v) const {
    size_t n = 0;
    for (auto& e : v.t) n += (e.first != 0 && !b_wires.count(e.first)) ? 1 : 0;
    return n;
  }
  void orient(LC& a, LC& bb) {
    if (new_in_b(a) < new_in_b(bb)) std::swap(a, bb);
    for (auto& e : bb.t) if (e.first) b_wires.insert(e.first);
  }
  // Peephole for power maps (Poseidon's x^5 is compiled by nargo as x2 = x x, x4 = x2 x2, x5 = x4 x): a wire known to be a
  // square x x that is itself squared is rewritten (x2 x) x -- one more row, but x2 never enters B (see orient), which is
  // what circuit.cpp's hand-written S-box does.  squares: wire -> the factor x it is the square of.
  std::map<uint32_t, LC> squares;
  static bool same(const LC& a, const LC& bb) {
    if (a.t.size() != bb.t.size()) return false;
    for (size_t i = 0; i < a.t.size(); i++)
      if (a.t[i].first != bb.t[i].first || !(a.t[i].second == bb.t[i].second)) return false;
    return true;
  }
  static bool single_wire(const LC& a, uint32_t* w) {
    if (a.t.size() != 1 || a.t[0].first == 0 || !(a.t[0].second == Fr::one())) return false;
    *w = a.t[0].first;
    return true;
  }
  // rewrites (a, bb) in place when it is the square of a recorded square; returns true if it did
  bool peephole = true;
  bool split_fourth_power(LC& a, LC& bb) {
    uint32_t w;
    if (!peephole) return false;
    if (!same(a, bb) || !single_wire(a, &w)) return false;
    auto it = squares.find(w);
    if (it == squares.end()) return false;
    const LC x = it->second;
    for (auto& e : x.t) if (e.first) b_wires.insert(e.first);
    a = b.mul(LC::wire(w), x, true, false);        // x3 = x2 * x   (A = x2, B = x)
    bb = x;
    return true;
  }
  LC product(const LC& x, const LC& y) {           // helper wire t = x y (folds when a factor is constant)
    if (x.is_constant() || y.is_constant()) return b.mul(x, y);
    LC a = x, bb = y;
    orient(a, bb);
    return b.mul(a, bb);
  }
  // value of an expression whose witnesses are all known (products become wires)
  bool expr_value(const Expr& e, LC* out) {
    LC s = LC::constant(e.c);
    for (auto& t : e.lin) {
      if (!known(t.w)) { err = "hint input uses an unsolved witness"; return false; }
      s = s + val(t.w).scaled(t.q);
    }
    for (auto& t : e.mul) {
      if (!known(t.a) || !known(t.b)) { err = "hint input uses an unsolved witness"; return false; }
      s = s + product(val(t.a), val(t.b)).scaled(t.q);
    }
    *out = s;
    return true;
  }

  bool assert_zero(const Expr& e, uint32_t index) {
    // unknown witnesses
    std::vector<uint32_t> unk;
    auto note = [&](uint32_t w) {
      if (!known(w) && std::find(unk.begin(), unk.end(), w) == unk.end()) unk.push_back(w);
    };
    for (auto& t : e.mul) { note(t.a); note(t.b); }
    for (auto& t : e.lin) note(t.w);
    if (unk.size() > 1) { err = "opcode " + std::to_string(index) + ": more than one unsolved witness"; return false; }
    const bool has_u = unk.size() == 1;
    const uint32_t u = has_u ? unk[0] : 0;
    // split: the product that contains u (if any), the other products, the linear part without u, the coefficient of u
    LC rest = LC::constant(e.c);
    Fr qu = Fr::zero();
    for (auto& t : e.lin) {
      if (has_u && t.w == u) qu = qu + t.q;
      else rest = rest + val(t.w).scaled(t.q);
    }
    LC den;                  // coefficient of u coming from products: sum q * (other operand)
    bool u_in_product = false;
    std::vector<const MulTerm*> plain;
    for (auto& t : e.mul) {
      const bool ua = has_u && t.a == u, ub = has_u && t.b == u;
      if (ua && ub) { err = "opcode " + std::to_string(index) + ": quadratic in its unsolved witness"; return false; }
      if (ua || ub) {
        u_in_product = true;
        den = den + val(ua ? t.b : t.a).scaled(t.q);
      } else {
        plain.push_back(&t);
      }
    }
    if (has_u && u_in_product) {
      // u * (den + qu) = -(rest + plain products)
      for (auto* t : plain) rest = rest + product(val(t->a), val(t->b)).scaled(t->q);
      LC d = den + LC::constant(qu);
      for (auto& e : d.t) if (e.first) b_wires.insert(e.first);
      define(u, b.div(rest.neg(), d));
      return true;
    }
    // keep ONE plain product for the row, turn the others into helper wires
    const MulTerm* main = plain.empty() ? nullptr : plain.back();
    for (size_t i = 0; i + 1 < plain.size(); i++) rest = rest + product(val(plain[i]->a), val(plain[i]->b)).scaled(plain[i]->q);
    LC ma, mb;
    if (main) {
      ma = val(main->a);
      mb = val(main->b);
      if (ma.is_constant() || mb.is_constant()) {      // a product with a constant factor is linear
        rest = rest + (ma.is_constant() ? mb.scaled(ma.constant_value()) : ma.scaled(mb.constant_value())).scaled(main->q);
        main = nullptr;
      } else if (!split_fourth_power(ma, mb)) {
        orient(ma, mb);
      }
    }
    if (has_u) {
      if (qu.is_zero()) { err = "opcode " + std::to_string(index) + ": unsolved witness with zero coefficient"; return false; }
      const Fr k = qu.inv().neg();                       // u = k * (q a b + rest)
      if (main) {
        const bool is_square = same(val(main->a), val(main->b)) && same(ma, mb) && rest.t.empty() && (main->q * k == Fr::one());
        define(u, b.mul_sub(ma.scaled(main->q * k), mb, rest.scaled(k).neg(), true, false));
        uint32_t w;
        if (is_square && single_wire(wmap[u], &w)) squares[w] = mb;
      } else define(u, rest.scaled(k));
      return true;
    }
    if (main) b.constrain(ma.scaled(main->q), mb, rest.neg());
    else b.constrain(rest, LC::constant(Fr::one()), LC());
    return true;
  }

  bool range(uint32_t w, uint32_t bits, uint32_t index) {
    if (!known(w)) { err = "opcode " + std::to_string(index) + ": RANGE on an unsolved witness"; return false; }
    if (bits == 0 || bits > 253) { err = "opcode " + std::to_string(index) + ": RANGE width " + std::to_string(bits); return false; }
    const LC v = val(w);
    if (bits == 1) {
      b.constrain(v, v, v);
      return true;
    }
    const uint32_t nl = (bits + 7) / 8, rem = bits % 8;
    std::vector<LC> limbs = b.to_limbs8(v, nl);
    if (rem) b.lookup8(limbs.back().scaled_u64(1u << (8 - rem)));     // top limb < 2^rem
    return true;
  }

  // an output of a black box or of an unconstrained helper must be a witness nothing has defined yet: a blob that names an input
  // or an already solved witness there would silently re-map it (spp_circuit_build_acir is a public entry point)
  bool fresh(uint32_t w, uint32_t index) {
    if (!known(w)) return true;
    err = "opcode " + std::to_string(index) + ": output witness " + std::to_string(w) + " is already defined";
    return false;
  }

  bool msm(uint32_t lo, uint32_t hi, uint32_t ox, uint32_t oy, uint32_t oinf, uint32_t index) {
    if (!known(lo) || !known(hi)) { err = "opcode " + std::to_string(index) + ": MSM scalar unsolved"; return false; }
    if (!fresh(ox, index) || !fresh(oy, index) || !fresh(oinf, index)) return false;
    // 256 consecutive bit wires: to_bits allocates its outputs back to back
    std::vector<LC> bits = b.to_bits(val(lo), 128);
    std::vector<LC> hbits = b.to_bits(val(hi), 128);
    bits.insert(bits.end(), hbits.begin(), hbits.end());
    auto pt = gadget_grumpkin_fixed_base(b, bits, true);
    wmap[ox] = pt.first;
    wmap[oy] = pt.second;
    wmap[oinf] = LC();        // the ladder has no representation of the point at infinity: scalar 0 is unsatisfiable
    return true;
  }
};

}  // namespace

bool build_acir_circuit(const uint8_t* blob, size_t len, uint32_t circuit_id, Circuit* out, std::string* err) {
  BlobReader r{blob, len};
  auto bail = [&](const std::string& m) { if (err) *err = m; return false; };
  if (r.u32() != 0x31524341u) return bail("not an ACR1 blob");
  const uint32_t n_pub = r.u32(), n_sec = r.u32(), n_ops = r.u32();
  if (!r.ok || n_pub == 0 || n_pub > 64 || n_sec > (1u << 20)) return bail("bad header");
  Compiler c(circuit_id);
  if (const char* e = getenv("SPP_ACIR_COMPACT")) c.compact_above = (size_t)atoi(e);
  if (getenv("SPP_ACIR_NO_PEEPHOLE")) c.peephole = false;
  for (uint32_t i = 0; i < n_pub; i++) c.wmap[i] = c.b.public_input();
  for (uint32_t i = 0; i < n_sec; i++) c.wmap[n_pub + i] = c.b.secret_input();
  for (uint32_t k = 0; k < n_ops; k++) {
    const uint32_t kind = r.u32();
    if (!r.ok) return bail("truncated blob");
    bool ok = true;
    switch (kind) {
      case 0: {
        Expr e = read_expr(r);
        ok = r.ok && c.assert_zero(e, k);
        break;
      }
      case 1: {
        const uint32_t w = r.u32(), bits = r.u32();
        ok = r.ok && c.range(w, bits, k);
        break;
      }
      case 2: {
        const uint32_t lo = r.u32(), hi = r.u32(), ox = r.u32(), oy = r.u32(), oi = r.u32();
        ok = r.ok && c.msm(lo, hi, ox, oy, oi, k);
        break;
      }
      case 3: {
        Expr e = read_expr(r);
        const uint32_t lg = r.u32(), oq = r.u32(), orr = r.u32();
        LC v;
        ok = r.ok && lg % 8 == 0 && lg > 0 && lg < 256 && c.expr_value(e, &v) && c.fresh(oq, k) && c.fresh(orr, k) && oq != orr;
        if (ok) {
          std::vector<LC> limbs = c.b.limbs8_hint(v, 32);
          LC q, rem;
          Fr pw = Fr::one();
          const Fr k256 = Fr::from_u64(256);
          for (uint32_t i = 0; i < lg / 8; i++) { rem = rem + limbs[i].scaled(pw); pw = pw * k256; }
          pw = Fr::one();
          for (uint32_t i = lg / 8; i < 32; i++) { q = q + limbs[i].scaled(pw); pw = pw * k256; }
          c.wmap[oq] = q;
          c.wmap[orr] = rem;
        } else if (c.err.empty()) c.err = "opcode " + std::to_string(k) + ": unsupported quotient hint";
        break;
      }
      case 4: {
        Expr e = read_expr(r);
        const uint32_t o = r.u32();
        LC v;
        ok = r.ok && c.expr_value(e, &v) && c.fresh(o, k);
        if (ok) c.wmap[o] = c.b.inv_hint(v);
        break;
      }
      case 5: {
        Expr e = read_expr(r);
        const uint32_t n = r.u32();
        LC v;
        ok = r.ok && n > 0 && n <= 254 && c.expr_value(e, &v);
        if (ok) {
          std::vector<LC> bits = c.b.bits_hint(v, n);
          for (uint32_t i = 0; i < n && ok; i++) {
            const uint32_t o = r.u32();
            ok = r.ok && c.fresh(o, k);
            if (ok) c.wmap[o] = bits[i];
          }
        }
        break;
      }
      default: return bail("unknown op kind " + std::to_string(kind));
    }
    if (!ok) return bail(c.err.empty() ? "malformed op " + std::to_string(k) : c.err);
  }
  if (r.o != len) return bail("trailing bytes in blob");
  *out = c.b.finish();
  return true;
}

}  // namespace spp

// R1CS builder + solver-program emitter + circuit container ("SPPC" file) -- host side of libspp.
//
// The reference proves whatever R1CS `sunspot compile` derives from the Noir ACIR
// (noir_circuit/prove_linux.sh:66-70, scripts/generate_audit.py:659-665); that compiler is an external
// Go binary and its output for the audit circuit is absent (.MISSING_LARGE_BLOBS). libspp therefore
// carries its own R1CS for the two circuits, written against the Noir sources
// (noir_circuit/src/main.nr:38-82, scripts/generate_audit.py:405-463), with exactly one BSB22-style
// commitment each so that the 388-byte proof layout (withdraw.rs:13) is kept.
//
// A circuit is: sparse matrices A,B,C (rows = constraints  <A_k,w>*<B_k,w> = <C_k,w>), a matrix H of
// auxiliary linear forms read by hints, a coefficient table, and a *solver program*: a list of
// instructions, executed in order by one GPU lane per proof (kernels_solve.hip), that computes every
// internal wire from the inputs.  Wire 0 is the constant 1, then public inputs, then secret inputs,
// then internal wires.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "bn254.hpp"

namespace spp {

// ---- solver opcodes (u32 stream; operands follow the opcode word) -------------------------------
enum : uint32_t {
  OP_END = 0,
  OP_SOLVE_C = 1,     // k            : w[out] = <A_k,w><B_k,w> - (<C_k,w> - w[out]); out = LAST term of C_k (coeff 1)
  OP_SOLVE_A = 2,     // k            : w[out] = <C_k,w> / <B_k,w>;  A_k = {out:1}
  OP_BATCH_DIV = 3,   // k0 n         : OP_SOLVE_A for constraints k0..k0+n-1 with one shared inversion
  OP_BITS = 4,        // h nbits out0 : out0+i = bit i of canonical(<H_h,w>)
  OP_LIMBS8 = 5,      // h n out0     : out0+i = byte i (little-endian) of canonical(<H_h,w>)
  OP_COUNT8 = 6,      // h0 n out0    : out0+v = #{ i<n : canonical(<H_{h0+i},w>) == v }, v<256
  OP_POSEIDON = 7,    // t h0 out0    : native Poseidon permutation of (<H_h0..>), writes x^2,x^4,x^5 per S-box
  OP_POSEIDON2 = 8,   // h0 out0      : native Poseidon2 t=4 permutation, same wire convention
  OP_COMMIT = 9,      //              : phase boundary: the challenge wire is filled in before continuing
  OP_GRUMPKIN = 10,   // bit0 nbits aux_off n  w_0..w_{n-1} : slopes of the fixed-base Grumpkin ladder (see circuit.cpp)
  OP_INV_H = 11,      // h out        : w[out] = 1 / <H_h,w>  (0 when the form is 0); an UNCONSTRAINED hint (ACIR Brillig inverse)
  OP_MASK = 12,       // out          : w[out] = fr.Hash(r || s): the random mask of the commitment (gnark: hints.Randomize inside
                      //                api.Commit) -- a committed wire no constraint touches, so that the commitment hides the others
  // ---- the solver of a DECODED gnark system (spp/ccs.py to_sppc_solved): rows as gnark's solver uses them ----
  OP_SOLVE_ROW = 13,  // k side inv odiv : the unknown is the LAST term of side (0 A, 1 B, 2 C) of row k, with coefficient cf:
                      //                side 2: (<A><B> - rest) / cf;  side 0: (<C> / <B> - rest) / cf;  side 1: (<C> / <A> - rest) / cf.
                      //                inv = index of 1/cf in coeffs (0xffffffff: cf = 1); odiv = index of the inverse of the other
                      //                factor when that is a constant (0xffffffff: inverted at run time; a zero factor gives 0)
  OP_LIMBS = 14,      // h n width out0 : out0+i = bits [i*width, (i+1)*width) of canonical(<H_h,w>), width <= 128; bit 31 of the width word:
                      //                the limbs in reverse wire order
                      //                (rangecheck.DecomposeHint, sw-grumpkin.decompose, the Brillig quotient / remainder by 2^128)
  OP_COUNTN = 15,     // h0 n out0 size : out0+v = #{ i<n : canonical(<H_{h0+i},w>) == v }, v < size <= 256 (logderivarg.countHint
                      //                over the table 0 .. size-1)
  OP_GK_MUL = 16,     // h_lo h_hi gy out_x out_y out_inf : (lo + 2^128 hi) * G on Grumpkin, G = (1, coeffs[gy]); the ACIR MultiScalarMul
                      //                black box over the generator: out = (x, y, 0), or (0, 0, 1) for the point at infinity
  OP_GLV = 17,        // h out0 c[28] : sw-grumpkin.decomposeScalar: out0..+3 = 64-bit limbs of s1, out0+4..+7 of s2 (gnark_hints.hpp)
  OP_EMUL = 18,       // h0 out0 c[16]: emulated.mulHint with b = 1, 64-bit limbs, 4-limb modulus q, 6 limb forms H_h0..: out0..+3 quotient,
                      //                +4..+7 remainder, +8..+13 carries
};

enum : uint32_t { CIRCUIT_WITHDRAW = 1, CIRCUIT_AUDIT = 2, CIRCUIT_ACIR = 5 };

struct Term {
  uint32_t wire;
  uint32_t coeff;  // index into Circuit::coeffs
};

struct Sparse {
  std::vector<uint32_t> rowptr{0};
  std::vector<Term> terms;
  uint32_t rows() const { return (uint32_t)rowptr.size() - 1; }
};

struct Circuit {
  uint32_t id = 0;
  uint32_t n_public = 0;   // including wire 0 (constant one)
  uint32_t n_secret = 0;
  uint32_t n_wires = 0;
  uint32_t n_constraints = 0;
  uint32_t domain_log = 0;
  uint32_t challenge_wire = 0;
  std::vector<Fr> coeffs;               // Montgomery form in memory; canonical LE in the file
  Sparse A, B, C, H;
  std::vector<uint32_t> committed;      // wires bound by the commitment (private)
  std::vector<uint32_t> program;
  std::vector<Fr> aux;                  // constants read by native hints (Grumpkin window tables)
  uint32_t n_inputs() const { return n_public - 1 + n_secret; }
  bool save(const std::string& path) const;
  bool load(const std::string& path);
};

// ---- linear combinations ---------------------------------------------------------------------------
struct LC {
  std::vector<std::pair<uint32_t, Fr>> t;  // sorted by wire, no zero coefficients
  LC() {}
  static LC wire(uint32_t w) {
    LC r;
    r.t.push_back({w, Fr::one()});
    return r;
  }
  static LC constant(const Fr& c) {
    LC r;
    if (!c.is_zero()) r.t.push_back({0, c});
    return r;
  }
  static LC constant_u64(uint64_t v) { return constant(Fr::from_u64(v)); }
  bool is_constant() const { return t.empty() || (t.size() == 1 && t[0].first == 0); }
  Fr constant_value() const { return t.empty() ? Fr::zero() : t[0].second; }
  LC scaled(const Fr& k) const {
    LC r;
    if (k.is_zero()) return r;
    r.t.reserve(t.size());
    for (auto& e : t) r.t.push_back({e.first, e.second * k});
    return r;
  }
  LC scaled_u64(uint64_t k) const { return scaled(Fr::from_u64(k)); }
  LC neg() const { return scaled(Fr::one().neg()); }
  friend LC operator+(const LC& a, const LC& b) {
    LC r;
    r.t.reserve(a.t.size() + b.t.size());
    size_t i = 0, j = 0;
    while (i < a.t.size() || j < b.t.size()) {
      if (j == b.t.size() || (i < a.t.size() && a.t[i].first < b.t[j].first)) {
        r.t.push_back(a.t[i++]);
      } else if (i == a.t.size() || b.t[j].first < a.t[i].first) {
        r.t.push_back(b.t[j++]);
      } else {
        Fr s = a.t[i].second + b.t[j].second;
        if (!s.is_zero()) r.t.push_back({a.t[i].first, s});
        i++;
        j++;
      }
    }
    return r;
  }
  friend LC operator-(const LC& a, const LC& b) { return a + b.neg(); }
};

// ---- builder ---------------------------------------------------------------------------------------
class Builder {
 public:
  explicit Builder(uint32_t circuit_id) { c_.id = circuit_id; c_.n_public = 1; next_wire_ = 1; }

  // inputs must be declared before any internal wire: first all public, then all secret
  LC public_input() { uint32_t w = next_wire_++; c_.n_public = next_wire_; return LC::wire(w); }
  LC secret_input() { c_.n_secret++; return LC::wire(next_wire_++); }

  uint32_t new_wire() { return next_wire_++; }
  uint32_t next_wire() const { return next_wire_; }
  uint32_t n_constraints() const { return c_.A.rows(); }

  // a*b = c, nothing solved
  uint32_t constrain(const LC& a, const LC& b, const LC& c) {
    push_row(c_.A, a);
    push_row(c_.B, b);
    push_row(c_.C, c);
    return c_.A.rows() - 1;
  }
  void assert_eq(const LC& a, const LC& b) { constrain(a - b, LC::constant(Fr::one()), LC()); }

  // out = a*b - rest   (constraint a*b = out + rest); solved unless `solve` is false (native hint fills it)
  // `fold`: constant operands collapse to a linear form without a constraint (gadgets with a fixed wire
  // layout pass fold=false so that every product owns a wire).
  LC mul_sub(const LC& a, const LC& b, const LC& rest, bool solve = true, bool fold = true) {
    if (fold && a.is_constant()) return b.scaled(a.constant_value()) - rest;
    if (fold && b.is_constant()) return a.scaled(b.constant_value()) - rest;
    uint32_t out = new_wire();
    LC c = rest;                           // out must be the LAST term of the C row: append manually
    push_row(c_.A, a);
    push_row(c_.B, b);
    push_row_with_tail(c_.C, c, out);
    if (solve) {
      c_.program.push_back(OP_SOLVE_C);
      c_.program.push_back(c_.A.rows() - 1);
    }
    return LC::wire(out);
  }
  LC mul(const LC& a, const LC& b, bool solve = true, bool fold = true) { return mul_sub(a, b, LC(), solve, fold); }

  // out = num/den  (constraint out*den = num). If `defer` the instruction is left to a later batch_div().
  LC div(const LC& num, const LC& den, bool defer = false) {
    uint32_t out = new_wire();
    last_div_wire_ = out;
    uint32_t k = constrain(LC::wire(out), den, num);
    if (!defer) {
      c_.program.push_back(OP_SOLVE_A);
      c_.program.push_back(k);
    }
    return LC::wire(out);
  }
  void emit_batch_div(uint32_t k0, uint32_t n) {
    c_.program.push_back(OP_BATCH_DIV);
    c_.program.push_back(k0);
    c_.program.push_back(n);
  }

  uint32_t hint_row(const LC& a) {
    push_row(c_.H, a);
    return c_.H.rows() - 1;
  }
  // unconstrained hints (what ACIR's Brillig calls are): fresh wires filled by the solver, bound only by later constraints
  LC inv_hint(const LC& a) {
    uint32_t h = hint_row(a), out = new_wire();
    c_.program.push_back(OP_INV_H);
    c_.program.push_back(h);
    c_.program.push_back(out);
    return LC::wire(out);
  }
  std::vector<LC> bits_hint(const LC& a, uint32_t nbits) {
    uint32_t h = hint_row(a), out0 = next_wire_;
    c_.program.push_back(OP_BITS);
    c_.program.push_back(h);
    c_.program.push_back(nbits);
    c_.program.push_back(out0);
    std::vector<LC> bits;
    for (uint32_t i = 0; i < nbits; i++) bits.push_back(LC::wire(new_wire()));
    return bits;
  }
  std::vector<LC> limbs8_hint(const LC& a, uint32_t n) {
    uint32_t h = hint_row(a), out0 = next_wire_;
    c_.program.push_back(OP_LIMBS8);
    c_.program.push_back(h);
    c_.program.push_back(n);
    c_.program.push_back(out0);
    std::vector<LC> limbs;
    for (uint32_t i = 0; i < n; i++) limbs.push_back(LC::wire(new_wire()));
    return limbs;
  }

  // little-endian bits of a (nbits), each constrained boolean, recomposition asserted
  std::vector<LC> to_bits(const LC& a, uint32_t nbits) {
    uint32_t h = hint_row(a);
    uint32_t out0 = next_wire_;
    c_.program.push_back(OP_BITS);
    c_.program.push_back(h);
    c_.program.push_back(nbits);
    c_.program.push_back(out0);
    std::vector<LC> bits;
    LC sum;
    Fr pw = Fr::one();
    for (uint32_t i = 0; i < nbits; i++) {
      LC b = LC::wire(new_wire());
      bits.push_back(b);
      sum = sum + b.scaled(pw);
      pw = pw.dbl();
    }
    for (uint32_t i = 0; i < nbits; i++) constrain(bits[i], bits[i], bits[i]);
    assert_eq(sum, a);
    return bits;
  }

  // value(bits) <= bound (bound given as little-endian bits of a constant), one constraint per bit
  void assert_bits_leq_const(const std::vector<LC>& bits, const uint32_t bound_limbs[8]) {
    int n = (int)bits.size();
    LC p = LC::constant(Fr::one());  // "all higher bits where bound=1 were 1"
    for (int i = n - 1; i >= 0; i--) {
      bool cb = (bound_limbs[i / 32] >> (i % 32)) & 1;
      if (cb) {
        p = mul(p, bits[i]);
      } else {
        // bit may be 1 only if some higher position already made the value strictly smaller: b*(p) == 0 ... p==1 forbids b
        constrain(bits[i], p, LC());
      }
    }
  }

  // 8-bit limbs of a (n limbs), each limb queued for the 8-bit lookup; recomposition asserted
  std::vector<LC> to_limbs8(const LC& a, uint32_t n) {
    uint32_t h = hint_row(a);
    uint32_t out0 = next_wire_;
    c_.program.push_back(OP_LIMBS8);
    c_.program.push_back(h);
    c_.program.push_back(n);
    c_.program.push_back(out0);
    std::vector<LC> limbs;
    LC sum;
    Fr pw = Fr::one();
    Fr k256 = Fr::from_u64(256);
    for (uint32_t i = 0; i < n; i++) {
      LC l = LC::wire(new_wire());
      limbs.push_back(l);
      sum = sum + l.scaled(pw);
      pw = pw * k256;
      lookup8(l);
    }
    assert_eq(sum, a);
    return limbs;
  }
  // queue v for membership in [0,256)
  void lookup8(const LC& v) { lookups_.push_back(v); }

  // Emits multiplicities, the commitment boundary and the log-derivative argument:
  //   sum_i 1/(X - v_i) == sum_j m_j/(X - j)   with X = H(commitment)
  void finalize_lookups();

  // native permutation hints (wires x^2,x^4,x^5 per S-box are created by the gadget with solve=false)
  void emit_poseidon_hint(uint32_t t, uint32_t h0, uint32_t out0) {
    c_.program.push_back(OP_POSEIDON);
    c_.program.push_back(t);
    c_.program.push_back(h0);
    c_.program.push_back(out0);
  }
  void emit_poseidon2_hint(uint32_t h0, uint32_t out0) {
    c_.program.push_back(OP_POSEIDON2);
    c_.program.push_back(h0);
    c_.program.push_back(out0);
  }

  Circuit finish();

  Circuit& raw() { return c_; }
  uint32_t last_div_wire() const { return last_div_wire_; }
  std::vector<uint32_t>& program() { return c_.program; }
  std::vector<Fr>& aux() { return c_.aux; }

 private:
  uint32_t coeff_index(const Fr& c) {
    std::string key((const char*)c.l, 32);
    auto it = coeff_map_.find(key);
    if (it != coeff_map_.end()) return it->second;
    uint32_t idx = (uint32_t)c_.coeffs.size();
    c_.coeffs.push_back(c);
    coeff_map_[key] = idx;
    return idx;
  }
  void push_row(Sparse& m, const LC& a) {
    for (auto& e : a.t) m.terms.push_back({e.first, coeff_index(e.second)});
    m.rowptr.push_back((uint32_t)m.terms.size());
  }
  void push_row_with_tail(Sparse& m, const LC& a, uint32_t tail_wire) {
    for (auto& e : a.t) m.terms.push_back({e.first, coeff_index(e.second)});
    m.terms.push_back({tail_wire, coeff_index(Fr::one())});
    m.rowptr.push_back((uint32_t)m.terms.size());
  }

  Circuit c_;
  uint32_t next_wire_;
  std::map<std::string, uint32_t> coeff_map_;
  std::vector<LC> lookups_;
  bool finalized_ = false;
  uint32_t last_div_wire_ = 0;
};

// ---- gadgets (circuit_gadgets.cpp) -----------------------------------------------------------------
struct PoseidonParams {
  int t, rf, rp;
  std::vector<Fr> rc;                 // (rf+rp)*t
  std::vector<std::vector<Fr>> mds;   // t x t
};
const PoseidonParams& poseidon_params(int t);          // Grain-LFSR generated, cached (t = 3, 5)
struct Poseidon2Params {
  std::vector<Fr> rc;   // 88
  Fr mu[4];
};
const Poseidon2Params& poseidon2_params();

LC gadget_poseidon_hash(Builder& b, const std::vector<LC>& inputs, bool native_hint);   // t = inputs+1
void gadget_poseidon2_permute(Builder& b, LC state[4], bool native_hint);
// Grumpkin fixed-base multiplication by the 254 little-endian bits of the scalar; returns (x, y)
std::pair<LC, LC> gadget_grumpkin_fixed_base(Builder& b, const std::vector<LC>& bits, bool native_hint);

GkAffine grumpkin_generator();
GkAffine grumpkin_offset();
Circuit build_withdraw_circuit(bool native_hints, uint32_t pad_to_constraints = 0, uint32_t depth = 16);
// Compiles a decoded ACIR program (spp/acir.py to_blob: AssertZero / RANGE / fixed-base Grumpkin MSM / Brillig hints) into an
// R1CS + solver program: the `sunspot compile <acir>` step (noir_circuit/prove_linux.sh:66-70) for nargo-compiled circuits.
// Returns false and sets *err on an unsupported construct.  csrc/circuit_acir.cpp
bool build_acir_circuit(const uint8_t* blob, size_t len, uint32_t circuit_id, Circuit* out, std::string* err);
Circuit build_audit_circuit(const uint32_t* pk_a, const uint32_t* pk_b, bool native_hints);

}  // namespace spp

// Circuit container I/O, lookup argument, Poseidon/Poseidon2 constants (Grain LFSR) and gadgets.
#include "circuit.hpp"

#include <algorithm>
#include <mutex>

namespace spp {

// =====================================================================================================
// SPPC file
// =====================================================================================================
static const uint32_t SPPC_MAGIC = 0x43505053u;  // "SPPC"
static const uint32_t SPPC_VERSION = 2;

static void put_u32(std::vector<uint8_t>& o, uint32_t v) {
  for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i)));
}
static void put_sparse(std::vector<uint8_t>& o, const Sparse& m) {
  put_u32(o, m.rows());
  put_u32(o, (uint32_t)m.terms.size());
  for (uint32_t v : m.rowptr) put_u32(o, v);
  for (auto& t : m.terms) {
    put_u32(o, t.wire);
    put_u32(o, t.coeff);
  }
}

bool Circuit::save(const std::string& path) const {
  std::vector<uint8_t> o;
  put_u32(o, SPPC_MAGIC);
  put_u32(o, SPPC_VERSION);
  put_u32(o, id);
  put_u32(o, n_public);
  put_u32(o, n_secret);
  put_u32(o, n_wires);
  put_u32(o, n_constraints);
  put_u32(o, domain_log);
  put_u32(o, challenge_wire);
  put_u32(o, (uint32_t)coeffs.size());
  put_u32(o, (uint32_t)committed.size());
  put_u32(o, (uint32_t)program.size());
  put_u32(o, (uint32_t)aux.size());
  for (auto& c : coeffs) {
    uint32_t cl[8];
    c.to_canonical(cl);
    for (int i = 0; i < 8; i++) put_u32(o, cl[i]);
  }
  put_sparse(o, A);
  put_sparse(o, B);
  put_sparse(o, C);
  put_sparse(o, H);
  for (uint32_t w : committed) put_u32(o, w);
  for (uint32_t w : program) put_u32(o, w);
  for (auto& a : aux) {
    uint32_t cl[8];
    a.to_canonical(cl);
    for (int i = 0; i < 8; i++) put_u32(o, cl[i]);
  }
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
  fclose(f);
  return ok;
}

namespace {
struct Reader {
  const uint8_t* p;
  size_t n, off = 0;
  bool ok = true;
  uint32_t u32() {
    if (off + 4 > n) {
      ok = false;
      return 0;
    }
    uint32_t v = (uint32_t)p[off] | ((uint32_t)p[off + 1] << 8) | ((uint32_t)p[off + 2] << 16) | ((uint32_t)p[off + 3] << 24);
    off += 4;
    return v;
  }
};
bool get_sparse(Reader& r, Sparse& m) {
  uint32_t rows = r.u32(), nnz = r.u32();
  if (!r.ok || (size_t)rows * 4 > r.n || (size_t)nnz * 8 > r.n) return false;
  m.rowptr.resize(rows + 1);
  for (auto& v : m.rowptr) v = r.u32();
  m.terms.resize(nnz);
  for (auto& t : m.terms) {
    t.wire = r.u32();
    t.coeff = r.u32();
  }
  return r.ok && m.rowptr[rows] == nnz;
}
}  // namespace

bool Circuit::load(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> buf((size_t)sz);
  bool rd = fread(buf.data(), 1, buf.size(), f) == buf.size();
  fclose(f);
  if (!rd) return false;
  Reader r{buf.data(), buf.size()};
  if (r.u32() != SPPC_MAGIC || r.u32() != SPPC_VERSION) return false;
  id = r.u32();
  n_public = r.u32();
  n_secret = r.u32();
  n_wires = r.u32();
  n_constraints = r.u32();
  domain_log = r.u32();
  challenge_wire = r.u32();
  uint32_t nc = r.u32(), ncm = r.u32(), np = r.u32(), naux = r.u32();
  if (!r.ok || (size_t)nc * 32 > buf.size()) return false;
  coeffs.resize(nc);
  for (auto& c : coeffs) {
    uint32_t cl[8];
    for (int i = 0; i < 8; i++) cl[i] = r.u32();
    c = Fr::from_canonical(cl);
  }
  if (!get_sparse(r, A) || !get_sparse(r, B) || !get_sparse(r, C) || !get_sparse(r, H)) return false;
  committed.resize(ncm);
  for (auto& w : committed) w = r.u32();
  program.resize(np);
  for (auto& w : program) w = r.u32();
  if ((size_t)naux * 32 > buf.size()) return false;
  aux.resize(naux);
  for (auto& a : aux) {
    uint32_t cl[8];
    for (int i = 0; i < 8; i++) cl[i] = r.u32();
    a = Fr::from_canonical(cl);
  }
  return r.ok;
}

// =====================================================================================================
// Builder: lookups + finish
// =====================================================================================================
void Builder::finalize_lookups() {
  if (finalized_) return;
  finalized_ = true;
  const uint32_t n = (uint32_t)lookups_.size();
  // hint rows of the looked-up values, consecutive
  uint32_t h0 = c_.H.rows();
  for (auto& v : lookups_) hint_row(v);
  // multiplicities m_0..m_255
  uint32_t m0 = next_wire_;
  c_.program.push_back(OP_COUNT8);
  c_.program.push_back(h0);
  c_.program.push_back(n);
  c_.program.push_back(m0);
  std::vector<LC> m;
  for (int j = 0; j < 256; j++) m.push_back(LC::wire(new_wire()));
  // committed wires: everything the looked-up values and multiplicities depend on
  std::vector<uint32_t> cw;
  for (auto& v : lookups_)
    for (auto& e : v.t)
      if (e.first != 0) cw.push_back(e.first);
  for (int j = 0; j < 256; j++) cw.push_back(m0 + j);
  // hiding: without it the commitment is a deterministic function of the committed values (limbs of secrets in a compiled
  // program, the RLWE noise in the audit circuit) and two proofs over related witnesses could be linked through it.  gnark's
  // api.Commit adds the same wire (hints.Randomize); here its value is derived from the proof's blinding factors.
  const uint32_t mask = new_wire();
  c_.program.push_back(OP_MASK);
  c_.program.push_back(mask);
  cw.push_back(mask);
  // A committed wire's basis point is built from its columns of A, B, C; a wire in no constraint would get the point at infinity
  // and mask nothing.  gnark gives its mask the row  mask * 1 = mask  (row 11 937 of the reference's .ccs); so does this builder.
  constrain(LC::wire(mask), LC::constant(Fr::one()), LC::wire(mask));
  std::sort(cw.begin(), cw.end());
  cw.erase(std::unique(cw.begin(), cw.end()), cw.end());
  c_.committed = cw;
  // phase boundary; X = H(commitment)
  c_.program.push_back(OP_COMMIT);
  c_.challenge_wire = new_wire();
  LC X = LC::wire(c_.challenge_wire);
  uint32_t k0 = n_constraints();
  LC sum;
  LC one = LC::constant(Fr::one());
  for (auto& v : lookups_) sum = sum + div(one, X - v, true);
  for (int j = 0; j < 256; j++) sum = sum - div(m[j], X - LC::constant_u64((uint64_t)j), true);
  emit_batch_div(k0, n + 256);
  assert_eq(sum, LC());
}

Circuit Builder::finish() {
  finalize_lookups();
  // input-consistency rows (one per public wire incl. the constant): w_i * 0 = 0, keeps the public
  // A-polynomials linearly independent
  for (uint32_t i = 0; i < c_.n_public; i++) constrain(LC::wire(i), LC(), LC());
  c_.program.push_back(OP_END);
  c_.n_wires = next_wire_;
  c_.n_constraints = c_.A.rows();
  uint32_t lg = 1;
  while ((1u << lg) < c_.n_constraints) lg++;
  c_.domain_log = lg;
  return c_;
}

// =====================================================================================================
// Grain LFSR parameter generation (Poseidon reference generator)
// =====================================================================================================
namespace {
struct Grain {
  uint8_t s[80];
  Grain(int t, int rf, int rp) {
    int k = 0;
    auto put = [&](uint32_t v, int w) {
      for (int i = w - 1; i >= 0; i--) s[k++] = (v >> i) & 1;
    };
    put(1, 2);
    put(0, 4);
    put(254, 12);
    put((uint32_t)t, 12);
    put((uint32_t)rf, 10);
    put((uint32_t)rp, 10);
    for (int i = 0; i < 30; i++) s[k++] = 1;
    for (int i = 0; i < 160; i++) step();
  }
  int step() {
    int nb = s[0] ^ s[13] ^ s[23] ^ s[38] ^ s[51] ^ s[62];
    memmove(s, s + 1, 79);
    s[79] = (uint8_t)nb;
    return nb;
  }
  int bit() {
    for (;;) {
      int a = step();
      int b = step();
      if (a) return b;
    }
  }
  // 254-bit sample, MSB first, as 8 canonical limbs
  void sample(uint32_t out[8]) {
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (int i = 253; i >= 0; i--)
      if (bit()) out[i / 32] |= 1u << (i % 32);
  }
  Fr field_rejection() {
    uint32_t v[8];
    for (;;) {
      sample(v);
      if (!Fr::geq_mod(v)) return Fr::from_canonical(v);
    }
  }
  Fr field_mod() {
    uint32_t v[8];
    sample(v);
    return Fr::from_u256(v);
  }
};
std::mutex g_param_mu;
std::map<int, PoseidonParams> g_poseidon;
Poseidon2Params* g_poseidon2 = nullptr;
}  // namespace

const PoseidonParams& poseidon_params(int t) {
  std::lock_guard<std::mutex> lk(g_param_mu);
  auto it = g_poseidon.find(t);
  if (it != g_poseidon.end()) return it->second;
  PoseidonParams p;
  p.t = t;
  p.rf = 8;
  p.rp = (t == 3) ? 57 : (t == 5) ? 60 : 56;
  Grain g(t, p.rf, p.rp);
  for (int i = 0; i < (p.rf + p.rp) * t; i++) p.rc.push_back(g.field_rejection());
  std::vector<Fr> xy;
  for (;;) {
    xy.clear();
    for (int i = 0; i < 2 * t; i++) xy.push_back(g.field_mod());
    bool dup = false;
    for (int i = 0; i < 2 * t && !dup; i++)
      for (int j = i + 1; j < 2 * t; j++)
        if (xy[i] == xy[j]) dup = true;
    if (!dup) break;
  }
  p.mds.assign(t, std::vector<Fr>(t));
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) p.mds[i][j] = (xy[i] + xy[t + j]).inv();
  return g_poseidon[t] = p;
}

const Poseidon2Params& poseidon2_params() {
  std::lock_guard<std::mutex> lk(g_param_mu);
  if (g_poseidon2) return *g_poseidon2;
  auto* p = new Poseidon2Params();
  Grain g(4, 8, 56);
  for (int i = 0; i < 88; i++) p->rc.push_back(g.field_rejection());
  for (int cand = 0; cand < 5; cand++) {
    Fr d[4];
    for (int i = 0; i < 4; i++) d[i] = g.field_mod();
    if (cand == 4)
      for (int i = 0; i < 4; i++) p->mu[i] = d[i] - Fr::one();
  }
  g_poseidon2 = p;
  return *p;
}

// =====================================================================================================
// Gadgets
// =====================================================================================================
// x^5 as x2 = x*x, x3 = x2*x, x4 = x3*x, x5 = x4*x: four constraints whose B side is always the S-box input, so the
// intermediate powers never enter the B matrix.  A wire on the B side costs a G1 and a G2 window walk (the G2 one
// three times as expensive); x2*x2 = x4 would save a constraint but put x2 there.  Net: -1 B wire, +1 A/K wire per
// S-box, about 8 % less MSM work per proof for the withdraw circuit.
// The audit circuit's Poseidon2 sponge (4 664 S-boxes) uses the same form since round 2: 29 178 constraints instead of
// 24 513, but 4 664 fewer wires in B -- the B1/B2 tables shrink by 21 GB, A/K grow by 14 GB, every set keeps its 11-bit
// windows, and a proof needs about 8 % fewer additions (G1-equivalent).  sbox5_compact is kept for reference.
static LC sbox5(Builder& b, const LC& x, bool solve) {
  LC x2 = b.mul(x, x, solve, false);
  LC x3 = b.mul(x2, x, solve, false);
  LC x4 = b.mul(x3, x, solve, false);
  return b.mul(x4, x, solve, false);
}
[[maybe_unused]] static LC sbox5_compact(Builder& b, const LC& x, bool solve) {
  LC x2 = b.mul(x, x, solve, false);
  LC x4 = b.mul(x2, x2, solve, false);
  return b.mul(x4, x, solve, false);
}

// noir_circuit/src/main.nr:1-9 (poseidon bn254 hash_2 / hash_4): state = [0, inputs...], out = state[0]
LC gadget_poseidon_hash(Builder& b, const std::vector<LC>& inputs, bool native_hint) {
  const int t = (int)inputs.size() + 1;
  const PoseidonParams& pp = poseidon_params(t);
  std::vector<LC> s(t);
  for (int i = 1; i < t; i++) s[i] = inputs[i - 1];
  if (native_hint) {
    uint32_t h0 = 0;
    for (int i = 0; i < t; i++) {
      uint32_t h = b.hint_row(s[i]);
      if (i == 0) h0 = h;
    }
    b.emit_poseidon_hint((uint32_t)t, h0, b.next_wire());
  }
  const bool solve = !native_hint;
  for (int r = 0; r < pp.rf + pp.rp; r++) {
    for (int i = 0; i < t; i++) s[i] = s[i] + LC::constant(pp.rc[r * t + i]);
    bool full = r < pp.rf / 2 || r >= pp.rf / 2 + pp.rp;
    if (full) {
      for (int i = 0; i < t; i++) s[i] = sbox5(b, s[i], solve);
    } else {
      s[0] = sbox5(b, s[0], solve);
    }
    std::vector<LC> n(t);
    for (int i = 0; i < t; i++)
      for (int j = 0; j < t; j++) n[i] = n[i] + s[j].scaled(pp.mds[i][j]);
    s = n;
  }
  return s[0];
}

static void p2_external(LC s[4]) {
  static const uint32_t ME[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
  LC n[4];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) n[i] = n[i] + s[j].scaled_u64(ME[i][j]);
  for (int i = 0; i < 4; i++) s[i] = n[i];
}

// ct_helper/src/main.nr:3,23,33 (std::hash::poseidon2_permutation, t=4)
void gadget_poseidon2_permute(Builder& b, LC s[4], bool native_hint) {
  const Poseidon2Params& pp = poseidon2_params();
  if (native_hint) {
    uint32_t h0 = b.hint_row(s[0]);
    for (int i = 1; i < 4; i++) b.hint_row(s[i]);
    b.emit_poseidon2_hint(h0, b.next_wire());
  }
  const bool solve = !native_hint;
  p2_external(s);
  int k = 0;
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 4; i++) s[i] = sbox5(b, s[i] + LC::constant(pp.rc[k + i]), solve);
    k += 4;
    p2_external(s);
  }
  for (int r = 0; r < 56; r++) {
    s[0] = sbox5(b, s[0] + LC::constant(pp.rc[k]), solve);
    k++;
    LC tot = s[0] + s[1] + s[2] + s[3];
    for (int i = 0; i < 4; i++) s[i] = s[i].scaled(pp.mu[i]) + tot;
  }
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 4; i++) s[i] = sbox5(b, s[i] + LC::constant(pp.rc[k + i]), solve);
    k += 4;
    p2_external(s);
  }
}

// ---- Grumpkin fixed-base multiplication (noir_circuit/src/main.nr:54-59, std fixed_base_scalar_mul) ----
// 64 windows of 4 bits; window j selects T_j[d] = (d+1)*16^j*G from constants (8 constraints), then one
// incomplete affine addition (3 constraints) onto an accumulator that starts at a fixed offset point O
// of unknown discrete log; the constant O + sum_j 16^j G is removed at the end.
static const char* GK_OFFSET_X = "203d7c39681b09f7b2a19414b3602833b72bd817490b6cb971aba33aec96fbb3";
static const char* GK_OFFSET_Y = "138ac000841409427e5bc866e15250e22cc65fc693498682491db022dbf0a37c";
static const char* GK_GEN_Y = "0000000000000002cf135e7506a45d632d270d45f1181294833fc48d823f272c";

Fr fr_from_hex(const char* h) {
  uint32_t c[8];
  for (int i = 0; i < 8; i++) {
    char buf[9];
    memcpy(buf, h + 8 * i, 8);
    buf[8] = 0;
    c[7 - i] = (uint32_t)strtoul(buf, nullptr, 16);
  }
  return Fr::from_canonical(c);
}

GkAffine grumpkin_generator() { return {Fr::one(), fr_from_hex(GK_GEN_Y)}; }
GkAffine grumpkin_offset() { return {fr_from_hex(GK_OFFSET_X), fr_from_hex(GK_OFFSET_Y)}; }

static std::pair<LC, LC> affine_add_incomplete(Builder& b, const LC& x1, const LC& y1, const LC& x2, const LC& y2, bool defer = false) {
  if (x1.is_constant() && y1.is_constant() && x2.is_constant() && y2.is_constant()) {
    GkXYZZ a = GkXYZZ::from_affine({x1.constant_value(), y1.constant_value()});
    a.madd({x2.constant_value(), y2.constant_value()});
    GkAffine r = a.to_affine();
    return {LC::constant(r.x), LC::constant(r.y)};
  }
  LC lam = b.div(y2 - y1, x2 - x1, defer);
  LC x3 = b.mul_sub(lam, lam, x1 + x2);
  LC y3 = b.mul_sub(lam, x1 - x3, y1);
  return {x3, y3};
}

std::pair<LC, LC> gadget_grumpkin_fixed_base(Builder& b, const std::vector<LC>& bits_in, bool native_hint) {
  std::vector<LC> bits = bits_in;
  // native hint: all 65 slopes of the ladder are produced by OP_GRUMPKIN from the scalar bits (two shared
  // inversions instead of 65); operands: first bit wire, #bits, aux offset, #slopes, slope wires (patched below)
  size_t patch = 0;
  uint32_t aux_off = (uint32_t)b.aux().size();
  std::vector<uint32_t> lam_wires;
  if (native_hint) {
    auto& pr = b.program();
    pr.push_back(OP_GRUMPKIN);
    pr.push_back(bits_in[0].t[0].first);
    pr.push_back((uint32_t)bits_in.size());
    pr.push_back(aux_off);
    pr.push_back(65);
    patch = pr.size();
    for (int i = 0; i < 65; i++) pr.push_back(0);
    for (size_t i = 1; i < bits_in.size(); i++)
      if (bits_in[i].t.size() != 1 || bits_in[i].t[0].first != bits_in[0].t[0].first + i) abort();  // consecutive bit wires
    b.aux().resize(aux_off + 4 + 64 * 16 * 2);
  }
  while (bits.size() < 256) bits.push_back(LC());
  GkAffine G = grumpkin_generator();
  GkAffine O = grumpkin_offset();
  LC ax = LC::constant(O.x), ay = LC::constant(O.y);
  if (native_hint) { b.aux()[aux_off] = O.x; b.aux()[aux_off + 1] = O.y; }
  GkXYZZ base = GkXYZZ::from_affine(G);       // 16^j * G
  GkXYZZ corr = GkXYZZ::from_affine(O);       // O + sum_j 16^j G
  for (int j = 0; j < 64; j++) {
    // table T[d] = (d+1)*base
    GkAffine T[16];
    GkXYZZ run = base;
    GkAffine base_aff = base.to_affine();
    for (int d = 0; d < 16; d++) {
      T[d] = run.to_affine();
      run.madd(base_aff);
    }
    corr.madd(base_aff);
    if (native_hint)
      for (int d = 0; d < 16; d++) {
        b.aux()[aux_off + 4 + (j * 16 + d) * 2] = T[d].x;
        b.aux()[aux_off + 4 + (j * 16 + d) * 2 + 1] = T[d].y;
      }
    // base for the next window is 16*base = T[15]
    base = GkXYZZ::from_affine(T[15]);
    const LC& b0 = bits[4 * j + 0];
    const LC& b1 = bits[4 * j + 1];
    const LC& b2 = bits[4 * j + 2];
    const LC& b3 = bits[4 * j + 3];
    LC b01 = b.mul(b0, b1);
    LC b23 = b.mul(b2, b3);
    LC sel[2];
    for (int coord = 0; coord < 2; coord++) {
      Fr v[16];
      for (int d = 0; d < 16; d++) v[d] = coord == 0 ? T[d].x : T[d].y;
      LC L[4];
      for (int h = 0; h < 4; h++) {
        const Fr* q = v + 4 * h;
        L[h] = LC::constant(q[0]) + b0.scaled(q[1] - q[0]) + b1.scaled(q[2] - q[0]) +
               b01.scaled(q[3] - q[2] - q[1] + q[0]);
      }
      LC p1 = b.mul(b2, L[1] - L[0]);
      LC p2 = b.mul(b3, L[2] - L[0]);
      LC p3 = b.mul(b23, L[3] - L[2] - L[1] + L[0]);
      sel[coord] = L[0] + p1 + p2 + p3;
    }
    auto r = affine_add_incomplete(b, ax, ay, sel[0], sel[1], native_hint);
    lam_wires.push_back(b.last_div_wire());
    ax = r.first;
    ay = r.second;
  }
  GkAffine N = corr.to_affine().neg();
  auto res = affine_add_incomplete(b, ax, ay, LC::constant(N.x), LC::constant(N.y), native_hint);
  lam_wires.push_back(b.last_div_wire());
  if (native_hint) {
    b.aux()[aux_off + 2] = N.x;
    b.aux()[aux_off + 3] = N.y;
    for (int i = 0; i < 65; i++) b.program()[patch + i] = lam_wires[i];
  }
  return res;
}

// =====================================================================================================
// Withdraw circuit: noir_circuit/src/main.nr:38-82
// =====================================================================================================
Circuit build_withdraw_circuit(bool native_hints, uint32_t pad_to_constraints, uint32_t depth) {
  Builder b(CIRCUIT_WITHDRAW);
  // public inputs, in the .pw order (withdraw.rs:74-90)
  LC root = b.public_input();
  LC nullifier = b.public_input();
  LC recipient = b.public_input();
  LC amount = b.public_input();
  LC wa_commitment = b.public_input();
  // private inputs, Prover.toml order (client/proof.helper.ts:41-50)
  LC secret_key = b.secret_input();
  LC owner_x = b.secret_input();
  LC owner_y = b.secret_input();
  LC randomness = b.secret_input();
  LC index = b.secret_input();
  std::vector<LC> siblings;
  for (uint32_t i = 0; i < depth; i++) siblings.push_back(b.secret_input());   // 16 in the reference (main.nr:11,49)

  // 1. secret_key * G == (owner_x, owner_y)        main.nr:52-62
  //    canonical 254-bit decomposition (lo = bits 0..127, hi = bits 128..253)
  std::vector<LC> skbits = b.to_bits(secret_key, 254);
  uint32_t rm1[8];
  for (int i = 0; i < 8; i++) rm1[i] = FrParams::MOD(i);
  rm1[0] -= 1;
  b.assert_bits_leq_const(skbits, rm1);
  auto pk = gadget_grumpkin_fixed_base(b, skbits, native_hints);
  b.assert_eq(pk.first, owner_x);
  b.assert_eq(pk.second, owner_y);

  // 2. wa_commitment == Poseidon(owner_x, owner_y)  main.nr:64-67
  b.assert_eq(gadget_poseidon_hash(b, {owner_x, owner_y}, native_hints), wa_commitment);

  // amount: pub u64                                  main.nr:43
  b.to_limbs8(amount, 8);

  // 3. commitment = Poseidon(owner_x, owner_y, amount, randomness)   main.nr:69-70
  LC commitment = gadget_poseidon_hash(b, {owner_x, owner_y, amount, randomness}, native_hints);

  // 4. nullifier == Poseidon(secret_key, index)      main.nr:72-74
  b.assert_eq(gadget_poseidon_hash(b, {secret_key, index}, native_hints), nullifier);

  // 5. Merkle membership, depth 16 in the reference  main.nr:11-29,76-78
  std::vector<LC> path = b.to_bits(index, depth);
  LC cur = commitment;
  for (uint32_t i = 0; i < depth; i++) {
    LC d = b.mul(path[i], siblings[i] - cur);  // bit ? sibling-cur : 0
    LC left = cur + d;
    LC right = siblings[i] - d;
    cur = gadget_poseidon_hash(b, {left, right}, native_hints);
  }
  b.assert_eq(cur, root);

  // 6. recipient != 0                                main.nr:80-81
  b.div(LC::constant(Fr::one()), recipient);

  // Optional ballast (SPP_CIRCUIT_WITHDRAW_REFSHAPE): the same statement padded to the size of the reference's gnark
  // R1CS (12 452 constraints, domain 2^14; its Grumpkin arithmetic runs on an emulated field).  Each step is one
  // multiplication of full-size values, t <- (t + k) * t, so the added wires cost the MSMs and NTTs what real
  // constraints cost.  Soundness of the statement is unchanged: the chain constrains only its own wires.
  if (pad_to_constraints) {
    Circuit probe = Builder(b).finish();          // lookups/commitment add constraints at finish(): measure them
    uint32_t have = probe.n_constraints;
    LC t = randomness;
    for (uint32_t k = 0; have + k < pad_to_constraints; k++) t = b.mul(t + LC::constant(Fr::from_u64(k + 1)), t);
  }
  return b.finish();
}

}  // namespace spp

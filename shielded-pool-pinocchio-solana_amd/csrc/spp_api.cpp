// libspp C ABI (include/spp.h), core: contexts, circuit loading (window tables in HBM), the batched proving pipeline and the
// trusted setup.  Everything heavy runs on the GPU; the host parses containers, derives one-time constants and enqueues
// kernels.  There is deliberately no CPU implementation of the hot path here.
#include "spp_internal.hpp"
#include <chrono>

thread_local char g_spp_err[512] = "";
extern "C" const char* spp_last_error(void) { return g_spp_err; }
extern "C" const char* spp_version(void) { return "libspp 0.2 (gfx950)"; }

template <class F>
struct MsmSet {
  uint32_t N = 0;
  Affine<F>* table = nullptr;
  uint32_t* rows = nullptr;
  bool from_h = false;   // scalars come from the h array instead of the witness
  uint32_t c = 0;        // window bits of this set's table
  uint32_t Wt = 0;       // table rows per base: msm_windows(c) = one per window (no passes), 1 = one row and msm_windows(c) passes
};
template <class F>
struct MsmBuf {
  XYZZ<F>* partial = nullptr;
  XYZZ<F>* out = nullptr;
  size_t partial_cap = 0;   // elements allocated in `partial`
  MsmPlan plan{};           // lane layout of the last launch (the fold needs it)
};
struct Workspace {
  hipStream_t st = nullptr;
  hipStream_t st2 = nullptr;          // side stream: the G2 MSM only needs the witness, so it runs beside matrix eval / NTT / G1 MSMs
  hipStream_t own_st = nullptr, own_st2 = nullptr;   // the streams of the pipelined mode (st / st2 point at them unless serialised)
  hipStream_t own_st2p = nullptr;                    // side stream with a priority of its own: used by batches (see its creation)
  std::pair<hipEvent_t, hipEvent_t> g2_ev{nullptr, nullptr};   // dispatch timestamps of the G2 MSM kernel
  hipEvent_t ev_w = nullptr, ev_b2 = nullptr;
  size_t cap = 0, last_P = 0;
  Fr *W = nullptr, *abc = nullptr, *scratch = nullptr;
  G1Affine* commit_affine = nullptr;
  uint8_t *d_inputs = nullptr, *d_rs = nullptr, *d_proofs = nullptr, *d_pws = nullptr;
  uint32_t* d_status = nullptr;
  uint32_t* counters = nullptr;   // [256][P] lookup histogram
  MsmBuf<Fq> A, B1, K, Z, CB, CS;
  MsmBuf<Fq> sA, rB;                  // small batches: s*Ar and r*Bs1 as table sums over the scaled witness (Ws, Wr)
  Fr *Ws = nullptr, *Wr = nullptr;
  MsmBuf<Fq2> B2;
  // signed-digit planes of the scalars of one MSM (kernels_msm.hip): dig1 is shared by the G1 sets, which run one after the
  // other on `st`; the G2 set runs beside them on the side stream and has its own
  int16_t *dig1 = nullptr, *dig2 = nullptr;
  size_t dig1_cap = 0, dig2_cap = 0;
  int16_t* small = nullptr;           // [sm_nslots][P]: byte-ranged wires as integers (small rows of the matrix evaluation)
  void* audit_scratch = nullptr;      // temporaries of the audit input pipeline (spp_prove_audit_from_secrets_device), P = cap
  size_t audit_scratch_cap = 0;
  std::vector<void*> owned;
  hipEvent_t ev[8] = {};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> msm_ev;
  size_t msm_ev_used = 0;
};

template <class F>
struct PendingTable {
  std::vector<Affine<F>> pts;
  Affine<F>* table;
  uint32_t c, Wt;
};
struct SolveStep {
  enum Kind { SEQ, BATCH_DIV, COUNT8, COMMIT } kind;
  uint32_t a = 0, b = 0, c = 0;   // SEQ: [pc_begin, pc_end) ; BATCH_DIV: k0, n ; COUNT8: h0, n, out0
  // SEQ: the same stretch as items of the cooperative solver (small batches), dealt over independent tracks (coop_plan)
  uint32_t ntracks = 0, tr_begin[COOP_TRACKS] = {}, tr_end[COOP_TRACKS] = {};
};
// batches up to this size are solved by one wave per proof (k_solve_coop); above it the wave-per-64-proofs solver has the
// better throughput (a cooperative wave runs ~1/3 of the dependent instructions, but 64 times as many waves)
static const uint32_t COOP_MAX_BATCH = [] {   // SPP_COOP_MAX (experiment) overrides
  const char* e = getenv("SPP_COOP_MAX");
  return e ? (uint32_t)atoi(e) : 1024u;
}();
// Up to a batch size that depends on the circuit s*Ar and r*Bs1 are two more fixed-base sums (sets A and B1 over the witness scaled by s and r) instead of
// 254 doublings on one lane each: 3 ms of a single proof's 9.  The sums cost a third of a proof's table additions, so a batch
// keeps the per-lane multiplication (its latency is shared by the whole batch).
// Measured (profiles/batch_size_sweep.py): the two extra sums cost ~15 us per withdraw proof and ~60 us per audit proof, the
// per-lane multiplication 3.3 ms per batch whatever its size -- so the switch is on the number of scaled scalars, P * (N_A + N_B1).
static constexpr uint64_t SCALED_BLIND_MAX_SCALARS = 1500000;
static uint32_t scaled_blind_max_batch(uint32_t n_a, uint32_t n_b1) {
  const uint64_t n = (uint64_t)n_a + n_b1;
  return n ? (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(1024, SCALED_BLIND_MAX_SCALARS / n)) : 1;
}
struct spp_circuit {
  spp_ctx* ctx = nullptr;
  std::vector<SolveStep> schedule;
  Circuit circ;
  DevCircuit dc{};
  uint32_t c_bits = 10, n = 0, logn = 0;
  uint32_t max_batch_div = SOLVE_SCRATCH_MIN_ROWS;
  DevCoop coop{};
  // How h = (A B - C) / Z reaches Krs (SPP_H_MODE, default 2):
  //   0  gnark's computeH: 3 inverse + 3 coset-forward + 1 coset-inverse transform, h coefficients against pk.G1.Z
  //   1  the H bases moved to the evaluation basis on the coset g*H at load: six transforms
  //   2  product form: h is the HIGH HALF of the product polynomial A(X) B(X) (A B = h (X^n - 1) + C with deg C < n), whose
  //      coefficients are a linear functional of its values on the 2n-th roots of unity H u zeta*H.  On H the values are a_i b_i =
  //      c_i = <C_i, w> -- linear in the witness, folded into per-wire bases at load; on zeta*H they need the transforms of A and
  //      B only: FOUR transforms, no transform of C, the same group element (spp_load_circuit, "product form")
  int h_mode = 2;
  bool generic_solver = false;    // the program is the solver of a decoded gnark system (OP_SOLVE_ROW ...): ~12 K dependent row solves per
                                  // proof on one lane -- a batch's solver phase outlasts the rest of it, so three batches take turns
  bool no_coop = false;           // SPP_NO_COOP=1 (diagnostic): always the one-lane-per-proof solver
  bool trace_items = false;       // SPP_COOP_TRACE=1 (diagnostic): one launch per item of the cooperative solver
  bool one_track = false;         // SPP_COOP_ONE_TRACK=1 (diagnostic): the independent tracks of a stretch one after the other
  bool no_level_stream = false;   // SPP_NO_LEVEL_STREAM=1 (diagnostic): table-driven level items instead of the LDS-staged stream
  uint32_t row_r = 0, row_s = 0, row_rs = 0, n_rows = 0;
  uint64_t table_bytes = 0;
  MsmSet<Fq> A, B1, K, Z, CB, CS;
  MsmSet<Fq2> B2;
  Fr *tw_fwd = nullptr, *tw_inv = nullptr, *coset_br = nullptr, *coset_inv_br = nullptr;
  Fr zinv;
  // device copies owned here
  std::vector<void*> owned;
  Workspace ws[SPP_NWS];
  int next_ws = 0, last_ws = 0, prev_ws = 0;   // prev_ws: the workspace of the batch before the last one (spp_timings which = 1)
  std::vector<PendingTable<Fq>> pending1;    // tables allocated but not yet built (spp_load_circuit)
  std::vector<PendingTable<Fq2>> pending2;
};
static std::vector<PendingTable<Fq>>& pending(spp_circuit* c, Fq*) { return c->pending1; }
static std::vector<PendingTable<Fq2>>& pending(spp_circuit* c, Fq2*) { return c->pending2; }

template <class T>
static int own_upload(spp_circuit* c, T** dst, const std::vector<T>& src) {
  HIP_TRY(dev_upload(dst, src));
  c->owned.push_back((void*)*dst);
  return 0;
}
// Item list of the cooperative solver (kernels_solve.hip, k_solve_coop) for every sequential stretch of the schedule.
// Nothing here changes what is computed: permutations become their lane-parallel form, runs of SOLVE_C rows are ordered
// by dependency level (level of a row = 1 + the highest level among the rows of the run that write one of its inputs),
// runs of independent BITS / LIMBS8 / INV_H instructions go one per lane, the rest stays on lane 0.  Items that share no
// wire (directly or through other items of the stretch) form independent components; the components are dealt over up to
// COOP_TRACKS waves per proof, longest first (the Merkle chain beside the key derivation; the ciphertext sponge beside the
// rest of the audit circuit).  Components that use the per-proof scratch rows stay together on track 0.
static int coop_plan(spp_circuit* c) {
  const Circuit& circ = c->circ;
  const auto& pr = circ.program;
  struct Item { uint32_t kind, a, b; };
  std::vector<uint32_t> items, par, lvl_ptr{0}, lvl_rows, stream;
  const Fr f_one = Fr::one(), f_mone = Fr::one().neg();
  auto coeff_word = [&](uint32_t ci) -> uint32_t {
    return ci | (circ.coeffs[ci] == f_one ? COEFF_ONE : circ.coeffs[ci] == f_mone ? COEFF_MINUS_ONE : 0u);
  };
  // one row of the level stream (see DevCoop::lvl_stream)
  auto row_record = [&](uint32_t k, std::vector<uint32_t>& out) {
    const bool square = [&] {
      const uint32_t a0 = circ.A.rowptr[k], a1 = circ.A.rowptr[k + 1], b0 = circ.B.rowptr[k], b1 = circ.B.rowptr[k + 1];
      if (a1 - a0 != b1 - b0) return false;
      for (uint32_t i = 0; i < a1 - a0; i++)
        if (circ.A.terms[a0 + i].wire != circ.B.terms[b0 + i].wire || circ.A.terms[a0 + i].coeff != circ.B.terms[b0 + i].coeff) return false;
      return true;
    }();
    const uint32_t nA = square ? 0 : circ.A.rowptr[k + 1] - circ.A.rowptr[k], nB = circ.B.rowptr[k + 1] - circ.B.rowptr[k],
                   nC = circ.C.rowptr[k + 1] - circ.C.rowptr[k] - 1;
    out.push_back(circ.C.terms[circ.C.rowptr[k + 1] - 1].wire);
    out.push_back(nA | (square ? 0x80000000u : 0u));
    out.push_back(nB);
    out.push_back(nC);
    auto put = [&](const Sparse& m, uint32_t n) {
      for (uint32_t t = m.rowptr[k]; t < m.rowptr[k] + n; t++) { out.push_back(m.terms[t].wire); out.push_back(coeff_word(m.terms[t].coeff)); }
    };
    put(circ.A, nA); put(circ.B, nB); put(circ.C, nC);
  };
  std::vector<uint32_t> level_of(circ.n_wires + 3, 0), stamp(circ.n_wires + 3, 0), writer(circ.n_wires + 3, 0);
  uint32_t epoch = 0;
  auto op_len = [&](size_t pc) -> uint32_t {
    switch (pr[pc]) {
      case OP_SOLVE_C: case OP_SOLVE_A: case OP_MASK: return 2;
      case OP_BATCH_DIV: case OP_POSEIDON2: case OP_INV_H: return 3;
      case OP_COUNT8: case OP_BITS: case OP_LIMBS8: case OP_POSEIDON: return 4;
      case OP_COMMIT: return 1;
      case OP_GRUMPKIN: return 5 + pr[pc + 4];
      default: return 0;
    }
  };
  auto is_par_op = [&](uint32_t op) { return op == OP_BITS || op == OP_LIMBS8 || op == OP_INV_H; };
  // wires an instruction reads / writes, a rough cost in microseconds of a lone wave, whether it uses the scratch rows
  struct RW { std::vector<uint32_t> rd, wr; double cost = 0; bool scratch = false; };
  auto row_rd = [&](RW& x, const Sparse& m, uint32_t k, uint32_t skip_last) {
    for (uint32_t t = m.rowptr[k]; t + skip_last < m.rowptr[k + 1]; t++) x.rd.push_back(m.terms[t].wire);
  };
  auto solve_c_rw = [&](RW& x, uint32_t k) {
    row_rd(x, circ.A, k, 0); row_rd(x, circ.B, k, 0); row_rd(x, circ.C, k, 1);
    x.wr.push_back(circ.C.terms[circ.C.rowptr[k + 1] - 1].wire);
  };
  auto div_rw = [&](RW& x, uint32_t k) {
    row_rd(x, circ.B, k, 0); row_rd(x, circ.C, k, 0);
    x.wr.push_back(circ.A.terms[circ.A.rowptr[k]].wire);
  };
  auto op_rw = [&](RW& x, size_t pc, bool coop_form) {
    switch (pr[pc]) {
      case OP_SOLVE_C: solve_c_rw(x, pr[pc + 1]); x.cost += 5; break;
      case OP_SOLVE_A: div_rw(x, pr[pc + 1]); x.cost += 60; x.scratch = true; break;
      case OP_BATCH_DIV:
        for (uint32_t k = 0; k < pr[pc + 2]; k++) div_rw(x, pr[pc + 1] + k);
        x.cost += 60 + 10.0 * pr[pc + 2]; x.scratch = true;
        break;
      case OP_BITS: case OP_LIMBS8:
        row_rd(x, circ.H, pr[pc + 1], 0);
        for (uint32_t i = 0; i < pr[pc + 2]; i++) x.wr.push_back(pr[pc + 3] + i);
        x.cost += 5 + 0.2 * pr[pc + 2];
        break;
      case OP_INV_H: row_rd(x, circ.H, pr[pc + 1], 0); x.wr.push_back(pr[pc + 2]); x.cost += 40; break;
      case OP_MASK: x.wr.push_back(pr[pc + 1]); x.cost += 5; break;
      case OP_POSEIDON: {
        const uint32_t t = pr[pc + 1], nsbox = 8 * t + (t == 3 ? 57 : 60);
        for (uint32_t i = 0; i < t; i++) row_rd(x, circ.H, pr[pc + 2] + i, 0);
        for (uint32_t i = 0; i < 4 * nsbox; i++) x.wr.push_back(pr[pc + 3] + i);
        x.cost += coop_form ? 175 : 450;
        break;
      }
      case OP_POSEIDON2:
        for (uint32_t i = 0; i < 4; i++) row_rd(x, circ.H, pr[pc + 1] + i, 0);
        for (uint32_t i = 0; i < 4 * 88; i++) x.wr.push_back(pr[pc + 2] + i);
        x.cost += coop_form ? 185 : 480;
        break;
      case OP_GRUMPKIN:
        for (uint32_t i = 0; i < pr[pc + 2]; i++) x.rd.push_back(pr[pc + 1] + i);
        for (uint32_t i = 0; i < pr[pc + 4]; i++) x.wr.push_back(pr[pc + 5 + i]);
        x.cost += coop_form ? 300 : 1800;
        x.scratch = x.scratch || !coop_form;
        break;
      default: break;
    }
  };
  for (SolveStep& st : c->schedule) {
    if (st.kind != SolveStep::SEQ) continue;
    std::vector<Item> its;
    size_t pc = st.a, seq0 = st.a;
    auto push = [&](uint32_t kind, uint32_t a, uint32_t b) { its.push_back({kind, a, b}); };
    auto flush = [&](size_t end) {
      if (end > seq0) push(COOP_SEQ, (uint32_t)seq0, (uint32_t)end);
    };
    while (pc < st.b) {
      const uint32_t op = pr[pc];
      if (op == OP_POSEIDON || op == OP_POSEIDON2 || (op == OP_GRUMPKIN && pr[pc + 4] >= 64 && pr[pc + 4] <= 65)) {
        flush(pc);
        push(op == OP_POSEIDON ? COOP_POSEIDON : op == OP_POSEIDON2 ? COOP_POSEIDON2 : COOP_GRUMPKIN, (uint32_t)pc, 0);
        pc += op_len(pc);
        seq0 = pc;
      } else if (op == OP_SOLVE_C) {
        size_t e = pc;
        while (e < st.b && pr[e] == OP_SOLVE_C) e += 2;
        const size_t nrows = (e - pc) / 2;
        if (nrows < 8) { pc = e; continue; }
        flush(pc);
        epoch++;
        std::vector<std::pair<uint32_t, uint32_t>> rows;   // (level, constraint)
        uint32_t max_level = 0;
        for (size_t q = pc; q < e; q += 2) {
          const uint32_t k = pr[q + 1];
          uint32_t lv = 0;
          auto scan = [&](const Sparse& m, uint32_t skip_last) {
            for (uint32_t t = m.rowptr[k]; t + skip_last < m.rowptr[k + 1]; t++) {
              const uint32_t w = m.terms[t].wire;
              if (stamp[w] == epoch) lv = std::max(lv, level_of[w]);
            }
          };
          scan(circ.A, 0); scan(circ.B, 0); scan(circ.C, 1);
          const uint32_t out = circ.C.terms[circ.C.rowptr[k + 1] - 1].wire;
          stamp[out] = epoch;
          level_of[out] = lv + 1;
          rows.push_back({lv, k});
          max_level = std::max(max_level, lv);
        }
        // rows of the run that share no wire written inside it are independent of each other: one LEVELS item per connected
        // component (small ones lumped together), so that the tracks below can take them apart (a compiled program keeps the
        // key derivation and the hash chain in the same run of rows)
        std::vector<uint32_t> rp(rows.size());
        for (size_t i = 0; i < rows.size(); i++) rp[i] = (uint32_t)i;
        auto rfind = [&](uint32_t x) { while (rp[x] != x) x = rp[x] = rp[rp[x]]; return x; };
        epoch++;
        for (size_t i = 0; i < rows.size(); i++) {
          const uint32_t k = rows[i].second;
          auto link = [&](const Sparse& m, uint32_t skip_last) {
            for (uint32_t t = m.rowptr[k]; t + skip_last < m.rowptr[k + 1]; t++) {
              const uint32_t w = m.terms[t].wire;
              if (stamp[w] == epoch) { const uint32_t ra = rfind((uint32_t)i), rb = rfind(writer[w]); if (ra != rb) rp[ra] = rb; }
            }
          };
          link(circ.A, 0); link(circ.B, 0); link(circ.C, 1);
          const uint32_t out = circ.C.terms[circ.C.rowptr[k + 1] - 1].wire;
          stamp[out] = epoch;
          writer[out] = (uint32_t)i;
        }
        std::vector<uint32_t> comp_size(rows.size(), 0), comp_id(rows.size(), 0);
        for (size_t i = 0; i < rows.size(); i++) comp_size[rfind((uint32_t)i)]++;
        const uint32_t MISC = 0xffffffffu;
        std::vector<uint32_t> comp_order;     // big components in order of first appearance, then the lump of small ones
        bool any_misc = false;
        for (size_t i = 0; i < rows.size(); i++) {
          const uint32_t r = rfind((uint32_t)i);
          if (comp_size[r] < 32) { comp_id[i] = MISC; any_misc = true; continue; }
          comp_id[i] = r;
          if (std::find(comp_order.begin(), comp_order.end(), r) == comp_order.end()) comp_order.push_back(r);
        }
        if (any_misc) comp_order.push_back(MISC);
        std::vector<size_t> order(rows.size());
        for (size_t i = 0; i < rows.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return rows[x].first < rows[y].first; });
        for (uint32_t cid : comp_order) {
          const uint32_t l0 = (uint32_t)lvl_ptr.size() - 1;
          uint32_t cur = 0xffffffffu;
          std::vector<std::vector<uint32_t>> by_level;      // constraints of this component, level by level
          for (size_t oi : order) {
            if (comp_id[oi] != cid) continue;
            if (rows[oi].first != cur) {
              if (cur != 0xffffffffu) lvl_ptr.push_back((uint32_t)lvl_rows.size());
              cur = rows[oi].first;
              by_level.emplace_back();
            }
            lvl_rows.push_back(rows[oi].second);
            by_level.back().push_back(rows[oi].second);
          }
          lvl_ptr.push_back((uint32_t)lvl_rows.size());
          // the streamed form: levels as self-contained records in chunks of COOP_CHUNK words; a level too big for a chunk is cut
          // into consecutive sub-levels (its rows are independent), a single row too big for one sends the component down the
          // table-driven path
          std::vector<uint32_t> local;          // this component's chunks
          uint32_t used = 0;                    // words used in the current chunk
          bool fits = !c->no_level_stream;
          auto close_chunk = [&] {
            if (used < COOP_CHUNK) local.push_back(0xffffffffu), used++;
            local.resize(local.size() + (COOP_CHUNK - used), 0xffffffffu);
            used = 0;
          };
          for (const auto& lv : by_level) {
            if (!fits) break;
            size_t i = 0;
            while (i < lv.size() && fits) {
              // greedily take rows while the sub-level record fits one chunk
              std::vector<std::vector<uint32_t>> recs;
              uint32_t words = 2;
              while (i < lv.size()) {
                std::vector<uint32_t> rec;
                row_record(lv[i], rec);
                if (words + 1 + rec.size() > COOP_CHUNK - 1) break;
                words += 1 + (uint32_t)rec.size();
                recs.push_back(std::move(rec));
                i++;
              }
              if (recs.empty()) { fits = false; break; }
              if (used + words > COOP_CHUNK - 1 && used) close_chunk();
              // lanes per row: enough for the longest linear form of the sub-level (every extra doubling costs three shuffle-add
              // rounds), at most 16, and rows x lanes within the wave when possible
              uint32_t longest = 1;
              for (const auto& rec : recs) longest = std::max({longest, rec[1] & 0x7fffffffu, rec[2], rec[3]});
              uint32_t G = 1;
              while (G < 16 && G < longest && recs.size() * (G * 2) <= 64) G *= 2;
              uint32_t need = 0;
              for (const auto& rec : recs) need |= ((rec[1] & 0x7fffffffu) ? 1u << 29 : 0u) | (rec[3] ? 1u << 30 : 0u);
              local.push_back((uint32_t)recs.size() | (G << 24) | need);
              local.push_back(words);
              uint32_t off = 2 + (uint32_t)recs.size();
              for (const auto& rec : recs) { local.push_back(off); off += (uint32_t)rec.size(); }
              for (const auto& rec : recs) local.insert(local.end(), rec.begin(), rec.end());
              used += words;
            }
          }
          if (fits && !local.empty()) {
            if (used) close_chunk();
            const uint32_t chunk0 = (uint32_t)(stream.size() / COOP_CHUNK);
            stream.insert(stream.end(), local.begin(), local.end());
            push(COOP_LEVEL_STREAM, chunk0, (uint32_t)(local.size() / COOP_CHUNK));
          } else {
            push(COOP_LEVELS, l0, (uint32_t)lvl_ptr.size() - 1);
          }
        }
        pc = e;
        seq0 = pc;
      } else if (is_par_op(op)) {
        // maximal run of lane-independent instructions: none may read a wire an earlier one of the run writes
        size_t e = pc;
        epoch++;
        std::vector<uint32_t> group;
        while (e < st.b && is_par_op(pr[e])) {
          const uint32_t h = pr[e + 1];
          bool dep = false;
          for (uint32_t t = circ.H.rowptr[h]; t < circ.H.rowptr[h + 1]; t++) dep = dep || stamp[circ.H.terms[t].wire] == epoch;
          if (dep) break;
          if (pr[e] == OP_INV_H) stamp[pr[e + 2]] = epoch;
          else for (uint32_t i = 0; i < pr[e + 2]; i++) stamp[pr[e + 3] + i] = epoch;
          group.push_back((uint32_t)e);
          e += op_len(e);
        }
        if (group.size() < 4) { pc = group.empty() ? pc + op_len(pc) : e; continue; }
        flush(pc);
        const uint32_t g0 = (uint32_t)(par.size() / 2);
        for (uint32_t q : group) { par.push_back(q); par.push_back(q + op_len(q)); }
        push(COOP_PAR, g0, g0 + (uint32_t)group.size());
        pc = e;
        seq0 = pc;
      } else {
        const uint32_t n = op_len(pc);
        if (n == 0) return fail(SPP_ERR_FORMAT, "bad opcode %u in solver program", op);
        pc += n;
      }
    }
    flush(st.b);

    // ---- independent components of this stretch -> tracks ----
    const size_t n = its.size();
    std::vector<RW> rw(n);
    for (size_t i = 0; i < n; i++) {
      const Item& it = its[i];
      switch (it.kind) {
        case COOP_SEQ:
          for (size_t q = it.a; q < it.b; q += op_len(q)) op_rw(rw[i], q, false);
          break;
        case COOP_PAR:
          for (uint32_t g = it.a; g < it.b; g++) op_rw(rw[i], par[2 * g], false);
          rw[i].cost = 10 + rw[i].cost / 32;
          break;
        case COOP_LEVELS:
          for (uint32_t r = lvl_ptr[it.a]; r < lvl_ptr[it.b]; r++) solve_c_rw(rw[i], lvl_rows[r]);
          rw[i].cost = 4.5 * (it.b - it.a);
          break;
        case COOP_LEVEL_STREAM: {
          uint32_t nlev = 0;
          for (uint32_t ch = it.a; ch < it.a + it.b; ch++) {
            const uint32_t* sb = stream.data() + (size_t)ch * COOP_CHUNK;
            for (uint32_t pos = 0; pos < COOP_CHUNK && sb[pos] != 0xffffffffu; pos += sb[pos + 1]) {
              nlev++;
              for (uint32_t r = 0; r < (sb[pos] & 0xffffffu); r++) {
                const uint32_t base = pos + sb[pos + 2 + r];
                const uint32_t nt = (sb[base + 1] & 0x7fffffffu) + sb[base + 2] + sb[base + 3];
                rw[i].wr.push_back(sb[base]);
                for (uint32_t t = 0; t < nt; t++) rw[i].rd.push_back(sb[base + 4 + 2 * t]);
              }
            }
          }
          rw[i].cost = 3.0 * nlev;
          break;
        }
        default: op_rw(rw[i], it.a, true); break;
      }
    }
    std::vector<uint32_t> parent(n);
    for (size_t i = 0; i < n; i++) parent[i] = (uint32_t)i;
    auto find = [&](uint32_t x) { while (parent[x] != x) x = parent[x] = parent[parent[x]]; return x; };
    epoch++;
    for (size_t i = 0; i < n; i++) {
      for (uint32_t w : rw[i].rd)
        if (stamp[w] == epoch) { const uint32_t ra = find((uint32_t)i), rb = find(writer[w]); if (ra != rb) parent[ra] = rb; }
      for (uint32_t w : rw[i].wr) { stamp[w] = epoch; writer[w] = (uint32_t)i; }
    }
    std::vector<double> comp_cost(n, 0.0);
    std::vector<char> comp_scratch(n, 0);
    for (size_t i = 0; i < n; i++) { const uint32_t r = find((uint32_t)i); comp_cost[r] += rw[i].cost; comp_scratch[r] |= rw[i].scratch; }
    std::vector<uint32_t> roots;
    for (size_t i = 0; i < n; i++) if (find((uint32_t)i) == i) roots.push_back((uint32_t)i);
    std::sort(roots.begin(), roots.end(), [&](uint32_t x, uint32_t y) { return comp_cost[x] > comp_cost[y]; });
    double load[COOP_TRACKS] = {};
    std::vector<uint32_t> track_of(n, 0);
    for (uint32_t r : roots) if (comp_scratch[r]) { track_of[r] = 0; load[0] += comp_cost[r]; }
    for (uint32_t r : roots) {
      if (comp_scratch[r]) continue;
      uint32_t best = 0;
      for (uint32_t t = 1; t < COOP_TRACKS; t++) if (load[t] < load[best]) best = t;
      track_of[r] = best;
      load[best] += comp_cost[r];
    }
    st.ntracks = 0;
    for (uint32_t t = 0; t < COOP_TRACKS; t++) {
      st.tr_begin[t] = (uint32_t)(items.size() / 3);
      for (size_t i = 0; i < n; i++)
        if (track_of[find((uint32_t)i)] == t) { items.push_back(its[i].kind); items.push_back(its[i].a); items.push_back(its[i].b); }
      st.tr_end[t] = (uint32_t)(items.size() / 3);
      if (st.tr_end[t] > st.tr_begin[t]) st.ntracks = t + 1;
    }
  }
  if (items.empty()) items.assign(3, 0);
  if (par.empty()) par.assign(2, 0);
  if (lvl_rows.empty()) lvl_rows.push_back(0);
  if (stream.empty()) stream.assign(COOP_CHUNK, 0xffffffffu);
  uint32_t *d_items, *d_par, *d_lp, *d_lr, *d_ls;
  int e;
  if ((e = own_upload(c, &d_items, items)) || (e = own_upload(c, &d_par, par)) || (e = own_upload(c, &d_lp, lvl_ptr)) ||
      (e = own_upload(c, &d_lr, lvl_rows)) || (e = own_upload(c, &d_ls, stream)))
    return e;
  c->coop.items = d_items; c->coop.par = d_par; c->coop.lvl_ptr = d_lp; c->coop.lvl_rows = d_lr; c->coop.lvl_stream = d_ls;
  return 0;
}

// Small rows of the matrix evaluation (DevCircuit::sm_*).  A wire is "byte-ranged" when one of the looked-up values of an OP_COUNT8
// range check is exactly that wire, or that wire plus a small constant: the log-derivative argument then holds only if the wire's
// value lies in [-c, 255 - c].  A row of A, B or C with at least SMALL_ROW_MIN terms, all of them (small integer coefficient) x
// (byte-ranged wire or the constant one), is evaluated in 64-bit integer arithmetic from an int16 copy of those wires: the audit
// circuit's 1 088 quotient equations (1 024 public-key coefficients each) are 1.13 M of the 1.8 M matrix terms of a proof, and
// re-read the same 1 024 witness rows 1 088 times -- 73 GB of L2 misses per 2 048-proof batch in the general kernel (20 ms,
// profiles/round2_audit_b2048_pmc_hbm.json); as integers over a 13 MB array they take well under a millisecond.
// SPP_NO_SMALL_ROWS=1 (diagnostic): off.
static constexpr uint32_t SMALL_ROW_MIN = 64, SMALL_ROW_REST = 32;
static int small_rows_plan(spp_circuit* c, std::vector<uint8_t>& flags_out) {
  const Circuit& circ = c->circ;
  c->dc.sm_nrows = 0;
  c->dc.sm_nslots = 0;
  c->dc.row_small = nullptr;
  if (getenv("SPP_NO_SMALL_ROWS")) return 0;
  // signed small value of a coefficient-table entry, if it has one
  auto small_of = [&](uint32_t ci, int64_t* out) {
    uint32_t v[8];
    circ.coeffs[ci].to_canonical(v);
    bool hi0 = true;
    for (int k = 1; k < 8; k++) hi0 = hi0 && v[k] == 0;
    if (hi0 && v[0] < (1u << 30)) { *out = (int64_t)v[0]; return true; }
    circ.coeffs[ci].neg().to_canonical(v);
    hi0 = true;
    for (int k = 1; k < 8; k++) hi0 = hi0 && v[k] == 0;
    if (hi0 && v[0] < (1u << 30)) { *out = -(int64_t)v[0]; return true; }
    return false;
  };
  std::vector<int32_t> slot_of(circ.n_wires, -1);
  std::vector<uint32_t> wires{0};
  std::vector<int32_t> lo{0};
  slot_of[0] = 0;   // the constant one
  for (const SolveStep& st : c->schedule) {
    if (st.kind != SolveStep::COUNT8) continue;
    for (uint32_t h = st.a; h < st.a + st.b && h < circ.H.rows(); h++) {
      uint32_t w = 0, nw = 0;
      int64_t cst = 0;
      bool ok = true;
      for (uint32_t t = circ.H.rowptr[h]; t < circ.H.rowptr[h + 1] && ok; t++) {
        const Term& tm = circ.H.terms[t];
        int64_t v;
        if (!small_of(tm.coeff, &v)) { ok = false; break; }
        if (tm.wire == 0) cst += v;
        else if (v == 1) { w = tm.wire; nw++; }
        else ok = false;
      }
      if (!ok || nw != 1 || cst < -32000 || cst > 32000 || slot_of[w] >= 0) continue;
      slot_of[w] = (int32_t)wires.size();
      wires.push_back(w);
      lo.push_back((int32_t)-cst);
    }
  }
  if (wires.size() < 2) return 0;
  std::vector<uint32_t> rowptr{0}, slots, row_out, rest_ptr{0}, rest_wire, rest_coeff;
  std::vector<int32_t> coefs;
  std::vector<uint8_t> flags(std::max<uint32_t>(circ.n_constraints, 1), 0);
  const Sparse* mats[3] = {&circ.A, &circ.B, &circ.C};
  for (uint32_t mi = 0; mi < 3; mi++) {
    const Sparse& m = *mats[mi];
    for (uint32_t k = 0; k < circ.n_constraints; k++) {
      const uint32_t b = m.rowptr[k], e = m.rowptr[k + 1];
      if (e - b < SMALL_ROW_MIN || e - b > (1u << 20)) continue;
      // terms that qualify (small coefficient x byte-ranged wire) go to the integer sum, at most SMALL_ROW_REST others stay
      // field arithmetic (a quotient equation has nine: k * q and the eight message bits times Delta * 2^i)
      uint32_t n_small = 0;
      for (uint32_t t = b; t < e; t++) {
        int64_t v;
        if (slot_of[m.terms[t].wire] >= 0 && small_of(m.terms[t].coeff, &v)) n_small++;
      }
      if (n_small < SMALL_ROW_MIN || (e - b) - n_small > SMALL_ROW_REST) continue;
      for (uint32_t t = b; t < e; t++) {
        int64_t v = 0;
        if (slot_of[m.terms[t].wire] >= 0 && small_of(m.terms[t].coeff, &v)) {
          slots.push_back((uint32_t)slot_of[m.terms[t].wire]);
          coefs.push_back((int32_t)v);
        } else {
          rest_wire.push_back(m.terms[t].wire);
          rest_coeff.push_back(m.terms[t].coeff);
        }
      }
      rowptr.push_back((uint32_t)slots.size());
      rest_ptr.push_back((uint32_t)rest_wire.size());
      row_out.push_back((mi << 30) | k);
      flags[k] |= (uint8_t)(1u << mi);
    }
  }
  if (row_out.empty()) return 0;
  // a run of constraints shares ONE B evaluation (its first row's): the flag of the first row decides for the run, and the rows of
  // a run have identical B rows, so they qualify together
  uint32_t *d_w, *d_rp, *d_sl, *d_ro, *d_xp, *d_xw, *d_xc;
  int32_t *d_lo, *d_co;
  int e;
  if (rest_wire.empty()) { rest_wire.push_back(0); rest_coeff.push_back(0); }   // never read: keeps the uploads non-empty
  if ((e = own_upload(c, &d_w, wires)) || (e = own_upload(c, &d_lo, lo)) || (e = own_upload(c, &d_rp, rowptr)) || (e = own_upload(c, &d_sl, slots)) ||
      (e = own_upload(c, &d_co, coefs)) || (e = own_upload(c, &d_ro, row_out)) ||
      (e = own_upload(c, &d_xp, rest_ptr)) || (e = own_upload(c, &d_xw, rest_wire)) || (e = own_upload(c, &d_xc, rest_coeff)))
    return e;
  c->dc.sm_rest_ptr = d_xp; c->dc.sm_rest_wire = d_xw; c->dc.sm_rest_coeff = d_xc;
  c->dc.sm_wires = d_w; c->dc.sm_lo = d_lo; c->dc.sm_nslots = (uint32_t)wires.size();
  c->dc.sm_rowptr = d_rp; c->dc.sm_slot = d_sl; c->dc.sm_coef = d_co; c->dc.sm_row_out = d_ro; c->dc.sm_nrows = (uint32_t)row_out.size();
  flags_out = flags;
  return 0;
}

// the "already in abc" bits of k_spmv_check: small rows (above) and long rows (DevCircuit::lg_rows)
static constexpr uint32_t LONG_ROW_MIN = 512;
static int row_paths_plan(spp_circuit* c) {
  const Circuit& circ = c->circ;
  std::vector<uint8_t> small_flags;
  if (int e = small_rows_plan(c, small_flags)) return e;
  const uint32_t nc = std::max<uint32_t>(circ.n_constraints, 1);
  if (small_flags.empty()) small_flags.assign(nc, 0);
  std::vector<uint8_t> long_flags(nc, 0);
  std::vector<uint32_t> lg;
  c->dc.lg_n = 0;
  c->dc.lg_rows = nullptr;
  c->dc.row_long = nullptr;
  if (!getenv("SPP_NO_LONG_ROWS")) {
    const Sparse* mats[3] = {&circ.A, &circ.B, &circ.C};
    for (uint32_t mi = 0; mi < 3; mi++)
      for (uint32_t k = 0; k < circ.n_constraints; k++)
        if (mats[mi]->rowptr[k + 1] - mats[mi]->rowptr[k] > LONG_ROW_MIN && !(small_flags[k] & (1u << mi))) {
          lg.push_back((mi << 30) | k);
          long_flags[k] |= (uint8_t)(1u << mi);
        }
  }
  int e;
  if (!lg.empty()) {
    uint32_t* d_lg;
    uint8_t* d_lf;
    if ((e = own_upload(c, &d_lg, lg)) || (e = own_upload(c, &d_lf, long_flags))) return e;
    c->dc.lg_rows = d_lg;
    c->dc.lg_n = (uint32_t)lg.size();
    c->dc.row_long = d_lf;
  }
  if (c->dc.sm_nrows) {
    for (uint32_t k = 0; k < nc; k++) small_flags[k] |= long_flags[k];
    uint8_t* d_fl;
    if ((e = own_upload(c, &d_fl, small_flags))) return e;
    c->dc.row_small = d_fl;
  } else {
    c->dc.row_small = c->dc.row_long;
  }
  return 0;
}

static int upload_sparse(spp_circuit* c, const Circuit& circ, const Sparse& m, DevSparse* out) {
  std::vector<uint32_t> wire(m.terms.size()), coeff(m.terms.size()), lit(m.terms.size(), 0);
  Fr one = Fr::one(), mone = Fr::one().neg();
  // small literals, per coefficient-table entry: canonical value v < 2^28, or p - v < 2^28
  std::vector<uint32_t> small(circ.coeffs.size(), 0);
  for (size_t ci = 0; ci < circ.coeffs.size(); ci++) {
    uint32_t v[8], nv[8];
    circ.coeffs[ci].to_canonical(v);
    bool hi0 = true;
    for (int k = 1; k < 8; k++) hi0 = hi0 && v[k] == 0;
    if (hi0 && v[0] != 0 && v[0] < (1u << 28)) { small[ci] = v[0]; continue; }
    circ.coeffs[ci].neg().to_canonical(nv);
    hi0 = true;
    for (int k = 1; k < 8; k++) hi0 = hi0 && nv[k] == 0;
    if (hi0 && nv[0] != 0 && nv[0] < (1u << 28)) small[ci] = nv[0] | 0x80000000u;
  }
  for (size_t i = 0; i < m.terms.size(); i++) {
    wire[i] = m.terms[i].wire;
    uint32_t ci = m.terms[i].coeff;
    uint32_t flag = 0;
    if (circ.coeffs[ci] == one) flag = COEFF_ONE;
    else if (circ.coeffs[ci] == mone) flag = COEFF_MINUS_ONE;
    else lit[i] = small[ci];
    coeff[i] = ci | flag;
  }
  uint32_t *rp, *w, *co, *li;
  if (int e = own_upload(c, &li, lit)) return e;
  out->lit = li;
  if (int e = own_upload(c, &rp, m.rowptr)) return e;
  if (int e = own_upload(c, &w, wire)) return e;
  if (int e = own_upload(c, &co, coeff)) return e;
  out->rowptr = rp;
  out->wire = w;
  out->coeff = co;
  return 0;
}

// window tables: allocate first (all sets), then build with temporaries sized from the HBM that is left, so that
// each launch has enough rows (>= tens of thousands of lanes) to fill the chip
template <class F>
static int alloc_table(spp_circuit* c, size_t N, uint32_t cbits, uint32_t Wt, Affine<F>** table_out) {
  size_t table_elems = std::max<size_t>(msm_table_elems((uint32_t)N, cbits, Wt), 1);
  Affine<F>* table;
  HIP_TRY(hipMalloc((void**)&table, table_elems * sizeof(Affine<F>)));
  c->owned.push_back(table);
  c->table_bytes += table_elems * sizeof(Affine<F>);
  *table_out = table;
  return 0;
}
template <class F>
static int build_table(spp_circuit* c, const std::vector<Affine<F>>& pts, uint32_t cbits, uint32_t Wt, Affine<F>* table, size_t temp_budget) {
  hipStream_t st = c->ctx->stream;
  const uint32_t Wn = Wt, E = 1u << (cbits - 1);
  const size_t N = pts.size();
  if (N == 0) return 0;
  const size_t rows_total = ((N * Wn + 63) / 64) * 64;
  const size_t per_row = (size_t)E * (sizeof(XYZZ<F>) + sizeof(F));
  size_t chunk = std::max<size_t>(64, ((temp_budget / per_row) / 64) * 64);
  chunk = std::min(chunk, (size_t)65536);   // larger launches only add TLB misses (the d-stride is chunk * 128 B)
  chunk = std::min(chunk, rows_total);
  DevBuf d_bases, tmp, tmp_pre;   // released on every return path
  HIP_TRY(d_bases.alloc(N * sizeof(Affine<F>)));
  HIP_TRY(hipMemcpy(d_bases.p, pts.data(), N * sizeof(Affine<F>), hipMemcpyHostToDevice));
  HIP_TRY(tmp.alloc(chunk * E * sizeof(XYZZ<F>)));
  HIP_TRY(tmp_pre.alloc(chunk * E * sizeof(F)));
  for (size_t r0 = 0; r0 < rows_total; r0 += chunk) {
    uint32_t cnt = (uint32_t)std::min(chunk, rows_total - r0);
    launch_build_table<F>(st, d_bases.as<Affine<F>>(), (uint32_t)N, cbits, Wt, (uint32_t)r0, cnt, table, tmp.as<XYZZ<F>>(), tmp_pre.as<F>());
  }
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return 0;
}
static size_t table_temp_budget() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)2 << 30;
  size_t b = free_b / 2;                       // leave room for the batch workspaces
  b = std::min(b, (size_t)48 << 30);
  return std::max(b, (size_t)1 << 28);
}
template <class F>
static int build_table_chunked(spp_circuit* c, const std::vector<Affine<F>>& pts, uint32_t cbits, uint32_t Wt, Affine<F>** table_out) {
  if (int e = alloc_table<F>(c, pts.size(), cbits, Wt, table_out)) return e;
  return build_table<F>(c, pts, cbits, Wt, *table_out, std::min(table_temp_budget(), (size_t)2 << 30));
}


template <class F>
static int make_set(spp_circuit* c, MsmSet<F>* set, const std::vector<uint32_t>& rows, const std::vector<Affine<F>>& pts, bool from_h,
                    uint32_t cbits, bool flat) {
  set->N = (uint32_t)pts.size();
  set->from_h = from_h;
  set->c = cbits;
  set->Wt = flat ? 1 : msm_windows(cbits);
  if (int e = own_upload(c, &set->rows, rows)) return e;
  if (int e = alloc_table<F>(c, pts.size(), cbits, set->Wt, &set->table)) return e;
  pending(c, (F*)nullptr).push_back({pts, set->table, cbits, set->Wt});
  return 0;
}
static int build_pending(spp_circuit* c) {
  const size_t budget = table_temp_budget();
  int e = 0;
  for (auto& p : pending(c, (Fq*)nullptr)) if (!e) e = build_table<Fq>(c, p.pts, p.c, p.Wt, p.table, budget);
  for (auto& p : pending(c, (Fq2*)nullptr)) if (!e) e = build_table<Fq2>(c, p.pts, p.c, p.Wt, p.table, budget);
  pending(c, (Fq*)nullptr).clear();
  pending(c, (Fq2*)nullptr).clear();
  return e;
}

// -----------------------------------------------------------------------------------------------------
// circuit construction (host only)
// -----------------------------------------------------------------------------------------------------
extern "C" int spp_circuit_build(int circuit_id, const uint32_t* aux, const char* out_path, uint32_t* n_constraints) {
  if (!out_path) return fail(SPP_ERR_BAD_INPUT, "out_path is NULL");
  Circuit c;
  if (circuit_id == SPP_CIRCUIT_WITHDRAW) {
    c = build_withdraw_circuit(true);
  } else if (circuit_id == SPP_CIRCUIT_WITHDRAW_REFSHAPE) {
    c = build_withdraw_circuit(true, 12452);
  } else if (circuit_id == SPP_CIRCUIT_WITHDRAW_DEPTH20) {
    c = build_withdraw_circuit(true, 0, 20);
  } else if (circuit_id == SPP_CIRCUIT_AUDIT) {
    if (!aux) return fail(SPP_ERR_BAD_INPUT, "audit circuit needs the RLWE public key (aux)");
    c = build_audit_circuit(aux, aux + 1024, true);
  } else {
    return fail(SPP_ERR_BAD_INPUT, "unknown circuit id %d", circuit_id);
  }
  if (n_constraints) *n_constraints = c.n_constraints;
  if (!c.save(out_path)) return fail(SPP_ERR_IO, "cannot write %s", out_path);
  return SPP_OK;
}

// `sunspot compile <acir>` for a nargo-compiled program: blob = spp/acir.py to_blob() (the decoded opcode list)
extern "C" int spp_circuit_build_acir(const uint8_t* blob, size_t blob_len, int circuit_id, const char* out_path, uint32_t* n_constraints) {
  if (!blob || !out_path) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  Circuit c;
  std::string err;
  if (!build_acir_circuit(blob, blob_len, circuit_id > 0 ? (uint32_t)circuit_id : CIRCUIT_ACIR, &c, &err))
    return fail(SPP_ERR_FORMAT, "ACIR program not supported: %s", err.c_str());
  if (n_constraints) *n_constraints = c.n_constraints;
  if (!c.save(out_path)) return fail(SPP_ERR_IO, "cannot write %s", out_path);
  return SPP_OK;
}

// -----------------------------------------------------------------------------------------------------
// context
// -----------------------------------------------------------------------------------------------------
static int pick_concurrent_stream(hipStream_t ref, hipStream_t* out);
extern "C" int spp_init(int device, spp_ctx** out) {
  if (!out) return fail(SPP_ERR_BAD_INPUT, "out is NULL");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(SPP_ERR_NO_DEVICE, "no HIP device visible: libspp has no CPU fallback");
  if (device < 0 || device >= count) return fail(SPP_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, count);
  HIP_TRY(hipSetDevice(device));
  spp_ctx* ctx = new spp_ctx();
  ctx->device = device;
  ctx->stream = nullptr;
  for (int k = 0; k < SPP_NWS; k++) ctx->pstream[k] = nullptr;
  struct Guard {   // a failure below must not leak the context and the streams created so far
    spp_ctx* ctx;
    ~Guard() { if (ctx) spp_free_ctx(ctx); }
  } guard{ctx};
  HIP_TRY(hipStreamCreate(&ctx->stream));
  HIP_TRY(hipStreamCreate(&ctx->pstream[0]));
  for (int k = 1; k < SPP_NWS; k++)
    if (int e = pick_concurrent_stream(ctx->pstream[0], &ctx->pstream[k])) return e;
  guard.ctx = nullptr;
  *out = ctx;
  return SPP_OK;
}
// A new stream that really runs beside `ref`.  HIP multiplexes the streams of one priority onto a few hardware queues in
// creation order; two streams that share a queue execute in submission order, and which ones do depends on how many streams
// the process (torch included) created before -- seen as run-to-run differences of 4 % on pipelined batches and 0.5 ms on a
// single proof whose G2 sum ran in front of the matrix evaluation instead of beside it.  Probe: a one-lane kernel that waits
// 2 ms on `ref`, a trivial kernel on the candidate; the candidate is kept if its kernel finishes while the other still waits.
static int pick_concurrent_stream(hipStream_t ref, hipStream_t* out) {
  static const bool probe = getenv("SPP_NO_STREAM_PROBE") == nullptr;
  hipStream_t rejected[6];
  int nrej = 0;
  *out = nullptr;
  for (int attempt = 0; attempt < 6 && probe; attempt++) {
    hipStream_t cand;
    HIP_TRY(hipStreamCreate(&cand));
    HIP_TRY(hipStreamSynchronize(ref));
    launch_spin(ref, 200000, nullptr);            // 2 ms of the 100 MHz wall clock
    const auto t0 = std::chrono::steady_clock::now();
    launch_touch(cand, nullptr);
    HIP_TRY(hipStreamSynchronize(cand));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    HIP_TRY(hipStreamSynchronize(ref));
    if (ms < 1.0) {
      *out = cand;
      break;
    }
    rejected[nrej++] = cand;                      // keep it alive until the search ends: destroying it would free its queue slot
  }
  for (int i = 0; i < nrej; i++) hipStreamDestroy(rejected[i]);
  if (!*out) HIP_TRY(hipStreamCreate(out));
  return 0;
}

extern "C" void spp_free_ctx(spp_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  for (int k = 0; k < SPP_NWS; k++)
    if (ctx->pstream[k]) hipStreamDestroy(ctx->pstream[k]);
  for (void* p : ctx->owned) hipFree(p);
  if (ctx->audit_scratch) hipFree(ctx->audit_scratch);
  delete ctx;
}

// -----------------------------------------------------------------------------------------------------
// pk container
// -----------------------------------------------------------------------------------------------------
namespace {
struct PkFile {
  uint32_t circuit_id, n_wires, domain_log, n_public, challenge_wire;
  G1Affine alpha1, beta1, delta1;
  G2Affine beta2, delta2;
  std::vector<uint32_t> A_w, B1_w, B2_w, K_w, CB_w, CS_w;
  std::vector<G1Affine> A, B1, K, Z, CB, CS;
  std::vector<G2Affine> B2;
};
struct Rd {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  uint32_t u32() {
    if (p + 4 > end) { ok = false; return 0; }
    uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    p += 4;
    return v;
  }
  const uint8_t* take(size_t n) {
    if (p + n > end) { ok = false; return nullptr; }
    const uint8_t* q = p;
    p += n;
    return q;
  }
};
bool rd_g1_section(Rd& r, std::vector<uint32_t>* wires, std::vector<G1Affine>& pts) {
  uint32_t n = r.u32();
  if (!r.ok || (size_t)n * 64 > (size_t)(r.end - r.p)) return false;
  if (wires) {
    wires->resize(n);
    for (auto& w : *wires) w = r.u32();
  }
  pts.resize(n);
  for (auto& pt : pts) {
    const uint8_t* b = r.take(64);
    if (!b) return false;
    pt = g1_from_raw(b);
  }
  return r.ok;
}
bool parse_pk(const std::vector<uint8_t>& buf, PkFile& k) {
  Rd r{buf.data(), buf.data() + buf.size()};
  if (r.u32() != 0x4b505053u || r.u32() != 1) return false;
  k.circuit_id = r.u32(); k.n_wires = r.u32(); k.domain_log = r.u32(); k.n_public = r.u32(); k.challenge_wire = r.u32();
  const uint8_t* b;
  if (!(b = r.take(64))) return false; k.alpha1 = g1_from_raw(b);
  if (!(b = r.take(64))) return false; k.beta1 = g1_from_raw(b);
  if (!(b = r.take(64))) return false; k.delta1 = g1_from_raw(b);
  if (!(b = r.take(128))) return false; k.beta2 = g2_from_raw(b);
  if (!(b = r.take(128))) return false; k.delta2 = g2_from_raw(b);
  if (!rd_g1_section(r, &k.A_w, k.A)) return false;
  if (!rd_g1_section(r, &k.B1_w, k.B1)) return false;
  uint32_t n2 = r.u32();
  if (!r.ok || (size_t)n2 * 128 > (size_t)(r.end - r.p)) return false;
  k.B2_w.resize(n2);
  for (auto& w : k.B2_w) w = r.u32();
  k.B2.resize(n2);
  for (auto& pt : k.B2) {
    if (!(b = r.take(128))) return false;
    pt = g2_from_raw(b);
  }
  if (!rd_g1_section(r, &k.K_w, k.K)) return false;
  if (!rd_g1_section(r, nullptr, k.Z)) return false;
  if (!rd_g1_section(r, &k.CB_w, k.CB)) return false;
  if (!rd_g1_section(r, &k.CS_w, k.CS)) return false;
  return r.ok && r.p == r.end;
}
}  // namespace

static void destroy_circuit(spp_circuit* c);
static void free_workspace(Workspace& w);

// merge `extra` into the entry of `wire` (or append one)
template <class F>
static void merge_point(std::vector<uint32_t>& wires, std::vector<Affine<F>>& pts, uint32_t wire, const Affine<F>& extra) {
  for (size_t i = 0; i < wires.size(); i++)
    if (wires[i] == wire) {
      pts[i] = host_add(pts[i], extra);
      return;
    }
  wires.push_back(wire);
  pts.push_back(extra);
}

// Greedy split of an HBM budget over the throughput-layout (one table row per base) MSM sets of one OR SEVERAL circuits: start
// every set at 6 bits and repeatedly widen the set whose next window bit removes the most mixed-addition work per extra byte
// (a G2 addition is weighted 3 G1 additions, as measured); `fixed` sets keep their bits.  With several circuits the unit of work
// is one proof of each (the relayer's pair: an audit proof and a withdraw proof per withdrawal,
// demo-frontend/app/api/relay/withdraw/route.ts:238-276), so their sets simply compete in one list.
namespace {
struct PlanSet {
  double n, esz, wgt;   // bases, bytes per table entry, weight of an addition (0: fixed)
  bool flat;            // one row per base (else one row per window)
  int bits;
};
double plan_bytes(const PlanSet& s, int cb) { return s.n * s.esz * (s.flat ? 1.0 : (double)msm_windows((uint32_t)cb)) * (double)(1u << (cb - 1)); }
void plan_greedy(std::vector<PlanSet>& sets, double budget, int cmax) {
  double used = 0;
  for (auto& s : sets) used += plan_bytes(s, s.bits);
  for (;;) {
    int best = -1;
    double best_gain = 0;
    for (size_t i = 0; i < sets.size(); i++) {
      const PlanSet& s = sets[i];
      if (s.wgt == 0 || s.bits >= cmax || s.n == 0) continue;
      const double extra = plan_bytes(s, s.bits + 1) - plan_bytes(s, s.bits);
      if (used + extra > budget) continue;
      const double saved = s.wgt * s.n * ((double)msm_windows((uint32_t)s.bits) - (double)msm_windows((uint32_t)s.bits + 1));
      double gain = saved / extra;
      if (saved <= 0) gain = 1e-30;   // a bit that does not change the window count yet may enable the next one
      if (gain > best_gain) { best_gain = gain; best = (int)i; }
    }
    if (best < 0) break;
    used += plan_bytes(sets[best], sets[best].bits + 1) - plan_bytes(sets[best], sets[best].bits);
    sets[best].bits++;
  }
}
const double PLAN_ESZ[7] = {64, 64, 64, 64, 64, 64, 128}, PLAN_WGT[7] = {1, 1, 1, 1, 0, 0, 3.0};   // A, B1, K, Z, CB, CS, B2
}  // namespace

static int load_circuit_impl(spp_ctx* ctx, const char* circuit_path, const char* pk_path, int window_bits, const uint32_t* forced_bits,
                             spp_circuit** out);
extern "C" int spp_load_circuit(spp_ctx* ctx, const char* circuit_path, const char* pk_path, int window_bits, spp_circuit** out) {
  return load_circuit_impl(ctx, circuit_path, pk_path, window_bits, nullptr, out);
}
extern "C" int spp_load_circuit_with_windows(spp_ctx* ctx, const char* circuit_path, const char* pk_path, const uint32_t bits[7],
                                             spp_circuit** out) {
  if (!bits) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  for (int s = 0; s < 7; s++)
    if (bits[s] < 4 || bits[s] > 16) return fail(SPP_ERR_BAD_INPUT, "window bits %u of set %d outside [4,16]", bits[s], s);
  return load_circuit_impl(ctx, circuit_path, pk_path, 0, bits, out);
}
static int load_circuit_impl(spp_ctx* ctx, const char* circuit_path, const char* pk_path, int window_bits, const uint32_t* forced_bits,
                             spp_circuit** out) {
  if (!ctx || !circuit_path || !pk_path || !out) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (window_bits != 0 && (window_bits < 4 || window_bits > 16)) return fail(SPP_ERR_BAD_INPUT, "window_bits %d outside [4,16]", window_bits);
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  spp_circuit* c = new spp_circuit();
  c->ctx = ctx;
  c->c_bits = (uint32_t)window_bits;
  // every early return below releases what has been allocated so far
  struct Guard {
    spp_circuit* c;
    ~Guard() { if (c) destroy_circuit(c); }
  } guard{c};
  if (!c->circ.load(circuit_path)) return fail(SPP_ERR_IO, "cannot read circuit %s", circuit_path);
  std::vector<uint8_t> pkbuf;
  PkFile pk;
  if (!read_file(pk_path, pkbuf)) return fail(SPP_ERR_IO, "cannot read proving key %s", pk_path);
  if (!parse_pk(pkbuf, pk)) return fail(SPP_ERR_FORMAT, "malformed proving key %s", pk_path);
  const Circuit& circ = c->circ;
  if (pk.circuit_id != circ.id || pk.n_wires != circ.n_wires || pk.domain_log != circ.domain_log)
    return fail(SPP_ERR_FORMAT, "proving key does not match the circuit");
  // Window bits and table layout per MSM set.
  //  * window_bits given: every set gets one table row per window (msm_windows(c) rows of 2^(c-1) multiples per base) -- small
  //    tables, a single pass, no Horner step: the layout of the one-proof latency path (the drop-in helpers load 8 bits).
  //  * window_bits = 0 (throughput): the five big sets keep ONE row per base and walk it once per window ("flat", see
  //    kernels_msm.hip); the window of every set is a greedy split of the HBM budget (env SPP_TABLE_BUDGET_GB, default 240 of the
  //    288 GB, capped at 85 % of the free HBM): repeatedly widen the set whose next window bit removes the most mixed-addition
  //    work per extra byte (a G2 addition is weighted 3 G1 additions, as measured).  A flat G1 row at 16 bits is 2 MB per base
  //    and costs 16 additions per full-size scalar; the row-per-window layout of rounds 1-2 afforded 11-12 bits (22-24
  //    additions) in the same bytes (SPP_FLAT=0 brings it back for comparison).  The two commitment sets only ever see bytes /
  //    small counters and sit on the critical path of the challenge: row-per-window tables at 9 bits, no passes.
  uint32_t cw[7];   // A, B1, K, Z, CB, CS, B2
  bool flat[7] = {false, false, false, false, false, false, false};
  {
    const double nset[7] = {(double)pk.A.size() + 2, (double)pk.B1.size() + 2, (double)pk.K.size() + 1, (double)pk.Z.size(),
                            (double)pk.CB.size(), (double)pk.CS.size(), (double)pk.B2.size() + 2};
    if (forced_bits) {           // spp_load_circuit_with_windows: the caller planned the windows (spp_plan_windows), throughput layout
      for (int s = 0; s < 7; s++) {
        cw[s] = forced_bits[s];
        flat[s] = PLAN_WGT[s] != 0;
      }
    } else if (window_bits != 0) {
      for (int s = 0; s < 7; s++) cw[s] = (uint32_t)window_bits;
    } else {
      const char* fe = getenv("SPP_FLAT");
      const bool use_flat = !(fe && fe[0] == '0');
      size_t free_b = 0, total_b = 0;
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
      double budget = 240e9;
      if (const char* env = getenv("SPP_TABLE_BUDGET_GB")) budget = atof(env) * 1e9;
      budget = std::min(budget, 0.85 * (double)free_b);
      std::vector<PlanSet> sets;
      for (int s = 0; s < 7; s++) {
        flat[s] = use_flat && PLAN_WGT[s] != 0;
        sets.push_back({nset[s], PLAN_ESZ[s], PLAN_WGT[s], flat[s], PLAN_WGT[s] != 0 ? 6 : 9});
      }
      plan_greedy(sets, budget, use_flat ? 16 : 15);
      for (int s = 0; s < 7; s++) cw[s] = (uint32_t)sets[s].bits;
    }
    c->c_bits = cw[3];   // reported window = that of the largest set (Z)
  }
  c->logn = circ.domain_log;
  c->n = 1u << c->logn;
  c->row_r = circ.n_wires;
  c->row_s = circ.n_wires + 1;
  c->row_rs = circ.n_wires + 2;
  c->n_rows = circ.n_wires + 3;

  // ---- R1CS + program + hash constants ----
  int e;
  if ((e = upload_sparse(c, circ, circ.A, &c->dc.A)) || (e = upload_sparse(c, circ, circ.B, &c->dc.B)) ||
      (e = upload_sparse(c, circ, circ.C, &c->dc.C)) || (e = upload_sparse(c, circ, circ.H, &c->dc.H)))
    return e;
  Fr* d_coeffs;
  Fr* d_aux;
  uint32_t* d_prog;
  if ((e = own_upload(c, &d_coeffs, circ.coeffs)) || (e = own_upload(c, &d_prog, circ.program)) || (e = own_upload(c, &d_aux, circ.aux))) return e;
  c->dc.coeffs = d_coeffs;
  c->dc.aux = d_aux;
  c->dc.program = d_prog;
  c->dc.n_wires = circ.n_wires;
  c->dc.n_constraints = circ.n_constraints;
  {
    // runs of consecutive constraints with identical B rows (at most 8 long, so lanes stay comparable in cost)
    std::vector<uint32_t> runs;
    auto same_b = [&](uint32_t k) {
      const uint32_t a0 = circ.B.rowptr[k - 1], a1 = circ.B.rowptr[k], b1 = circ.B.rowptr[k + 1];
      if (a1 - a0 != b1 - a1 || a1 == a0) return false;
      for (uint32_t t = 0; t < a1 - a0; t++)
        if (circ.B.terms[a0 + t].wire != circ.B.terms[a1 + t].wire || circ.B.terms[a0 + t].coeff != circ.B.terms[a1 + t].coeff) return false;
      return true;
    };
    uint32_t len = 0;
    for (uint32_t k = 0; k < circ.n_constraints; k++) {
      if (k == 0 || len >= 8 || !same_b(k)) { runs.push_back(k); len = 0; }
      len++;
    }
    const uint32_t n_runs = (uint32_t)runs.size();
    runs.push_back(circ.n_constraints);
    uint32_t* d_runs;
    if ((e = own_upload(c, &d_runs, runs))) return e;
    c->dc.run_start = d_runs;
    c->dc.n_runs = n_runs;
    uint32_t longest = 0;
    for (const Sparse* m : {&circ.A, &circ.B, &circ.C})
      for (uint32_t k = 0; k < circ.n_constraints; k++) longest = std::max(longest, m->rowptr[k + 1] - m->rowptr[k]);
    c->dc.max_row_terms = longest;
    std::vector<uint8_t> flags(std::max<uint32_t>(circ.n_constraints, 1), 0);
    for (uint32_t k = 0; k < circ.n_constraints; k++) {
      if (k > 0 && same_b(k)) flags[k] |= 1;
      const uint32_t a0 = circ.A.rowptr[k], a1 = circ.A.rowptr[k + 1], b0 = circ.B.rowptr[k], b1 = circ.B.rowptr[k + 1];
      bool eq = a1 - a0 == b1 - b0 && a1 != a0;
      for (uint32_t t = 0; eq && t < a1 - a0; t++)
        eq = circ.A.terms[a0 + t].wire == circ.B.terms[b0 + t].wire && circ.A.terms[a0 + t].coeff == circ.B.terms[b0 + t].coeff;
      if (eq) flags[k] |= 2;
    }
    uint8_t* d_flags;
    if ((e = own_upload(c, &d_flags, flags))) return e;
    c->dc.row_flags = d_flags;
  }
  c->dc.n_public = circ.n_public;
  c->dc.n_inputs = circ.n_inputs();
  c->dc.challenge_wire = circ.challenge_wire;
  {
    auto flat = [](const PoseidonParams& pp) {
      std::vector<Fr> m;
      for (auto& row : pp.mds)
        for (auto& v : row) m.push_back(v);
      return m;
    };
    const PoseidonParams& p3 = poseidon_params(3);
    const PoseidonParams& p5 = poseidon_params(5);
    const Poseidon2Params& p2 = poseidon2_params();
    Fr *a, *b, *cc, *d, *f, *g;
    std::vector<Fr> mu(p2.mu, p2.mu + 4);
    auto canon = [](std::vector<Fr> v) {   // words < p: dev_poseidon29's value bounds rely on it
      for (auto& x : v) x = x.canonical();
      return v;
    };
    if ((e = own_upload(c, &a, canon(p3.rc))) || (e = own_upload(c, &b, flat(p3))) || (e = own_upload(c, &cc, canon(p5.rc))) ||
        (e = own_upload(c, &d, flat(p5))) || (e = own_upload(c, &f, p2.rc)) || (e = own_upload(c, &g, mu)))
      return e;
    std::vector<Fr> bytes(256);
    for (int i = 0; i < 256; i++) bytes[i] = Fr::from_u64((uint64_t)i);
    Fr* bm;
    if ((e = own_upload(c, &bm, bytes))) return e;
    c->dc.byte_mont = bm;
    auto flat29 = [](const PoseidonParams& pp) {
      std::vector<uint32_t> m;
      for (auto& row : pp.mds)
        for (auto& v : row) {
          const F29<FrParams> x = F29<FrParams>::from_fp(v);     // v * 2^261, normalised, < 1.1 p
          for (int k = 0; k < 9; k++) m.push_back(x.l[k]);
        }
      return m;
    };
    uint32_t *m3, *m5;
    if ((e = own_upload(c, &m3, flat29(p3))) || (e = own_upload(c, &m5, flat29(p5)))) return e;
    c->dc.pos3_mds29 = m3;
    c->dc.pos5_mds29 = m5;
    c->dc.pos3_rc = a; c->dc.pos3_mds = b; c->dc.pos5_rc = cc; c->dc.pos5_mds = d; c->dc.p2_rc = f; c->dc.p2_mu = g;
  }
  // program scan: split into sequential segments (one lane per proof) and wide steps (data-parallel instructions
  // that get their own kernels: batch divisions and lookup histograms), with the commitment boundary in between
  bool generic_ops = false;
  {
    const auto& pr = circ.program;
    size_t pc = 0, seg = 0;
    auto flush = [&](size_t end) {
      if (end > seg) c->schedule.push_back({SolveStep::SEQ, (uint32_t)seg, (uint32_t)end, 0});
    };
    // SPP_SOLVE_TRACE=1 (diagnostic): one launch per instruction class run, so a kernel trace of a proof shows where the
    // sequential solver spends its time
    const bool trace_ops = getenv("SPP_SOLVE_TRACE") != nullptr;
    c->no_coop = getenv("SPP_NO_COOP") != nullptr || trace_ops;
    c->trace_items = getenv("SPP_COOP_TRACE") != nullptr;
    c->one_track = getenv("SPP_COOP_ONE_TRACK") != nullptr;
    c->no_level_stream = getenv("SPP_NO_LEVEL_STREAM") != nullptr;
    uint32_t prev_op = OP_END;
    while (pc < pr.size() && pr[pc] != OP_END) {
      if (trace_ops && pr[pc] != prev_op) {
        flush(pc);
        seg = std::max(seg, pc);
      }
      prev_op = pr[pc];
      switch (pr[pc]) {
        case OP_SOLVE_C: case OP_SOLVE_A: case OP_MASK: pc += 2; break;
        case OP_BATCH_DIV:
          c->max_batch_div = std::max(c->max_batch_div, pr[pc + 2]);
          if (pr[pc + 2] >= 64) {
            flush(pc);
            c->schedule.push_back({SolveStep::BATCH_DIV, pr[pc + 1], pr[pc + 2], 0});
            seg = pc + 3;
          }
          pc += 3;
          break;
        case OP_COUNT8:
          flush(pc);
          c->schedule.push_back({SolveStep::COUNT8, pr[pc + 1], pr[pc + 2], pr[pc + 3]});
          seg = pc + 4;
          pc += 4;
          break;
        case OP_BITS: case OP_LIMBS8: case OP_POSEIDON: pc += 4; break;
        case OP_POSEIDON2: case OP_INV_H: pc += 3; break;
        case OP_COMMIT:
          flush(pc);
          c->schedule.push_back({SolveStep::COMMIT, 0, 0, 0});
          pc += 1;
          seg = pc;
          break;
        case OP_GRUMPKIN: pc += 5 + pr[pc + 4]; break;
        // the solver of a decoded gnark system (spp/ccs.py to_sppc_solved): one lane per proof, whatever the batch size -- the
        // cooperative planner knows nothing of these instructions
        case OP_SOLVE_ROW: case OP_LIMBS: case OP_COUNTN: pc += 5; generic_ops = true; break;
        case OP_GK_MUL: pc += 7; generic_ops = true; break;
        case OP_GLV: pc += 3 + 28; generic_ops = true; break;
        case OP_EMUL: pc += 3 + 16; generic_ops = true; break;
        default: return fail(SPP_ERR_FORMAT, "bad opcode %u in solver program", pr[pc]);
      }
    }
    flush(pc);
  }

  c->generic_solver = generic_ops;
  if (generic_ops) c->no_coop = true;
  else if (int e = coop_plan(c)) return e;
  if (int e = row_paths_plan(c)) return e;

  // ---- NTT tables ----
  {
    const uint32_t n = c->n, logn = c->logn;
    Fr w = fr_root_of_unity(logn), wi = w.inv();
    std::vector<Fr> tf(n / 2), ti(n / 2), cb(n), cib(n);
    Fr a = Fr::one(), b = Fr::one();
    for (uint32_t k = 0; k < n / 2; k++) {
      tf[k] = a;
      ti[k] = b;
      a = a * w;
      b = b * wi;
    }
    {
      const char* hm = getenv("SPP_H_MODE");
      c->h_mode = hm ? atoi(hm) : (getenv("SPP_Z_COEFF") ? 0 : 2);
      if (c->h_mode < 0 || c->h_mode > 2) c->h_mode = 2;
    }
    // the coset: gnark's multiplicative generator 5, or -- product form -- zeta, the primitive 2n-th root of unity with
    // zeta^2 = w, so that H u zeta*H are the 2n-th roots of unity
    Fr g = c->h_mode == 2 ? fr_root_of_unity(logn + 1) : Fr::from_u64(5), gi = g.inv(), ninv = Fr::from_u64(n).inv();
    std::vector<Fr> gp(n), gip(n);
    Fr x = ninv, y = ninv;
    for (uint32_t i = 0; i < n; i++) {
      gp[i] = x;
      gip[i] = y;
      x = x * g;
      y = y * gi;
    }
    for (uint32_t pos = 0; pos < n; pos++) {
      uint32_t i = bitrev(pos, logn);
      cb[pos] = gp[i];
      cib[pos] = gip[i];
    }
    if ((e = own_upload(c, &c->tw_fwd, tf)) || (e = own_upload(c, &c->tw_inv, ti)) || (e = own_upload(c, &c->coset_br, cb)) ||
        (e = own_upload(c, &c->coset_inv_br, cib)))
      return e;
    Fr gn = g.pow_u64(n);
    c->zinv = (gn - Fr::one()).inv();
  }

  // ---- MSM sets ----
  {
    std::vector<uint32_t> w = pk.A_w;
    std::vector<G1Affine> p = pk.A;
    merge_point(w, p, 0u, pk.alpha1);
    w.push_back(c->row_r); p.push_back(pk.delta1);
    if ((e = make_set(c, &c->A, w, p, false, cw[0], flat[0]))) return e;
  }
  {
    std::vector<uint32_t> w = pk.B1_w;
    std::vector<G1Affine> p = pk.B1;
    merge_point(w, p, 0u, pk.beta1);
    w.push_back(c->row_s); p.push_back(pk.delta1);
    if ((e = make_set(c, &c->B1, w, p, false, cw[1], flat[1]))) return e;
  }
  {
    std::vector<uint32_t> w = pk.B2_w;
    std::vector<G2Affine> p = pk.B2;
    merge_point(w, p, 0u, pk.beta2);
    w.push_back(c->row_s); p.push_back(pk.delta2);
    if ((e = make_set(c, &c->B2, w, p, false, cw[6], flat[6]))) return e;
  }
  // ---- the H bases (and, in the product form, the per-wire column sums that join the K set) ----
  std::vector<uint32_t> z_w, xk_w;
  std::vector<G1Affine> z_p, xk_p;
  {
    if (pk.Z.size() != (size_t)c->n - 1) return fail(SPP_ERR_FORMAT, "Z section has %zu points, expected %u", pk.Z.size(), c->n - 1);
    const uint32_t n = c->n;
    hipStream_t st = ctx->stream;
    // out[i] = sum_j scale[j] w^(-ij) Z_j, natural order (group DFT on the device, kernels_msm.hip)
    auto eval_basis = [&](const std::vector<Fr>& scale, std::vector<G1Affine>& out) -> int {
      DevBuf d_pts, d_scale, d_work, d_out;
      HIP_TRY(d_pts.alloc(pk.Z.size() * sizeof(G1Affine)));
      HIP_TRY(d_scale.alloc((size_t)n * sizeof(Fr)));
      HIP_TRY(d_work.alloc((size_t)n * sizeof(G1XYZZ)));
      HIP_TRY(d_out.alloc((size_t)n * sizeof(G1Affine)));
      HIP_TRY(hipMemcpyAsync(d_pts.p, pk.Z.data(), pk.Z.size() * sizeof(G1Affine), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(d_scale.p, scale.data(), (size_t)n * sizeof(Fr), hipMemcpyHostToDevice, st));
      launch_g1_eval_basis(st, d_pts.as<G1Affine>(), (uint32_t)pk.Z.size(), c->logn, d_scale.as<Fr>(), c->tw_inv, d_work.as<G1XYZZ>(),
                           d_out.as<G1Affine>());
      std::vector<G1Affine> br(n);
      HIP_TRY(hipMemcpyAsync(br.data(), d_out.p, (size_t)n * sizeof(G1Affine), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipGetLastError());
      out.resize(n);
      for (uint32_t pos = 0; pos < n; pos++) out[bitrev(pos, c->logn)] = br[pos];     // the DIF stages leave element i at bitrev(i)
      return 0;
    };
    if (c->h_mode == 0) {
      // h comes out of the last DIF pass in bit-reversed order: row `pos` holds h_{bitrev(pos)}
      for (uint32_t pos = 0; pos < n; pos++) {
        uint32_t i = bitrev(pos, c->logn);
        if (i == n - 1) continue;
        z_w.push_back(pos);
        z_p.push_back(pk.Z[i]);
      }
    } else if (c->h_mode == 1) {
      // sum_j h_j Z_j = sum_i h(g w^i) Z'_i with Z'_i = sum_j (g^-j / n) w^(-ij) Z_j; row i of the a-slot holds h(g w^i)
      Fr gi = Fr::from_u64(5).inv(), x = Fr::from_u64(n).inv();
      std::vector<Fr> scale(n);
      for (uint32_t j = 0; j < n; j++) { scale[j] = x; x = x * gi; }
      if ((e = eval_basis(scale, z_p))) return e;
      for (uint32_t i = 0; i < n; i++) z_w.push_back(i);
    } else {
      // Product form.  P = A B has degree <= 2n - 2 and h_j = P_{n+j}; over D = the 2n-th roots of unity P_k = (1/2n) sum_{x in D}
      // P(x) x^-k, hence  sum_j h_j Z_j = sum_{x in D} P(x) W_x  with  W_x = (1/2n) sum_j x^-(n+j) Z_j:
      //   x = w^i        (x^-n = 1):   W_i  =  (1/2n) sum_j w^(-ij) Z_j,              P(x) = a_i b_i = c_i = <C_i, witness>
      //   x = zeta w^i   (x^-n = -1):  W'_i = -(1/2n) sum_j zeta^-j w^(-ij) Z_j,      P(x) = A(x) B(x) from two coset transforms
      // The first sum is linear in the witness: sum_i c_i W_i = sum_wire w_wire X_wire, X_wire = sum_i C[i][wire] W_i -- a point per
      // wire, computed here once and added to the wire's base in the K set (wires without one -- public, committed, the
      // challenge -- join the set with X_wire alone: the sum is part of Krs whatever the wire's class).
      const Fr inv2n = Fr::from_u64(2 * (uint64_t)n).inv();
      std::vector<Fr> scale(n, inv2n);
      std::vector<G1Affine> WH;
      if ((e = eval_basis(scale, WH))) return e;
      Fr zi = fr_root_of_unity(c->logn + 1).inv(), x = inv2n.neg();
      for (uint32_t j = 0; j < n; j++) { scale[j] = x; x = x * zi; }
      if ((e = eval_basis(scale, z_p))) return e;
      for (uint32_t i = 0; i < n; i++) z_w.push_back(i);
      // column sums of C against W
      struct Tm { uint32_t wire, row; Fr cf; };
      std::vector<Tm> tms;
      for (uint32_t k = 0; k < circ.n_constraints; k++)
        for (uint32_t t = circ.C.rowptr[k]; t < circ.C.rowptr[k + 1]; t++) tms.push_back({circ.C.terms[t].wire, k, circ.coeffs[circ.C.terms[t].coeff]});
      std::stable_sort(tms.begin(), tms.end(), [](const Tm& a, const Tm& b) { return a.wire < b.wire; });
      std::vector<uint32_t> rows(tms.size()), seg{0};
      std::vector<Fr> cfs(tms.size());
      for (size_t t = 0; t < tms.size(); t++) {
        rows[t] = tms[t].row;
        cfs[t] = tms[t].cf;
        if (t + 1 == tms.size() || tms[t + 1].wire != tms[t].wire) {
          xk_w.push_back(tms[t].wire);
          seg.push_back((uint32_t)t + 1);
        }
      }
      if (!tms.empty()) {
        DevBuf d_base, d_rows, d_cfs, d_seg, d_work, d_out;
        HIP_TRY(d_base.alloc((size_t)n * sizeof(G1Affine)));
        HIP_TRY(d_rows.alloc(rows.size() * 4));
        HIP_TRY(d_cfs.alloc(cfs.size() * sizeof(Fr)));
        HIP_TRY(d_seg.alloc(seg.size() * 4));
        HIP_TRY(d_work.alloc(tms.size() * sizeof(G1XYZZ)));
        HIP_TRY(d_out.alloc(xk_w.size() * sizeof(G1Affine)));
        HIP_TRY(hipMemcpyAsync(d_base.p, WH.data(), (size_t)n * sizeof(G1Affine), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_rows.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_cfs.p, cfs.data(), cfs.size() * sizeof(Fr), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_seg.p, seg.data(), seg.size() * 4, hipMemcpyHostToDevice, st));
        launch_g1_column_sums(st, d_base.as<G1Affine>(), d_rows.as<uint32_t>(), d_cfs.as<Fr>(), (uint32_t)tms.size(), d_seg.as<uint32_t>(),
                              (uint32_t)xk_w.size(), d_work.as<G1XYZZ>(), d_out.as<G1Affine>());
        xk_p.resize(xk_w.size());
        HIP_TRY(hipMemcpyAsync(xk_p.data(), d_out.p, xk_p.size() * sizeof(G1Affine), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipGetLastError());
      }
    }
  }
  {
    std::vector<uint32_t> w = pk.K_w;
    std::vector<G1Affine> p = pk.K;
    if (!xk_w.empty()) {   // product form: K_wire + X_wire (one pass over a wire -> position map; merge_point is linear per call)
      std::vector<int32_t> at(circ.n_wires + 3, -1);
      for (size_t i = 0; i < w.size(); i++) at[w[i]] = (int32_t)i;
      for (size_t i = 0; i < xk_w.size(); i++) {
        if (xk_p[i].is_inf()) continue;
        if (at[xk_w[i]] >= 0) p[at[xk_w[i]]] = host_add(p[at[xk_w[i]]], xk_p[i]);
        else {
          at[xk_w[i]] = (int32_t)w.size();
          w.push_back(xk_w[i]);
          p.push_back(xk_p[i]);
        }
      }
    }
    w.push_back(c->row_rs); p.push_back(pk.delta1.neg());
    if ((e = make_set(c, &c->K, w, p, false, cw[2], flat[2]))) return e;
  }
  if ((e = make_set(c, &c->Z, z_w, z_p, true, cw[3], flat[3]))) return e;
  if ((e = make_set(c, &c->CB, pk.CB_w, pk.CB, false, cw[4], false))) return e;
  if ((e = make_set(c, &c->CS, pk.CS_w, pk.CS, false, cw[5], false))) return e;
  if ((e = build_pending(c))) return e;

  for (int k = 0; k < SPP_NWS; k++) {
    Workspace& w = c->ws[k];
    // SPP_SERIAL=1 (profiling aid): one stream for everything, so per-stage / per-kernel times are not stretched by
    // the other batch or by the G2 side stream
    const bool serial = getenv("SPP_SERIAL") != nullptr;
    w.own_st = ctx->pstream[k];
    if (int e = pick_concurrent_stream(w.own_st, &w.own_st2)) return e;
    {
      // Batches run the G2 sum on a side stream with a priority of its own.  Streams of one priority share a few hardware queues
      // round-robin; when st and st2 land on the same one the G2 sum runs in front of the matrix evaluation instead of beside it.
      // Measured on 2048-proof audit batches (same box, alternating): 4 747-4 760 proofs/s with the priority stream, 4 662-4 707
      // without.  Small batches keep the default-priority side stream: with a second queue class in use every dispatch of a single
      // proof's ~120 short kernels started later (audit 12.3 -> 13.6 ms).  SPP_ST2_PRIORITY=0 (diagnostic): never use it.
      int lo = 0, hi = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
      const char* pe = getenv("SPP_ST2_PRIORITY");
      if (!(pe && pe[0] == '0') && hi < lo) HIP_TRY(hipStreamCreateWithPriority(&w.own_st2p, hipStreamDefault, hi));
    }
    w.st = serial ? ctx->pstream[0] : w.own_st;
    w.st2 = serial ? w.st : w.own_st2;
    HIP_TRY(hipEventCreate(&w.g2_ev.first));
    HIP_TRY(hipEventCreate(&w.g2_ev.second));
    HIP_TRY(hipEventCreateWithFlags(&w.ev_w, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&w.ev_b2, hipEventDisableTiming));
    for (auto& evt : w.ev) HIP_TRY(hipEventCreate(&evt));
    w.msm_ev.resize(8);
    for (auto& pr : w.msm_ev) {
      HIP_TRY(hipEventCreate(&pr.first));
      HIP_TRY(hipEventCreate(&pr.second));
    }
  }
  guard.c = nullptr;
  *out = c;
  return SPP_OK;
}

static void free_workspace(Workspace& w) {
  if (w.audit_scratch) hipFree(w.audit_scratch);
  w.audit_scratch = nullptr;
  w.audit_scratch_cap = 0;
  for (void* p : w.owned) hipFree(p);
  w.owned.clear();
  w.cap = 0;
}
static void destroy_circuit(spp_circuit* c) {
  if (!c) return;
  hipSetDevice(c->ctx->device);
  hipStreamSynchronize(c->ctx->stream);
  for (auto& w : c->ws) {
    if (w.st) hipStreamSynchronize(w.st);
    if (w.own_st2) { hipStreamSynchronize(w.own_st2); hipStreamDestroy(w.own_st2); }
    if (w.own_st2p) { hipStreamSynchronize(w.own_st2p); hipStreamDestroy(w.own_st2p); }
    if (w.g2_ev.first) hipEventDestroy(w.g2_ev.first);
    if (w.g2_ev.second) hipEventDestroy(w.g2_ev.second);
    if (w.ev_w) hipEventDestroy(w.ev_w);
    if (w.ev_b2) hipEventDestroy(w.ev_b2);
    free_workspace(w);
    for (auto& evt : w.ev) if (evt) hipEventDestroy(evt);
    for (auto& pr : w.msm_ev) {
      if (pr.first) hipEventDestroy(pr.first);
      if (pr.second) hipEventDestroy(pr.second);
    }
  }
  for (void* p : c->owned) hipFree(p);
  delete c;
}
extern "C" void spp_free_circuit(spp_circuit* c) { destroy_circuit(c); }
extern "C" int spp_circuit_info(const spp_circuit* c, uint32_t info[8]) {
  if (!c || !info) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  info[0] = c->circ.id; info[1] = c->circ.n_public - 1; info[2] = c->circ.n_secret; info[3] = c->circ.n_wires;
  info[4] = c->circ.n_constraints; info[5] = c->circ.domain_log; info[6] = c->circ.n_inputs(); info[7] = c->c_bits;
  return SPP_OK;
}
extern "C" uint64_t spp_circuit_table_bytes(const spp_circuit* c) { return c ? c->table_bytes : 0; }
extern "C" int spp_circuit_small_rows(const spp_circuit* c, uint32_t out[2]) {
  if (!c || !out) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  out[0] = c->dc.sm_nrows;
  out[1] = c->dc.sm_nslots;
  return SPP_OK;
}
extern "C" int spp_circuit_msm_windows(const spp_circuit* c, uint32_t bits[7]) {
  if (!c || !bits) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  bits[0] = c->A.c; bits[1] = c->B1.c; bits[2] = c->K.c; bits[3] = c->Z.c; bits[4] = c->CB.c; bits[5] = c->CS.c; bits[6] = c->B2.c;
  return SPP_OK;
}
extern "C" int spp_circuit_msm_table_rows(const spp_circuit* c, uint32_t rows[7]) {
  if (!c || !rows) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  rows[0] = c->A.Wt; rows[1] = c->B1.Wt; rows[2] = c->K.Wt; rows[3] = c->Z.Wt; rows[4] = c->CB.Wt; rows[5] = c->CS.Wt; rows[6] = c->B2.Wt;
  return SPP_OK;
}
extern "C" int spp_circuit_msm_sizes(const spp_circuit* c, uint32_t sizes[7]) {
  if (!c || !sizes) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  sizes[0] = c->A.N; sizes[1] = c->B1.N; sizes[2] = c->K.N; sizes[3] = c->Z.N; sizes[4] = c->CB.N; sizes[5] = c->CS.N; sizes[6] = c->B2.N;
  return SPP_OK;
}

// bases per MSM set of a proving key file, as spp_circuit_msm_sizes reports them after loading (A, B1, K, Z, CB, CS, B2): what
// spp_plan_windows needs before anything is loaded
extern "C" int spp_pk_msm_sizes(const char* pk_path, uint32_t sizes[7]) {
  if (!pk_path || !sizes) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  std::vector<uint8_t> pkbuf;
  PkFile pk;
  if (!read_file(pk_path, pkbuf)) return fail(SPP_ERR_IO, "cannot read proving key %s", pk_path);
  if (!parse_pk(pkbuf, pk)) return fail(SPP_ERR_FORMAT, "malformed proving key %s", pk_path);
  sizes[0] = (uint32_t)pk.A.size() + 2; sizes[1] = (uint32_t)pk.B1.size() + 2; sizes[2] = (uint32_t)pk.K.size() + 1;
  sizes[3] = (uint32_t)pk.Z.size(); sizes[4] = (uint32_t)pk.CB.size(); sizes[5] = (uint32_t)pk.CS.size(); sizes[6] = (uint32_t)pk.B2.size() + 2;
  return SPP_OK;
}
// window bits for the sets of n_circuits circuits that are to live on one GPU TOGETHER: sizes / bits = n_circuits x 7 (the order
// above); one greedy split of budget_bytes over the union of their sets (see plan_greedy).  Host only.
extern "C" int spp_plan_windows(uint32_t n_circuits, const uint32_t* sizes, double budget_bytes, uint32_t* bits) {
  if (!sizes || !bits || n_circuits == 0 || n_circuits > 16) return fail(SPP_ERR_BAD_INPUT, "bad argument");
  std::vector<PlanSet> sets;
  for (uint32_t k = 0; k < n_circuits; k++)
    for (int s = 0; s < 7; s++) sets.push_back({(double)sizes[7 * k + s], PLAN_ESZ[s], PLAN_WGT[s], PLAN_WGT[s] != 0, PLAN_WGT[s] != 0 ? 6 : 9});
  double floor_bytes = 0;
  for (auto& ps : sets) floor_bytes += plan_bytes(ps, ps.bits);
  if (floor_bytes > budget_bytes) return fail(SPP_ERR_BAD_INPUT, "the budget does not hold even 6-bit tables (%.1f GB needed)", floor_bytes / 1e9);
  plan_greedy(sets, budget_bytes, 16);
  for (size_t i = 0; i < sets.size(); i++) bits[i] = (uint32_t)sets[i].bits;
  return SPP_OK;
}

template <class T>
static int ws_alloc(Workspace& w, T** p, size_t count) {
  HIP_TRY(hipMalloc((void**)p, sizeof(T) * std::max<size_t>(count, 1)));
  w.owned.push_back((void*)*p);
  return 0;
}
template <class F>
static int ws_set(Workspace& w, const MsmSet<F>* s, MsmBuf<F>* b, size_t P) {
  // R * Sg(P') * P' <= lane target + R * P' for every P' <= P (msm_plan); small batches: up to 64K (item, pass) lanes
  const uint32_t occ = sizeof(F) > sizeof(Fq) ? 1 : 2;
  const uint32_t R = msm_plan(s->N, (uint32_t)P, s->c, s->Wt, occ).R;
  b->partial_cap = (size_t)256 * 4 * 8 * 64 + 65536 + (size_t)(R + 1) * (P + 64);
  for (size_t q = P; q >= 1; q /= 2)   // and the exact need at the sizes most likely to be used
    b->partial_cap = std::max(b->partial_cap, msm_plan(s->N, (uint32_t)q, s->c, s->Wt, occ).partial_elems((uint32_t)q));
  int e;
  if ((e = ws_alloc(w, &b->partial, b->partial_cap))) return e;
  return ws_alloc(w, &b->out, P);
}
static int ensure_workspace(spp_circuit* c, Workspace& w, size_t P) {
  if (P <= w.cap) return 0;
  HIP_TRY(hipStreamSynchronize(w.st));
  free_workspace(w);
  int e;
  const size_t npub = c->circ.n_public - 1;
  if ((e = ws_alloc(w, &w.W, (size_t)c->n_rows * P)) || (e = ws_alloc(w, &w.abc, (size_t)3 * c->n * P)) ||
      (e = ws_alloc(w, &w.scratch, (size_t)c->max_batch_div * P)) || (e = ws_alloc(w, &w.commit_affine, P)) ||
      (e = ws_alloc(w, &w.d_inputs, (size_t)c->circ.n_inputs() * 32 * P)) || (e = ws_alloc(w, &w.d_rs, 64 * P)) ||
      (e = ws_alloc(w, &w.d_proofs, (size_t)SPP_PROOF_LEN * P)) || (e = ws_alloc(w, &w.d_pws, (12 + 32 * npub) * P)) ||
      (e = ws_alloc(w, &w.d_status, P)) || (e = ws_alloc(w, &w.counters, 256 * P)))
    return e;
  if ((e = ws_set(w, &c->A, &w.A, P)) || (e = ws_set(w, &c->B1, &w.B1, P)) || (e = ws_set(w, &c->B2, &w.B2, P)) ||
      (e = ws_set(w, &c->K, &w.K, P)) || (e = ws_set(w, &c->Z, &w.Z, P)) || (e = ws_set(w, &c->CB, &w.CB, P)) ||
      (e = ws_set(w, &c->CS, &w.CS, P)))
    return e;
  {
    const size_t Ps = std::min<size_t>(P, scaled_blind_max_batch(c->A.N, c->B1.N));
    if ((e = ws_set(w, &c->A, &w.sA, Ps)) || (e = ws_set(w, &c->B1, &w.rB, Ps)) || (e = ws_alloc(w, &w.Ws, (size_t)c->n_rows * Ps)) ||
        (e = ws_alloc(w, &w.Wr, (size_t)c->n_rows * Ps)))
      return e;
  }
  {
    // digit planes: the G1 sets share one buffer (they run one after the other on `st`), the G2 set has its own
    size_t d1 = 0;
    for (const MsmSet<Fq>* s : {&c->A, &c->B1, &c->K, &c->Z, &c->CB, &c->CS}) d1 = std::max(d1, msm_digit_elems(s->N, (uint32_t)P, s->c));
    w.dig1_cap = d1;
    w.dig2_cap = msm_digit_elems(c->B2.N, (uint32_t)P, c->B2.c);
    if ((e = ws_alloc(w, &w.dig1, w.dig1_cap)) || (e = ws_alloc(w, &w.dig2, w.dig2_cap))) return e;
    if (c->dc.sm_nrows && (e = ws_alloc(w, &w.small, (size_t)c->dc.sm_nslots * P))) return e;
  }
  w.cap = P;
  return 0;
}

static int16_t* ws_dig(Workspace& w, Fq*) { return w.dig1; }
static int16_t* ws_dig(Workspace& w, Fq2*) { return w.dig2; }
// digits + accumulate of one set; the caller folds (several sets share the fold launches): b.plan holds the lane layout
template <class F>
static void run_msm(spp_circuit* c, Workspace& w, const MsmSet<F>& s, MsmBuf<F>& b, uint32_t P, bool timed, hipStream_t st_override = nullptr,
                    std::pair<hipEvent_t, hipEvent_t>* ev_override = nullptr, bool fold = true, const Fr* scal_override = nullptr) {
  hipStream_t st = st_override ? st_override : w.st;
  const Fr* scal = scal_override ? scal_override : s.from_h ? w.abc : w.W;
  MsmPlan pl = msm_plan(s.N, P, s.c, s.Wt, sizeof(F) > sizeof(Fq) ? 1 : 2);
  while (pl.Sg > 1 && pl.partial_elems(P) > b.partial_cap) pl.Sg--;  // never exceed the allocated partial buffer
  b.plan = pl;
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  if (timed && w.msm_ev_used < w.msm_ev.size()) ev = &w.msm_ev[w.msm_ev_used++];
  if (ev_override) ev = ev_override;
  int16_t* dig = ws_dig(w, (F*)nullptr);
  launch_msm_digits(st, s.rows, scal, dig, s.N, P, s.c);
  // the event pair receives the dispatch's own start/stop timestamps (what rocprofv3 reports as the kernel's duration)
  launch_msm_accumulate<F>(st, s.table, dig, b.partial, s.N, P, s.c, pl, ev ? ev->first : nullptr, ev ? ev->second : nullptr);
  if (fold) launch_msm_reduce<F>(st, b.partial, b.out, P, pl, s.c, s.N == 0);
}

static int prove_on_device(spp_circuit* c, Workspace& w, uint32_t P, const uint8_t* d_inputs, const uint8_t* d_rs, uint8_t* d_proofs,
                           uint8_t* d_pws, uint32_t* d_status) {
  hipStream_t st = w.st;
  const Circuit& circ = c->circ;
  const uint32_t n = c->n;
  w.msm_ev_used = 0;
  w.last_P = P;
  const bool scaled_blind = P <= scaled_blind_max_batch(c->A.N, c->B1.N) && !c->no_coop;
  HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(uint32_t) * P, st));
  hipEventRecord(w.ev[0], st);
  // 1. inputs, solver phase 1, commitment, challenge, solver phase 2
  launch_load_inputs(st, d_inputs, d_rs, w.W, circ.n_inputs(), circ.n_wires, P);
  for (const SolveStep& s : c->schedule) {
    switch (s.kind) {
      case SolveStep::SEQ:
        if (P <= COOP_MAX_BATCH && !c->no_coop && c->trace_items) {
          for (uint32_t t = 0; t < s.ntracks; t++)
            for (uint32_t it = s.tr_begin[t]; it < s.tr_end[t]; it++) {
              CoopTracks one{};
              one.n = 1; one.begin[0] = it; one.end[0] = it + 1;
              launch_solve_coop(st, c->dc, c->coop, w.W, w.scratch, one, P);
            }
        } else if (P <= COOP_MAX_BATCH && !c->no_coop) {
          CoopTracks tr{};
          tr.n = c->one_track ? 1 : s.ntracks;
          for (uint32_t t = 0; t < s.ntracks; t++) { tr.begin[t] = s.tr_begin[t]; tr.end[t] = s.tr_end[t]; }
          if (c->one_track) {   // SPP_COOP_ONE_TRACK=1 (diagnostic): the tracks one after the other
            for (uint32_t t = 0; t < s.ntracks; t++) {
              CoopTracks one{};
              one.n = 1; one.begin[0] = s.tr_begin[t]; one.end[0] = s.tr_end[t];
              launch_solve_coop(st, c->dc, c->coop, w.W, w.scratch, one, P);
            }
          } else launch_solve_coop(st, c->dc, c->coop, w.W, w.scratch, tr, P);
        }
        else launch_solve(st, c->dc, w.W, w.scratch, s.a, s.b, P);
        break;
      case SolveStep::BATCH_DIV: launch_batch_div(st, c->dc, w.W, w.scratch, s.a, s.b, P); break;
      case SolveStep::COUNT8: launch_count8(st, c->dc, w.W, w.counters, s.a, s.b, s.c, P); break;
      case SolveStep::COMMIT:
        run_msm(c, w, c->CB, w.CB, P, true);
        launch_challenge(st, w.CB.out, w.W, circ.challenge_wire, P, w.commit_affine, d_status);
        break;
    }
  }
  hipEventRecord(w.ev[1], st);
  // the G2 MSM depends on the witness only: start it now on the side stream
  hipEventRecord(w.ev_w, st);
  static const bool no_side = getenv("SPP_NO_SIDE") != nullptr;   // experiment: the G2 sum on the batch's own stream
  hipStream_t side = (w.st2 != w.st && w.own_st2p && P > COOP_MAX_BATCH) ? w.own_st2p : w.st2;
  if (no_side && P <= COOP_MAX_BATCH) side = st;
  hipStreamWaitEvent(side, w.ev_w, 0);
  run_msm(c, w, c->B2, w.B2, P, false, side, &w.g2_ev);
  hipEventRecord(w.ev_b2, side);
  // 2. constraint evaluations + satisfaction check
  launch_spmv_check(st, c->dc, w.W, w.abc, n, P, d_status, w.small);
  hipEventRecord(w.ev[2], st);
  // 3. h = (a*b - c)/Z  (coefficients land bit-reversed in the a-slot of abc)
  const size_t bs = (size_t)n * P;
  // the coset shifts ride on the stores of the inverse transforms' last pass (no separate pass over the arrays)
  const uint32_t nt = c->h_mode == 2 ? 2 : 3;      // product form: A and B only
  launch_ntt(st, w.abc, c->logn, P, c->tw_inv, true, nt, bs, c->coset_br);
  launch_ntt(st, w.abc, c->logn, P, c->tw_fwd, false, nt, bs);
  if (c->h_mode == 2) launch_qap_product(st, w.abc, n, P);
  else launch_qap_pointwise(st, w.abc, n, P, c->zinv);
  if (c->h_mode == 0) launch_ntt(st, w.abc, c->logn, P, c->tw_inv, true, 1, bs, c->coset_inv_br);   // else: the Z bases are in the evaluation basis
  hipEventRecord(w.ev[3], st);
  // 4. MSMs
  {
    MsmFoldSets<Fq> fs{};
    const MsmSet<Fq>* sets[7] = {&c->A, &c->B1, &c->K, &c->Z, &c->CS, &c->A, &c->B1};
    MsmBuf<Fq>* bufs[7] = {&w.A, &w.B1, &w.K, &w.Z, &w.CS, &w.sA, &w.rB};
    const Fr* scal[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, w.Ws, w.Wr};
    const int nsets = scaled_blind ? 7 : 5;
    if (scaled_blind) launch_scale_witness(st, w.W, w.Ws, w.Wr, c->n_rows, c->row_r, c->row_s, P);
    for (int i = 0; i < nsets; i++) {
      run_msm(c, w, *sets[i], *bufs[i], P, i < 5, nullptr, nullptr, false, scal[i]);
      fs.partial[i] = bufs[i]->partial;
      fs.out[i] = bufs[i]->out;
      fs.Sg[i] = sets[i]->N ? bufs[i]->plan.Sg : 0;
      fs.R[i] = bufs[i]->plan.R;
      fs.c[i] = sets[i]->c;
    }
    launch_msm_reduce_multi<Fq>(st, fs, nsets, P);   // the slice sums are folded level by level in shared launches, then Horner
  }
  hipEventRecord(w.ev[4], st);
  hipStreamWaitEvent(st, w.ev_b2, 0);   // join the G2 MSM
  hipEventRecord(w.ev[5], st);
  // 5. assembly
  AssembleArgs a;
  a.mA = w.A.out; a.mB1 = w.B1.out; a.mB2 = w.B2.out; a.mK = w.K.out; a.mZ = w.Z.out; a.mPok = w.CS.out;
  a.commit_affine = w.commit_affine;
  a.W = w.W; a.row_r = c->row_r; a.row_s = c->row_s; a.n_public = circ.n_public;
  a.sAr = scaled_blind ? w.sA.out : nullptr;
  a.rBs1 = scaled_blind ? w.rB.out : nullptr;
  a.proofs = d_proofs; a.pws = d_pws; a.P = P;
  launch_assemble(st, a);
  hipEventRecord(w.ev[6], st);
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}

static int ws_depth(size_t count, bool generic_solver = false) {
  static const int forced = [] {   // SPP_DEPTH (experiment): batches in flight, 1 .. SPP_NWS
    const char* e = getenv("SPP_DEPTH");
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= SPP_NWS ? v : 0;
  }();
  if (forced) return forced;
  return count <= 256 ? 6 : count <= 768 ? 4 : generic_solver ? 3 : 2;
}
extern "C" int spp_prove_batch_device(spp_circuit* c, size_t count, const void* d_inputs, const void* d_rs, void* d_proofs, void* d_pws,
                                      void* d_status) {
  if (!c || !d_inputs || !d_rs || !d_proofs || !d_pws || !d_status) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (count == 0) return SPP_OK;
  if (count > (1u << 20)) return fail(SPP_ERR_BAD_INPUT, "batch too large");
  std::lock_guard<std::mutex> lk(c->ctx->mu);
  HIP_TRY(hipSetDevice(c->ctx->device));
  // Batches in flight: two for big batches (more adds nothing once the latency-bound phases are covered: DESIGN 8.3); small
  // batches -- 128 proofs are one of 8 ranks' share of BASELINE.json configs[2] -- spend a larger part of their time in
  // latency-bound kernels (11 ms of sponge chain in the solver, the Horner combines), so up to six take turns
  // (128-proof audit batches, ms per step on one box: 25.3 with four in flight, 23.3 with five, 23.1 with six).
  const int depth = ws_depth(count, c->generic_solver);
  if (c->next_ws >= depth) c->next_ws = 0;
  const int wi = c->next_ws, wo = (wi + 1) % depth;
  Workspace& w = c->ws[wi];
  c->prev_ws = c->last_ws;
  c->last_ws = wi;
  c->next_ws = wo;
  // size the workspaces on the first call, so that no allocation ever lands inside a caller's timed / pipelined region
  for (int k = 0; k < depth; k++)
    if (int e = ensure_workspace(c, c->ws[k], count)) return e;
  // A batch that is not a multiple of the wave width is cut into a 64-aligned body and a tail of < 64 proofs.  Every kernel
  // of the path maps 64 proofs to a wave, so 1025 proofs used to cost a 17th wave per (slice, window) everywhere -- and before the
  // lanes were padded to waves, every wave of the MSM straddled two slices (1024 -> 1025 proofs: +23 % time, profiles/
  // round2_batch_size_sweep.txt).  The tail takes the small-batch paths (cooperative solver, lanes per (base, proof)) on the
  // OTHER proving stream, beside the body; the next call starts on that stream, behind the short tail.  SPP_NO_SPLIT=1: off.
  static const bool no_split = getenv("SPP_NO_SPLIT") != nullptr;
  const size_t tail = count % 64;
  if (count > 64 && tail && !no_split) {
    const size_t body = count - tail, nin = c->circ.n_inputs(), pwl = 12 + 32 * (size_t)(c->circ.n_public - 1);
    if (int e = prove_on_device(c, w, (uint32_t)body, (const uint8_t*)d_inputs, (const uint8_t*)d_rs, (uint8_t*)d_proofs, (uint8_t*)d_pws,
                                (uint32_t*)d_status))
      return e;
    return prove_on_device(c, c->ws[wo], (uint32_t)tail, (const uint8_t*)d_inputs + body * nin * 32, (const uint8_t*)d_rs + body * 64,
                           (uint8_t*)d_proofs + body * SPP_PROOF_LEN, (uint8_t*)d_pws + body * pwl, (uint32_t*)d_status + body);
  }
  return prove_on_device(c, w, (uint32_t)count, (const uint8_t*)d_inputs, (const uint8_t*)d_rs, (uint8_t*)d_proofs, (uint8_t*)d_pws,
                         (uint32_t*)d_status);
}
// End to end: the audit proof from the prover's raw secrets.  The input pipeline of scripts/generate_audit.py:468-641 (keygen,
// wa_commitment, RLWE encryption, quotients, packing, ct_commitment) is enqueued on the batch's own proving stream in front of the
// solver, into the workspace's input rows: nothing returns to the host between the secrets and the proof bytes, and the
// pipelining of consecutive calls is that of spp_prove_batch_device.
extern "C" int spp_prove_audit_from_secrets_device(spp_circuit* c, size_t count, const void* d_pk_a, const void* d_pk_b, const void* d_sk,
                                                   const void* d_r, const void* d_e1, const void* d_e2, const void* d_rs, void* d_proofs,
                                                   void* d_pws, void* d_status) {
  if (!c || !d_pk_a || !d_pk_b || !d_sk || !d_r || !d_e1 || !d_e2 || !d_rs || !d_proofs || !d_pws || !d_status)
    return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (c->circ.id != SPP_CIRCUIT_AUDIT || c->circ.n_inputs() != 3360) return fail(SPP_ERR_BAD_INPUT, "not the audit circuit");
  if (count == 0) return SPP_OK;
  if (count > (1u << 20)) return fail(SPP_ERR_BAD_INPUT, "batch too large");
  std::lock_guard<std::mutex> lk(c->ctx->mu);
  HIP_TRY(hipSetDevice(c->ctx->device));
  if (int e = spp_ensure_ctx_consts(c->ctx)) return e;
  const int depth = ws_depth(count);
  if (c->next_ws >= depth) c->next_ws = 0;
  const int wi = c->next_ws;
  Workspace& w = c->ws[wi];
  c->prev_ws = c->last_ws;
  c->last_ws = wi;
  c->next_ws = (wi + 1) % depth;
  for (int k = 0; k < depth; k++)
    if (int e = ensure_workspace(c, c->ws[k], count)) return e;
  const size_t need = spp_audit_scratch_bytes(count);
  if (need > w.audit_scratch_cap) {
    HIP_TRY(hipStreamSynchronize(w.st));
    if (w.audit_scratch) HIP_TRY(hipFree(w.audit_scratch));
    w.audit_scratch = nullptr;
    w.audit_scratch_cap = 0;
    HIP_TRY(hipMalloc(&w.audit_scratch, need));
    w.audit_scratch_cap = need;
  }
  if (int e = spp_audit_inputs_enqueue(c->ctx, w.st, w.audit_scratch, (const uint32_t*)d_pk_a, (const uint32_t*)d_pk_b, (uint32_t)count,
                                       (const uint8_t*)d_sk, (const int8_t*)d_r, (const int8_t*)d_e1, (const int8_t*)d_e2, w.d_inputs))
    return e;
  return prove_on_device(c, w, (uint32_t)count, w.d_inputs, (const uint8_t*)d_rs, (uint8_t*)d_proofs, (uint8_t*)d_pws, (uint32_t*)d_status);
}
extern "C" int spp_commitment_challenge(spp_circuit* c, size_t count, const uint8_t* inputs, uint8_t* challenges) {
  if (!c || !inputs || !challenges) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (count == 0) return SPP_OK;
  if (count > 4096) return fail(SPP_ERR_BAD_INPUT, "at most 4096 rows per call");
  if (c->CB.N == 0) return fail(SPP_ERR_BAD_INPUT, "the circuit has no commitment");
  std::lock_guard<std::mutex> lk(c->ctx->mu);
  HIP_TRY(hipSetDevice(c->ctx->device));
  Workspace& w = c->ws[c->next_ws];
  if (int e = ensure_workspace(c, w, count)) return e;
  hipStream_t st = w.st;
  const uint32_t P = (uint32_t)count;
  const size_t nin = c->circ.n_inputs();
  HIP_TRY(hipMemcpyAsync(w.d_inputs, inputs, nin * 32 * count, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(w.d_rs, 0, 64 * count, st));
  launch_load_inputs(st, w.d_inputs, w.d_rs, w.W, (uint32_t)nin, c->circ.n_wires, P);
  // whatever the program computes before the commitment (nothing for an all-inputs system), then commit and hash
  for (const SolveStep& s : c->schedule) {
    if (s.kind == SolveStep::COMMIT) break;
    switch (s.kind) {
      case SolveStep::SEQ: launch_solve(st, c->dc, w.W, w.scratch, s.a, s.b, P); break;
      case SolveStep::BATCH_DIV: launch_batch_div(st, c->dc, w.W, w.scratch, s.a, s.b, P); break;
      case SolveStep::COUNT8: launch_count8(st, c->dc, w.W, w.counters, s.a, s.b, s.c, P); break;
      default: break;
    }
  }
  run_msm(c, w, c->CB, w.CB, P, false);
  launch_challenge(st, w.CB.out, w.W, c->circ.challenge_wire, P, w.commit_affine, w.d_status);
  std::vector<Fr> out(count);
  HIP_TRY(hipMemcpyAsync(out.data(), w.W + (size_t)c->circ.challenge_wire * P, sizeof(Fr) * count, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for (size_t i = 0; i < count; i++) out[i].to_bytes_be(challenges + 32 * i);
  return SPP_OK;
}
extern "C" int spp_sync(spp_circuit* c) {
  if (!c) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  HIP_TRY(hipSetDevice(c->ctx->device));
  for (auto& w : c->ws)
    if (w.st) HIP_TRY(hipStreamSynchronize(w.st));
  return SPP_OK;
}
extern "C" int spp_last_timings(spp_circuit* c, float ms[9]) { return spp_timings(c, 0, ms); }
extern "C" int spp_timings(spp_circuit* c, int which, float ms[9]) {
  if (!c || !ms) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  HIP_TRY(hipSetDevice(c->ctx->device));
  Workspace& w = c->ws[which ? c->prev_ws : c->last_ws];
  if (w.cap == 0) return fail(SPP_ERR_BAD_INPUT, "no such batch");
  HIP_TRY(hipStreamSynchronize(w.st));
  for (int i = 0; i < 6; i++) {
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, w.ev[i], w.ev[i + 1]));
    ms[i] = t;
  }
  float tot = 0;
  HIP_TRY(hipEventElapsedTime(&tot, w.ev[0], w.ev[6]));
  ms[6] = tot;
  float sum = 0;
  for (size_t i = 0; i < w.msm_ev_used; i++) {
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, w.msm_ev[i].first, w.msm_ev[i].second));
    sum += t;
  }
  ms[7] = w.msm_ev_used ? sum / (float)w.msm_ev_used : 0.f;
  ms[8] = (float)w.msm_ev_used;
  return SPP_OK;
}

// per-launch durations of the MSM kernels of one batch, in launch order: commitment (CB), A, B1, K, Z, PoK (CS), then the G2 set
extern "C" int spp_msm_kernel_ms(spp_circuit* c, int which, float ms[7]) {
  if (!c || !ms) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  HIP_TRY(hipSetDevice(c->ctx->device));
  Workspace& w = c->ws[which ? c->prev_ws : c->last_ws];
  if (w.cap == 0) return fail(SPP_ERR_BAD_INPUT, "no such batch");
  HIP_TRY(hipStreamSynchronize(w.st));
  HIP_TRY(hipStreamSynchronize(w.st2));
  if (w.own_st2p) HIP_TRY(hipStreamSynchronize(w.own_st2p));
  for (int i = 0; i < 7; i++) ms[i] = 0.f;
  for (size_t i = 0; i < w.msm_ev_used && i < 6; i++) HIP_TRY(hipEventElapsedTime(&ms[i], w.msm_ev[i].first, w.msm_ev[i].second));
  if (c->B2.N) HIP_TRY(hipEventElapsedTime(&ms[6], w.g2_ev.first, w.g2_ev.second));
  return SPP_OK;
}
// on = 1: both batch workspaces and the G2 MSM run on ONE stream (kernel durations are then not stretched by another stream
// sharing the chip: what a roofline figure needs); on = 0: the pipelined default.  Drains the device first.
extern "C" int spp_set_serial(spp_circuit* c, int on) {
  if (!c) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  std::lock_guard<std::mutex> lk(c->ctx->mu);
  HIP_TRY(hipSetDevice(c->ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  for (int k = 0; k < SPP_NWS; k++) {
    Workspace& w = c->ws[k];
    w.st = on ? c->ws[0].own_st : w.own_st;
    w.st2 = on ? w.st : w.own_st2;
  }
  return SPP_OK;
}

extern "C" int spp_prove_batch(spp_circuit* c, size_t count, const uint8_t* inputs, const uint8_t* rs, uint8_t* proofs, uint8_t* pws,
                               int32_t* status) {
  if (!c || !inputs || !proofs || !pws) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (count == 0) return SPP_OK;
  std::vector<uint8_t> rnd;
  if (!rs) {
    rnd.resize(64 * count);
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f || fread(rnd.data(), 1, rnd.size(), f) != rnd.size()) {
      if (f) fclose(f);
      return fail(SPP_ERR_IO, "cannot read /dev/urandom");
    }
    fclose(f);
    rs = rnd.data();
  }
  std::vector<uint32_t> st(count);
  {
    // Large host batches are cut into chunks that alternate between the two workspaces / proving streams, so the
    // copies and the solver of chunk k+1 overlap the MSMs of chunk k exactly as consecutive spp_prove_batch_device calls
    // do, and the workspaces never grow beyond one chunk.
    std::lock_guard<std::mutex> lk(c->ctx->mu);
    HIP_TRY(hipSetDevice(c->ctx->device));
    const size_t pref = c->circ.n_wires <= 16384 ? 4096 : 2048;  // batch sizes at which the per-launch overheads are amortised
    const size_t chunk = count <= pref + pref / 2 ? count : pref;
    const size_t nin = c->circ.n_inputs(), npub = c->circ.n_public - 1, pwl = 12 + 32 * npub;
    // copies back to pageable host memory block the caller until their stream has drained, so the results of chunk k are
    // fetched only after chunk k+1 has been enqueued on the other stream
    auto fetch = [&](Workspace& w, size_t off, size_t n) -> int {
      hipStream_t s = w.st;
      HIP_TRY(hipMemcpyAsync(proofs + (size_t)SPP_PROOF_LEN * off, w.d_proofs, (size_t)SPP_PROOF_LEN * n, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(pws + pwl * off, w.d_pws, pwl * n, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(st.data() + off, w.d_status, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      return 0;
    };
    Workspace* prev_w = nullptr;
    size_t prev_off = 0, prev_n = 0;
    // chunk list: a last chunk that is not a multiple of the wave width is cut into a 64-aligned body and a tail (see
    // spp_prove_batch_device); the two alternate workspaces like any other pair of chunks
    std::vector<std::pair<size_t, size_t>> chunks;
    for (size_t off = 0; off < count; off += chunk) {
      const size_t n = std::min(chunk, count - off), t = n % 64;
      if (n > 64 && t && !getenv("SPP_NO_SPLIT")) {
        chunks.push_back({off, n - t});
        chunks.push_back({off + n - t, t});
      } else chunks.push_back({off, n});
    }
    for (const auto& ch : chunks) {
      const size_t off = ch.first, n = ch.second;
      if (c->next_ws >= 2) c->next_ws = 0;
      Workspace& w = c->ws[c->next_ws];
      c->prev_ws = c->last_ws;
      c->last_ws = c->next_ws;
      c->next_ws ^= 1;
      if (int e = ensure_workspace(c, w, n)) return e;
      hipStream_t s = w.st;
      HIP_TRY(hipMemcpyAsync(w.d_inputs, inputs + nin * 32 * off, nin * 32 * n, hipMemcpyHostToDevice, s));
      HIP_TRY(hipMemcpyAsync(w.d_rs, rs + 64 * off, 64 * n, hipMemcpyHostToDevice, s));
      if (int e = prove_on_device(c, w, (uint32_t)n, w.d_inputs, w.d_rs, w.d_proofs, w.d_pws, w.d_status)) return e;
      if (prev_w)
        if (int e = fetch(*prev_w, prev_off, prev_n)) return e;
      prev_w = &w;
      prev_off = off;
      prev_n = n;
    }
    if (prev_w)
      if (int e = fetch(*prev_w, prev_off, prev_n)) return e;
    for (auto& w : c->ws)
      if (w.st) HIP_TRY(hipStreamSynchronize(w.st));
  }
  int rc = SPP_OK;
  for (size_t i = 0; i < count; i++) {
    int32_t v = st[i] ? SPP_ERR_UNSAT : SPP_OK;
    if (status) status[i] = v;
    if (v && rc == SPP_OK) rc = fail(SPP_ERR_UNSAT, "proof %zu: inputs do not satisfy the circuit", i);
    if (v) memset(proofs + (size_t)SPP_PROOF_LEN * i, 0, SPP_PROOF_LEN);
  }
  return rc;
}

extern "C" int spp_prove_withdraw(spp_circuit* c, const spp_withdraw_inputs* in, const uint8_t rs_seed[64], uint8_t proof[SPP_PROOF_LEN],
                                  uint8_t pw[SPP_WITHDRAW_PW_LEN]) {
  if (!c || !in || !proof || !pw) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (c->circ.id != SPP_CIRCUIT_WITHDRAW) return fail(SPP_ERR_BAD_INPUT, "not a withdraw circuit");
  if (c->circ.n_inputs() != 10 + SPP_TREE_DEPTH) return fail(SPP_ERR_BAD_INPUT, "this entry point serves the depth-16 circuit; use spp_prove_batch");
  std::vector<uint8_t> buf(26 * 32, 0);
  auto put = [&](int i, const uint8_t* v) { memcpy(buf.data() + 32 * i, v, 32); };
  auto put64 = [&](int i, uint64_t v) { for (int k = 0; k < 8; k++) buf[32 * i + 31 - k] = (uint8_t)(v >> (8 * k)); };
  put(0, in->root); put(1, in->nullifier); put(2, in->recipient); put64(3, in->amount); put(4, in->wa_commitment);
  put(5, in->secret_key); put(6, in->owner_x); put(7, in->owner_y); put(8, in->randomness); put64(9, in->index);
  for (int i = 0; i < SPP_TREE_DEPTH; i++) put(10 + i, in->siblings[i]);
  int32_t st = 0;
  return spp_prove_batch(c, 1, buf.data(), rs_seed, proof, pw, &st);
}

extern "C" int spp_debug_witness(spp_circuit* c, uint8_t* out, size_t n_wires) {
  if (!c || !out) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  std::lock_guard<std::mutex> lk(c->ctx->mu);
  Workspace& w = c->ws[c->last_ws];
  if (w.cap == 0) return fail(SPP_ERR_BAD_INPUT, "no batch has been proved yet");
  HIP_TRY(hipSetDevice(c->ctx->device));
  HIP_TRY(hipStreamSynchronize(w.st));
  size_t P = w.last_P;   // column 0 of W at the stride of the last batch
  std::vector<Fr> col(std::min<size_t>(n_wires, c->circ.n_wires));
  for (size_t i = 0; i < col.size(); i++) HIP_TRY(hipMemcpy(&col[i], w.W + i * P, sizeof(Fr), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < col.size(); i++) col[i].to_bytes_be(out + 32 * i);
  return SPP_OK;
}

// -----------------------------------------------------------------------------------------------------
// setup on the GPU
// -----------------------------------------------------------------------------------------------------
static void wr32(std::vector<uint8_t>& o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static void wr32be(std::vector<uint8_t>& o, uint32_t v) { for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i))); }
static void wr_g1(std::vector<uint8_t>& o, const G1Affine& p) { uint8_t b[64]; g1_to_raw(p, b); o.insert(o.end(), b, b + 64); }
static void wr_g2(std::vector<uint8_t>& o, const G2Affine& p) { uint8_t b[128]; g2_to_raw(p, b); o.insert(o.end(), b, b + 128); }
static bool write_file(const char* path, const std::vector<uint8_t>& o) {
  FILE* f = fopen(path, "wb");
  if (!f) return false;
  bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
  fclose(f);
  return ok;
}

extern "C" int spp_setup(spp_ctx* ctx, const char* circuit_path, const uint8_t seed[32], const char* pk_path, const char* vk_path) {
  if (!ctx || !circuit_path || !seed || !pk_path || !vk_path) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  Circuit circ;
  if (!circ.load(circuit_path)) return fail(SPP_ERR_IO, "cannot read circuit %s", circuit_path);
  // toxic waste = hash_to_fr(seed, "spp-groth16-setup-v1", 7)
  Fr tox[7];
  {
    const char* dst = "spp-groth16-setup-v1";
    uint8_t u[7 * 48];
    expand_message_xmd(seed, 32, (const uint8_t*)dst, strlen(dst), u, sizeof u);
    for (int i = 0; i < 7; i++) {
      uint32_t w[12];
      for (int k = 0; k < 12; k++) {
        const uint8_t* q = u + 48 * i + 4 * k;
        w[k] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
      }
      tox[i] = fr_from_wide48(w);
    }
  }
  const Fr tau = tox[0], alpha = tox[1], beta = tox[2], gamma = tox[3], delta = tox[4], sigma = tox[5], rho = tox[6];
  const uint32_t logn = circ.domain_log, n = 1u << logn, W = circ.n_wires;
  // Lagrange basis at tau
  std::vector<Fr> L(n), den(n), pre(n);
  {
    Fr omega = fr_root_of_unity(logn);
    Fr zt = tau.pow_u64(n) - Fr::one();
    Fr scale = zt * Fr::from_u64(n).inv();
    Fr wk = Fr::one(), acc = Fr::one();
    for (uint32_t k = 0; k < n; k++) {
      den[k] = tau - wk;
      pre[k] = acc;
      acc = acc * den[k];
      L[k] = wk;
      wk = wk * omega;
    }
    Fr ia = acc.inv();
    for (uint32_t k = n; k-- > 0;) {
      Fr di = ia * pre[k];
      ia = ia * den[k];
      L[k] = L[k] * di * scale;
    }
  }
  std::vector<Fr> aw(W, Fr::zero()), bw(W, Fr::zero()), cw(W, Fr::zero());
  {
    const Sparse* M[3] = {&circ.A, &circ.B, &circ.C};
    std::vector<Fr>* O[3] = {&aw, &bw, &cw};
    for (int m = 0; m < 3; m++)
      for (uint32_t k = 0; k < circ.n_constraints; k++)
        for (uint32_t i = M[m]->rowptr[k]; i < M[m]->rowptr[k + 1]; i++) {
          const Term& t = M[m]->terms[i];
          (*O[m])[t.wire] = (*O[m])[t.wire] + circ.coeffs[t.coeff] * L[k];
        }
  }
  std::vector<uint8_t> cls(W, 0);
  for (uint32_t j = 0; j < circ.n_public; j++) cls[j] = 1;
  cls[circ.challenge_wire] = 1;
  for (uint32_t w : circ.committed) cls[w] = 2;
  const Fr gi = gamma.inv(), di = delta.inv();
  // scalar vectors, one fixed-base multiplication each:
  //   G1: [A(W) | B1(W) | K(W) | S(W) | Z(n-1) | alpha beta delta]    G2: [B2(W) | beta gamma delta rho -rho*sigma]
  std::vector<Fr> s1, s2;
  s1.reserve((size_t)4 * W + n + 3);
  for (uint32_t j = 0; j < W; j++) s1.push_back(aw[j]);
  for (uint32_t j = 0; j < W; j++) s1.push_back(bw[j]);
  std::vector<Fr> kk(W);
  for (uint32_t j = 0; j < W; j++) kk[j] = (beta * aw[j] + alpha * bw[j] + cw[j]) * (cls[j] ? gi : di);
  for (uint32_t j = 0; j < W; j++) s1.push_back(kk[j]);
  for (uint32_t j = 0; j < W; j++) s1.push_back(cls[j] == 2 ? kk[j] * sigma : Fr::zero());
  {
    Fr zt = tau.pow_u64(n) - Fr::one();
    Fr pw = zt * di;
    for (uint32_t i = 0; i + 1 < n; i++) {
      s1.push_back(pw);
      pw = pw * tau;
    }
  }
  s1.push_back(alpha); s1.push_back(beta); s1.push_back(delta);
  for (uint32_t j = 0; j < W; j++) s2.push_back(bw[j]);
  s2.push_back(beta); s2.push_back(gamma); s2.push_back(delta); s2.push_back(rho); s2.push_back((rho * sigma).neg());

  // generator tables (c = 8) and the batched fixed-base multiplications on the GPU
  const uint32_t cb = 8, Wn = msm_windows(cb), E = 1u << (cb - 1);
  G1Affine g1{Fq::from_u64(1), Fq::from_u64(2)};
  auto fq_dec = [](const char* dec) {
    Fq acc = Fq::zero(), ten = Fq::from_u64(10);
    for (const char* ch = dec; *ch; ch++) acc = acc * ten + Fq::from_u64((uint64_t)(*ch - '0'));
    return acc;
  };
  G2Affine g2;
  g2.x.c0 = fq_dec("10857046999023057135944570762232829481370756359578518086990519993285655852781");
  g2.x.c1 = fq_dec("11559732032986387107991004021392285783925812861821192530917403151452391805634");
  g2.y.c0 = fq_dec("8495653923123431417604973247489272438418190587263600148770280649306958101930");
  g2.y.c1 = fq_dec("4082367875863433681332203403145435568316851327593401208105741076214120093531");
  DevBuf d_g1, d_g2, t1, t2, tmp1, tmp2, pre1, pre2, d_s1, d_s2, o1, o2;   // released on every return path
  HIP_TRY(d_g1.alloc(sizeof g1)); HIP_TRY(hipMemcpy(d_g1.p, &g1, sizeof g1, hipMemcpyHostToDevice));
  HIP_TRY(d_g2.alloc(sizeof g2)); HIP_TRY(hipMemcpy(d_g2.p, &g2, sizeof g2, hipMemcpyHostToDevice));
  const size_t ge = msm_table_elems(1, cb, Wn), gr = ((size_t)Wn + 63) / 64 * 64;
  HIP_TRY(t1.alloc(sizeof(G1Affine) * ge)); HIP_TRY(t2.alloc(sizeof(G2Affine) * ge));
  HIP_TRY(tmp1.alloc(sizeof(G1XYZZ) * gr * E)); HIP_TRY(tmp2.alloc(sizeof(G2XYZZ) * gr * E));
  HIP_TRY(pre1.alloc(sizeof(Fq) * gr * E)); HIP_TRY(pre2.alloc(sizeof(Fq2) * gr * E));
  HIP_TRY(d_s1.alloc(sizeof(Fr) * s1.size())); HIP_TRY(d_s2.alloc(sizeof(Fr) * s2.size()));
  HIP_TRY(o1.alloc(sizeof(G1Affine) * s1.size())); HIP_TRY(o2.alloc(sizeof(G2Affine) * s2.size()));
  HIP_TRY(hipMemcpyAsync(d_s1.p, s1.data(), sizeof(Fr) * s1.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_s2.p, s2.data(), sizeof(Fr) * s2.size(), hipMemcpyHostToDevice, st));
  launch_build_table<Fq>(st, d_g1.as<G1Affine>(), 1, cb, Wn, 0, (uint32_t)gr, t1.as<G1Affine>(), tmp1.as<G1XYZZ>(), pre1.as<Fq>());
  launch_build_table<Fq2>(st, d_g2.as<G2Affine>(), 1, cb, Wn, 0, (uint32_t)gr, t2.as<G2Affine>(), tmp2.as<G2XYZZ>(), pre2.as<Fq2>());
  launch_fixed_base_mul<Fq>(st, t1.as<G1Affine>(), cb, d_s1.as<Fr>(), (uint32_t)s1.size(), o1.as<G1Affine>(), nullptr);
  launch_fixed_base_mul<Fq2>(st, t2.as<G2Affine>(), cb, d_s2.as<Fr>(), (uint32_t)s2.size(), o2.as<G2Affine>(), nullptr);
  std::vector<G1Affine> p1(s1.size());
  std::vector<G2Affine> p2(s2.size());
  HIP_TRY(hipMemcpyAsync(p1.data(), o1.p, sizeof(G1Affine) * p1.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(p2.data(), o2.p, sizeof(G2Affine) * p2.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());

  const G1Affine* pA = p1.data();
  const G1Affine* pB1 = pA + W;
  const G1Affine* pK = pB1 + W;
  const G1Affine* pS = pK + W;
  const G1Affine* pZ = pS + W;
  const G1Affine* pC = pZ + (n - 1);
  const G2Affine* pB2 = p2.data();
  const G2Affine* pC2 = pB2 + W;

  std::vector<uint8_t> o;
  wr32(o, 0x4b505053u); wr32(o, 1);
  wr32(o, circ.id); wr32(o, W); wr32(o, logn); wr32(o, circ.n_public); wr32(o, circ.challenge_wire);
  wr_g1(o, pC[0]); wr_g1(o, pC[1]); wr_g1(o, pC[2]); wr_g2(o, pC2[0]); wr_g2(o, pC2[2]);
  auto sec1 = [&](const G1Affine* pts, auto pred) {
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < W; j++) cnt += pred(j) ? 1 : 0;
    wr32(o, cnt);
    for (uint32_t j = 0; j < W; j++) if (pred(j)) wr32(o, j);
    for (uint32_t j = 0; j < W; j++) if (pred(j)) wr_g1(o, pts[j]);
  };
  sec1(pA, [&](uint32_t j) { return !pA[j].is_inf(); });
  sec1(pB1, [&](uint32_t j) { return !pB1[j].is_inf(); });
  {
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < W; j++) cnt += !pB2[j].is_inf();
    wr32(o, cnt);
    for (uint32_t j = 0; j < W; j++) if (!pB2[j].is_inf()) wr32(o, j);
    for (uint32_t j = 0; j < W; j++) if (!pB2[j].is_inf()) wr_g2(o, pB2[j]);
  }
  sec1(pK, [&](uint32_t j) { return cls[j] == 0 && !pK[j].is_inf(); });
  wr32(o, n - 1);
  for (uint32_t i = 0; i + 1 < n; i++) wr_g1(o, pZ[i]);
  wr32(o, (uint32_t)circ.committed.size());
  for (uint32_t w : circ.committed) wr32(o, w);
  for (uint32_t w : circ.committed) wr_g1(o, pK[w]);
  wr32(o, (uint32_t)circ.committed.size());
  for (uint32_t w : circ.committed) wr32(o, w);
  for (uint32_t w : circ.committed) wr_g1(o, pS[w]);
  if (!write_file(pk_path, o)) return fail(SPP_ERR_IO, "cannot write %s", pk_path);

  std::vector<uint8_t> v;
  wr_g1(v, pC[0]); wr_g1(v, pC[1]); wr_g2(v, pC2[0]); wr_g2(v, pC2[1]); wr_g1(v, pC[2]); wr_g2(v, pC2[2]);
  wr32be(v, circ.n_public + 1);
  for (uint32_t j = 0; j < circ.n_public; j++) wr_g1(v, pK[j]);
  wr_g1(v, pK[circ.challenge_wire]);
  wr32be(v, 1); wr32be(v, 0); wr32be(v, 1);
  wr_g2(v, pC2[3]); wr_g2(v, pC2[4]);
  if (!write_file(vk_path, v)) return fail(SPP_ERR_IO, "cannot write %s", vk_path);
  return SPP_OK;
}

// -----------------------------------------------------------------------------------------------------
// table-based MSM over caller-supplied bases (unit entry point; uses the table builder above)
// -----------------------------------------------------------------------------------------------------
template <class F> static Affine<F> point_from_raw(const uint8_t* b);
template <> Affine<Fq> point_from_raw<Fq>(const uint8_t* b) { return g1_from_raw(b); }
template <> Affine<Fq2> point_from_raw<Fq2>(const uint8_t* b) { return g2_from_raw(b); }
static void point_to_raw(const G1Affine& p, uint8_t* b) { g1_to_raw(p, b); }
static void point_to_raw(const G2Affine& p, uint8_t* b) { g2_to_raw(p, b); }
template <class F, size_t PT_BYTES>
static int msm_fixed_unit(spp_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, int window_bits, uint8_t* out) {
  if (!ctx || !out || (n && (!bases || !scalars))) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (window_bits == 0) window_bits = 8;
  if (window_bits < 4 || window_bits > 16) return fail(SPP_ERR_BAD_INPUT, "window_bits outside [4,16]");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint32_t cb = (uint32_t)window_bits, Wn = msm_windows(cb), E = 1u << (cb - 1);
  (void)E;
  if ((uint64_t)msm_table_elems((uint32_t)n, cb, Wn) * PT_BYTES > ((uint64_t)64 << 30)) return fail(SPP_ERR_BAD_INPUT, "table would exceed 64 GiB; use a smaller window");
  std::vector<Affine<F>> pts(n);
  std::vector<Fr> sc(n);
  std::vector<uint32_t> rows(n);
  for (size_t i = 0; i < n; i++) {
    pts[i] = point_from_raw<F>(bases + PT_BYTES * i);
    sc[i] = Fr::from_bytes_be(scalars + 32 * i);
    rows[i] = (uint32_t)i;
  }
  spp_circuit tmpc;   // only used as an owner of device allocations
  tmpc.ctx = ctx;
  tmpc.c_bits = cb;
  Affine<F>* table = nullptr;
  int e = build_table_chunked<F>(&tmpc, pts, cb, Wn, &table);
  Fr* d_sc = nullptr;
  uint32_t* d_rows = nullptr;
  XYZZ<F> *partial = nullptr, *d_out = nullptr;
  DevBuf dig;
  const MsmPlan pl = msm_plan((uint32_t)n, 1, cb, Wn);
  if (!e) e = own_upload(&tmpc, &d_sc, sc);
  if (!e) e = own_upload(&tmpc, &d_rows, rows);
  if (!e && hipMalloc((void**)&partial, sizeof(XYZZ<F>) * std::max<size_t>(pl.partial_elems(1), 1)) != hipSuccess) e = fail(SPP_ERR_HIP, "hipMalloc");
  if (!e && hipMalloc((void**)&d_out, sizeof(XYZZ<F>)) != hipSuccess) e = fail(SPP_ERR_HIP, "hipMalloc");
  if (!e && dig.alloc(sizeof(int16_t) * std::max<size_t>(msm_digit_elems((uint32_t)n, 1, cb), 1)) != hipSuccess) e = fail(SPP_ERR_HIP, "hipMalloc");
  XYZZ<F> res = XYZZ<F>::infinity();
  if (!e) {
    launch_msm_digits(st, d_rows, d_sc, dig.as<int16_t>(), (uint32_t)n, 1, cb);
    launch_msm_accumulate<F>(st, table, dig.as<int16_t>(), partial, (uint32_t)n, 1, cb, pl);
    launch_msm_reduce<F>(st, partial, d_out, 1, pl, cb, n == 0);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) e = fail(SPP_ERR_HIP, "msm kernels failed");
    else if (hipMemcpy(&res, d_out, sizeof res, hipMemcpyDeviceToHost) != hipSuccess) e = fail(SPP_ERR_HIP, "copy back failed");
  }
  for (void* p : tmpc.owned) hipFree(p);
  if (partial) hipFree(partial);
  if (d_out) hipFree(d_out);
  if (e) return e;
  point_to_raw(res.to_affine(), out);
  return SPP_OK;
}
extern "C" int spp_msm_g1(spp_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, int window_bits, uint8_t out[64]) {
  return msm_fixed_unit<Fq, 64>(ctx, bases, scalars, n, window_bits, out);
}
// the same walk over G2 bases (128 B, gnark raw X.A1|X.A0|Y.A1|Y.A0): what Bs of a proof comes from (k_msm_fixed<Fq2>)
extern "C" int spp_msm_g2(spp_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, int window_bits, uint8_t out[128]) {
  return msm_fixed_unit<Fq2, 128>(ctx, bases, scalars, n, window_bits, out);
}

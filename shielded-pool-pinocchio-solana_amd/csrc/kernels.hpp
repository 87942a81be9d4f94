// Launch wrappers of the HIP kernels (one .hip file per kernel family), shared by spp_api.cpp.
// Data layout in HBM, used by every kernel of the proving path ("batch-minor"):
//   witness   W   [rows][P]      Fr Montgomery, 32 B; row = wire (plus 3 blinding rows r, s, rs)
//   abc           [3][n][P]      constraint evaluations <A_k,w>, <B_k,w>, <C_k,w>, zero padded to n
//   MSM tables    [N][Wn][E]     affine multiples (d+1) * 2^(c*j) * Base_i,  E = 2^(c-1)
//   partials      [S][P]         XYZZ partial sums of one MSM, reduced over S by msm_reduce
// P (proofs in the batch) is the fastest index everywhere, so a wavefront whose 64 lanes hold 64 proofs
// reads/writes 2 KiB contiguous per field element and executes one uniform instruction stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bn254.hpp"
#include "rlwe_ntt.hpp"

namespace spp {

struct DevSparse {
  const uint32_t* rowptr;
  const uint32_t* wire;
  const uint32_t* coeff;   // bit31: coefficient is +1, bit30: coefficient is -1, low bits: table index
  const uint32_t* lit;     // 0, or the coefficient itself when |c| < 2^28: magnitude, bit31 = negative (matrix evaluation
                           // accumulates those terms as wide integers instead of doing a field multiplication each)
};
static constexpr uint32_t COEFF_ONE = 0x80000000u;
static constexpr uint32_t COEFF_MINUS_ONE = 0x40000000u;
static constexpr uint32_t COEFF_MASK = 0x3fffffffu;

struct DevCircuit {
  DevSparse A, B, C, H;
  const Fr* coeffs;
  const Fr* aux;            // hint constants (Grumpkin window tables)
  const uint32_t* program;
  uint32_t n_wires, n_constraints, n_public, n_inputs, challenge_wire;
  // matrix evaluation: constraints grouped into runs of consecutive rows whose B rows are identical (the four
  // constraints of a Poseidon S-box share B = the S-box input): run r covers rows run_start[r] .. run_start[r+1]-1
  const uint32_t* run_start;
  uint32_t n_runs;
  uint32_t max_row_terms;   // longest row of A, B, C (small batches of circuits with long rows take the 16-lanes-per-row evaluation)
  // solver shortcuts per constraint: bit 0 = the B row is identical to the B row of constraint k - 1, bit 1 = the A row is
  // identical to the B row (a square).  A solver lane that has just evaluated row k - 1 reuses the value instead of walking
  // the same linear form again: the rows of a power map share their B side, and compiled (ACIR) circuits have long forms.
  const uint8_t* row_flags;
  // "small rows" of the matrix evaluation (spp_api.cpp, small_rows_plan): rows whose every term is a small integer coefficient
  // times a wire that the lookup argument bounds to a byte-sized range (the audit circuit's 1 088 quotient equations: 1 024
  // public-key coefficients < 2^28 times noise values in [-3, 3], generate_audit.py:57-66,236-243, plus ciphertext bytes).  The
  // bounded wires are extracted once per batch as int16 (k_small_extract), the rows are summed as 64-bit integers
  // (k_spmv_small_rows) and land in the a / b / c arrays before k_spmv_check runs, which then skips them (row_small bits).
  const uint32_t* sm_wires;     // wire of small slot s (slot 0 = wire 0, the constant one)
  const int32_t* sm_lo;         // lowest value the lookup argument allows for slot s (highest = lo + 255)
  uint32_t sm_nslots;
  const uint32_t* sm_rowptr;    // per small row: terms [rowptr[r], rowptr[r+1])
  const uint32_t* sm_slot;      // slot of a term
  const int32_t* sm_coef;       // its coefficient
  const uint32_t* sm_rest_ptr;  // per small row: its few OTHER terms (any wire, any coefficient) [rest_ptr[r], rest_ptr[r+1]),
  const uint32_t* sm_rest_wire; //   added with field arithmetic (the quotient k * q and the message bits of a quotient equation)
  const uint32_t* sm_rest_coeff;//   index into coeffs
  const uint32_t* sm_row_out;   // matrix (0 = A, 1 = B, 2 = C) << 30 | constraint
  uint32_t sm_nrows;
  const uint8_t* row_small;     // per constraint: bit 0 A, bit 1 B, bit 2 C is a small row (nullptr: none)
  // LONG rows (more than LONG_ROW_MIN terms and not small: the audit circuit's lookup sum has 6 720): one lane per (run, proof)
  // would leave 32 waves walking such a row alone for milliseconds after the rest of the grid has drained.  k_spmv_long_rows
  // evaluates them first, 16 lanes per (row, proof) meeting through LDS, and they carry the same "already in abc" bits.
  const uint32_t* lg_rows;      // (matrix << 30) | constraint
  uint32_t lg_n;
  const uint8_t* row_long;      // the bits of the long rows alone (what k_spmv_check reads when the small-row path is off)
  // hash constants (Montgomery)
  const Fr* pos3_rc;  const Fr* pos3_mds;   // t=3: 195 rc, 9 mds (row-major)
  const Fr* pos5_rc;  const Fr* pos5_mds;   // t=5: 340 rc, 25 mds
  const Fr* p2_rc;    const Fr* p2_mu;      // 88 rc, 4 mu
  const Fr* byte_mont;                      // Montgomery forms of 0..255
  // Poseidon MDS matrices in the 9x29-bit form, scaled by 2^261 (f29.hpp): mont29(state word, entry) stays an x*2^256 word
  const uint32_t* pos3_mds29;               // 9 x 9 limbs
  const uint32_t* pos5_mds29;               // 25 x 9 limbs
};

// Poseidon / Poseidon2 constants in HBM (Montgomery form), shared by the solver and the stand-alone hash kernels
struct HashConsts {
  const Fr* pos3_rc;  const Fr* pos3_mds;   // rc words canonical (< p): poseidon29.hpp's value bounds rely on it
  const Fr* pos5_rc;  const Fr* pos5_mds;
  const Fr* p2_rc;    const Fr* p2_mu;
  const uint32_t* pos3_mds29;               // MDS entries * 2^261 in 9 x 29-bit limbs (poseidon29.hpp)
  const uint32_t* pos5_mds29;
};

// ---- stand-alone witness-input kernels (kernels_witness.hip) ----
// device-resident constants of the RLWE witness kernel (rlwe_ntt.hpp) + scratch for the transformed public key
struct RlwePkDev {
  int32_t hat[2][2][1024];    // [a | b][field][i]: NTT(pk psi^j)[i] / 1024, Montgomery form, in (-p, p)
  uint32_t nzeros[2];
  uint16_t zeros[2][1024];    // positions of zero coefficients (wrap correction)
};
struct RlweDev {
  RnTables tb;
  int32_t pk_scale[2];
  RlwePkDev* pk = nullptr;
};
void launch_rlwe_witness(hipStream_t st, const RlweDev& rd, const uint32_t* pk_a, const uint32_t* pk_b, const int8_t* r, const int8_t* e1,
                         const int8_t* e2, const uint8_t* msg, uint32_t* c0, uint32_t* c1, int32_t* k0, int32_t* k1, uint8_t* packed_be,
                         uint32_t count);
void launch_poseidon_hash(hipStream_t st, HashConsts hc, const uint8_t* in_be, uint32_t arity, uint8_t* out_be, uint32_t count);
void launch_merkle_path(hipStream_t st, HashConsts hc, const uint8_t* leaf_be, const uint64_t* index, const uint8_t* siblings_be,
                        uint32_t depth, uint8_t* root_be, uint32_t count);
void launch_merkle_level(hipStream_t st, HashConsts hc, const Fr* children, uint32_t n_children, Fr dflt, Fr* parents, uint32_t n_parents);
// incremental Poseidon-Merkle tree resident in HBM (spp_merkle_tree_*): level[l] holds count[l] = ceil(n_leaves / 2^l) nodes
struct MerkleTreeDev {
  Fr* level[33];
  uint64_t count[33];
  Fr dflt[33];
  uint32_t depth;
};
void launch_merkle_defaults(hipStream_t st, HashConsts hc, Fr* out, uint32_t depth);
void launch_merkle_update(hipStream_t st, HashConsts hc, const Fr* children, uint64_t n_children, const Fr* dflt_level, Fr* parents,
                          uint64_t first, uint32_t n);
void launch_merkle_gather(hipStream_t st, const MerkleTreeDev* t, uint32_t depth, const uint64_t* indices, uint32_t nq, uint8_t* out_be);
void launch_fr_from_be(hipStream_t st, const uint8_t* in, Fr* out, uint32_t n);
void launch_fr_to_be(hipStream_t st, const Fr* in, uint8_t* out, uint32_t n);
void launch_grumpkin_keygen(hipStream_t st, const GkAffine* table, const uint8_t* sk_be, uint8_t* xy_be, uint32_t count);
void launch_poseidon2_sponge(hipStream_t st, HashConsts hc, const uint8_t* in_be, uint32_t n, uint8_t* out_be, uint32_t count);
void launch_rlwe_decrypt(hipStream_t st, const uint32_t* sk_mod_q, const uint32_t* c0, const uint32_t* c1, uint8_t* msg, uint32_t count);
void launch_shamir_combine(hipStream_t st, const Fr* lambda, const uint8_t* ys_be, uint32_t t, uint32_t n, uint8_t* secret_be,
                           uint32_t* sk_mod_q);
void launch_audit_msg(hipStream_t st, const uint8_t* xy_be, uint8_t* msg, uint32_t count);
void launch_audit_assemble(hipStream_t st, const uint8_t* wa_be, const uint8_t* ct_be, const uint8_t* packed_be, const uint8_t* sk_be,
                           const int8_t* r, const int8_t* e1, const int8_t* e2, const int32_t* k0, const int32_t* k1, uint8_t* rows,
                           uint32_t count);

// ---- witness ----
void launch_load_inputs(hipStream_t st, const uint8_t* d_inputs_be, const uint8_t* d_rs_be, Fr* W, uint32_t n_inputs,
                        uint32_t n_wires, uint32_t P);
// runs the solver program from word `pc` until OP_COMMIT / OP_END; scratch: [SOLVE_SCRATCH_ROWS or more][P] Fr
static constexpr uint32_t SOLVE_SCRATCH_MIN_ROWS = 324;
void launch_solve(hipStream_t st, DevCircuit dc, Fr* W, Fr* scratch, uint32_t pc_begin, uint32_t pc_end, uint32_t P);
// cooperative solver for small batches (one wave per proof; kernels_solve.hip): items = (kind, a, b) triples
enum : uint32_t { COOP_SEQ = 0, COOP_PAR = 1, COOP_LEVELS = 2, COOP_POSEIDON = 3, COOP_POSEIDON2 = 4, COOP_GRUMPKIN = 5, COOP_LEVEL_STREAM = 6 };
struct DevCoop {
  const uint32_t* items;      // 3 words per item
  const uint32_t* par;        // COOP_PAR: (pc_begin, pc_end) pairs of independent instructions
  const uint32_t* lvl_ptr;    // COOP_LEVELS: rows lvl_rows[lvl_ptr[l] .. lvl_ptr[l+1]) form dependency level l
  const uint32_t* lvl_rows;
  // COOP_LEVEL_STREAM (a = first chunk, b = chunks): the same levels flattened into self-contained records, in chunks of
  // COOP_CHUNK words that the wave stages through LDS one ahead.  Level: [rows n | lanes per row << 24 | some A form << 29 | some C rest << 30][words in the level][offset of row 0..n-1]
  // then per row [output wire][terms of A | bit 31: A = B][terms of B][terms of C without the output][(wire, coefficient word) ...];
  // 0xffffffff instead of n: the rest of the chunk is padding.  No level crosses a chunk boundary.
  const uint32_t* lvl_stream;
};
static constexpr uint32_t COOP_CHUNK = 1024;
// independent item ranges of one stretch: track t (blockIdx.y) runs items [begin[t], end[t]); only track 0 may use `scratch`
static constexpr uint32_t COOP_TRACKS = 4;
struct CoopTracks { uint32_t n; uint32_t begin[COOP_TRACKS], end[COOP_TRACKS]; };
void launch_solve_coop(hipStream_t st, DevCircuit dc, DevCoop co, Fr* W, Fr* scratch, CoopTracks tracks, uint32_t P);
// wide forms of the two data-parallel solver instructions (one lane per (element chunk, proof) instead of one per proof)
void launch_batch_div(hipStream_t st, DevCircuit dc, Fr* W, Fr* scratch, uint32_t k0, uint32_t n, uint32_t P);
void launch_count8(hipStream_t st, DevCircuit dc, Fr* W, uint32_t* counters, uint32_t h0, uint32_t n, uint32_t out0, uint32_t P);
// a,b,c evaluation + satisfaction check (status[p] |= 1 when some row fails)
// ---- batched verification (kernels_verify.hip); the structures live in pairing_fast.hpp ----
struct VerifyKeyDev;
void launch_verify(hipStream_t st, const VerifyKeyDev* vk, const uint8_t* proofs, const uint8_t* pws, uint32_t pw_len, uint32_t count,
                   int32_t* ok);
struct PairingCheckDev;
void launch_pairing_check(hipStream_t st, const PairingCheckDev* a, int32_t* ok);
// small: [sm_nslots][P] int16 scratch of the small rows (may be nullptr when the circuit has none)
void launch_spmv_check(hipStream_t st, DevCircuit dc, const Fr* W, Fr* abc, uint32_t n, uint32_t P, uint32_t* status, int16_t* small = nullptr);

// ---- NTT / QAP ----
// in-place radix-2 passes over data [n][P]; dif: natural->bitreversed with table tw (w^-k or w^k), else DIT
// post (optional): n row factors multiplied into the output rows by the last pass (fused coset shift)
void launch_ntt(hipStream_t st, Fr* data, uint32_t logn, uint32_t P, const Fr* tw, bool dif, uint32_t nbatch, size_t batch_stride,
                const Fr* post = nullptr);
void launch_scale_rows(hipStream_t st, Fr* data, const Fr* table, uint32_t n, uint32_t P, uint32_t nbatch, size_t batch_stride);
void launch_qap_pointwise(hipStream_t st, Fr* abc, uint32_t n, uint32_t P, Fr zinv);
void launch_qap_product(hipStream_t st, Fr* abc, uint32_t n, uint32_t P);   // a <- a * b

// ---- MSM with precomputed window tables (kernels_msm.hip) ----
// A base carries Wt table rows of 2^(c-1) affine multiples; row m = multiples of 2^(c*R*m) * Base, R = ceil(W / Wt) window passes
// (W = ceil(254 / c) windows per scalar).  Wt = W: one row per window, no passes (the small-table latency layout).  Wt = 1: one row
// per base and W passes whose sums are combined by Horner (the throughput layout: widest window for the bytes).
uint32_t msm_windows(uint32_t c);
struct MsmPlan {
  uint32_t W;        // windows per scalar
  uint32_t Wt, R;    // table rows per base, passes
  uint32_t Q, Wq;    // small batches: Q lanes share the rows of a base, Wq rows each
  uint32_t Sg;       // slices of the item range per pass
  uint32_t Pp;       // proofs per slice row: P rounded up to a wave (P >= 64), else P
  size_t partial_elems(uint32_t P) const { return (size_t)R * Sg * P; }
};
MsmPlan msm_plan(uint32_t N, uint32_t P, uint32_t c, uint32_t Wt, uint32_t occ = 2);   // occ: resident waves per SIMD of the kernel (G1 2, G2 1)
// builds rows [row0, row0 + nrows) (row = base * Wt + m; row0 a multiple of 64) of the table of N bases;
// tmp / tmp_pre: nrows * 2^(c-1) elements each.  Layout: see kernels_msm.hip.
template <class F>
void launch_build_table(hipStream_t st, const Affine<F>* bases, uint32_t N, uint32_t c, uint32_t Wt, uint32_t row0, uint32_t nrows,
                        Affine<F>* table, XYZZ<F>* tmp, F* tmp_pre);
size_t msm_table_elems(uint32_t N, uint32_t c, uint32_t Wt);
// signed c-bit digits of scalars[rows[i]][p] as int16 planes dig[j][i][p] (Pp per row); msm_digit_elems = W * N * Pp
size_t msm_digit_elems(uint32_t N, uint32_t P, uint32_t c);
void launch_msm_digits(hipStream_t st, const uint32_t* rows, const Fr* scalars, int16_t* dig, uint32_t N, uint32_t P, uint32_t c);
// lane g -> (pass, slice, proof); partial[R * Sg][P]
template <class F>
void launch_msm_accumulate(hipStream_t st, const Affine<F>* table, const int16_t* dig, XYZZ<F>* partial, uint32_t N, uint32_t P, uint32_t c,
                           const MsmPlan& pl, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// out[p] = sum over slices and passes (empty: out[p] = infinity); folds in place: `partial` is scratch afterwards
template <class F>
void launch_msm_reduce(hipStream_t st, XYZZ<F>* partial, XYZZ<F>* out, uint32_t P, const MsmPlan& pl, uint32_t c, bool empty);
// several sets folded by the same launches (one launch per level for all of them, one Horner launch)
static constexpr uint32_t MSM_FOLD_SETS = 8;
static constexpr uint32_t MSM_FOLD_RADIX = 8;   // slices summed per lane and fold level
template <class F>
struct MsmFoldSets {
  XYZZ<F>* partial[MSM_FOLD_SETS];
  XYZZ<F>* out[MSM_FOLD_SETS];
  uint32_t Sg[MSM_FOLD_SETS], R[MSM_FOLD_SETS], c[MSM_FOLD_SETS];   // per set: slices per pass (0 = empty set), passes, window bits
  uint32_t cur[MSM_FOLD_SETS], half[MSM_FOLD_SETS];                 // per level: slices before / after it (launch_msm_reduce_multi)
};
template <class F>
void launch_msm_reduce_multi(hipStream_t st, MsmFoldSets<F> fs, uint32_t nsets, uint32_t P);

// the H bases of a proving key in the evaluation basis on the coset (kernels_msm.hip): out[bitrev(i)] = Z'_i
void launch_g1_eval_basis(hipStream_t st, const G1Affine* pts, uint32_t n_pts, uint32_t logn, const Fr* scale, const Fr* tw_inv, G1XYZZ* work,
                          G1Affine* out);

// out[s] = sum over t in [seg[s], seg[s+1]) of coeff[t] * base[row[t]]  (work: nterms points)
void launch_g1_column_sums(hipStream_t st, const G1Affine* base, const uint32_t* row, const Fr* coeff, uint32_t nterms, const uint32_t* seg,
                           uint32_t nseg, G1XYZZ* work, G1Affine* out);

// ---- general-base Pippenger (kernels_pippenger.hip) ----
size_t pippenger_workspace_bytes(uint32_t n);
size_t pippenger_workspace_bytes_g2(uint32_t n);
void launch_pippenger_g2(hipStream_t st, const G2Affine* bases, const Fr* scalars, uint32_t n, void* workspace, G2XYZZ** out_windows,
                         hipEvent_t ev0, hipEvent_t ev1);
uint32_t pippenger_windows();
void launch_pippenger_g1(hipStream_t st, const G1Affine* bases, const Fr* scalars, uint32_t n, void* workspace, G1XYZZ** out_windows,
                         hipEvent_t ev0, hipEvent_t ev1);

// ---- commitment challenge, proof assembly ----
void launch_challenge(hipStream_t st, const G1XYZZ* commit, Fr* W, uint32_t challenge_wire, uint32_t P, G1Affine* commit_affine,
                      uint32_t* status);
struct AssembleArgs {
  const G1XYZZ* mA; const G1XYZZ* mB1; const G2XYZZ* mB2; const G1XYZZ* mK; const G1XYZZ* mZ; const G1XYZZ* mPok;
  const G1Affine* commit_affine;
  const Fr* W; uint32_t row_r, row_s; uint32_t n_public;
  // small batches: s*Ar and r*Bs1 arrive as two more fixed-base sums over the scaled witness (nullptr: the lanes multiply)
  const G1XYZZ* sAr; const G1XYZZ* rBs1;
  uint8_t* proofs;   // [P][388]
  uint8_t* pws;      // [P][12+32*(n_public-1)]
  uint32_t P;
};
void launch_assemble(hipStream_t st, AssembleArgs a);
void launch_spin(hipStream_t st, uint64_t ticks_100mhz, uint32_t* sink);   // one lane busy-waits (bounded); stream-concurrency probe
void launch_touch(hipStream_t st, uint32_t* sink);
// Ws[row][p] = s_p * W[row][p], Wr[row][p] = r_p * W[row][p] for rows < n_rows (r, s = rows row_r, row_s of W)
void launch_scale_witness(hipStream_t st, const Fr* W, Fr* Ws, Fr* Wr, uint32_t n_rows, uint32_t row_r, uint32_t row_s, uint32_t P);

// ---- setup helpers ----
// out[i] = scalars[i] * G for a generator table built with launch_build_table (N=1)
template <class F>
void launch_fixed_base_mul(hipStream_t st, const Affine<F>* gen_table, uint32_t c, const Fr* scalars, uint32_t n, Affine<F>* out,
                           XYZZ<F>* tmp);

}  // namespace spp

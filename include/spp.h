/* libspp -- C ABI of the MI355X-native Groth16 prover for the shielded-pool circuits.
 *
 * Drop-in boundary for the proving path of Ham3798/shielded-pool-pinocchio-solana. Each entry point
 * names the reference interface it replaces (paths relative to the reference tree):
 *
 *   spp_prove_withdraw        client/proof.helper.ts:28-72  generateProof(): the two child processes
 *                             `nargo execute` (:55) + `sunspot prove` (:64) and the .proof/.pw reads (:68-69)
 *   spp_prove_batch           client/payroll-demo.ts:326-352 (Promise.all over generateProof) and
 *                             audit_circuit/prove_audit.sh:74-99 / scripts/generate_audit.py:668-685
 *   spp_setup                 `sunspot setup <ccs>`   noir_circuit/prove_linux.sh:72-79, generate_audit.py:670-677
 *   spp_circuit_build         `sunspot compile <acir>` noir_circuit/prove_linux.sh:66-70, generate_audit.py:659-665
 *   spp_rlwe_witness_batch    scripts/generate_audit.py:507-584 (encrypt + quotient witnesses + packing),
 *                             demo-frontend/app/lib/rlwe.ts:157-247
 *   spp_poseidon_*            client/merkle.ts:22-38,119-140,165-221 (circomlibjs Poseidon, Merkle tree)
 *   spp_grumpkin_keygen_batch client/merkle.ts:98-113 generateIdentityKeypair
 *   spp_audit_inputs_batch    scripts/generate_audit.py:468-641 (everything before `nargo execute`)
 *   spp_verify                `sunspot verify` noir_circuit/prove_linux.sh:86-87, audit_circuit/prove_audit.sh:98-99
 *   spp_verify_batch          the same for many proofs on the GPU
 *   spp_shamir_reconstruct / spp_rlwe_decrypt_batch   scripts/rlwe_decrypt.py:61-132, demo-frontend/app/lib/shamir.ts:97-169
 *   spp_msm_g1(_pippenger) / spp_ntt_fr   micro-benchmark entry points (BASELINE.json configs[4]); no reference equivalent
 *
 * Conventions: field elements cross the boundary as 32-byte big-endian canonical integers (the encoding
 * of the reference's .pw files, shielded_pool_program/src/instructions/withdraw.rs:74-90); points as
 * gnark raw uncompressed bytes (64 B G1, 128 B G2). The caller owns every buffer it passes; the library
 * owns device memory inside spp_circuit. All functions return 0 on success or a negative SPP_ERR_* code,
 * with a thread-local message available from spp_last_error(). Calls on one spp_ctx are serialised.
 * There is no CPU fallback: without a HIP device spp_init fails with SPP_ERR_NO_DEVICE.
 */
#ifndef SPP_H
#define SPP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SPP_OK 0
#define SPP_ERR_BAD_INPUT (-1)
#define SPP_ERR_NO_DEVICE (-2)
#define SPP_ERR_IO (-3)
#define SPP_ERR_UNSAT (-4)       /* the inputs do not satisfy the circuit (proof refused before the MSMs) */
#define SPP_ERR_HIP (-5)
#define SPP_ERR_NOT_IMPLEMENTED (-6)
#define SPP_ERR_FORMAT (-7)

#define SPP_CIRCUIT_WITHDRAW 1   /* noir_circuit/src/main.nr */
#define SPP_CIRCUIT_AUDIT 2      /* audit_circuit (scripts/generate_audit.py:246-465) */
/* spp_circuit_build only: the withdraw statement padded with ballast multiplications to the dimensions of the
 * reference's gnark R1CS (noir_circuit/target/shielded_pool_verifier.ccs: 12 452 constraints, domain 2^14), for
 * like-for-like throughput figures.  The container it writes is an ordinary SPP_CIRCUIT_WITHDRAW circuit. */
#define SPP_CIRCUIT_WITHDRAW_REFSHAPE 3
/* spp_circuit_build only: the withdraw statement over a depth-20 tree (20 siblings, 25 secret inputs) -- the synthetic
 * variant of SURVEY 8d Config 2 / BASELINE.json configs[1]; the reference's circuit is depth 16 (main.nr:11). */
#define SPP_CIRCUIT_WITHDRAW_DEPTH20 4

#define SPP_PROOF_LEN 388        /* withdraw.rs:13, submit_audit.rs:18 */
#define SPP_WITHDRAW_PW_LEN 172  /* withdraw.rs:14-16 */
#define SPP_AUDIT_PW_LEN 76      /* submit_audit.rs:19-21 */
#define SPP_TREE_DEPTH 16        /* noir_circuit/src/main.nr:5 */

typedef struct spp_ctx spp_ctx;
typedef struct spp_circuit spp_circuit;

/* Inputs of the withdraw circuit, field order of ShieldedPoolInputs (client/proof.helper.ts:6-21). */
typedef struct {
  uint8_t root[32], nullifier[32], recipient[32];
  uint64_t amount;
  uint8_t wa_commitment[32];
  uint8_t secret_key[32], owner_x[32], owner_y[32], randomness[32];
  uint64_t index;
  uint8_t siblings[SPP_TREE_DEPTH][32];
} spp_withdraw_inputs;

const char* spp_last_error(void);
const char* spp_version(void);

/* ---- host-only: circuit construction (no GPU needed) ---- */
/* Writes the R1CS + solver program container ("SPPC"). aux: for SPP_CIRCUIT_AUDIT the RLWE public key as
 * 2048 uint32 (a[1024] then b[1024], demo-frontend/public/rlwe/rlwe_pk.json); NULL for withdraw.
 * Prints nothing; *n_constraints (optional) receives the constraint count (`nbConstraints=` of sunspot compile). */
int spp_circuit_build(int circuit_id, const uint32_t* aux, const char* out_path, uint32_t* n_constraints);

/* `sunspot compile <acir>` (noir_circuit/prove_linux.sh:66-70) for a program compiled by nargo: lowers the ACIR opcodes --
 * AssertZero, RANGE, the fixed-base Grumpkin MultiScalarMul and the Brillig hints of the reference's withdraw circuit
 * (noir_circuit/target/shielded_pool_verifier.json) -- to an R1CS + solver program in the same SPPC container, so the
 * reference's OWN compiled circuit is what gets set up and proved.  blob: the decoded opcode list written by
 * spp/acir.py:to_blob (bincode decoding stays on the host side).  circuit_id: id stored in the container (0 = generic;
 * SPP_CIRCUIT_WITHDRAW when the program has the withdraw circuit's ABI, so that spp_prove_withdraw accepts it). */
#define SPP_CIRCUIT_ACIR 5
int spp_circuit_build_acir(const uint8_t* blob, size_t blob_len, int circuit_id, const char* out_path, uint32_t* n_constraints);

/* ---- device context ---- */
int spp_init(int device, spp_ctx** out);
void spp_free_ctx(spp_ctx* ctx);

/* Deterministic trusted setup on the GPU from a 32-byte seed: writes pk ("SPPK") and vk (gnark raw layout). */
int spp_setup(spp_ctx* ctx, const char* circuit_path, const uint8_t seed[32], const char* pk_path, const char* vk_path);

/* Loads R1CS + proving key, builds the window tables in HBM. window_bits in [4,16] = the same window for every MSM set;
 * 0 = per-set windows chosen greedily within env SPP_TABLE_BUDGET_GB (default 240) and 85 % of the free HBM, single-row tables walked once per window (16 bits = 16 additions per scalar); env SPP_SERIAL=1 (profiling aid) puts both batch workspaces and the G2 MSM on one stream. */
int spp_load_circuit(spp_ctx* ctx, const char* circuit_path, const char* pk_path, int window_bits, spp_circuit** out);
/* Several circuits on ONE GPU at the same time (the reference's relayer submits an audit proof AND a withdraw proof per withdrawal,
 * demo-frontend/app/api/relay/withdraw/route.ts:238-276): plan the windows of all their MSM sets under one HBM budget, then load
 * each circuit with its share.  sizes / bits: n_circuits x 7 in the order of spp_circuit_msm_sizes; spp_pk_msm_sizes reads the
 * sizes from a proving-key file; both are host-only.  spp_load_circuit_with_windows = spp_load_circuit(window_bits = 0) with
 * the planned bits instead of a budget of its own (single-row tables; the two commitment sets keep one row per window). */
int spp_pk_msm_sizes(const char* pk_path, uint32_t sizes[7]);
int spp_plan_windows(uint32_t n_circuits, const uint32_t* sizes, double budget_bytes, uint32_t* bits);
int spp_load_circuit_with_windows(spp_ctx* ctx, const char* circuit_path, const char* pk_path, const uint32_t bits[7], spp_circuit** out);
void spp_free_circuit(spp_circuit* c);
/* info[0..7] = id, n_public (without the constant), n_secret, n_wires, n_constraints, domain_log, n_inputs, window_bits */
int spp_circuit_info(const spp_circuit* c, uint32_t info[8]);
/* number of bases per MSM of one proof: G1 sets A, B1, K, Z, commitment basis, commitment basis^sigma; then the G2 set B2 */
int spp_circuit_msm_sizes(const spp_circuit* c, uint32_t sizes[7]);
/* window bits of the table of each of those sets (same order) */
int spp_circuit_msm_windows(const spp_circuit* c, uint32_t bits[7]);
/* table rows per base of each of those sets (same order): 1 = one row of 2^(bits-1) multiples, walked once per window
 * (the throughput layout chosen with window_bits = 0); ceil(254 / bits) = one row per window (explicit window_bits) */
int spp_circuit_msm_table_rows(const spp_circuit* c, uint32_t rows[7]);
/* out[0] = rows of A / B / C that the matrix evaluation sums in integer arithmetic (every term a small coefficient times a wire the
 * lookup argument bounds to a byte range: the audit circuit's 1 088 quotient equations), out[1] = such wires (+1: the constant) */
int spp_circuit_small_rows(const spp_circuit* c, uint32_t out[2]);
/* exact bytes of HBM held by the window tables */
uint64_t spp_circuit_table_bytes(const spp_circuit* c);

/* ---- proving ---- */
/* Generic batch: inputs = count * n_inputs * 32 B (public then secret, big-endian), rs = count * 64 B blinding
 * (r || s, reduced mod r; NULL = OS randomness). Outputs: proofs count*388, pws count*(12+32*n_public),
 * status[count] (0 ok, SPP_ERR_UNSAT). Returns 0 if every proof was produced, else the first error. */
int spp_prove_batch(spp_circuit* c, size_t count, const uint8_t* inputs, const uint8_t* rs, uint8_t* proofs, uint8_t* pws,
                    int32_t* status);
/* Same with every buffer already resident in HBM (device pointers); asynchronous until spp_sync().
 * Consecutive calls rotate over several HIP streams and workspaces -- two for large batches, four for up to 768 proofs, six for
 * up to 256 -- so the latency-bound phases of a batch (witness solver, Horner combines) overlap the MSMs of the others: output
 * buffers must not be shared by calls that may be in flight together (up to six consecutive calls).
 * d_status: uint32 per proof, nonzero = unsatisfied. */
int spp_prove_batch_device(spp_circuit* c, size_t count, const void* d_inputs, const void* d_rs, void* d_proofs, void* d_pws,
                           void* d_status);
int spp_sync(spp_circuit* c);
/* The value of the circuit's commitment challenge for `count` (partial) input rows: loads the rows, commits to the committed
 * wires (BSB22 / Pedersen, the commitment basis of the proving key) and hashes the commitment to the field exactly as the prover
 * does between its two solver phases; out = count x 32 B big-endian.  For systems whose witness is completed OUTSIDE the library --
 * the reference's own gnark R1CS (noir_circuit/target/shielded_pool_verifier.ccs decoded by spp/ccs.py, every wire an input): the
 * wires after the commitment (the lookup argument's) depend on this value, gnark's solver gets it from the
 * Bsb22CommitmentComputePlaceholder hint (`sunspot prove`, client/proof.helper.ts:58-64).  Wires not known yet are passed as 0.
 * (Circuits built by spp_circuit_build* derive one committed wire, the hiding mask, from the blinding factors of the proof; this
 * call has none and takes them as zero, so for those circuits its result is not the challenge of any real proof.) */
int spp_commitment_challenge(spp_circuit* c, size_t count, const uint8_t* inputs, uint8_t* challenges);
/* per-stage device time of the last spp_prove_batch_device call, milliseconds:
 * [0] witness solve (+commitment), [1] matrix eval, [2] NTT/QAP, [3] MSM G1, [4] wait for the G2 MSM (it runs on a side
 * stream from the end of [0]), [5] assembly, [6] total;
 * [7] = average duration of one k_msm_fixed<G1> launch (the dominant kernel), [8] = number of such launches */
int spp_last_timings(spp_circuit* c, float ms[9]);
/* which = 0: the last enqueued batch, 1: the one before it (consecutive batches alternate between two streams,
 * so reading batch k-1 while batch k runs does not drain the pipeline) */
int spp_timings(spp_circuit* c, int which, float ms[9]);

/* durations (ms) of the MSM kernel launches of one batch (which: as spp_timings), from the dispatches' own timestamps, in launch
 * order: commitment, A, B1, K, Z, proof of knowledge (the six k_msm_fixed<G1> launches), then the G2 launch */
int spp_msm_kernel_ms(spp_circuit* c, int which, float ms[7]);
/* on = 1: run everything of this circuit on one stream (profiling / roofline probe: a kernel's duration is then its own, not
 * stretched by the other batch or by the G2 side stream sharing the chip); on = 0: back to the pipelined default.
 * Synchronises the device. */
int spp_set_serial(spp_circuit* c, int on);

int spp_prove_withdraw(spp_circuit* c, const spp_withdraw_inputs* in, const uint8_t rs_seed[64], uint8_t proof[SPP_PROOF_LEN],
                       uint8_t pw[SPP_WITHDRAW_PW_LEN]);

/* `sunspot verify <vk> <proof> <pw>` (noir_circuit/prove_linux.sh:86-87, audit_circuit/prove_audit.sh:98-99,
 * scripts/generate_audit.py:687-691): host-side Groth16 + BSB22 check. *ok = 1 accepted, 0 rejected; the return value
 * is an error only for malformed inputs. Needs no GPU. */
int spp_verify(const uint8_t* vk, size_t vk_len, const uint8_t* proof, size_t proof_len, const uint8_t* pw, size_t pw_len, int* ok);

/* debug / parity: full witness of proof 0 of the last batch, n_wires * 32 B big-endian */
int spp_debug_witness(spp_circuit* c, uint8_t* out, size_t n_wires);

/* ---- witness-input generation (what the reference computes on the client before proving) ---- */
/* RLWE encryption + quotient witnesses for `count` instances (scripts/generate_audit.py:507-554, rlwe.ts:157-247):
 * pk_a, pk_b: 1024 coefficients in [0,q); r, e2: count*1024 int8; e1: count*64 int8; msg: count*64 bytes.
 * Out: c0 count*64, c1 count*1024 (in [0,q)); k0 count*64, k1 count*1024 (signed quotients);
 * packed_be (optional): count * 157 fields of 32 B big-endian = pack_values(c0) ++ pack_values(c1) (:154-163). */
int spp_rlwe_witness_batch(spp_ctx* ctx, const uint32_t* pk_a, const uint32_t* pk_b, size_t count, const int8_t* r, const int8_t* e1,
                           const int8_t* e2, const uint8_t* msg, uint32_t* c0, uint32_t* c1, int32_t* k0, int32_t* k1,
                           uint8_t* packed_be);
/* same, every pointer a device pointer; asynchronous on the context stream until spp_ctx_sync() */
int spp_rlwe_witness_batch_device(spp_ctx* ctx, const void* d_pk_a, const void* d_pk_b, size_t count, const void* d_r, const void* d_e1,
                                  const void* d_e2, const void* d_msg, void* d_c0, void* d_c1, void* d_k0, void* d_k1,
                                  void* d_packed_be);
int spp_ctx_sync(spp_ctx* ctx);
/* Poseidon hash_2 / hash_4 (client/merkle.ts:22-38): in = count * arity * 32 B, out = count * 32 B */
int spp_poseidon_hash_batch(spp_ctx* ctx, size_t count, int arity, const uint8_t* in, uint8_t* out);
/* compute_merkle_root (noir_circuit/src/main.nr:11-29) for count paths: siblings = count * depth * 32 B */
int spp_merkle_root_batch(spp_ctx* ctx, size_t count, uint32_t depth, const uint8_t* leaves, const uint64_t* indices,
                          const uint8_t* siblings, uint8_t* roots);
/* ShieldedPoolMerkleTree.getRoot + getProof (client/merkle.ts:165-221): tree of n_leaves inserted leaves, missing
 * nodes = default hashes (:150-156); siblings_out = n_queries * depth * 32 B */
int spp_merkle_build(spp_ctx* ctx, size_t n_leaves, uint32_t depth, const uint8_t* leaves, size_t n_queries,
                     const uint64_t* query_indices, uint8_t* siblings_out, uint8_t* root_out);
/* The same tree kept RESIDENT in HBM and updated incrementally: insert() appends leaves and recomputes only the
 * touched paths -- O(count + depth) Poseidon hashes per call instead of the O(2^depth) recomputation of every getRoot() /
 * getProof() in client/merkle.ts:165-221; getRoot is one read, getProof `depth` reads per query.  Leaves must be canonical
 * field elements (32 B big-endian).  Calls on one tree are serialised with the other calls on its context. */
typedef struct spp_merkle_tree spp_merkle_tree;
int spp_merkle_tree_new(spp_ctx* ctx, uint32_t depth, spp_merkle_tree** out);
void spp_merkle_tree_free(spp_merkle_tree* t);
uint64_t spp_merkle_tree_size(const spp_merkle_tree* t);
/* ShieldedPoolMerkleTree.insert (client/merkle.ts:158-163) for `count` leaves; *first_index (optional) = index of the first */
int spp_merkle_tree_insert(spp_merkle_tree* t, size_t count, const uint8_t* leaves, uint64_t* first_index);
int spp_merkle_tree_root(spp_merkle_tree* t, uint8_t root[32]);
/* getProof for n leaf indices (any index below 2^depth, inserted or not): siblings_out = n * depth * 32 B */
int spp_merkle_tree_proofs(spp_merkle_tree* t, size_t n, const uint64_t* indices, uint8_t* siblings_out);

/* generateIdentityKeypair's sk * G on Grumpkin (client/merkle.ts:98-113; scalar = the canonical field element,
 * as noir_circuit/src/main.nr:54-59): sk count * 32 B -> (x, y) count * 64 B */
int spp_grumpkin_keygen_batch(spp_ctx* ctx, size_t count, const uint8_t* sk, uint8_t* xy);
/* ct_commitment sponge (ct_helper/src/main.nr:15-34): in = count * n * 32 B, out = count * 32 B */
int spp_poseidon2_sponge_batch(spp_ctx* ctx, size_t count, uint32_t n, const uint8_t* in, uint8_t* out);

/* Everything scripts/generate_audit.py:468-641 computes before `nargo execute`, for `count` instances, on the GPU:
 * keygen, wa_commitment, message slots, RLWE encryption + quotients, packing, ct_commitment -> rows of 3360 fields
 * (32 B big-endian, main()'s parameter order :405-417) ready for spp_prove_batch on the audit circuit. */
int spp_audit_inputs_batch(spp_ctx* ctx, const uint32_t* pk_a, const uint32_t* pk_b, size_t count, const uint8_t* sk, const int8_t* r,
                           const int8_t* e1, const int8_t* e2, uint8_t* rows);
int spp_audit_inputs_batch_device(spp_ctx* ctx, const void* d_pk_a, const void* d_pk_b, size_t count, const void* d_sk, const void* d_r,
                                  const void* d_e1, const void* d_e2, void* d_rows);

/* End to end on the device: audit proofs from the provers' raw secrets.  Replaces the whole of scripts/generate_audit.py:468-691
 * for `count` instances -- keygen, wa_commitment, RLWE encryption with the public key (pk_a, pk_b: 1024 x u32 each), quotient
 * witnesses, packing, ct_commitment (:468-641), then `nargo execute` + `sunspot prove` (:668-685, audit_circuit/prove_audit.sh:
 * 74-95) -- without a host round trip: all pointers are device memory (sk count*32 B big-endian, r / e2 count*1024 int8,
 * e1 count*64 int8, rs count*64 B), outputs as spp_prove_batch_device.  Asynchronous, pipelined like spp_prove_batch_device. */
int spp_prove_audit_from_secrets_device(spp_circuit* c, size_t count, const void* d_pk_a, const void* d_pk_b, const void* d_sk, const void* d_r,
                                        const void* d_e1, const void* d_e2, const void* d_rs, void* d_proofs, void* d_pws, void* d_status);

/* Batched verification on the GPU (`sunspot verify` for many proofs against one key, the checks of the
 * deployed verifier withdraw.rs:63-90 / submit_audit.rs:41-54): proofs = count * 388 B, pws = count * pw_len B (host
 * buffers), ok[i] = 1 iff proof i verifies.  Same decisions as spp_verify; one lane per proof. kernel_ms (optional):
 * duration of the verification kernel. */
int spp_verify_batch(spp_ctx* ctx, const uint8_t* vk, size_t vk_len, size_t count, const uint8_t* proofs, const uint8_t* pws, size_t pw_len,
                     int32_t* ok, float* kernel_ms);

/* prod_k e(P_k, Q_k) == 1 for 1..4 caller-supplied pairs (G1 64 B, G2 128 B, gnark raw uncompressed), computed on the GPU
 * with the device pairing code of spp_verify_batch (curve + subgroup checks included; *ok = 0 when a point is invalid).
 * No call site in the reference: it exists so that the only gnark-made curve data the reference holds -- its verifying keys
 * noir_circuit/target/shielded_pool_verifier.vk and audit_circuit/target/rlwe_audit.vk -- can be put through the device
 * pairing path (e(beta1, G2) == e(G1, beta2) etc.; tests/test_vk_pins.py).  The _host variant runs the single-proof host
 * pairing of spp_verify instead and needs no GPU. */
int spp_pairing_check(spp_ctx* ctx, uint32_t n_pairs, const uint8_t* g1s, const uint8_t* g2s, int* ok);
int spp_pairing_check_host(uint32_t n_pairs, const uint8_t* g1s, const uint8_t* g2s, int* ok);

/* ---- auditor side (scripts/rlwe_decrypt.py:61-132, demo-frontend/app/lib/shamir.ts:97-169) ---- */
/* Shamir reconstruction at 0 over BN254 Fr for n coefficients from t shares: xs[t] share indices, ys = t * n * 32 B
 * (share-major, big-endian). secret_be (optional): n * 32 B; sk_mod_q (optional): the centred value reduced mod q
 * (reconstructSk, shamir.ts:97-120). */
int spp_shamir_reconstruct(spp_ctx* ctx, uint32_t t, const uint32_t* xs, const uint8_t* ys, size_t n, uint8_t* secret_be,
                           uint32_t* sk_mod_q);
/* rlweDecrypt (shamir.ts:134-169 / rlwe_decrypt.py:106-132) for `count` ciphertexts: c0 count*64, c1 count*1024 in
 * [0,q); msg = count * 64 recovered byte slots (owner_x = slots 0..31 little-endian, owner_y = slots 32..63). */
int spp_rlwe_decrypt_batch(spp_ctx* ctx, const uint32_t* sk_mod_q, size_t count, const uint32_t* c0, const uint32_t* c1, uint8_t* msg);

/* ---- micro-benchmark / unit entry points ---- */
/* data: n = 2^logn elements, 32 B big-endian each, natural order in and out */
int spp_ntt_fr(spp_ctx* ctx, uint8_t* data, uint32_t logn, int inverse);
/* sum_i scalars[i] * bases[i]; bases 64 B, scalars 32 B (big-endian); out 64 B. Table-based path. */
int spp_msm_g1(spp_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, int window_bits, uint8_t out[64]);
/* the same over G2: bases 128 B (gnark raw X.A1 | X.A0 | Y.A1 | Y.A0), out 128 B -- the table walk that produces a proof's Bs */
int spp_msm_g2(spp_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, int window_bits, uint8_t out[128]);

/* General-base Pippenger (16-bit signed windows, bucket sort + accumulate + reduce) for large n; same conventions. */
int spp_msm_g1_pippenger(spp_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[64]);
/* the same over G2 (bases and out 128 B, gnark raw X.A1 | X.A0 | Y.A1 | Y.A0): the digit / count / scatter kernels are shared,
 * the bucket kernels run on the G2 accumulator of the table walk */
int spp_msm_g2_pippenger(spp_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[128]);
/* Synthetic, device-resident form of the same MSM (BASELINE.json configs[4]: n = 2^24): bases k_i*G and scalars from
 * an LCG of `seed`; scale_be (optional) multiplies every scalar (linearity checks). Mean ms over `iters` runs. */
int spp_msm_g1_pippenger_bench(spp_ctx* ctx, size_t n, uint64_t seed, const uint8_t scale_be[32], int iters, uint8_t out[64],
                               float* ms_total, float* ms_bucket_kernel);
/* Same with a "witness-like" scalar distribution: small_permille / 1000 of the scalars are byte-sized (SURVEY 8d Config 5:
 * 70 % of a gnark witness is small), the rest uniform. */
int spp_msm_g1_pippenger_bench_dist(spp_ctx* ctx, size_t n, uint64_t seed, uint32_t small_permille, const uint8_t scale_be[32], int iters,
                                    uint8_t out[64], float* ms_total, float* ms_bucket_kernel);
/* Points [first, first + count) of the same n_total-point synthetic MSM: what one of N ranks computes when the 2^24 points of
 * BASELINE.json configs[4] are cut over the GPUs of a node; the N partial sums are gathered and added by the caller
 * (spp/multi.py msm_g1_sharded).  out = this share's partial sum. */
int spp_msm_g1_pippenger_bench_shard(spp_ctx* ctx, size_t n_total, size_t first, size_t count, uint64_t seed, uint32_t small_permille,
                                     const uint8_t scale_be[32], int iters, uint8_t out[64], float* ms_total, float* ms_bucket_kernel);

#ifdef __cplusplus
}
#endif
#endif

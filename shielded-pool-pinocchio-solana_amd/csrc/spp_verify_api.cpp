// libspp C ABI, verification: `sunspot verify` on the host (spp_verify), the batched GPU verifier (spp_verify_batch) and the
// pairing-product checks that pin the pairing code to the reference's gnark-made verifying keys.
#include "spp_internal.hpp"

// -----------------------------------------------------------------------------------------------------
// verification (host): `sunspot verify <vk> <proof> <pw>`
// -----------------------------------------------------------------------------------------------------
extern "C" int spp_verify(const uint8_t* vk, size_t vk_len, const uint8_t* proof, size_t proof_len, const uint8_t* pw, size_t pw_len,
                          int* ok) {
  if (!vk || !proof || !pw || !ok) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  *ok = 0;
  if (proof_len != SPP_PROOF_LEN) return fail(SPP_ERR_FORMAT, "proof must be %d bytes", SPP_PROOF_LEN);
  auto be32 = [](const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; };
  if (vk_len < 576 + 4) return fail(SPP_ERR_FORMAT, "verifying key too short");
  G1Affine alpha1 = g1_from_raw(vk);
  G2Affine beta2 = g2_from_raw(vk + 128), gamma2 = g2_from_raw(vk + 256), delta2 = g2_from_raw(vk + 448);
  uint32_t nk = be32(vk + 576);
  size_t off = 580;
  if (nk < 2 || vk_len != off + (size_t)nk * 64 + 12 + 256) return fail(SPP_ERR_FORMAT, "verifying key has the wrong length");
  std::vector<G1Affine> K(nk);
  for (uint32_t i = 0; i < nk; i++) K[i] = g1_from_raw(vk + off + 64 * (size_t)i);
  off += (size_t)nk * 64;
  if (be32(vk + off) != 1 || be32(vk + off + 4) != 0 || be32(vk + off + 8) != 1) return fail(SPP_ERR_FORMAT, "unsupported commitment layout");
  G2Affine pedG = g2_from_raw(vk + off + 12), pedGS = g2_from_raw(vk + off + 12 + 128);
  if (pw_len < 12) return fail(SPP_ERR_FORMAT, "public witness too short");
  uint32_t npub = be32(pw);
  if (be32(pw + 4) != 0 || be32(pw + 8) != npub || pw_len != 12 + 32 * (size_t)npub || npub + 2 != nk)
    return fail(SPP_ERR_FORMAT, "public witness does not match the verifying key");
  if (be32(proof + 256) != 1) return fail(SPP_ERR_FORMAT, "proof must carry exactly one commitment");
  // canonical encodings only (gnark's readers refuse a coordinate >= q or a witness word >= r; reducing them would make
  // v and v + r two byte strings for the same nullifier)
  for (size_t o : {0, 32, 64, 96, 128, 160, 192, 224, 260, 292, 324, 356})
    if (!be_is_canonical<FqParams>(proof + o)) return SPP_OK;   // ok = 0
  for (uint32_t i = 0; i < npub; i++)
    if (!be_is_canonical<FrParams>(pw + 12 + 32 * (size_t)i)) return SPP_OK;
  G1Affine Ar = g1_from_raw(proof), Krs = g1_from_raw(proof + 192), Cm = g1_from_raw(proof + 260), Pok = g1_from_raw(proof + 324);
  G2Affine Bs = g2_from_raw(proof + 64);
  if (!g1_on_curve(Ar) || !g1_on_curve(Krs) || !g1_on_curve(Cm) || !g1_on_curve(Pok) || !g2_on_curve(Bs)) return SPP_OK;   // ok = 0
  if (!g2_in_subgroup(Bs)) return SPP_OK;   // the twist has a large cofactor: Bs must lie in the order-r subgroup
  // Pedersen proof of knowledge of the commitment, gnark-crypto's current convention (VerifyingKey{G, GSigmaNeg = -sigma G},
  // PoK = sum v_i * sigma Basis_i):  e(Cm, GSigmaNeg) * e(PoK, G) == 1
  if (!pairing_product_is_one({{Cm, pedGS}, {Pok, pedG}})) return SPP_OK;
  // challenge = hash_to_field(commitment, "bsb22-commitment")
  const char* dst = "bsb22-commitment";
  uint8_t u[48];
  expand_message_xmd(proof + 260, 64, (const uint8_t*)dst, strlen(dst), u, 48);
  uint32_t w12[12];
  for (int k = 0; k < 12; k++) w12[k] = be32(u + 4 * k);
  Fr challenge = fr_from_wide48(w12);
  G1XYZZ ksum = G1XYZZ::from_affine(K[0]);
  for (uint32_t i = 0; i <= npub; i++) {
    Fr v = i < npub ? Fr::from_bytes_be(pw + 12 + 32 * (size_t)i) : challenge;
    uint32_t lim[8];
    v.to_canonical(lim);
    ksum.add(scalar_mul(K[i + 1], lim));
  }
  ksum.madd(Cm);
  if (pairing_product_is_one({{Ar, Bs}, {alpha1.neg(), beta2}, {ksum.to_affine().neg(), gamma2}, {Krs.neg(), delta2}})) *ok = 1;
  return SPP_OK;
}

// -----------------------------------------------------------------------------------------------------
// batched verification on the GPU: same decisions as spp_verify above, one lane per proof
// -----------------------------------------------------------------------------------------------------
extern "C" int spp_verify_batch(spp_ctx* ctx, const uint8_t* vk, size_t vk_len, size_t count, const uint8_t* proofs, const uint8_t* pws,
                                size_t pw_len, int32_t* ok, float* kernel_ms) {
  if (!ctx || !vk || !ok || (count && (!proofs || !pws))) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (kernel_ms) *kernel_ms = 0;
  if (count == 0) return SPP_OK;
  if (count > (1u << 24)) return fail(SPP_ERR_BAD_INPUT, "count too large");
  auto be32 = [](const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; };
  if (vk_len < 576 + 4) return fail(SPP_ERR_FORMAT, "verifying key too short");
  const uint32_t nk = be32(vk + 576);
  size_t off = 580;
  if (nk < 2 || vk_len != off + (size_t)nk * 64 + 12 + 256) return fail(SPP_ERR_FORMAT, "verifying key has the wrong length");
  if (pw_len != 12 + 32 * (size_t)(nk - 2)) return fail(SPP_ERR_FORMAT, "public witness length does not match the verifying key");
  const G1Affine alpha1 = g1_from_raw(vk);
  const G2Affine beta2 = g2_from_raw(vk + 128), gamma2 = g2_from_raw(vk + 256), delta2 = g2_from_raw(vk + 448);
  std::vector<G1Affine> K(nk);
  for (uint32_t i = 0; i < nk; i++) K[i] = g1_from_raw(vk + off + 64 * (size_t)i);
  off += (size_t)nk * 64;
  if (be32(vk + off) != 1 || be32(vk + off + 4) != 0 || be32(vk + off + 8) != 1) return fail(SPP_ERR_FORMAT, "unsupported commitment layout");
  const G2Affine pedG = g2_from_raw(vk + off + 12), pedGS = g2_from_raw(vk + off + 12 + 128);
  for (const G2Affine* q : {&beta2, &gamma2, &delta2, &pedG, &pedGS})
    if (q->is_inf() || !g2_on_curve(*q)) return fail(SPP_ERR_FORMAT, "verifying key holds an invalid G2 point");
  if (!pairing_fast_consts_consistent()) return fail(SPP_ERR_HIP, "internal: Frobenius constants are not two-term");

  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // per-key preparation on the host: line tables of the four key-side G2 points, e(-alpha, beta), constants
  VerifyKeyDev h;
  h.pc = make_pairing_fast_consts();
  h.e_alpha_beta = f12_from(miller_loop(alpha1.neg(), beta2));
  h.twist_b = twist_b();
  h.nk = nk;
  DevBuf dtab[4], dK, dvk, dproofs, dpws, dok;
  const G2Affine* qs[4] = {&gamma2, &delta2, &pedG, &pedGS};
  std::vector<LineStep> tabs_host[4];   // stay alive until the stream has been synchronised
  for (int k = 0; k < 4; k++) {
    tabs_host[k] = build_line_table(*qs[k]);
    UP(dtab[k], tabs_host[k].data(), tabs_host[k].size() * sizeof(LineStep));
    h.tab[k] = dtab[k].as<LineStep>();
  }
  UP(dK, K.data(), K.size() * sizeof(G1Affine));
  h.K = dK.as<G1Affine>();
  UP(dvk, &h, sizeof h);
  UP(dproofs, proofs, count * (size_t)SPP_PROOF_LEN);
  UP(dpws, pws, count * pw_len);
  HIP_TRY(dok.alloc(count * sizeof(int32_t)));
  struct Ev {   // destroyed on every return path
    hipEvent_t e = nullptr;
    ~Ev() { if (e) hipEventDestroy(e); }
  } ev0, ev1;
  HIP_TRY(hipEventCreate(&ev0.e));
  HIP_TRY(hipEventCreate(&ev1.e));
  hipEvent_t e0 = ev0.e, e1 = ev1.e;
  hipEventRecord(e0, st);
  launch_verify(st, dvk.as<VerifyKeyDev>(), dproofs.as<uint8_t>(), dpws.as<uint8_t>(), (uint32_t)pw_len, (uint32_t)count, dok.as<int32_t>());
  hipEventRecord(e1, st);
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  if (kernel_ms) *kernel_ms = ms;
  HIP_TRY(hipMemcpy(ok, dok.p, count * sizeof(int32_t), hipMemcpyDeviceToHost));
  return SPP_OK;
}

// prod_k e(P_k, Q_k) == 1 on the GPU with the device pairing code of the batched verifier (k_pairing_check).
extern "C" int spp_pairing_check(spp_ctx* ctx, uint32_t n_pairs, const uint8_t* g1s, const uint8_t* g2s, int* ok) {
  if (!ctx || !g1s || !g2s || !ok) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  *ok = 0;
  if (n_pairs < 1 || n_pairs > 4) return fail(SPP_ERR_BAD_INPUT, "1 to 4 pairs");
  for (uint32_t k = 0; k < n_pairs; k++) {
    for (int o = 0; o < 64; o += 32)
      if (!be_is_canonical<FqParams>(g1s + 64 * k + o)) return fail(SPP_ERR_FORMAT, "G1 coordinate not below q");
    for (int o = 0; o < 128; o += 32)
      if (!be_is_canonical<FqParams>(g2s + 128 * k + o)) return fail(SPP_ERR_FORMAT, "G2 coordinate not below q");
  }
  if (!pairing_fast_consts_consistent()) return fail(SPP_ERR_HIP, "internal: Frobenius constants are not two-term");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  PairingCheckDev h;
  h.pc = make_pairing_fast_consts();
  h.twist_b = twist_b();
  h.n = n_pairs;
  DevBuf dtab[3], darg, dok;
  std::vector<LineStep> tabs_host[3];
  for (uint32_t k = 0; k < 4; k++) {
    h.P[k] = k < n_pairs ? g1_from_raw(g1s + 64 * k) : G1Affine::infinity();
    h.Q[k] = k < n_pairs ? g2_from_raw(g2s + 128 * k) : G2Affine::infinity();
  }
  for (uint32_t k = 1; k < 4; k++) {
    h.tab[k - 1] = nullptr;
    if (k >= n_pairs) continue;
    if (h.Q[k].is_inf() || !g2_on_curve(h.Q[k])) return SPP_OK;   // ok = 0 (a line table needs a point of the twist)
    tabs_host[k - 1] = build_line_table(h.Q[k]);
    UP(dtab[k - 1], tabs_host[k - 1].data(), tabs_host[k - 1].size() * sizeof(LineStep));
    h.tab[k - 1] = dtab[k - 1].as<LineStep>();
  }
  UP(darg, &h, sizeof h);
  HIP_TRY(dok.alloc(sizeof(int32_t)));
  launch_pairing_check(st, darg.as<PairingCheckDev>(), dok.as<int32_t>());
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  int32_t res = 0;
  HIP_TRY(hipMemcpy(&res, dok.p, sizeof res, hipMemcpyDeviceToHost));
  *ok = res;
  return SPP_OK;
}

// same product on the host with the single-proof pairing (pairing.hpp): needs no GPU
extern "C" int spp_pairing_check_host(uint32_t n_pairs, const uint8_t* g1s, const uint8_t* g2s, int* ok) {
  if (!g1s || !g2s || !ok) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  *ok = 0;
  if (n_pairs < 1 || n_pairs > 8) return fail(SPP_ERR_BAD_INPUT, "1 to 8 pairs");
  std::vector<std::pair<G1Affine, G2Affine>> pairs;
  for (uint32_t k = 0; k < n_pairs; k++) {
    for (int o = 0; o < 64; o += 32)
      if (!be_is_canonical<FqParams>(g1s + 64 * k + o)) return fail(SPP_ERR_FORMAT, "G1 coordinate not below q");
    for (int o = 0; o < 128; o += 32)
      if (!be_is_canonical<FqParams>(g2s + 128 * k + o)) return fail(SPP_ERR_FORMAT, "G2 coordinate not below q");
    G1Affine P = g1_from_raw(g1s + 64 * k);
    G2Affine Q = g2_from_raw(g2s + 128 * k);
    if (!g1_on_curve(P) || Q.is_inf() || !g2_on_curve(Q) || !g2_in_subgroup(Q)) return SPP_OK;
    pairs.push_back({P, Q});
  }
  *ok = pairing_product_is_one(pairs) ? 1 : 0;
  return SPP_OK;
}

// libspp C ABI, unit / micro-benchmark entry points: Fr NTT, table-based G1 MSM, general-base Pippenger (BASELINE.json
// configs[4]).  No reference equivalent: gnark's NTT / MSM are internal to `sunspot prove`.
#include "spp_internal.hpp"

// -----------------------------------------------------------------------------------------------------
// micro-benchmark / unit entry points
// -----------------------------------------------------------------------------------------------------
extern "C" int spp_ntt_fr(spp_ctx* ctx, uint8_t* data, uint32_t logn, int inverse) {
  if (!ctx || !data || logn == 0 || logn > 24) return fail(SPP_ERR_BAD_INPUT, "bad argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  const uint32_t n = 1u << logn;
  Fr w = fr_root_of_unity(logn);
  if (inverse) w = w.inv();
  std::vector<Fr> tw(n / 2 ? n / 2 : 1), host(n);
  Fr a = Fr::one();
  for (uint32_t k = 0; k < n / 2; k++) { tw[k] = a; a = a * w; }
  for (uint32_t i = 0; i < n; i++) host[i] = Fr::from_bytes_be(data + 32 * (size_t)i);
  DevBuf d_tw, d_x;
  HIP_TRY(d_tw.alloc(sizeof(Fr) * tw.size()));
  HIP_TRY(d_x.alloc(sizeof(Fr) * n));
  HIP_TRY(hipMemcpy(d_tw.p, tw.data(), sizeof(Fr) * tw.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_x.p, host.data(), sizeof(Fr) * n, hipMemcpyHostToDevice));
  launch_ntt(ctx->stream, d_x.as<Fr>(), logn, 1, d_tw.as<Fr>(), true, 1, 0);   // DIF: natural in, bit-reversed out
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(host.data(), d_x.p, sizeof(Fr) * n, hipMemcpyDeviceToHost));
  Fr ninv = inverse ? Fr::from_u64(n).inv() : Fr::one();
  for (uint32_t pos = 0; pos < n; pos++) {
    Fr v = host[pos];
    if (inverse) v = v * ninv;
    v.to_bytes_be(data + 32 * (size_t)bitrev(pos, logn));
  }
  return SPP_OK;
}

// -----------------------------------------------------------------------------------------------------
// general-base Pippenger MSM (BASELINE.json configs[4])
// -----------------------------------------------------------------------------------------------------
static G1Affine pippenger_finish(const G1XYZZ* d_windows) {
  // Horner over the 16 window sums: r = sum_j 2^(16 j) W_j
  std::vector<G1XYZZ> w(pippenger_windows());
  hipMemcpy(w.data(), d_windows, sizeof(G1XYZZ) * w.size(), hipMemcpyDeviceToHost);
  G1XYZZ r = G1XYZZ::infinity();
  for (int j = (int)w.size() - 1; j >= 0; j--) {
    for (int k = 0; k < 16; k++) r.dbl_inplace();
    r.add(w[j]);
  }
  return r.to_affine();
}

extern "C" int spp_msm_g1_pippenger(spp_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[64]) {
  if (!ctx || !out || (n && (!bases || !scalars))) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (n >= (1u << 31)) return fail(SPP_ERR_BAD_INPUT, "too many points");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  std::vector<G1Affine> pts(n);
  std::vector<Fr> sc(n);
  for (size_t i = 0; i < n; i++) {
    pts[i] = g1_from_raw(bases + 64 * i);
    sc[i] = Fr::from_bytes_be(scalars + 32 * i);
  }
  DevBuf dp, ds, dw;
  UP(dp, pts.data(), n * sizeof(G1Affine));
  UP(ds, sc.data(), n * sizeof(Fr));
  HIP_TRY(dw.alloc(pippenger_workspace_bytes((uint32_t)n)));
  G1XYZZ* win = nullptr;
  launch_pippenger_g1(st, dp.as<G1Affine>(), ds.as<Fr>(), (uint32_t)n, dw.p, &win, nullptr, nullptr);
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  g1_to_raw(pippenger_finish(win), out);
  return SPP_OK;
}

extern "C" int spp_msm_g2_pippenger(spp_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[128]) {
  if (!ctx || !out || (n && (!bases || !scalars))) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (n >= (1u << 31)) return fail(SPP_ERR_BAD_INPUT, "too many points");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  std::vector<G2Affine> pts(n);
  std::vector<Fr> sc(n);
  for (size_t i = 0; i < n; i++) {
    pts[i] = g2_from_raw(bases + 128 * i);
    sc[i] = Fr::from_bytes_be(scalars + 32 * i);
  }
  DevBuf dp, ds, dw;
  UP(dp, pts.data(), n * sizeof(G2Affine));
  UP(ds, sc.data(), n * sizeof(Fr));
  HIP_TRY(dw.alloc(pippenger_workspace_bytes_g2((uint32_t)n)));
  G2XYZZ* win = nullptr;
  launch_pippenger_g2(st, dp.as<G2Affine>(), ds.as<Fr>(), (uint32_t)n, dw.p, &win, nullptr, nullptr);
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  std::vector<G2XYZZ> w(pippenger_windows());
  HIP_TRY(hipMemcpy(w.data(), win, sizeof(G2XYZZ) * w.size(), hipMemcpyDeviceToHost));
  G2XYZZ r = G2XYZZ::infinity();
  for (int j = (int)w.size() - 1; j >= 0; j--) {
    for (int k = 0; k < 16; k++) r.dbl_inplace();
    r.add(w[j]);
  }
  g2_to_raw(r.to_affine(), out);
  return SPP_OK;
}

// Synthetic micro-benchmark, everything on the device: bases_i = k_i * G (k_i from a 64-bit LCG of `seed`), scalars
// uniform 254-bit values from the same generator; runs `iters` MSMs, returns the result of the last one, the mean
// wall time of one MSM and the mean duration of the bucket-accumulation kernel (HIP events).
// scale: if nonzero, every scalar is multiplied by it first (linearity checks: MSM(scale * s) = scale * MSM(s)).
extern "C" int spp_msm_g1_pippenger_bench(spp_ctx* ctx, size_t n, uint64_t seed, const uint8_t scale_be[32], int iters, uint8_t out[64],
                                          float* ms_total, float* ms_bucket_kernel) {
  return spp_msm_g1_pippenger_bench_dist(ctx, n, seed, 0, scale_be, iters, out, ms_total, ms_bucket_kernel);
}
extern "C" int spp_msm_g1_pippenger_bench_dist(spp_ctx* ctx, size_t n, uint64_t seed, uint32_t small_permille, const uint8_t scale_be[32],
                                               int iters, uint8_t out[64], float* ms_total, float* ms_bucket_kernel) {
  return spp_msm_g1_pippenger_bench_shard(ctx, n, 0, n, seed, small_permille, scale_be, iters, out, ms_total, ms_bucket_kernel);
}
// The same synthetic MSM cut over GPUs (BASELINE.json configs[4] on N GPUs): points [first, first + count) of the SAME
// n_total-point sequence -- every rank proves its contiguous share, the N partial sums (64 B each) are gathered and added
// (spp/multi.py msm_g1_sharded).  out = the partial sum of the share.
extern "C" int spp_msm_g1_pippenger_bench_shard(spp_ctx* ctx, size_t n_total, size_t first, size_t count, uint64_t seed, uint32_t small_permille,
                                                const uint8_t scale_be[32], int iters, uint8_t out[64], float* ms_total,
                                                float* ms_bucket_kernel) {
  if (!ctx || !out || n_total == 0 || count == 0 || first + count > n_total || iters <= 0 || small_permille > 1000)
    return fail(SPP_ERR_BAD_INPUT, "bad argument");
  if (n_total > (1u << 26)) return fail(SPP_ERR_BAD_INPUT, "n too large");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n = count;
  // host-generated scalars (n * 32 B; 512 MiB at 2^24) uploaded once; the generator is stepped through the points before `first`
  std::vector<Fr> ks(n), sc(n);
  uint64_t x = seed * 6364136223846793005ull + 1442695040888963407ull;
  auto next = [&]() { x = x * 6364136223846793005ull + 1442695040888963407ull; return x; };
  Fr scale = Fr::one();
  if (scale_be) scale = Fr::from_bytes_be(scale_be);
  for (size_t gi = 0; gi < first + n; gi++) {
    const uint64_t kv = next() | 1;
    uint32_t w[8];
    for (int k = 0; k < 8; k += 2) { uint64_t v = next(); w[k] = (uint32_t)v; w[k + 1] = (uint32_t)(v >> 32); }
    const bool small = small_permille && (next() >> 20) % 1000 < small_permille;
    if (gi < first) continue;
    const size_t i = gi - first;
    ks[i] = Fr::from_u64(kv);
    w[7] &= 0x1fffffffu;   // < 2^253 < r
    Fr s;
    for (int k = 0; k < 8; k++) s.l[k] = w[k];
    // witness-like: a byte-sized VALUE (Config 5); the uniform ones are raw words of a random element anyway
    if (small) s = Fr::from_u64(w[0] & 0xffu);
    sc[i] = scale_be ? s * scale : s;   // both are fixed representations of the same field element family
  }
  DevBuf dk, ds, dp, dw, dt, dg, dtmp, dpre;
  UP(dk, ks.data(), n * sizeof(Fr));
  UP(ds, sc.data(), n * sizeof(Fr));
  HIP_TRY(dp.alloc(n * sizeof(G1Affine)));
  HIP_TRY(dw.alloc(pippenger_workspace_bytes((uint32_t)n)));
  // bases = k_i * G through the generator's window table
  const uint32_t cb = 8, Wn = msm_windows(cb), E = 1u << (cb - 1);
  G1Affine g1{Fq::from_u64(1), Fq::from_u64(2)};
  UP(dg, &g1, sizeof g1);
  const size_t gr = ((size_t)Wn + 63) / 64 * 64;
  HIP_TRY(dt.alloc(sizeof(G1Affine) * msm_table_elems(1, cb, Wn)));
  HIP_TRY(dtmp.alloc(sizeof(G1XYZZ) * gr * E));
  HIP_TRY(dpre.alloc(sizeof(Fq) * gr * E));
  launch_build_table<Fq>(st, dg.as<G1Affine>(), 1, cb, Wn, 0, (uint32_t)gr, dt.as<G1Affine>(), dtmp.as<G1XYZZ>(), dpre.as<Fq>());
  launch_fixed_base_mul<Fq>(st, dt.as<G1Affine>(), cb, dk.as<Fr>(), (uint32_t)n, dp.as<G1Affine>(), nullptr);
  HIP_TRY(hipStreamSynchronize(st));
  hipEvent_t e0, e1, k0, k1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventCreate(&k0)); HIP_TRY(hipEventCreate(&k1));
  G1XYZZ* win = nullptr;
  launch_pippenger_g1(st, dp.as<G1Affine>(), ds.as<Fr>(), (uint32_t)n, dw.p, &win, nullptr, nullptr);   // warm-up
  HIP_TRY(hipStreamSynchronize(st));
  float tot = 0, kern = 0;
  for (int it = 0; it < iters; it++) {
    hipEventRecord(e0, st);
    launch_pippenger_g1(st, dp.as<G1Affine>(), ds.as<Fr>(), (uint32_t)n, dw.p, &win, k0, k1);
    hipEventRecord(e1, st);
    HIP_TRY(hipStreamSynchronize(st));
    float a = 0, b = 0;
    hipEventElapsedTime(&a, e0, e1);
    hipEventElapsedTime(&b, k0, k1);
    tot += a;
    kern += b;
  }
  HIP_TRY(hipGetLastError());
  hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(k0); hipEventDestroy(k1);
  if (ms_total) *ms_total = tot / iters;
  if (ms_bucket_kernel) *ms_bucket_kernel = kern / iters;
  g1_to_raw(pippenger_finish(win), out);
  return SPP_OK;
}

// libspp C ABI, witness-input side: what the reference computes on the client before it can call the prover (RLWE encryption
// + quotient witnesses, Poseidon / Merkle / Grumpkin, the ct_commitment sponge, whole audit input rows) and the auditor side
// (Shamir reconstruction, BFV decryption), batched on the GPU.
#include "spp_internal.hpp"

int spp_ensure_ctx_consts(spp_ctx* ctx) {
  if (ctx->consts_ready) return 0;
  auto flat = [](const PoseidonParams& pp) {
    std::vector<Fr> m;
    for (auto& row : pp.mds)
      for (auto& v : row) m.push_back(v);
    return m;
  };
  const PoseidonParams& p3 = poseidon_params(3);
  const PoseidonParams& p5 = poseidon_params(5);
  const Poseidon2Params& p2 = poseidon2_params();
  std::vector<Fr> mu(p2.mu, p2.mu + 4);
  Fr *a, *b, *c, *d, *f, *g;
  int e;
  auto canon = [](std::vector<Fr> v) {
    for (auto& x : v) x = x.canonical();
    return v;
  };
  auto flat29 = [](const PoseidonParams& pp) {
    std::vector<uint32_t> m;
    for (auto& row : pp.mds)
      for (auto& v : row) {
        const F29<FrParams> x = F29<FrParams>::from_fp(v);
        for (int k = 0; k < 9; k++) m.push_back(x.l[k]);
      }
    return m;
  };
  uint32_t *m3, *m5;
  if ((e = ctx_upload(ctx, &a, canon(p3.rc))) || (e = ctx_upload(ctx, &b, flat(p3))) || (e = ctx_upload(ctx, &c, canon(p5.rc))) ||
      (e = ctx_upload(ctx, &d, flat(p5))) || (e = ctx_upload(ctx, &f, p2.rc)) || (e = ctx_upload(ctx, &g, mu)) ||
      (e = ctx_upload(ctx, &m3, flat29(p3))) || (e = ctx_upload(ctx, &m5, flat29(p5))))
    return e;
  ctx->hc = HashConsts{a, b, c, d, f, g, m3, m5};
  // Grumpkin window table T[j][d] = (d+1) * 16^j * G, j < 64, d < 16
  std::vector<GkAffine> tab(64 * 16);
  GkXYZZ base = GkXYZZ::from_affine(grumpkin_generator());
  for (int j = 0; j < 64; j++) {
    GkAffine ba = base.to_affine();
    GkXYZZ run = base;
    for (int dd = 0; dd < 16; dd++) {
      tab[j * 16 + dd] = run.to_affine();
      run.madd(ba);
    }
    base = GkXYZZ::from_affine(tab[j * 16 + 15]);
  }
  if ((e = ctx_upload(ctx, &ctx->gk_table, tab))) return e;
  ctx->consts_ready = true;
  return 0;
}
// twiddle / twist tables of the RLWE NTT kernel (32 KB) + the scratch that receives the transformed public key
int spp_ensure_rlwe(spp_ctx* ctx) {
  if (ctx->rlwe_ready) return 0;
  static RnHostTables h;   // 33 KB: not on the stack
  rn_build_tables(h);
  RlweDev& rd = ctx->rlwe;
  int e;
  for (int k = 0; k < 2; k++) {
    rd.tb.f[k] = h.f[k];
    rd.pk_scale[k] = h.pk_scale[k];
    int32_t *w0, *w1, *ps, *ips;
    if ((e = ctx_upload(ctx, &w0, std::vector<int32_t>(h.w[k][0], h.w[k][0] + 1024))) ||
        (e = ctx_upload(ctx, &w1, std::vector<int32_t>(h.w[k][1], h.w[k][1] + 1024))) ||
        (e = ctx_upload(ctx, &ps, std::vector<int32_t>(h.psi[k], h.psi[k] + 1024))) ||
        (e = ctx_upload(ctx, &ips, std::vector<int32_t>(h.ipsi[k], h.ipsi[k] + 1024))))
      return e;
    rd.tb.w[k][0] = w0; rd.tb.w[k][1] = w1; rd.tb.psi[k] = ps; rd.tb.ipsi[k] = ips;

  }
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, sizeof(RlwePkDev)));
  ctx->owned.push_back(p);
  rd.pk = (RlwePkDev*)p;
  ctx->rlwe_ready = true;
  return 0;
}

extern "C" int spp_rlwe_witness_batch(spp_ctx* ctx, const uint32_t* pk_a, const uint32_t* pk_b, size_t count, const int8_t* r,
                                      const int8_t* e1, const int8_t* e2, const uint8_t* msg, uint32_t* c0, uint32_t* c1, int32_t* k0,
                                      int32_t* k1, uint8_t* packed_be) {
  if (!ctx || !pk_a || !pk_b || !r || !e1 || !e2 || !msg || !c0 || !c1 || !k0 || !k1) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (count == 0) return SPP_OK;
  for (int i = 0; i < 1024; i++)
    if (pk_a[i] >= 167772161u || pk_b[i] >= 167772161u) return fail(SPP_ERR_BAD_INPUT, "public key coefficient not in [0, q)");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (int e = spp_ensure_rlwe(ctx)) return e;
  DevBuf da, db, dr, de1, de2, dm, dc0, dc1, dk0, dk1, dp;
  UP(da, pk_a, 4096); UP(db, pk_b, 4096);
  UP(dr, r, count * 1024); UP(de1, e1, count * 64); UP(de2, e2, count * 1024); UP(dm, msg, count * 64);
  HIP_TRY(dc0.alloc(count * 64 * 4)); HIP_TRY(dc1.alloc(count * 1024 * 4)); HIP_TRY(dk0.alloc(count * 64 * 4)); HIP_TRY(dk1.alloc(count * 1024 * 4));
  if (packed_be) HIP_TRY(dp.alloc(count * 157 * 32));
  launch_rlwe_witness(st, ctx->rlwe, da.as<uint32_t>(), db.as<uint32_t>(), dr.as<int8_t>(), de1.as<int8_t>(), de2.as<int8_t>(), dm.as<uint8_t>(),
                      dc0.as<uint32_t>(), dc1.as<uint32_t>(), dk0.as<int32_t>(), dk1.as<int32_t>(), packed_be ? dp.as<uint8_t>() : nullptr,
                      (uint32_t)count);
  HIP_TRY(hipMemcpyAsync(c0, dc0.p, count * 64 * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(c1, dc1.p, count * 1024 * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(k0, dk0.p, count * 64 * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(k1, dk1.p, count * 1024 * 4, hipMemcpyDeviceToHost, st));
  if (packed_be) HIP_TRY(hipMemcpyAsync(packed_be, dp.p, count * 157 * 32, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}
// device-resident form (micro-benchmark: BASELINE.json configs[3]); all pointers are device pointers
extern "C" int spp_rlwe_witness_batch_device(spp_ctx* ctx, const void* d_pk_a, const void* d_pk_b, size_t count, const void* d_r,
                                             const void* d_e1, const void* d_e2, const void* d_msg, void* d_c0, void* d_c1, void* d_k0,
                                             void* d_k1, void* d_packed_be) {
  if (!ctx) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  if (int e = spp_ensure_rlwe(ctx)) return e;
  launch_rlwe_witness(ctx->stream, ctx->rlwe, (const uint32_t*)d_pk_a, (const uint32_t*)d_pk_b, (const int8_t*)d_r, (const int8_t*)d_e1,
                      (const int8_t*)d_e2, (const uint8_t*)d_msg, (uint32_t*)d_c0, (uint32_t*)d_c1, (int32_t*)d_k0, (int32_t*)d_k1,
                      (uint8_t*)d_packed_be, (uint32_t)count);
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}
extern "C" int spp_ctx_sync(spp_ctx* ctx) {
  if (!ctx) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SPP_OK;
}

extern "C" int spp_poseidon_hash_batch(spp_ctx* ctx, size_t count, int arity, const uint8_t* in, uint8_t* out) {
  if (!ctx || !in || !out) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (arity != 2 && arity != 4) return fail(SPP_ERR_BAD_INPUT, "arity must be 2 or 4");
  if (count == 0) return SPP_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  if (int e = spp_ensure_ctx_consts(ctx)) return e;
  hipStream_t st = ctx->stream;
  DevBuf di, dout;
  UP(di, in, count * arity * 32);
  HIP_TRY(dout.alloc(count * 32));
  launch_poseidon_hash(st, ctx->hc, di.as<uint8_t>(), (uint32_t)arity, dout.as<uint8_t>(), (uint32_t)count);
  HIP_TRY(hipMemcpyAsync(out, dout.p, count * 32, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" int spp_merkle_root_batch(spp_ctx* ctx, size_t count, uint32_t depth, const uint8_t* leaves, const uint64_t* indices,
                                     const uint8_t* siblings, uint8_t* roots) {
  if (!ctx || !leaves || !indices || !siblings || !roots) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (depth == 0 || depth > 64) return fail(SPP_ERR_BAD_INPUT, "depth out of range");
  if (count == 0) return SPP_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  if (int e = spp_ensure_ctx_consts(ctx)) return e;
  hipStream_t st = ctx->stream;
  DevBuf dl, di, ds, dr;
  UP(dl, leaves, count * 32); UP(di, indices, count * 8); UP(ds, siblings, count * depth * 32);
  HIP_TRY(dr.alloc(count * 32));
  launch_merkle_path(st, ctx->hc, dl.as<uint8_t>(), di.as<uint64_t>(), ds.as<uint8_t>(), depth, dr.as<uint8_t>(), (uint32_t)count);
  HIP_TRY(hipMemcpyAsync(roots, dr.p, count * 32, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}

// ShieldedPoolMerkleTree.getRoot + getProof (client/merkle.ts:165-221) for a tree of n_leaves inserted leaves
extern "C" int spp_merkle_build(spp_ctx* ctx, size_t n_leaves, uint32_t depth, const uint8_t* leaves, size_t n_queries,
                                const uint64_t* query_indices, uint8_t* siblings_out, uint8_t* root_out) {
  if (!ctx || !root_out || (n_leaves && !leaves) || (n_queries && (!query_indices || !siblings_out)))
    return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (depth == 0 || depth > 32 || n_leaves > ((size_t)1 << depth)) return fail(SPP_ERR_BAD_INPUT, "bad depth / too many leaves");
  for (size_t q = 0; q < n_queries; q++)
    if (query_indices[q] >= ((uint64_t)1 << depth)) return fail(SPP_ERR_BAD_INPUT, "query index out of range");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  if (int e = spp_ensure_ctx_consts(ctx)) return e;
  hipStream_t st = ctx->stream;
  // level sizes
  std::vector<size_t> cnt(depth + 1), off(depth + 2, 0);
  cnt[0] = n_leaves;
  for (uint32_t i = 0; i < depth; i++) cnt[i + 1] = (cnt[i] + 1) / 2;
  for (uint32_t i = 0; i <= depth; i++) off[i + 1] = off[i] + std::max<size_t>(cnt[i], 1);
  DevBuf dleaves, dnodes, ddef;
  UP(dleaves, leaves, n_leaves * 32);
  HIP_TRY(dnodes.alloc(off[depth + 1] * sizeof(Fr)));
  HIP_TRY(ddef.alloc((depth + 1) * sizeof(Fr)));
  Fr* nodes = dnodes.as<Fr>();
  launch_fr_from_be(st, dleaves.as<uint8_t>(), nodes, (uint32_t)n_leaves);
  std::vector<Fr> dflt(depth + 1);
  dflt[0] = Fr::zero();
  for (uint32_t i = 0; i < depth; i++) {
    // default hash of the next level: H(d_i, d_i) -- one lane, read back (depth <= 32 round trips at tree-build time)
    launch_merkle_level(st, ctx->hc, nullptr, 0, dflt[i], ddef.as<Fr>() + i + 1, 1);
    HIP_TRY(hipMemcpyAsync(&dflt[i + 1], ddef.as<Fr>() + i + 1, sizeof(Fr), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    launch_merkle_level(st, ctx->hc, nodes + off[i], (uint32_t)cnt[i], dflt[i], nodes + off[i + 1], (uint32_t)cnt[i + 1]);
  }
  std::vector<Fr> host(off[depth + 1]);
  HIP_TRY(hipMemcpyAsync(host.data(), nodes, host.size() * sizeof(Fr), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  Fr root = cnt[depth] ? host[off[depth]] : dflt[depth];
  root.to_bytes_be(root_out);
  for (size_t q = 0; q < n_queries; q++) {
    uint64_t idx = query_indices[q];
    for (uint32_t i = 0; i < depth; i++) {
      uint64_t sib = idx ^ 1;
      Fr v = sib < cnt[i] ? host[off[i] + sib] : dflt[i];
      v.to_bytes_be(siblings_out + (q * depth + i) * 32);
      idx >>= 1;
    }
  }
  return SPP_OK;
}

// -----------------------------------------------------------------------------------------------------
// incremental tree: ShieldedPoolMerkleTree (client/merkle.ts:146-222) with the levels kept in HBM.  insert() appends leaves and
// recomputes only the touched path(s): O(count + depth) hashes instead of the reference's O(2^depth) per getRoot / getProof.
// -----------------------------------------------------------------------------------------------------
struct spp_merkle_tree {
  spp_ctx* ctx = nullptr;
  uint32_t depth = 0;
  uint64_t n_leaves = 0, cap_leaves = 0;       // capacity of level 0 (level l holds cap_leaves >> l, + 1)
  MerkleTreeDev host{};                        // host mirror of the device descriptor
  MerkleTreeDev* dev = nullptr;
  Fr* d_dflt = nullptr;                        // depth + 1 default hashes
};
static size_t mt_level_cap(uint64_t cap_leaves, uint32_t l) { return (size_t)(cap_leaves >> l) + 1; }
static int mt_reserve(spp_merkle_tree* t, uint64_t want_leaves) {
  if (want_leaves <= t->cap_leaves) return 0;
  uint64_t cap = std::max<uint64_t>(t->cap_leaves ? t->cap_leaves : 1024, 1);
  while (cap < want_leaves) cap *= 2;
  cap = std::min<uint64_t>(cap, (uint64_t)1 << t->depth);
  hipStream_t st = t->ctx->stream;
  for (uint32_t l = 0; l <= t->depth; l++) {
    Fr* nw = nullptr;
    HIP_TRY(hipMalloc((void**)&nw, mt_level_cap(cap, l) * sizeof(Fr)));
    if (t->host.level[l]) {
      if (t->host.count[l]) HIP_TRY(hipMemcpyAsync(nw, t->host.level[l], t->host.count[l] * sizeof(Fr), hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));
      hipFree(t->host.level[l]);
    }
    t->host.level[l] = nw;
  }
  t->cap_leaves = cap;
  return 0;
}
static int mt_push_descriptor(spp_merkle_tree* t) {
  HIP_TRY(hipMemcpyAsync(t->dev, &t->host, sizeof(MerkleTreeDev), hipMemcpyHostToDevice, t->ctx->stream));
  return 0;
}
extern "C" int spp_merkle_tree_new(spp_ctx* ctx, uint32_t depth, spp_merkle_tree** out) {
  if (!ctx || !out) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (depth == 0 || depth > 32) return fail(SPP_ERR_BAD_INPUT, "depth must be 1..32");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  if (int e = spp_ensure_ctx_consts(ctx)) return e;
  spp_merkle_tree* t = new spp_merkle_tree();
  t->ctx = ctx;
  t->depth = depth;
  t->host.depth = depth;
  auto bail = [&](int e) { spp_merkle_tree_free(t); return e; };
  if (hipMalloc((void**)&t->dev, sizeof(MerkleTreeDev)) != hipSuccess || hipMalloc((void**)&t->d_dflt, 33 * sizeof(Fr)) != hipSuccess)
    return bail(fail(SPP_ERR_HIP, "hipMalloc"));
  launch_merkle_defaults(ctx->stream, ctx->hc, t->d_dflt, depth);
  if (hipMemcpyAsync(t->host.dflt, t->d_dflt, (depth + 1) * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return bail(fail(SPP_ERR_HIP, "default hashes"));
  if (int e = mt_reserve(t, 1024)) return bail(e);
  if (int e = mt_push_descriptor(t)) return bail(e);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *out = t;
  return SPP_OK;
}
extern "C" void spp_merkle_tree_free(spp_merkle_tree* t) {
  if (!t) return;
  hipSetDevice(t->ctx->device);
  hipStreamSynchronize(t->ctx->stream);
  for (uint32_t l = 0; l <= 32; l++) if (t->host.level[l]) hipFree(t->host.level[l]);
  if (t->dev) hipFree(t->dev);
  if (t->d_dflt) hipFree(t->d_dflt);
  delete t;
}
extern "C" uint64_t spp_merkle_tree_size(const spp_merkle_tree* t) { return t ? t->n_leaves : 0; }
// insert(commitment) (merkle.ts:158-163) for `count` leaves at once; *first_index receives the index of the first one
extern "C" int spp_merkle_tree_insert(spp_merkle_tree* t, size_t count, const uint8_t* leaves, uint64_t* first_index) {
  if (!t || (count && !leaves)) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  for (size_t i = 0; i < count; i++)
    if (!be_is_canonical<FrParams>(leaves + 32 * i)) return fail(SPP_ERR_BAD_INPUT, "leaf %zu is not a canonical field element", i);
  spp_ctx* ctx = t->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);   // the size, the first index handed out and the "full" check are read under the lock:
                                             // two inserting threads must not get the same index
  if (first_index) *first_index = t->n_leaves;
  if (count == 0) return SPP_OK;
  if (count > ((uint64_t)1 << t->depth) - t->n_leaves) return fail(SPP_ERR_BAD_INPUT, "tree is full");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (int e = mt_reserve(t, t->n_leaves + count)) return e;
  DevBuf dl;
  UP(dl, leaves, count * 32);
  const uint64_t first = t->n_leaves, last = first + count - 1;
  launch_fr_from_be(st, dl.as<uint8_t>(), t->host.level[0] + first, (uint32_t)count);
  t->n_leaves += count;
  t->host.count[0] = t->n_leaves;
  for (uint32_t l = 0; l < t->depth; l++) {
    const uint64_t p0 = first >> (l + 1), p1 = last >> (l + 1);
    launch_merkle_update(st, ctx->hc, t->host.level[l], t->host.count[l], t->d_dflt + l, t->host.level[l + 1], p0, (uint32_t)(p1 - p0 + 1));
    t->host.count[l + 1] = (t->host.count[l] + 1) / 2;
  }
  if (int e = mt_push_descriptor(t)) return e;
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}
// getRoot (merkle.ts:165-176): one 32-byte read
extern "C" int spp_merkle_tree_root(spp_merkle_tree* t, uint8_t root[32]) {
  if (!t || !root) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  std::lock_guard<std::mutex> lk(t->ctx->mu);
  HIP_TRY(hipSetDevice(t->ctx->device));
  Fr v = t->host.dflt[t->depth];
  if (t->n_leaves) HIP_TRY(hipMemcpy(&v, t->host.level[t->depth], sizeof(Fr), hipMemcpyDeviceToHost));
  v.to_bytes_be(root);
  return SPP_OK;
}
// getProof (merkle.ts:198-221) for n indices: siblings_out = n * depth * 32 B; no hashing, depth reads per query
extern "C" int spp_merkle_tree_proofs(spp_merkle_tree* t, size_t n, const uint64_t* indices, uint8_t* siblings_out) {
  if (!t || (n && (!indices || !siblings_out))) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (n == 0) return SPP_OK;
  if (n > (1u << 24)) return fail(SPP_ERR_BAD_INPUT, "too many queries in one call");
  for (size_t q = 0; q < n; q++)
    if (indices[q] >= ((uint64_t)1 << t->depth)) return fail(SPP_ERR_BAD_INPUT, "query index out of range");
  spp_ctx* ctx = t->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf di, dout;
  UP(di, indices, n * sizeof(uint64_t));
  HIP_TRY(dout.alloc(n * t->depth * 32));
  launch_merkle_gather(st, t->dev, t->depth, di.as<uint64_t>(), (uint32_t)n, dout.as<uint8_t>());
  HIP_TRY(hipMemcpyAsync(siblings_out, dout.p, n * t->depth * 32, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" int spp_grumpkin_keygen_batch(spp_ctx* ctx, size_t count, const uint8_t* sk, uint8_t* xy) {
  if (!ctx || !sk || !xy) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (count == 0) return SPP_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  if (int e = spp_ensure_ctx_consts(ctx)) return e;
  hipStream_t st = ctx->stream;
  DevBuf ds, dx;
  UP(ds, sk, count * 32);
  HIP_TRY(dx.alloc(count * 64));
  launch_grumpkin_keygen(st, ctx->gk_table, ds.as<uint8_t>(), dx.as<uint8_t>(), (uint32_t)count);
  HIP_TRY(hipMemcpyAsync(xy, dx.p, count * 64, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" int spp_poseidon2_sponge_batch(spp_ctx* ctx, size_t count, uint32_t n, const uint8_t* in, uint8_t* out) {
  if (!ctx || !in || !out) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (count == 0) return SPP_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  if (int e = spp_ensure_ctx_consts(ctx)) return e;
  hipStream_t st = ctx->stream;
  DevBuf di, dout;
  UP(di, in, count * n * 32);
  HIP_TRY(dout.alloc(count * 32));
  launch_poseidon2_sponge(st, ctx->hc, di.as<uint8_t>(), n, dout.as<uint8_t>(), (uint32_t)count);
  HIP_TRY(hipMemcpyAsync(out, dout.p, count * 32, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}
// -----------------------------------------------------------------------------------------------------
// audit inputs end to end on the device: (sk, r, e1, e2) -> 3360-field rows for spp_prove_batch(_device)
// -----------------------------------------------------------------------------------------------------
namespace {
struct ScratchPiece {
  void* p = nullptr;
  template <class T> T* as() { return (T*)p; }
};
const size_t AUDIT_PIECES = 10;
void audit_piece_sizes(size_t count, size_t sizes[AUDIT_PIECES]) {
  const size_t v[AUDIT_PIECES] = {count * 64, count * 64, count * 64 * 4, count * 1024 * 4, count * 64 * 4, count * 1024 * 4, count * 157 * 32, count * 32,
                                  count * 32, sizeof(RlwePkDev)};
  for (size_t i = 0; i < AUDIT_PIECES; i++) sizes[i] = (v[i] + 255) / 256 * 256;
}
}  // namespace
size_t spp_audit_scratch_bytes(size_t count) {
  size_t sizes[AUDIT_PIECES], total = 0;
  audit_piece_sizes(count, sizes);
  for (size_t sz : sizes) total += sz;
  return total;
}
// Enqueues the whole input pipeline of scripts/generate_audit.py:468-641 on `st` (no synchronisation): the rows are complete
// for whatever is enqueued on `st` next.  scratch: spp_audit_scratch_bytes(count) bytes of device memory that stay untouched
// until those kernels have run.
int spp_audit_inputs_enqueue(spp_ctx* ctx, hipStream_t st, void* scratch, const uint32_t* d_pk_a, const uint32_t* d_pk_b, uint32_t count,
                             const uint8_t* d_sk, const int8_t* d_r, const int8_t* d_e1, const int8_t* d_e2, uint8_t* d_rows) {
  ScratchPiece xy, msg, c0, c1, k0, k1, packed, ct, wa, pkhat;
  {
    size_t sizes[AUDIT_PIECES], off = 0;
    audit_piece_sizes(count, sizes);
    ScratchPiece* pieces[AUDIT_PIECES] = {&xy, &msg, &c0, &c1, &k0, &k1, &packed, &ct, &wa, &pkhat};
    for (size_t i = 0; i < AUDIT_PIECES; i++) {
      pieces[i]->p = (uint8_t*)scratch + off;
      off += sizes[i];
    }
  }
  if (int e = spp_ensure_rlwe(ctx)) return e;
  RlweDev rd = ctx->rlwe;
  rd.pk = pkhat.as<RlwePkDev>();   // the transformed public key of THIS call (calls in flight on other streams may use other keys)
  launch_grumpkin_keygen(st, ctx->gk_table, d_sk, xy.as<uint8_t>(), count);                       // generate_audit.py:482
  launch_poseidon_hash(st, ctx->hc, xy.as<uint8_t>(), 2, wa.as<uint8_t>(), count);                 // wa_commitment
  launch_audit_msg(st, xy.as<uint8_t>(), msg.as<uint8_t>(), count);                                // :489-496
  launch_rlwe_witness(st, rd, d_pk_a, d_pk_b, d_r, d_e1, d_e2, msg.as<uint8_t>(), c0.as<uint32_t>(), c1.as<uint32_t>(), k0.as<int32_t>(),
                      k1.as<int32_t>(), packed.as<uint8_t>(), count);                              // :507-584
  launch_poseidon2_sponge(st, ctx->hc, packed.as<uint8_t>(), 157, ct.as<uint8_t>(), count);        // ct_commitment :587
  launch_audit_assemble(st, wa.as<uint8_t>(), ct.as<uint8_t>(), packed.as<uint8_t>(), d_sk, d_r, d_e1, d_e2, k0.as<int32_t>(),
                        k1.as<int32_t>(), d_rows, count);                                          // Prover.toml order :630-641
  return SPP_OK;
}
static int audit_inputs_on_device(spp_ctx* ctx, const uint32_t* d_pk_a, const uint32_t* d_pk_b, uint32_t count, const uint8_t* d_sk,
                                  const int8_t* d_r, const int8_t* d_e1, const int8_t* d_e2, uint8_t* d_rows) {
  hipStream_t st = ctx->stream;
  // temporaries kept between calls (grow only): a hipFree per call would drain every stream of the device
  const size_t total = spp_audit_scratch_bytes(count);
  if (total > ctx->audit_scratch_cap) {
    if (ctx->audit_scratch) HIP_TRY(hipFree(ctx->audit_scratch));
    ctx->audit_scratch = nullptr;
    ctx->audit_scratch_cap = 0;
    HIP_TRY(hipMalloc(&ctx->audit_scratch, total));
    ctx->audit_scratch_cap = total;
  }
  if (int e = spp_audit_inputs_enqueue(ctx, st, ctx->audit_scratch, d_pk_a, d_pk_b, count, d_sk, d_r, d_e1, d_e2, d_rows)) return e;
  HIP_TRY(hipStreamSynchronize(st));   // the rows are complete when the call returns (the caller hands them to a proving stream)
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}
extern "C" int spp_audit_inputs_batch_device(spp_ctx* ctx, const void* d_pk_a, const void* d_pk_b, size_t count, const void* d_sk,
                                             const void* d_r, const void* d_e1, const void* d_e2, void* d_rows) {
  if (!ctx || !d_pk_a || !d_pk_b || !d_sk || !d_r || !d_e1 || !d_e2 || !d_rows) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (count == 0) return SPP_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  if (int e = spp_ensure_ctx_consts(ctx)) return e;
  return audit_inputs_on_device(ctx, (const uint32_t*)d_pk_a, (const uint32_t*)d_pk_b, (uint32_t)count, (const uint8_t*)d_sk,
                                (const int8_t*)d_r, (const int8_t*)d_e1, (const int8_t*)d_e2, (uint8_t*)d_rows);
}
extern "C" int spp_audit_inputs_batch(spp_ctx* ctx, const uint32_t* pk_a, const uint32_t* pk_b, size_t count, const uint8_t* sk,
                                      const int8_t* r, const int8_t* e1, const int8_t* e2, uint8_t* rows) {
  if (!ctx || !pk_a || !pk_b || !sk || !r || !e1 || !e2 || !rows) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (count == 0) return SPP_OK;
  for (int i = 0; i < 1024; i++)
    if (pk_a[i] >= 167772161u || pk_b[i] >= 167772161u) return fail(SPP_ERR_BAD_INPUT, "public key coefficient not in [0, q)");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  if (int e = spp_ensure_ctx_consts(ctx)) return e;
  hipStream_t st = ctx->stream;
  DevBuf da, db, ds, dr, de1, de2, drows;
  UP(da, pk_a, 4096); UP(db, pk_b, 4096); UP(ds, sk, count * 32);
  UP(dr, r, count * 1024); UP(de1, e1, count * 64); UP(de2, e2, count * 1024);
  HIP_TRY(drows.alloc(count * 3360 * 32));
  if (int e = audit_inputs_on_device(ctx, da.as<uint32_t>(), db.as<uint32_t>(), (uint32_t)count, ds.as<uint8_t>(), dr.as<int8_t>(),
                                     de1.as<int8_t>(), de2.as<int8_t>(), drows.as<uint8_t>()))
    return e;
  HIP_TRY(hipMemcpy(rows, drows.p, count * 3360 * 32, hipMemcpyDeviceToHost));
  return SPP_OK;
}

// -----------------------------------------------------------------------------------------------------
// auditor side: Shamir reconstruction of the RLWE secret key, batch decryption of audit ciphertexts
// -----------------------------------------------------------------------------------------------------
extern "C" int spp_shamir_reconstruct(spp_ctx* ctx, uint32_t t, const uint32_t* xs, const uint8_t* ys, size_t n, uint8_t* secret_be,
                                      uint32_t* sk_mod_q) {
  if (!ctx || !xs || !ys || (!secret_be && !sk_mod_q)) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (t == 0 || t > 64) return fail(SPP_ERR_BAD_INPUT, "threshold out of range");
  for (uint32_t i = 0; i < t; i++)
    for (uint32_t j = 0; j < i; j++)
      if (xs[i] == xs[j]) return fail(SPP_ERR_BAD_INPUT, "duplicate share index");
  if (n == 0) return SPP_OK;
  // Lagrange coefficients at 0: lambda_i = prod_{j != i} (-x_j) / (x_i - x_j)   (rlwe_decrypt.py:38-51)
  std::vector<Fr> lam(t);
  for (uint32_t i = 0; i < t; i++) {
    Fr num = Fr::one(), den = Fr::one();
    for (uint32_t j = 0; j < t; j++) {
      if (i == j) continue;
      num = num * Fr::from_u64(xs[j]).neg();
      den = den * (Fr::from_u64(xs[i]) - Fr::from_u64(xs[j]));
    }
    lam[i] = num * den.inv();
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf dl, dy, ds, dq;
  UP(dl, lam.data(), sizeof(Fr) * t);
  UP(dy, ys, (size_t)t * n * 32);
  if (secret_be) HIP_TRY(ds.alloc(n * 32));
  if (sk_mod_q) HIP_TRY(dq.alloc(n * 4));
  launch_shamir_combine(st, dl.as<Fr>(), dy.as<uint8_t>(), t, (uint32_t)n, secret_be ? ds.as<uint8_t>() : nullptr,
                        sk_mod_q ? dq.as<uint32_t>() : nullptr);
  if (secret_be) HIP_TRY(hipMemcpyAsync(secret_be, ds.p, n * 32, hipMemcpyDeviceToHost, st));
  if (sk_mod_q) HIP_TRY(hipMemcpyAsync(sk_mod_q, dq.p, n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" int spp_rlwe_decrypt_batch(spp_ctx* ctx, const uint32_t* sk_mod_q, size_t count, const uint32_t* c0, const uint32_t* c1,
                                      uint8_t* msg) {
  if (!ctx || !sk_mod_q || !c0 || !c1 || !msg) return fail(SPP_ERR_BAD_INPUT, "NULL argument");
  if (count == 0) return SPP_OK;
  for (int i = 0; i < 1024; i++)
    if (sk_mod_q[i] >= 167772161u) return fail(SPP_ERR_BAD_INPUT, "secret key coefficient not in [0, q)");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf dsk, d0, d1, dm;
  UP(dsk, sk_mod_q, 4096); UP(d0, c0, count * 64 * 4); UP(d1, c1, count * 1024 * 4);
  HIP_TRY(dm.alloc(count * 64));
  launch_rlwe_decrypt(st, dsk.as<uint32_t>(), d0.as<uint32_t>(), d1.as<uint32_t>(), dm.as<uint8_t>(), (uint32_t)count);
  HIP_TRY(hipMemcpyAsync(msg, dm.p, count * 64, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return SPP_OK;
}

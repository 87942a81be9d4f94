// Stand-alone witness-INPUT kernels: everything the reference computes on the client before it can call the
// prover, batched on the GPU.
//
//   k_rlwe_witness        scripts/generate_audit.py:45-66,236-243,507-554 (negacyclic products, quotient witnesses)
//                         + pack_values :154-163  -- also demo-frontend/app/lib/rlwe.ts:157-247
//   k_poseidon_hash       client/merkle.ts:22-38 (poseidonHash2/4), noir_circuit/src/main.nr:1-9
//   k_merkle_path         noir_circuit/src/main.nr:11-29, client/merkle.ts:198-221
//   k_merkle_level        client/merkle.ts:165-176 (getRoot: one tree level per launch)
//   k_grumpkin_keygen     client/merkle.ts:98-113 (generateIdentityKeypair), main.nr:54-59
//   k_poseidon2_sponge    ct_helper/src/main.nr:15-34 (= scripts/generate_audit.py:355-374)
#include "kernels.hpp"
#include "poseidon29.hpp"
#include "lanes.hpp"
#include "rlwe_ntt.hpp"

namespace spp {

// ----------------------------------------------------------------------------------------------------
// RLWE witness generation: one wavefront per instance, exact negacyclic products through a 1024-point NTT held in LDS
// (rlwe_ntt.hpp: two prime fields q and 7*2^26+1, CRT digit = the quotient witness).  Per instance:
//   forward transform of r (twisted by psi^j) in both fields                                   -> R
//   R . NTT(a)/1024, inverse transform, untwist, CRT  -> c1[1024], k1[1024]   (generate_audit.py:517-518, 547-554)
//   R . NTT(b)/1024, inverse transform, untwist, CRT  -> c0[64],   k0[64]     (:513-514, 539-545; slots 0..63 only)
//   pack_values (:154-163) from LDS.
// Lane t owns coefficients t + 64 j: every global access of a wave is a contiguous 64-element row.  The transforms of the
// public key (k_rlwe_pk_ntt, once per call) use the same routine.
// ----------------------------------------------------------------------------------------------------
static constexpr int RL_N = 1024, RL_SLOTS = 64;
static constexpr long long RL_Q = 167772161ll, RL_DELTA = 655360ll;

template <bool REDUCE>
__device__ __forceinline__ void rn_ntt1(uint32_t lane, int32_t (&x)[16], int32_t* lds, const RnField& f, const int32_t* w, int dir) {
  rn_pass1<REDUCE>(lane, x, lds, f, w, dir);
  __syncthreads();
  rn_pass2_read(lane, x, lds);
  __syncthreads();
  rn_pass2<REDUCE>(lane, x, lds, f, w, dir);
  __syncthreads();
  rn_pass3(lane, x, lds, f, dir);
  __syncthreads();
}

// block 0: a, block 1: b.  hat[field][i] = NTT(pk psi^j)[i] / 1024 in Montgomery form, zero positions listed.
__global__ void __launch_bounds__(64) k_rlwe_pk_ntt(RnTables tb, const uint32_t* __restrict__ pk_a, const uint32_t* __restrict__ pk_b,
                                                    int32_t pk_scale0, int32_t pk_scale1, RlwePkDev* __restrict__ out) {
  __shared__ int32_t lds[RN_LDS_WORDS];
  __shared__ uint32_t nz;
  const uint32_t lane = threadIdx.x, poly = blockIdx.x;
  const uint32_t* src = poly == 0 ? pk_a : pk_b;
  if (lane == 0) nz = 0;
  __syncthreads();
  int32_t x[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint32_t i = lane + 64 * j, v = src[i];
    if (v == 0) out->zeros[poly][atomicAdd(&nz, 1u)] = (uint16_t)i;
    x[j] = rn_mul((int32_t)v, tb.psi[0][i], tb.f[0]);
  }
  rn_ntt1<true>(lane, x, lds, tb.f[0], tb.w[0][0], 0);
#pragma unroll
  for (int j = 0; j < 16; j++) out->hat[poly][0][lane + 64 * j] = rn_mul(x[j], pk_scale0, tb.f[0]);
#pragma unroll
  for (int j = 0; j < 16; j++) x[j] = rn_mul((int32_t)src[lane + 64 * j], tb.psi[1][lane + 64 * j], tb.f[1]);
  rn_ntt1<false>(lane, x, lds, tb.f[1], tb.w[1][0], 0);
#pragma unroll
  for (int j = 0; j < 16; j++) out->hat[poly][1][lane + 64 * j] = rn_mul(x[j], pk_scale1, tb.f[1]);
  __syncthreads();
  if (lane == 0) out->nzeros[poly] = nz;
}

// out[j] = (R . hat) transformed back (before the untwist) at coefficient lane + 64 j, field K
template <int K>
__device__ __forceinline__ void rn_product(uint32_t lane, const int32_t (&R)[16], const int32_t* __restrict__ hat, int32_t (&out)[16],
                                           int32_t* lds, const RnTables& tb) {
#pragma unroll
  for (int j = 0; j < 16; j++) out[j] = rn_mul(R[j], hat[lane + 64 * j], tb.f[K]);
  rn_ntt1<K == 0>(lane, out, lds, tb.f[K], tb.w[K][1], 1);
}

// coefficient `lane` (< 64) of the same product through the pruned inverse transform
template <int K>
__device__ __forceinline__ int32_t rn_product_first64(uint32_t lane, const int32_t (&R)[16], const int32_t* __restrict__ hat, int32_t* lds,
                                                      const RnTables& tb) {
  int32_t x[16];
#pragma unroll
  for (int j = 0; j < 16; j++) x[j] = rn_mul(R[j], hat[lane + 64 * j], tb.f[K]);
  rn_pass1<K == 0>(lane, x, lds, tb.f[K], tb.w[K][1], 1);
  __syncthreads();
  rn_pass2_read(lane, x, lds);
  __syncthreads();
  rn_pass2_first64<K == 0>(lane, x, lds, tb.f[K], tb.w[K][1], 1);
  __syncthreads();
  const int32_t v = rn_pass3_first64(lane, lds);
  __syncthreads();
  return v;
}

__global__ void __launch_bounds__(64, 4) k_rlwe_witness(RnTables tb, const RlwePkDev* __restrict__ pk, const int8_t* __restrict__ r_in,
                                                        const int8_t* __restrict__ e1_in, const int8_t* __restrict__ e2_in,
                                                        const uint8_t* __restrict__ msg_in, uint32_t* __restrict__ c0_out,
                                                        uint32_t* __restrict__ c1_out, int32_t* __restrict__ k0_out,
                                                        int32_t* __restrict__ k1_out, uint8_t* __restrict__ packed_be, uint32_t count) {
  __shared__ int32_t lds[RN_LDS_WORDS];       // exchange buffer of the transform in flight (one field at a time)
  __shared__ int32_t pre[RL_N + 64];          // prefix scan of r -> suffix sums at [64 + i]; then c0 | c1 for the packing epilogue
  __shared__ int8_t rbytes[RL_N];
  __shared__ int32_t tot[64], offs[64];
  const uint32_t inst = blockIdx.x, lane = threadIdx.x;
  if (inst >= count) return;
  // ---- r: one 16-byte load per lane, transposed through LDS to the lane layout i = lane + 64 j ----
  reinterpret_cast<uint4*>(rbytes)[lane] = reinterpret_cast<const uint4*>(r_in + (size_t)inst * RL_N)[lane];
  __syncthreads();
  int32_t R0[16], R1[16];
  int32_t suffix0;   // wrap correction of coefficient `lane` (needed again for the message slots)
  {
    int32_t rv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) rv[j] = rbytes[lane + 64 * j];
    // forward transforms of r psi^j in both fields
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t i = lane + 64 * j;
      R0[j] = rn_mul(rv[j], tb.psi[0][i], tb.f[0]);     // signed arithmetic: r_j in [-128, 127] goes in as it is
      R1[j] = rn_mul(rv[j], tb.psi[1][i], tb.f[1]);
    }
    // wrap correction: suffix sums of r (rlwe_ntt.hpp), left in LDS at pre[64 + i] (the slot that later receives c1[i])
    rn_scan_scatter(lane, rv, pre);
    __syncthreads();
    rn_scan_chunk(lane, pre, tot);
    __syncthreads();
    int32_t offset, total;
    rn_scan_offsets(lane, tot, offset, total);
    offs[lane] = offset;
    __syncthreads();
    int32_t suffix[16];
    rn_scan_gather(lane, total, pre, offs, suffix);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) pre[64 + lane + 64 * j] = suffix[j];
    suffix0 = suffix[0];
  }
  rn_ntt1<true>(lane, R0, lds, tb.f[0], tb.w[0][0], 0);
  rn_ntt1<false>(lane, R1, lds, tb.f[1], tb.w[1][0], 0);
  uint32_t* cs = reinterpret_cast<uint32_t*>(pre);   // c0 at [0,64), c1 at [64, 1088)
  // ---- a: c1 / k1 ----
  {
    int32_t s0[16], y[16];
    rn_product<0>(lane, R0, pk->hat[0][0], s0, lds, tb);
#pragma unroll
    for (int j = 0; j < 16; j++) s0[j] = rn_canon(rn_mul(s0[j], tb.ipsi[0][lane + 64 * j], tb.f[0]), tb.f[0]);
    rn_product<1>(lane, R1, pk->hat[0][1], y, lds, tb);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t i = lane + 64 * j;
      const int32_t s1 = rn_canon(rn_mul(y[j], tb.ipsi[1][i], tb.f[1]), tb.f[1]);
      const int32_t t = rn_crt_digit(s0[j], s1, tb.f[1]);
      int32_t k;
      uint32_t rem;
      rn_quot_rem(s0[j], t, (int32_t)e2_in[(size_t)inst * RL_N + i], k, rem);
      k += pre[64 + i];                 // own slot: read the suffix sum, then overwrite it with the remainder
      c1_out[(size_t)inst * RL_N + i] = rem;
      k1_out[(size_t)inst * RL_N + i] = k;
      cs[RL_SLOTS + i] = rem;
    }
  }
  // ---- b: c0 / k0 (message slots: coefficients 0..63 = element j = 0 of every lane) ----
  {
    const int32_t y0 = rn_product_first64<0>(lane, R0, pk->hat[1][0], lds, tb);
    const int32_t y1 = rn_product_first64<1>(lane, R1, pk->hat[1][1], lds, tb);
    const uint32_t i = lane;
    const int32_t s0 = rn_canon(rn_mul(y0, tb.ipsi[0][i], tb.f[0]), tb.f[0]);
    const int32_t s1 = rn_canon(rn_mul(y1, tb.ipsi[1][i], tb.f[1]), tb.f[1]);
    const int32_t t = rn_crt_digit(s0, s1, tb.f[1]);
    int32_t k;
    uint32_t rem;
    rn_quot_rem(s0, t, (int32_t)e1_in[(size_t)inst * RL_SLOTS + i] + (int32_t)RL_DELTA * (int32_t)msg_in[(size_t)inst * RL_SLOTS + i], k, rem);
    k += suffix0;
    c0_out[(size_t)inst * RL_SLOTS + i] = rem;
    k0_out[(size_t)inst * RL_SLOTS + i] = k;
    cs[i] = rem;
  }
  // ---- rare: the public key has zero coefficients (probability 1/q each): the wrapped term is 0, not q * r_j ----
  const uint32_t nza = pk->nzeros[0], nzb = pk->nzeros[1];
  if (nza | nzb) {
    if (nza)
#pragma unroll 1
      for (uint32_t j = 0; j < 16; j++) {
        const uint32_t i = lane + 64 * j;
        k1_out[(size_t)inst * RL_N + i] -= rn_zero_correction(i, pk->zeros[0], nza, rbytes);
      }
    if (nzb) k0_out[(size_t)inst * RL_SLOTS + lane] -= rn_zero_correction(lane, pk->zeros[1], nzb, rbytes);
  }
  __syncthreads();
  // ---- pack 7 x 32-bit per field (pack_values), 32-byte big-endian: 10 + 147 fields; one 32-bit word per lane and step ----
  if (packed_be) {
    uint32_t* out = reinterpret_cast<uint32_t*>(packed_be + (size_t)inst * 157 * 32);
    for (uint32_t w = lane; w < 157 * 8; w += 64) {
      const uint32_t f = w >> 3, jw = 7 - (w & 7);      // big-endian word w & 7 of field f  <->  little-endian word jw
      uint32_t v = 0;
      if (jw < 7) {
        if (f < 10) { const uint32_t idx = 7 * f + jw; if (idx < RL_SLOTS) v = cs[idx]; }
        else { const uint32_t idx = 7 * (f - 10) + jw; if (idx < RL_N) v = cs[RL_SLOTS + idx]; }
      }
      out[w] = __builtin_bswap32(v);
    }
  }
}
void launch_rlwe_witness(hipStream_t st, const RlweDev& rd, const uint32_t* pk_a, const uint32_t* pk_b, const int8_t* r, const int8_t* e1,
                         const int8_t* e2, const uint8_t* msg, uint32_t* c0, uint32_t* c1, int32_t* k0, int32_t* k1, uint8_t* packed_be,
                         uint32_t count) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_rlwe_pk_ntt, dim3(2), dim3(64), 0, st, rd.tb, pk_a, pk_b, rd.pk_scale[0], rd.pk_scale[1], rd.pk);
  hipLaunchKernelGGL(k_rlwe_witness, dim3(count), dim3(64), 0, st, rd.tb, rd.pk, r, e1, e2, msg, c0, c1, k0, k1, packed_be, count);
}

// ----------------------------------------------------------------------------------------------------
// Poseidon (t = 3, 5), permutation with the state in registers; one lane per hash
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ Fr load_be(const uint8_t* p) {
  uint8_t buf[32];
  for (int i = 0; i < 32; i++) buf[i] = p[i];
  return Fr::from_bytes_be(buf);
}
__device__ __forceinline__ void store_be(uint8_t* p, const Fr& v) {
  uint8_t buf[32];
  v.to_bytes_be(buf);
  for (int i = 0; i < 32; i++) p[i] = buf[i];
}
__device__ __forceinline__ Fr sbox5(const Fr& x) {
  Fr x2 = x.sqr();
  return x2.sqr() * x;
}
// the permutation itself lives in poseidon29.hpp (9x29-bit form, lazy MDS rows); nothing is emitted here
template <int T>
__device__ __noinline__ void poseidon_permute(Fr (&s)[T], const Fr* __restrict__ rc, const uint32_t* __restrict__ mds29, int rp) {
  poseidon_permute29<T, false>(s, rc, mds29, rp, PoseidonNoEmit{});
}
__device__ __forceinline__ Fr poseidon_hash2(const HashConsts& hc, const Fr& a, const Fr& b) {
  Fr s[3] = {Fr::zero(), a, b};
  poseidon_permute<3>(s, hc.pos3_rc, hc.pos3_mds29, 57);
  return s[0];
}

__global__ void __launch_bounds__(64) k_poseidon_hash(HashConsts hc, const uint8_t* __restrict__ in_be, uint32_t arity,
                                                      uint8_t* __restrict__ out_be, uint32_t count) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count) return;
  const uint8_t* in = in_be + (size_t)g * arity * 32;
  Fr h;
  if (arity == 2) {
    h = poseidon_hash2(hc, load_be(in), load_be(in + 32));
  } else {
    Fr s[5] = {Fr::zero(), load_be(in), load_be(in + 32), load_be(in + 64), load_be(in + 96)};
    poseidon_permute<5>(s, hc.pos5_rc, hc.pos5_mds29, 60);
    h = s[0];
  }
  store_be(out_be + (size_t)g * 32, h);
}
void launch_poseidon_hash(hipStream_t st, HashConsts hc, const uint8_t* in_be, uint32_t arity, uint8_t* out_be, uint32_t count) {
  if (count) hipLaunchKernelGGL(k_poseidon_hash, dim3((count + 63) / 64), dim3(64), 0, st, hc, in_be, arity, out_be, count);
}

// root of one authentication path per lane (main.nr:11-29)
__global__ void __launch_bounds__(64) k_merkle_path(HashConsts hc, const uint8_t* __restrict__ leaf_be, const uint64_t* __restrict__ index,
                                                    const uint8_t* __restrict__ siblings_be, uint32_t depth, uint8_t* __restrict__ root_be,
                                                    uint32_t count) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count) return;
  Fr cur = load_be(leaf_be + (size_t)g * 32);
  const uint64_t idx = index[g];
  for (uint32_t i = 0; i < depth; i++) {
    Fr sib = load_be(siblings_be + ((size_t)g * depth + i) * 32);
    cur = ((idx >> i) & 1) ? poseidon_hash2(hc, sib, cur) : poseidon_hash2(hc, cur, sib);
  }
  store_be(root_be + (size_t)g * 32, cur);
}
void launch_merkle_path(hipStream_t st, HashConsts hc, const uint8_t* leaf_be, const uint64_t* index, const uint8_t* siblings_be,
                        uint32_t depth, uint8_t* root_be, uint32_t count) {
  if (count) hipLaunchKernelGGL(k_merkle_path, dim3((count + 63) / 64), dim3(64), 0, st, hc, leaf_be, index, siblings_be, depth, root_be, count);
}

// one tree level: parents[j] = H(children[2j], children[2j+1]); a missing right/left child is the level's default hash
__global__ void __launch_bounds__(64) k_merkle_level(HashConsts hc, const Fr* __restrict__ children, uint32_t n_children, Fr dflt,
                                                     Fr* __restrict__ parents, uint32_t n_parents) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_parents) return;
  Fr l = 2 * g < n_children ? children[2 * g] : dflt;
  Fr r = 2 * g + 1 < n_children ? children[2 * g + 1] : dflt;
  parents[g] = poseidon_hash2(hc, l, r);
}
void launch_merkle_level(hipStream_t st, HashConsts hc, const Fr* children, uint32_t n_children, Fr dflt, Fr* parents, uint32_t n_parents) {
  if (n_parents) hipLaunchKernelGGL(k_merkle_level, dim3((n_parents + 63) / 64), dim3(64), 0, st, hc, children, n_children, dflt, parents, n_parents);
}
// ---- incremental tree (spp_merkle_tree_*): levels stay resident in HBM; an append touches O(count + depth) nodes ----
// default (empty-subtree) hashes: d_0 = 0, d_{l+1} = H(d_l, d_l)   (client/merkle.ts:150-156)
__global__ void __launch_bounds__(64) k_merkle_defaults(HashConsts hc, Fr* __restrict__ out, uint32_t depth) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Fr d = Fr::zero();
  out[0] = d;
  for (uint32_t l = 0; l < depth; l++) {
    d = poseidon_hash2(hc, d, d);
    out[l + 1] = d;
  }
}
// parents [first, first + n) of one level, recomputed from its children; children at or beyond n_children are the level default
__global__ void __launch_bounds__(64) k_merkle_update(HashConsts hc, const Fr* __restrict__ children, uint64_t n_children,
                                                      const Fr* __restrict__ dflt_level, Fr* __restrict__ parents, uint64_t first, uint32_t n) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const uint64_t j = first + g;
  const Fr d = *dflt_level;
  const Fr l = 2 * j < n_children ? children[2 * j] : d;
  const Fr r = 2 * j + 1 < n_children ? children[2 * j + 1] : d;
  parents[j] = poseidon_hash2(hc, l, r);
}
// getProof (client/merkle.ts:198-221) without any hashing: sibling of level l = stored node or the level default
__global__ void __launch_bounds__(64) k_merkle_gather(const MerkleTreeDev* __restrict__ t, const uint64_t* __restrict__ indices, uint32_t nq,
                                                      uint8_t* __restrict__ out_be) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t depth = t->depth;
  if (g >= nq * depth) return;
  const uint32_t q = g / depth, l = g % depth;
  const uint64_t sib = (indices[q] >> l) ^ 1;
  const Fr v = sib < t->count[l] ? t->level[l][sib] : t->dflt[l];
  store_be(out_be + ((size_t)q * depth + l) * 32, v);
}
void launch_merkle_defaults(hipStream_t st, HashConsts hc, Fr* out, uint32_t depth) {
  hipLaunchKernelGGL(k_merkle_defaults, dim3(1), dim3(64), 0, st, hc, out, depth);
}
void launch_merkle_update(hipStream_t st, HashConsts hc, const Fr* children, uint64_t n_children, const Fr* dflt_level, Fr* parents,
                          uint64_t first, uint32_t n) {
  if (n) hipLaunchKernelGGL(k_merkle_update, dim3((n + 63) / 64), dim3(64), 0, st, hc, children, n_children, dflt_level, parents, first, n);
}
void launch_merkle_gather(hipStream_t st, const MerkleTreeDev* t, uint32_t depth, const uint64_t* indices, uint32_t nq, uint8_t* out_be) {
  const uint32_t n = nq * depth;
  if (n) hipLaunchKernelGGL(k_merkle_gather, dim3((n + 63) / 64), dim3(64), 0, st, t, indices, nq, out_be);
}

__global__ void __launch_bounds__(256) k_fr_from_be(const uint8_t* __restrict__ in, Fr* __restrict__ out, uint32_t n) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) out[g] = load_be(in + (size_t)g * 32);
}
__global__ void __launch_bounds__(256) k_fr_to_be(const Fr* __restrict__ in, uint8_t* __restrict__ out, uint32_t n) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) store_be(out + (size_t)g * 32, in[g]);
}
void launch_fr_from_be(hipStream_t st, const uint8_t* in, Fr* out, uint32_t n) {
  if (n) hipLaunchKernelGGL(k_fr_from_be, dim3((n + 255) / 256), dim3(256), 0, st, in, out, n);
}
void launch_fr_to_be(hipStream_t st, const Fr* in, uint8_t* out, uint32_t n) {
  if (n) hipLaunchKernelGGL(k_fr_to_be, dim3((n + 255) / 256), dim3(256), 0, st, in, out, n);
}

// ----------------------------------------------------------------------------------------------------
// Grumpkin key generation: pk = sk * G with a 4-bit window table T[j][d] = (d+1) * 16^j * G (64 x 15 used)
// ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_grumpkin_keygen(const GkAffine* __restrict__ table, const uint8_t* __restrict__ sk_be,
                                                        uint8_t* __restrict__ xy_be, uint32_t count) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count) return;
  Fr sk = load_be(sk_be + (size_t)g * 32);
  uint32_t c[8];
  sk.to_canonical(c);
  GkXYZZ acc = GkXYZZ::infinity();
  uint32_t word = 0;
#pragma unroll 1
  for (uint32_t j = 0; j < 64; j++) {
    if ((j & 7) == 0) {
      const uint32_t li = j >> 3;
      word = li == 0 ? c[0] : li == 1 ? c[1] : li == 2 ? c[2] : li == 3 ? c[3] : li == 4 ? c[4] : li == 5 ? c[5] : li == 6 ? c[6] : c[7];
    }
    const uint32_t d = word & 15;
    word >>= 4;
    if (d) acc.madd(table[j * 16 + d - 1]);
  }
  GkAffine p = acc.to_affine();
  store_be(xy_be + (size_t)g * 64, p.x);
  store_be(xy_be + (size_t)g * 64 + 32, p.y);
}
void launch_grumpkin_keygen(hipStream_t st, const GkAffine* table, const uint8_t* sk_be, uint8_t* xy_be, uint32_t count) {
  if (count) hipLaunchKernelGGL(k_grumpkin_keygen, dim3((count + 63) / 64), dim3(64), 0, st, table, sk_be, xy_be, count);
}

// ----------------------------------------------------------------------------------------------------
// Poseidon2 t=4 sponge (rate 3) over n field elements per instance; one lane per instance
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void p2_external(Fr (&s)[4]) {
  Fr t01 = s[0] + s[1], t23 = s[2] + s[3];
  Fr d1 = s[1].dbl(), d3 = s[3].dbl();
  Fr q0 = s[0].dbl().dbl(), q1 = d1.dbl(), q2 = s[2].dbl().dbl(), q3 = d3.dbl();
  Fr n0 = q0 + s[0] + q1 + d1 + s[1] + s[2] + d3 + s[3];
  Fr n1 = q0 + q1 + d1 + t23;
  Fr n2 = s[0] + d1 + s[1] + q2 + s[2] + q3 + d3 + s[3];
  Fr n3 = t01 + q2 + q3 + d3;
  s[0] = n0; s[1] = n1; s[2] = n2; s[3] = n3;
}
__device__ __noinline__ void poseidon2_permute(Fr (&s)[4], const Fr* __restrict__ rc, const Fr* __restrict__ mu) {
  p2_external(s);
  int k = 0;
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    SPP_UNROLL for (int i = 0; i < 4; i++) s[i] = sbox5(s[i] + rc[k + i]);
    k += 4;
    p2_external(s);
  }
#pragma unroll 1
  for (int r = 0; r < 56; r++) {
    s[0] = sbox5(s[0] + rc[k]);
    k++;
    Fr tot = s[0] + s[1] + s[2] + s[3];
    SPP_UNROLL for (int i = 0; i < 4; i++) s[i] = mu[i] * s[i] + tot;
  }
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    SPP_UNROLL for (int i = 0; i < 4; i++) s[i] = sbox5(s[i] + rc[k + i]);
    k += 4;
    p2_external(s);
  }
}
__global__ void __launch_bounds__(64) k_poseidon2_sponge(HashConsts hc, const uint8_t* __restrict__ in_be, uint32_t n,
                                                         uint8_t* __restrict__ out_be, uint32_t count) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count) return;
  const uint8_t* in = in_be + (size_t)g * n * 32;
  Fr s[4] = {Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero()};
  const uint32_t full = n / 3;
  for (uint32_t i = 0; i < full; i++) {
    s[0] = s[0] + load_be(in + (size_t)(3 * i) * 32);
    s[1] = s[1] + load_be(in + (size_t)(3 * i + 1) * 32);
    s[2] = s[2] + load_be(in + (size_t)(3 * i + 2) * 32);
    poseidon2_permute(s, hc.p2_rc, hc.p2_mu);
  }
  const uint32_t rem = n - 3 * full;
  if (rem >= 1) s[0] = s[0] + load_be(in + (size_t)(3 * full) * 32);
  if (rem >= 2) s[1] = s[1] + load_be(in + (size_t)(3 * full + 1) * 32);
  poseidon2_permute(s, hc.p2_rc, hc.p2_mu);
  store_be(out_be + (size_t)g * 32, s[0]);
}
// Small counts (one audit proof through generateAuditProof, the tail of a batch): one WAVE per instance, the permutation in the
// lane-parallel form the cooperative solver uses (lanes.hpp: three dependent products per round instead of eight) -- 53 chained
// permutations are what a single instance waits for.  A wave spends 64 lanes' worth of issue slots on one instance, so large
// counts keep one lane per instance: 2 048 instances are 32 waves there (latency-bound, hidden behind the proving streams) and
// would be 2 048 waves x 53 permutations of chip time here.
__global__ void __launch_bounds__(64) k_poseidon2_sponge_coop(HashConsts hc, const uint8_t* __restrict__ in_be, uint32_t n,
                                                              uint8_t* __restrict__ out_be, uint32_t count) {
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  if (g >= count) return;
  const uint8_t* in = in_be + (size_t)g * n * 32;
  auto no_emit = [](uint32_t, const Fr&, const Fr&, const Fr&, const Fr&) {};
  Fr s = Fr::zero();
  const uint32_t full = n / 3, rem = n - 3 * full;
#pragma unroll 1
  for (uint32_t i = 0; i < full; i++) {
    if (lane < 3) s = s + load_be(in + (size_t)(3 * i + lane) * 32);
    s = coop_p2_permute(hc.p2_rc, hc.p2_mu, s, lane, no_emit);
  }
  if (lane < rem) s = s + load_be(in + (size_t)(3 * full + lane) * 32);
  s = coop_p2_permute(hc.p2_rc, hc.p2_mu, s, lane, no_emit);
  if (lane == 0) store_be(out_be + (size_t)g * 32, s);
}
void launch_poseidon2_sponge(hipStream_t st, HashConsts hc, const uint8_t* in_be, uint32_t n, uint8_t* out_be, uint32_t count) {
  if (count == 0) return;
  if (count <= 256) hipLaunchKernelGGL(k_poseidon2_sponge_coop, dim3(count), dim3(64), 0, st, hc, in_be, n, out_be, count);
  else hipLaunchKernelGGL(k_poseidon2_sponge, dim3((count + 63) / 64), dim3(64), 0, st, hc, in_be, n, out_be, count);
}


// ----------------------------------------------------------------------------------------------------
// Auditor side: scripts/rlwe_decrypt.py:61-132, demo-frontend/app/lib/shamir.ts:97-169
// ----------------------------------------------------------------------------------------------------
// msg[i] = round(centered((c0 + sk*c1 mod (X^n+1, q))[i]) / Delta) mod 256 for the 64 message slots; one 64-lane block per
// ciphertext, lane i owns slot i:  (sk*c1)[i] = sum_j SK2[(i - j) mod 2048] * c1[j],  SK2 = [sk, (q - sk) mod q].
// Products are 28 x 28 bits: the 64-bit accumulator is folded mod q every 128 terms.
__global__ void __launch_bounds__(64) k_rlwe_decrypt(const uint32_t* __restrict__ sk_mod_q, const uint32_t* __restrict__ c0,
                                                     const uint32_t* __restrict__ c1, uint8_t* __restrict__ msg, uint32_t count) {
  __shared__ uint32_t SK2[2 * RL_N];
  __shared__ uint32_t cs[RL_N];
  const uint32_t inst = blockIdx.x, t = threadIdx.x;
  if (inst >= count) return;
  for (int i = t; i < RL_N; i += 64) {
    const uint32_t s = sk_mod_q[i];
    SK2[i] = s;
    SK2[RL_N + i] = s ? (uint32_t)RL_Q - s : 0u;
    cs[i] = c1[(size_t)inst * RL_N + i];
  }
  __syncthreads();
  unsigned long long acc = 0, total = 0;
  for (int j = 0; j < RL_N; j++) {
    acc += (unsigned long long)SK2[(t - j) & 2047] * cs[j];
    if ((j & 127) == 127) { total += acc % (unsigned long long)RL_Q; acc = 0; }
  }
  const long long skc1 = (long long)(total % (unsigned long long)RL_Q);
  long long noisy = ((long long)c0[(size_t)inst * RL_SLOTS + t] + skc1) % RL_Q;
  if (noisy > RL_Q / 2) noisy -= RL_Q;                       // centered_mod (rlwe_decrypt.py:54-58)
  long long k = noisy / RL_DELTA, rem = noisy % RL_DELTA;    // floor division
  if (rem < 0) { rem += RL_DELTA; k -= 1; }
  if (2 * rem > RL_DELTA || (2 * rem == RL_DELTA && (k & 1))) k += 1;   // Python round(): half to even
  msg[(size_t)inst * RL_SLOTS + t] = (uint8_t)(((k % 256) + 256) % 256);
}
void launch_rlwe_decrypt(hipStream_t st, const uint32_t* sk_mod_q, const uint32_t* c0, const uint32_t* c1, uint8_t* msg, uint32_t count) {
  if (count) hipLaunchKernelGGL(k_rlwe_decrypt, dim3(count), dim3(64), 0, st, sk_mod_q, c0, c1, msg, count);
}

// Shamir reconstruction at 0: out[k] = sum_i lambda_i * y[i][k] over Fr, then centred and reduced mod q (shamir.ts:97-120)
__global__ void __launch_bounds__(256) k_shamir_combine(const Fr* __restrict__ lambda, const uint8_t* __restrict__ ys_be, uint32_t t,
                                                        uint32_t n, uint8_t* __restrict__ secret_be, uint32_t* __restrict__ sk_mod_q) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  Fr acc = Fr::zero();
  for (uint32_t i = 0; i < t; i++) acc = acc + lambda[i] * load_be(ys_be + ((size_t)i * n + k) * 32);
  if (secret_be) store_be(secret_be + (size_t)k * 32, acc);
  if (sk_mod_q) {
    uint32_t c[8];
    acc.to_canonical(c);
    // centred value must be small (|v| < 2^31) for a key coefficient; larger values are reduced limb-wise
    const bool neg = canonical_gt_half<FrParams>(c);
    uint32_t m[8];
    if (neg) canonical_negate<FrParams>(c, m);
    else { SPP_UNROLL for (int i = 0; i < 8; i++) m[i] = c[i]; }
    unsigned long long r = 0;   // |v| mod q by Horner over the limbs
    for (int i = 7; i >= 0; i--) r = ((r << 32) | m[i]) % (unsigned long long)RL_Q;
    sk_mod_q[k] = neg ? (r ? (uint32_t)(RL_Q - (long long)r) : 0u) : (uint32_t)r;
  }
}
void launch_shamir_combine(hipStream_t st, const Fr* lambda, const uint8_t* ys_be, uint32_t t, uint32_t n, uint8_t* secret_be,
                           uint32_t* sk_mod_q) {
  if (n) hipLaunchKernelGGL(k_shamir_combine, dim3((n + 255) / 256), dim3(256), 0, st, lambda, ys_be, t, n, secret_be, sk_mod_q);
}

// ----------------------------------------------------------------------------------------------------
// audit input assembly: (sk, r, e1, e2) -> the 3360-field input row of the audit circuit, all on the device
// (scripts/generate_audit.py:468-641: keygen, message slots, encryption + quotients, packing, commitments, Prover.toml)
// ----------------------------------------------------------------------------------------------------
// message = little-endian bytes of owner_x then owner_y (generate_audit.py:489-496); xy_be is 64 B big-endian per key
__global__ void __launch_bounds__(256) k_audit_msg(const uint8_t* __restrict__ xy_be, uint8_t* __restrict__ msg, uint32_t count) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count * 64) return;
  const uint32_t inst = g / 64, s = g % 64;
  msg[g] = xy_be[(size_t)inst * 64 + (s < 32 ? 31 - s : 32 + 31 - (s - 32))];
}
__device__ __forceinline__ void store_small_signed_be(uint8_t* o, int32_t v) {
  // field encoding of a small signed integer (format_field, generate_audit.py:77-82): v >= 0 -> v, v < 0 -> r - |v|
  const uint8_t RBE[32] = {0x30, 0x64, 0x4e, 0x72, 0xe1, 0x31, 0xa0, 0x29, 0xb8, 0x50, 0x45, 0xb6, 0x81, 0x81, 0x58, 0x5d,
                           0x28, 0x33, 0xe8, 0x48, 0x79, 0xb9, 0x70, 0x91, 0x43, 0xe1, 0xf5, 0x93, 0xf0, 0x00, 0x00, 0x01};
  uint32_t low;
  if (v >= 0) {
    for (int i = 0; i < 28; i++) o[i] = 0;
    low = (uint32_t)v;
  } else {
    for (int i = 0; i < 28; i++) o[i] = RBE[i];
    low = 0xf0000001u - (uint32_t)(-(int64_t)v);   // |v| <= 2^31 < 0xf0000001: no borrow into the upper limbs
  }
  o[28] = (uint8_t)(low >> 24); o[29] = (uint8_t)(low >> 16); o[30] = (uint8_t)(low >> 8); o[31] = (uint8_t)low;
}
static constexpr uint32_t AUDIT_NIN = 3360;
__global__ void __launch_bounds__(256) k_audit_assemble(const uint8_t* __restrict__ wa_be, const uint8_t* __restrict__ ct_be,
                                                        const uint8_t* __restrict__ packed_be, const uint8_t* __restrict__ sk_be,
                                                        const int8_t* __restrict__ r, const int8_t* __restrict__ e1, const int8_t* __restrict__ e2,
                                                        const int32_t* __restrict__ k0, const int32_t* __restrict__ k1,
                                                        uint8_t* __restrict__ rows, uint32_t count) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)count * AUDIT_NIN) return;
  const uint32_t inst = (uint32_t)(g / AUDIT_NIN), f = (uint32_t)(g % AUDIT_NIN);
  uint8_t* o = rows + g * 32;
  const uint8_t* src = nullptr;
  if (f == 0) src = wa_be + (size_t)inst * 32;
  else if (f == 1) src = ct_be + (size_t)inst * 32;
  else if (f < 2 + 157) src = packed_be + ((size_t)inst * 157 + (f - 2)) * 32;
  else if (f == 159) src = sk_be + (size_t)inst * 32;
  if (src) {
    for (int i = 0; i < 32; i++) o[i] = src[i];
    return;
  }
  uint32_t k = f - 160;
  int32_t v;
  if (k < 1024) v = r[(size_t)inst * 1024 + k];
  else if ((k -= 1024) < 64) v = e1[(size_t)inst * 64 + k];
  else if ((k -= 64) < 1024) v = e2[(size_t)inst * 1024 + k];
  else if ((k -= 1024) < 64) v = k0[(size_t)inst * 64 + k];
  else v = k1[(size_t)inst * 1024 + (k - 64)];
  store_small_signed_be(o, v);
}
void launch_audit_msg(hipStream_t st, const uint8_t* xy_be, uint8_t* msg, uint32_t count) {
  if (count) hipLaunchKernelGGL(k_audit_msg, dim3((count * 64 + 255) / 256), dim3(256), 0, st, xy_be, msg, count);
}
void launch_audit_assemble(hipStream_t st, const uint8_t* wa_be, const uint8_t* ct_be, const uint8_t* packed_be, const uint8_t* sk_be,
                           const int8_t* r, const int8_t* e1, const int8_t* e2, const int32_t* k0, const int32_t* k1, uint8_t* rows,
                           uint32_t count) {
  uint64_t lanes = (uint64_t)count * AUDIT_NIN;
  if (count) hipLaunchKernelGGL(k_audit_assemble, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, st, wa_be, ct_be, packed_be, sk_be, r, e1, e2, k0, k1, rows, count);
}

}  // namespace spp

// Internals shared by the translation units of libspp's C ABI (spp_api.cpp: contexts, circuits, setup, proving;
// spp_witness_api.cpp: witness-input kernels, Merkle trees, auditor side; spp_verify_api.cpp: verification and pairing checks;
// spp_micro_api.cpp: the NTT / MSM / Pippenger unit and micro-benchmark entry points).  Not installed; include/spp.h is the ABI.
#pragma once
#include "../../include/spp.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "circuit.hpp"
#include "kernels.hpp"
#include "sha256.hpp"
#include "f29.hpp"
#include "pairing.hpp"
#include "pairing_fast_host.hpp"

using namespace spp;

// -----------------------------------------------------------------------------------------------------
// errors: negative SPP_ERR_* codes + a thread-local message (spp_last_error)
// -----------------------------------------------------------------------------------------------------
extern thread_local char g_spp_err[512];
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_spp_err, sizeof g_spp_err, fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return fail(SPP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));     \
  } while (0)

// -----------------------------------------------------------------------------------------------------
// host helpers
// -----------------------------------------------------------------------------------------------------
inline Fr fr_pow_limbs(const Fr& base, const uint32_t e[8]) {
  Fr acc = Fr::one(), b = base;
  for (int w = 0; w < 8; w++)
    for (int i = 0; i < 32; i++) {
      if ((e[w] >> i) & 1) acc = acc * b;
      b = b.sqr();
    }
  return acc;
}
inline Fr fr_root_of_unity(uint32_t logn) {
  uint32_t e[8];
  for (int i = 0; i < 8; i++) e[i] = FrParams::MOD(i);
  e[0] -= 1;
  for (uint32_t s = 0; s < logn; s++) {
    for (int i = 0; i < 7; i++) e[i] = (e[i] >> 1) | (e[i + 1] << 31);
    e[7] >>= 1;
  }
  return fr_pow_limbs(Fr::from_u64(5), e);
}
inline uint32_t bitrev(uint32_t v, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
  return r;
}
inline G1Affine g1_from_raw(const uint8_t* b) {
  bool z = true;
  for (int i = 0; i < 64; i++) z &= b[i] == 0;
  if (z) return G1Affine::infinity();
  return {Fq::from_bytes_be(b), Fq::from_bytes_be(b + 32)};
}
inline G2Affine g2_from_raw(const uint8_t* b) {
  bool z = true;
  for (int i = 0; i < 128; i++) z &= b[i] == 0;
  if (z) return G2Affine::infinity();
  G2Affine p;
  p.x.c1 = Fq::from_bytes_be(b);
  p.x.c0 = Fq::from_bytes_be(b + 32);
  p.y.c1 = Fq::from_bytes_be(b + 64);
  p.y.c0 = Fq::from_bytes_be(b + 96);
  return p;
}
inline void g1_to_raw(const G1Affine& p, uint8_t* b) {
  if (p.is_inf()) { memset(b, 0, 64); return; }
  p.x.to_bytes_be(b);
  p.y.to_bytes_be(b + 32);
}
inline void g2_to_raw(const G2Affine& p, uint8_t* b) {
  if (p.is_inf()) { memset(b, 0, 128); return; }
  p.x.c1.to_bytes_be(b);
  p.x.c0.to_bytes_be(b + 32);
  p.y.c1.to_bytes_be(b + 64);
  p.y.c0.to_bytes_be(b + 96);
}
template <class F>
inline Affine<F> host_add(const Affine<F>& a, const Affine<F>& b) {
  XYZZ<F> x = XYZZ<F>::from_affine(a);
  x.madd(b);
  return x.to_affine();
}
inline bool read_file(const char* path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize((size_t)sz);
  bool ok = fread(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok;
}

template <class T>
inline hipError_t dev_upload(T** dst, const std::vector<T>& src) {
  *dst = nullptr;
  size_t bytes = sizeof(T) * std::max<size_t>(src.size(), 1);
  hipError_t e = hipMalloc((void**)dst, bytes);
  if (e != hipSuccess) return e;
  if (!src.empty()) e = hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice);
  return e;
}

// -----------------------------------------------------------------------------------------------------
// context / circuit objects
// -----------------------------------------------------------------------------------------------------
static constexpr int SPP_NWS = 6;   // batch workspaces / proving streams per circuit (big batches use two)
struct spp_ctx {
  int device;
  hipStream_t stream;        // setup / table construction
  hipStream_t pstream[SPP_NWS];   // proving: consecutive batches take the streams in turn, so the (latency-bound, few-wave)
                                  // witness solver of batch k+1 overlaps the MSMs of batch k; small batches use up to six
  std::mutex mu;
  // lazily created constants of the stand-alone witness kernels
  bool consts_ready = false;
  HashConsts hc{};
  GkAffine* gk_table = nullptr;
  bool rlwe_ready = false;
  RlweDev rlwe{};              // NTT tables of the RLWE witness kernel (rlwe_ntt.hpp)
  // temporaries of spp_audit_inputs_batch(_device), kept between calls (grow only): a hipFree per call would drain every
  // stream of the device, the proving streams of the batch in flight included
  void* audit_scratch = nullptr;
  size_t audit_scratch_cap = 0;
  std::vector<void*> owned;
};
template <class T>
inline int ctx_upload(spp_ctx* ctx, T** dst, const std::vector<T>& src) {
  HIP_TRY(dev_upload(dst, src));
  ctx->owned.push_back((void*)*dst);
  return 0;
}

// device buffer freed on every return path
namespace {
// RAII device buffer for the host-pointer convenience entry points
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <class T> T* as() { return (T*)p; }
};
}  // namespace
#define UP(buf, src, bytes)                                                              \
  do {                                                                                    \
    HIP_TRY(buf.alloc(bytes));                                                            \
    if (bytes) HIP_TRY(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, st));      \
  } while (0)


// the audit input pipeline as stream-ordered work (spp_witness_api.cpp); used by spp_prove_audit_from_secrets_device
size_t spp_audit_scratch_bytes(size_t count);
int spp_audit_inputs_enqueue(spp_ctx* ctx, hipStream_t st, void* scratch, const uint32_t* d_pk_a, const uint32_t* d_pk_b, uint32_t count,
                             const uint8_t* d_sk, const int8_t* d_r, const int8_t* d_e1, const int8_t* d_e2, uint8_t* d_rows);
// lazily built per-context constants (spp_witness_api.cpp)
int spp_ensure_ctx_consts(spp_ctx* ctx);   // Poseidon / Poseidon2 constants, Grumpkin window table
int spp_ensure_rlwe(spp_ctx* ctx);         // NTT tables of the RLWE witness kernel

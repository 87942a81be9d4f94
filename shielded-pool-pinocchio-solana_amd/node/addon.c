/* N-API binding of libspp's C ABI (include/spp.h) for the TypeScript host code.
 * Replaces the two execSync() calls of the reference's client/proof.helper.ts:55,64 with in-process calls.
 * Synchronous, like the reference; errors become thrown JS Errors whose message carries the SPP_ERR_* code. */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/spp.h"

#define NAPI_OK(call)                                            \
  do {                                                           \
    if ((call) != napi_ok) {                                     \
      napi_throw_error(env, NULL, "N-API call failed: " #call);  \
      return NULL;                                               \
    }                                                            \
  } while (0)

static napi_value throw_spp(napi_env env, int code) {
  char msg[640];
  snprintf(msg, sizeof msg, "libspp error %d: %s", code, spp_last_error());
  napi_throw_error(env, NULL, msg);
  return NULL;
}
static char* get_string(napi_env env, napi_value v) {
  size_t len = 0;
  if (napi_get_value_string_utf8(env, v, NULL, 0, &len) != napi_ok) return NULL;
  char* s = (char*)malloc(len + 1);
  napi_get_value_string_utf8(env, v, s, len + 1, &len);
  return s;
}

static spp_ctx* g_ctx = NULL;

/* init(device) */
static napi_value Init(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  int32_t dev = 0;
  if (argc >= 1) napi_get_value_int32(env, argv[0], &dev);
  if (!g_ctx) {
    int rc = spp_init(dev, &g_ctx);
    if (rc) return throw_spp(env, rc);
  }
  napi_value r;
  napi_get_boolean(env, 1, &r);
  return r;
}
/* buildCircuit(id, outPath, auxUint32Array?) -> nbConstraints */
static napi_value BuildCircuit(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  int32_t id = 0;
  napi_get_value_int32(env, argv[0], &id);
  char* path = get_string(env, argv[1]);
  const uint32_t* aux = NULL;
  if (argc >= 3) {
    bool is_ta = false;
    napi_is_typedarray(env, argv[2], &is_ta);
    if (is_ta) {
      napi_typedarray_type t;
      size_t len;
      void* data;
      napi_get_typedarray_info(env, argv[2], &t, &len, &data, NULL, NULL);
      if (t == napi_uint32_array && len == 2048) aux = (const uint32_t*)data;
    }
  }
  uint32_t n = 0;
  int rc = spp_circuit_build(id, aux, path, &n);
  free(path);
  if (rc) return throw_spp(env, rc);
  napi_value r;
  napi_create_uint32(env, n, &r);
  return r;
}
/* setup(circuitPath, seed32 Buffer, pkPath, vkPath) */
static napi_value Setup(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!g_ctx) { napi_throw_error(env, NULL, "call init() first"); return NULL; }
  char* c = get_string(env, argv[0]);
  void* seed; size_t slen;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &seed, &slen));
  char* pk = get_string(env, argv[2]);
  char* vk = get_string(env, argv[3]);
  int rc = slen == 32 ? spp_setup(g_ctx, c, (const uint8_t*)seed, pk, vk) : SPP_ERR_BAD_INPUT;
  free(c); free(pk); free(vk);
  if (rc) return throw_spp(env, rc);
  return NULL;
}
static void finalize_circuit(napi_env env, void* data, void* hint) { (void)env; (void)hint; spp_free_circuit((spp_circuit*)data); }
/* loadCircuit(circuitPath, pkPath, windowBits) -> handle */
static napi_value LoadCircuit(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!g_ctx) { napi_throw_error(env, NULL, "call init() first"); return NULL; }
  char* c = get_string(env, argv[0]);
  char* pk = get_string(env, argv[1]);
  int32_t wb = 0;
  if (argc >= 3) napi_get_value_int32(env, argv[2], &wb);
  spp_circuit* h = NULL;
  int rc = spp_load_circuit(g_ctx, c, pk, wb, &h);
  free(c); free(pk);
  if (rc) return throw_spp(env, rc);
  napi_value ext;
  NAPI_OK(napi_create_external(env, h, finalize_circuit, NULL, &ext));
  return ext;
}
/* circuitInfo(handle) -> Uint32 array of 8 */
static napi_value CircuitInfo(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void* h;
  NAPI_OK(napi_get_value_external(env, argv[0], &h));
  uint32_t inf[8];
  int rc = spp_circuit_info((spp_circuit*)h, inf);
  if (rc) return throw_spp(env, rc);
  napi_value arr;
  napi_create_array_with_length(env, 8, &arr);
  for (uint32_t i = 0; i < 8; i++) { napi_value v; napi_create_uint32(env, inf[i], &v); napi_set_element(env, arr, i, v); }
  return arr;
}
/* proveBatch(handle, count, inputs Buffer, rs Buffer|null) -> {proofs, publicWitnesses, status} */
static napi_value ProveBatch(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void* h;
  NAPI_OK(napi_get_value_external(env, argv[0], &h));
  uint32_t count = 0;
  napi_get_value_uint32(env, argv[1], &count);
  void* in; size_t inlen;
  NAPI_OK(napi_get_buffer_info(env, argv[2], &in, &inlen));
  void* rs = NULL; size_t rslen = 0;
  bool isbuf = false;
  if (argc >= 4) napi_is_buffer(env, argv[3], &isbuf);
  if (isbuf) NAPI_OK(napi_get_buffer_info(env, argv[3], &rs, &rslen));
  uint32_t inf[8];
  spp_circuit_info((spp_circuit*)h, inf);
  size_t pwlen = 12 + 32 * (size_t)inf[1];
  if (inlen != (size_t)count * inf[6] * 32 || (rs && rslen != (size_t)count * 64)) {
    napi_throw_error(env, NULL, "libspp error -1: input buffer has the wrong length");
    return NULL;
  }
  void *proofs, *pws;
  napi_value bproofs, bpws, st, out;
  NAPI_OK(napi_create_buffer(env, (size_t)count * SPP_PROOF_LEN, &proofs, &bproofs));
  NAPI_OK(napi_create_buffer(env, (size_t)count * pwlen, &pws, &bpws));
  int32_t* status = (int32_t*)calloc(count ? count : 1, sizeof(int32_t));
  int rc = spp_prove_batch((spp_circuit*)h, count, (const uint8_t*)in, (const uint8_t*)rs, (uint8_t*)proofs, (uint8_t*)pws, status);
  if (rc && rc != SPP_ERR_UNSAT) { free(status); return throw_spp(env, rc); }
  napi_create_array_with_length(env, count, &st);
  for (uint32_t i = 0; i < count; i++) { napi_value v; napi_create_int32(env, status[i], &v); napi_set_element(env, st, i, v); }
  free(status);
  napi_create_object(env, &out);
  napi_set_named_property(env, out, "proofs", bproofs);
  napi_set_named_property(env, out, "publicWitnesses", bpws);
  napi_set_named_property(env, out, "status", st);
  return out;
}
/* verify(vk Buffer, proof Buffer, publicWitness Buffer) -> boolean   (`sunspot verify`, prove_linux.sh:86-87; host only) */
static napi_value Verify(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void *vk, *pr, *pw;
  size_t vkl, prl, pwl;
  NAPI_OK(napi_get_buffer_info(env, argv[0], &vk, &vkl));
  NAPI_OK(napi_get_buffer_info(env, argv[1], &pr, &prl));
  NAPI_OK(napi_get_buffer_info(env, argv[2], &pw, &pwl));
  int ok = 0;
  int rc = spp_verify((const uint8_t*)vk, vkl, (const uint8_t*)pr, prl, (const uint8_t*)pw, pwl, &ok);
  if (rc) return throw_spp(env, rc);
  napi_value r;
  napi_get_boolean(env, ok != 0, &r);
  return r;
}
/* verifyBatch(vk Buffer, count, proofs Buffer, publicWitnesses Buffer) -> boolean[]   (after init(); one GPU lane per proof) */
static napi_value VerifyBatch(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (!g_ctx) { napi_throw_error(env, NULL, "call init() first"); return NULL; }
  void* ctx = g_ctx;
  void *vk, *pr, *pw;
  size_t vkl, prl, pwl;
  NAPI_OK(napi_get_buffer_info(env, argv[0], &vk, &vkl));
  uint32_t count = 0;
  napi_get_value_uint32(env, argv[1], &count);
  NAPI_OK(napi_get_buffer_info(env, argv[2], &pr, &prl));
  NAPI_OK(napi_get_buffer_info(env, argv[3], &pw, &pwl));
  if (prl != (size_t)count * SPP_PROOF_LEN || (count && pwl % count)) {
    napi_throw_error(env, NULL, "libspp error -1: proof / public-witness buffer has the wrong length");
    return NULL;
  }
  int32_t* ok = (int32_t*)calloc(count ? count : 1, sizeof(int32_t));
  int rc = spp_verify_batch((spp_ctx*)ctx, (const uint8_t*)vk, vkl, count, (const uint8_t*)pr, (const uint8_t*)pw, count ? pwl / count : 12, ok,
                            NULL);
  if (rc) { free(ok); return throw_spp(env, rc); }
  napi_value arr;
  napi_create_array_with_length(env, count, &arr);
  for (uint32_t i = 0; i < count; i++) { napi_value v; napi_get_boolean(env, ok[i] != 0, &v); napi_set_element(env, arr, i, v); }
  free(ok);
  return arr;
}
static napi_value Version(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value r;
  napi_create_string_utf8(env, spp_version(), NAPI_AUTO_LENGTH, &r);
  return r;
}

static napi_value ModuleInit(napi_env env, napi_value exports) {
  napi_property_descriptor d[] = {
      {"init", NULL, Init, NULL, NULL, NULL, napi_default, NULL},
      {"buildCircuit", NULL, BuildCircuit, NULL, NULL, NULL, napi_default, NULL},
      {"setup", NULL, Setup, NULL, NULL, NULL, napi_default, NULL},
      {"loadCircuit", NULL, LoadCircuit, NULL, NULL, NULL, napi_default, NULL},
      {"circuitInfo", NULL, CircuitInfo, NULL, NULL, NULL, napi_default, NULL},
      {"proveBatch", NULL, ProveBatch, NULL, NULL, NULL, napi_default, NULL},
      {"verify", NULL, Verify, NULL, NULL, NULL, napi_default, NULL},
      {"verifyBatch", NULL, VerifyBatch, NULL, NULL, NULL, napi_default, NULL},
      {"version", NULL, Version, NULL, NULL, NULL, napi_default, NULL},
  };
  napi_define_properties(env, exports, sizeof d / sizeof d[0], d);
  return exports;
}
NAPI_MODULE(spp_addon, ModuleInit)

// kmx_abi_on_oracle.cpp — the part of the katamx C ABI (include/katamx.h) that the reference-side binding calls, implemented
// on the CPU ORACLE (oracle/kmx_oracle.c).  TEST INFRASTRUCTURE ONLY.
//
// The binding (integration/katamxbackend.cpp, with this repo's NNEvaluator integration/katamx_nneval.cpp on top) is the
// translation unit a KataGo maintainer adds; it knows the C ABI and nothing else. Linked against libkatamx.so it is the MI355X
// backend (oracle/_ref/katago_hip, katago_hipx). Linked against THIS file and kmx_oracle.o instead it becomes
// oracle/_ref/katago_oracle / katago_oraclex: the reference's own host code and known-answer tests running on the oracle -
// which is how the oracle is pinned (tests/test_oracle_pinned.py) and how the evaluator's and featuriser's host logic is
// tested without a GPU (tests/test_nneval_own.py). Until round 4 the binding carried 29 #ifdef KMX_USE_ORACLE blocks for this;
// now the same object file serves both builds and the switch is which implementation of the ABI the linker is given.
//
// Semantics kept from the #ifdef build: fp32 arithmetic only (kmx_handle_precision / kmx_batcher_precision report
// KMX_PREC_FP32, the layer hooks refuse any other precision_mode with KMX_ERR_UNSUPPORTED - the reference's tests then
// skip their fp16 variants); a "batcher" evaluates each row synchronously inside submit (one row = one batch), wait returns
// at once. Only the entry points the binding uses are defined; nothing here is exported from libkatamx.so.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kmx_oracle.h"

struct kmx_model {
  okmx_model* m = nullptr;
};
struct kmx_context {
  int nnXLen = 0, nnYLen = 0;
};
struct kmx_handle {
  const kmx_context* ctx = nullptr;
  const kmx_model* model = nullptr;
  int maxBatch = 0;
  kmx_model_info info;
};
struct kmx_batcher {
  const kmx_context* ctx = nullptr;
  const kmx_model* model = nullptr;
  kmx_model_info info;
  int maxBatch = 0;
  std::atomic<uint64_t> rows{0};
};

namespace {
thread_local std::string tlsError;
int fail(int code, const std::string& what) {
  tlsError = what;
  return code;
}
// a failing okmx_* call has set the oracle's own message
int passOn(int status) {
  if(status != KMX_OK) tlsError = okmx_last_error();
  return status;
}
}  // namespace

extern "C" {

int kmx_abi_version(void) { return KMX_ABI_VERSION; }
int kmx_global_init(void) { return KMX_OK; }
void kmx_global_cleanup(void) {}
int kmx_device_count(void) { return 1; }
int kmx_device_name(int device, char* buf, size_t buflen) {
  if(device != 0 || buf == nullptr || buflen == 0) return fail(KMX_ERR_INVALID_ARG, "kmx_device_name: bad arguments");
  snprintf(buf, buflen, "CPU oracle standing in for the device (test build)");
  return KMX_OK;
}
const char* kmx_last_error(void) { return tlsError.c_str(); }

int kmx_model_load(const char* path, const char* expected_sha256, kmx_model** out) {
  if(out == nullptr) return fail(KMX_ERR_INVALID_ARG, "kmx_model_load: null output pointer");
  kmx_model* m = new(std::nothrow) kmx_model();
  if(m == nullptr) return fail(KMX_ERR_INTERNAL, "out of memory");
  const int rc = okmx_model_load(path, expected_sha256, &m->m);
  if(rc != KMX_OK) {
    delete m;
    return passOn(rc);
  }
  *out = m;
  return KMX_OK;
}
void kmx_model_free(kmx_model* model) {
  if(model == nullptr) return;
  okmx_model_free(model->m);
  delete model;
}
int kmx_model_info_get(const kmx_model* model, kmx_model_info* out) {
  if(model == nullptr || out == nullptr) return fail(KMX_ERR_INVALID_ARG, "kmx_model_info_get: null argument");
  return passOn(okmx_model_info_get(model->m, out));
}

int kmx_context_create(const int*, int, int nn_x_len, int nn_y_len, int, kmx_context** out) {
  if(out == nullptr || nn_x_len < 2 || nn_y_len < 2 || nn_x_len > 19 || nn_y_len > 19)
    return fail(KMX_ERR_INVALID_ARG, "kmx_context_create: nnXLen/nnYLen must be in 2..19");
  kmx_context* c = new kmx_context();
  c->nnXLen = nn_x_len;
  c->nnYLen = nn_y_len;
  *out = c;
  return KMX_OK;
}
void kmx_context_free(kmx_context* ctx) { delete ctx; }

int kmx_handle_create(kmx_context* ctx, const kmx_model* model, int max_batch_size, int, int, kmx_handle** out) {
  if(ctx == nullptr || model == nullptr || out == nullptr || max_batch_size < 1) return fail(KMX_ERR_INVALID_ARG, "kmx_handle_create: bad arguments");
  kmx_handle* h = new kmx_handle();
  h->ctx = ctx;
  h->model = model;
  h->maxBatch = max_batch_size;
  const int rc = okmx_model_info_get(model->m, &h->info);
  if(rc != KMX_OK) {
    delete h;
    return passOn(rc);
  }
  *out = h;
  return KMX_OK;
}
void kmx_handle_free(kmx_handle* handle) { delete handle; }
int kmx_handle_precision(const kmx_handle*) { return KMX_PREC_FP32; }

int kmx_eval_meta(kmx_handle* h, int n_rows, const float* const* row_spatial, const float* const* row_global, const float* const* row_meta,
                  const int* symmetry, const float* policy_optimism, float* const* out_policy, float* out_value, float* out_score,
                  float* const* out_ownership) {
  if(h == nullptr || n_rows < 1 || n_rows > h->maxBatch) return fail(KMX_ERR_INVALID_ARG, "kmx_eval_meta: batch size out of range for this handle");
  return passOn(okmx_eval_meta(h->model->m, h->ctx->nnXLen, h->ctx->nnYLen, n_rows, row_spatial, row_global, row_meta, symmetry, policy_optimism,
                               out_policy, out_value, out_score, out_ownership, 1));
}

int kmx_batcher_create(kmx_context* ctx, const kmx_model* model, int max_batch_size, int, int, kmx_batcher** out) {
  if(ctx == nullptr || model == nullptr || out == nullptr || max_batch_size < 1) return fail(KMX_ERR_INVALID_ARG, "kmx_batcher_create: bad arguments");
  kmx_batcher* b = new kmx_batcher();
  b->ctx = ctx;
  b->model = model;
  b->maxBatch = max_batch_size;
  const int rc = okmx_model_info_get(model->m, &b->info);
  if(rc != KMX_OK) {
    delete b;
    return passOn(rc);
  }
  *out = b;
  return KMX_OK;
}
void kmx_batcher_free(kmx_batcher* batcher) { delete batcher; }
int kmx_batcher_precision(const kmx_batcher*) { return KMX_PREC_FP32; }
// (no device, no granule: what was asked for is what a batch may hold; include/katamx.h, ABI 7)
int kmx_batcher_effective_batch(const kmx_batcher* b) { return b ? b->maxBatch : 0; }

int kmx_batcher_submit(kmx_batcher* b, const float* row_spatial, const float* row_global, const float* row_meta, int symmetry,
                       float policy_optimism, float* out_policy, float* out_value, float* out_score, float* out_ownership, uint64_t* ticket) {
  if(b == nullptr || ticket == nullptr) return fail(KMX_ERR_INVALID_ARG, "kmx_batcher_submit: null argument");
  const float* sp[1] = {row_spatial};
  const float* gl[1] = {row_global};
  const float* mt[1] = {row_meta};
  float* pol[1] = {out_policy};
  float* own[1] = {out_ownership};
  const int rc = okmx_eval_meta(b->model->m, b->ctx->nnXLen, b->ctx->nnYLen, 1, sp, gl, row_meta != nullptr ? mt : nullptr, &symmetry,
                                &policy_optimism, pol, out_value, out_score, own, 1);
  if(rc != KMX_OK) return passOn(rc);
  *ticket = b->rows.fetch_add(1) + 1;
  return KMX_OK;
}
int kmx_batcher_submit_packed(kmx_batcher* b, const uint8_t* row_packed, const float* row_global, const float* row_meta, int symmetry,
                              float policy_optimism, float* out_policy, float* out_value, float* out_score, float* out_ownership,
                              uint64_t* ticket) {
  if(b == nullptr || row_packed == nullptr) return fail(KMX_ERR_INVALID_ARG, "kmx_batcher_submit_packed: null argument");
  // the oracle knows fp32 rows only: expand the bits (what the device's input stage does, misc_kernels.hip inputExpand)
  const int cells = b->ctx->nnXLen * b->ctx->nnYLen, planeBytes = (cells + 7) / 8, numPlanes = b->info.num_input_channels;
  std::vector<float> row((size_t)cells * numPlanes);
  for(int pos = 0; pos < cells; pos++)
    for(int p = 0; p < numPlanes; p++)
      row[(size_t)pos * numPlanes + p] = (float)((row_packed[(size_t)p * planeBytes + (pos >> 3)] >> (7 - (pos & 7))) & 1);
  return kmx_batcher_submit(b, row.data(), row_global, row_meta, symmetry, policy_optimism, out_policy, out_value, out_score, out_ownership, ticket);
}
int kmx_batcher_wait(kmx_batcher*, uint64_t) { return KMX_OK; }
int kmx_batcher_stats(kmx_batcher* b, uint64_t* rows, uint64_t* batches) {
  if(b == nullptr || rows == nullptr || batches == nullptr) return fail(KMX_ERR_INVALID_ARG, "kmx_batcher_stats: null argument");
  *rows = *batches = b->rows.load();
  return KMX_OK;
}

#define FP32_ONLY(mode) \
  if((mode) != KMX_PREC_FP32) return fail(KMX_ERR_UNSUPPORTED, "the CPU oracle computes in fp32 only")
int kmx_test_conv(const kmx_conv_desc* desc, int batch, int nn_x_len, int nn_y_len, int precision_mode, const float* in_nhwc, float* out_nhwc) {
  FP32_ONLY(precision_mode);
  return passOn(okmx_test_conv(desc, batch, nn_x_len, nn_y_len, in_nhwc, out_nhwc));
}
int kmx_test_bnact(const kmx_bnact_desc* desc, int batch, int nn_x_len, int nn_y_len, int precision_mode, const float* in_nhwc,
                   const float* mask_nhw, float* out_nhwc) {
  FP32_ONLY(precision_mode);
  return passOn(okmx_test_bnact(desc, batch, nn_x_len, nn_y_len, in_nhwc, mask_nhw, out_nhwc));
}
int kmx_test_resblock(const kmx_resblock_desc* desc, int batch, int nn_x_len, int nn_y_len, int precision_mode, const float* in_nhwc,
                      const float* mask_nhw, float* out_nhwc) {
  FP32_ONLY(precision_mode);
  return passOn(okmx_test_resblock(desc, batch, nn_x_len, nn_y_len, in_nhwc, mask_nhw, out_nhwc));
}
int kmx_test_gpoolblock(const kmx_gpoolblock_desc* desc, int batch, int nn_x_len, int nn_y_len, int precision_mode, const float* in_nhwc,
                        const float* mask_nhw, float* out_nhwc) {
  FP32_ONLY(precision_mode);
  return passOn(okmx_test_gpoolblock(desc, batch, nn_x_len, nn_y_len, in_nhwc, mask_nhw, out_nhwc));
}

}  // extern "C"

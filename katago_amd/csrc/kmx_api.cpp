// kmx_api.cpp — the C ABI of include/katamx.h on top of the Engine. Every entry point converts C++
// exceptions into a status code + thread-local message; nothing here computes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>

#include "../../include/katamx.h"
#include "katamx_tuning.h"
#include "engine.h"
#include "model_desc.h"

using namespace kmx;

namespace kmx {
double benchConv(int ks, int wn, int variant, int cin, int cout, int batch, int X, int Y, int epilogueMode, int iters);  // conv_bench.hip
double benchConvStreams(int ks, int cfg, int cin, int cout, int batch, int nStreams, double delayUs, int launches, int epilogueMode);  // conv_bench.hip
double benchSeam(int batch, int iters, int timing);  // conv_bench.hip
double benchConvChain(int batch, int nConv, int chained, int iters, int timing);  // conv_bench.hip
double benchMfma(int wavesPerWg, int wgs, int mode, int steps, int iters, double* tflops, double* coreMhz);       // conv_bench.hip
double benchMfmaSustained(int wgs, int shape, int kind, int dtype, double seconds, double* tflops, double* coreMhz);                  // conv_bench.hip
double benchLaunchFloor(int wgs, int ldsBytes, int mode, int launches, int iters);                                 // conv_bench.hip
}

struct kmx_model {
  std::unique_ptr<ModelDesc> desc;
};
struct kmx_context {
  int nnXLen, nnYLen, precisionMode;
  std::vector<int> gpuIdxs;
};
// A handle owns one engine, or - for max_batch >= the split threshold - K engines on K streams that each evaluate 1/K
// of a large batch. The kernel streams run out of phase, so one part's memory-bound phases (1x1 convolutions, residual
// fetches, epilogue stores) overlap another part's MFMA loops instead of every work-group of the chip hitting HBM at
// the same moment. Weights are duplicated per engine (60 MB for b18c384nbt); activation memory is
// the same in total.
struct kmx_handle {
  std::vector<std::unique_ptr<Engine>> engines;  // [0] serves unsplit batches and the first part of split ones
  int precision;
  int maxBatch;
  int splitMin;          // batches of at least this many rows are split
  int splitMinAtCreate;  // the value the handle was created with (KMX_SPLIT_MIN or the default)
  int staggerOp = 0;     // the other parts start after this many launches of part 0 (0: all at entry)
  // fork / join of the streams around a split device-entry call: the other engines' streams wait for `fork` (recorded
  // on engines[0]'s stream, i.e. after everything the caller enqueued there), engines[0]'s stream waits for each `join`
  // (recorded after that part's last launch). kmx_handle_stream() is thereby an ordering point for the inputs and
  // outputs of ALL parts.
  hipEvent_t fork = nullptr;
  std::vector<hipEvent_t> join;
  uint64_t batches = 0;
  Engine& e0() const { return *engines[0]; }
  int ways() const { return (int)engines.size(); }
  bool splits(int n) const { return engines.size() > 1 && n >= splitMin && n >= ways(); }  // no empty parts
  // rows [begin, begin + count) of part i of an n-row batch: parts differ by at most one row
  void part(int n, int i, int* begin, int* count) const {
    const int k = ways(), base = n / k, extra = n % k;
    *begin = i * base + (i < extra ? i : extra);
    *count = base + (i < extra ? 1 : 0);
  }
  void syncAll() {
    for(auto& e : engines) e->sync();
  }
  ~kmx_handle() {
    if(fork) (void)hipEventDestroy(fork);
    for(hipEvent_t e : join)
      if(e) (void)hipEventDestroy(e);
  }
};

namespace {
thread_local std::string g_lastError;

int setError(int code, const std::string& msg) {
  g_lastError = msg;
  return code;
}
template <class F>
int guarded(F&& f) {
  try {
    f();
    return KMX_OK;
  }
  catch(const ModelError& e) { return setError(e.code, e.what()); }
  catch(const Error& e) { return setError(e.code, e.what()); }
  catch(const std::bad_alloc&) { return setError(KMX_ERR_INTERNAL, "out of host memory"); }
  catch(const std::exception& e) { return setError(KMX_ERR_INTERNAL, e.what()); }
  catch(...) { return setError(KMX_ERR_INTERNAL, "unknown exception"); }  // nothing may unwind through the C ABI
}
int dtypeForPrecision(int mode) {
  switch(mode) {
    case KMX_PREC_AUTO: return DT_BF16;
    case KMX_PREC_BF16: return DT_BF16;
    case KMX_PREC_FP16: return DT_F16;
    case KMX_PREC_FP32: return DT_F32;  // the verification mode (kernels.h DT_F32): plain fp32 kernels, correct and slow
    default: return -1;
  }
}
int deviceCountOrThrow() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if(e != hipSuccess || n <= 0)
    throw Error(KMX_ERR_DEVICE, std::string("no usable HIP device (katamx has no CPU fallback): ") + hipGetErrorString(e));
  return n;
}
}  // namespace

// what batcher.cpp (the other translation unit of the C ABI) needs from here
namespace kmx {
int apiGuarded(const std::function<void()>& f) { return guarded(f); }
int apiSetError(int code, const std::string& msg) { return setError(code, msg); }
const ModelDesc& apiModelDesc(const kmx_model* model) { return *model->desc; }
void apiContextDims(const kmx_context* ctx, int* x, int* y) {
  *x = ctx->nnXLen;
  *y = ctx->nnYLen;
}
// AUTO: fp16 wherever its five exponent bits are safe - convolutional nets that take the reference's 1/8 range transform
// (desc.cpp:2718-2736; model_desc.cpp scaledBy8, applied by the Engine), the reference's own answer to fp16 overflow - and for nets
// with transformer blocks or an RMSNorm trunk tip, which the reference also runs in plain fp16 and whose normalisations amplify
// bf16's 8-bit mantissa to 4.1x / 1.1x of the reference's reduced-precision limits (fp16: 0.24x / 0.06x; testgpuerror,
// profiles/r02/transformer/). What is left (standard-norm nets with an activation the transform has no counterpart for, i.e.
// SiLU) runs in bf16. Round 2 defaulted convolutional nets to bf16: 8x less accurate (1.67x of the strict limits against
// 0.21x) for ~3 % of speed.
int apiDtypeFor(const kmx_context* ctx, const kmx_model* model) {
  int dtype = dtypeForPrecision(ctx->precisionMode);
  if(ctx->precisionMode == KMX_PREC_AUTO &&
     (model->desc->hasTransformerBlocks || model->desc->trunkNormKind != 0 || model->desc->scale8Applies()))
    dtype = DT_F16;
  return dtype;
}
}  // namespace kmx

extern "C" {

int kmx_abi_version(void) { return KMX_ABI_VERSION; }
const char* kmx_last_error(void) { return g_lastError.c_str(); }

int kmx_global_init(void) {
  return guarded([] { (void)deviceCountOrThrow(); });
}
void kmx_global_cleanup(void) {}

int kmx_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if(e != hipSuccess) return setError(KMX_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}
int kmx_device_name(int device, char* buf, size_t buflen) {
  return guarded([&] {
    if(!buf || buflen == 0) throw Error(KMX_ERR_INVALID_ARG, "kmx_device_name: null buffer");
    hipDeviceProp_t prop;
    hipCheck(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    snprintf(buf, buflen, "%s (%s, %d CUs, %.1f GB)", prop.name, prop.gcnArchName, prop.multiProcessorCount,
             (double)prop.totalGlobalMem / 1e9);
  });
}

int kmx_model_load(const char* path, const char* expected_sha256, kmx_model** out) {
  return guarded([&] {
    if(!path || !out) throw Error(KMX_ERR_INVALID_ARG, "kmx_model_load: null argument");
    *out = nullptr;
    std::unique_ptr<kmx_model> m(new kmx_model());
    m->desc = ModelDesc::loadFromFile(path, expected_sha256 ? expected_sha256 : "");
    *out = m.release();
  });
}
void kmx_model_free(kmx_model* model) { delete model; }

int kmx_model_info_get(const kmx_model* model, kmx_model_info* out) {
  return guarded([&] {
    if(!model || !out) throw Error(KMX_ERR_INVALID_ARG, "kmx_model_info_get: null argument");
    const ModelDesc& d = *model->desc;
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", d.name.c_str());
    out->model_version = d.version;
    out->num_input_channels = d.numInputChannels;
    out->num_input_global_channels = d.numInputGlobalChannels;
    out->num_input_meta_channels = d.metaEncoderVersion > 0 ? d.numInputMetaChannels : 0;
    out->meta_encoder_version = d.metaEncoderVersion;
    out->num_policy_channels = d.numPolicyChannels;
    out->num_value_channels = d.numValueChannels;
    out->num_score_value_channels = d.numScoreValueChannels;
    out->num_ownership_channels = d.numOwnershipChannels;
    out->trunk_num_channels = d.trunkC;
    out->mid_num_channels = d.midC;
    out->num_blocks = d.numBlocks;
    out->td_score_multiplier = d.postProcess[0];
    out->score_mean_multiplier = d.postProcess[1];
    out->score_stdev_multiplier = d.postProcess[2];
    out->lead_multiplier = d.postProcess[3];
    out->variance_time_multiplier = d.postProcess[4];
    out->shortterm_value_error_multiplier = d.postProcess[5];
    out->shortterm_score_error_multiplier = d.postProcess[6];
    out->output_scale_multiplier = 1.0f;
    out->num_parameters = d.numParameters;
    out->flops_per_position = 2.0 * d.macPerPosition;
  });
}

int kmx_context_create(const int* gpu_idxs, int num_gpu_idxs, int nn_x_len, int nn_y_len, int precision_mode, kmx_context** out) {
  return guarded([&] {
    if(!out) throw Error(KMX_ERR_INVALID_ARG, "kmx_context_create: null argument");
    *out = nullptr;
    if(nn_x_len < 2 || nn_y_len < 2 || nn_x_len > 19 || nn_y_len > 19)
      throw Error(KMX_ERR_INVALID_ARG, "kmx_context_create: nnXLen/nnYLen must be in 2..19");
    if(dtypeForPrecision(precision_mode) < 0) throw Error(KMX_ERR_INVALID_ARG, "kmx_context_create: unknown precision mode");
    const int ndev = deviceCountOrThrow();
    std::unique_ptr<kmx_context> c(new kmx_context());
    c->nnXLen = nn_x_len;
    c->nnYLen = nn_y_len;
    c->precisionMode = precision_mode;
    for(int i = 0; i < num_gpu_idxs; i++) {
      int g = gpu_idxs ? gpu_idxs[i] : -1;
      if(g >= ndev) throw Error(KMX_ERR_DEVICE, "kmx_context_create: device index " + std::to_string(g) + " out of range (" + std::to_string(ndev) + " devices)");
      c->gpuIdxs.push_back(g);
    }
    *out = c.release();
  });
}
void kmx_context_free(kmx_context* ctx) { delete ctx; }

int kmx_handle_create(kmx_context* ctx, const kmx_model* model, int max_batch_size, int require_exact_nn_len, int gpu_idx,
                      kmx_handle** out) {
  (void)require_exact_nn_len;  // masking is always on: it costs one multiply in the conv epilogue
  return guarded([&] {
    if(!ctx || !model || !out) throw Error(KMX_ERR_INVALID_ARG, "kmx_handle_create: null argument");
    *out = nullptr;
    if(max_batch_size < 1) throw Error(KMX_ERR_INVALID_ARG, "kmx_handle_create: max_batch_size must be positive");
    const int ndev = deviceCountOrThrow();
    const int dev = gpu_idx < 0 ? 0 : gpu_idx;
    if(dev >= ndev) throw Error(KMX_ERR_DEVICE, "kmx_handle_create: device index out of range");
    std::unique_ptr<kmx_handle> h(new kmx_handle());
    const int dtype = apiDtypeFor(ctx, model);
    h->precision = dtype == DT_F16 ? KMX_PREC_FP16 : dtype == DT_F32 ? KMX_PREC_FP32 : KMX_PREC_BF16;
    h->maxBatch = max_batch_size;
    int splitMin = 224;  // parts of >= 56 boards: the 8-wave work-groups of all parts together fill >= 87 % of the CUs
    if(const char* e = getenv("KMX_SPLIT_MIN")) splitMin = atoi(e);  // 0 disables splitting
    int ways = 2;
    if(const char* e = getenv("KMX_SPLIT_WAYS")) ways = std::max(1, std::min(8, atoi(e)));
    h->splitMin = h->splitMinAtCreate = splitMin;
    if(const char* e = getenv("KMX_SPLIT_STAGGER")) h->staggerOp = atoi(e);
    h->engines.emplace_back(new Engine(*model->desc, ctx->nnXLen, ctx->nnYLen, max_batch_size, dtype, dev));
    if(splitMin > 1 && ways > 1 && max_batch_size >= splitMin && max_batch_size >= ways) {
      const int partMax = (max_batch_size + ways - 1) / ways;
      hipCheck(hipEventCreateWithFlags(&h->fork, hipEventDisableTiming), "hipEventCreate");
      for(int i = 1; i < ways; i++) {
        h->engines.emplace_back(new Engine(*model->desc, ctx->nnXLen, ctx->nnYLen, partMax, dtype, dev));
        hipEvent_t ev = nullptr;
        hipCheck(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        h->join.push_back(ev);
      }
    }
    *out = h.release();
  });
}
void kmx_handle_free(kmx_handle* handle) { delete handle; }
int kmx_handle_precision(const kmx_handle* handle) { return handle ? handle->precision : KMX_ERR_INVALID_ARG; }

int kmx_eval(kmx_handle* handle, int n_rows, const float* const* row_spatial, const float* const* row_global,
             const int* symmetry, const float* policy_optimism, float* const* out_policy, float* out_value, float* out_score,
             float* const* out_ownership) {
  return kmx_eval_meta(handle, n_rows, row_spatial, row_global, nullptr, symmetry, policy_optimism, out_policy, out_value, out_score,
                       out_ownership);
}

// kmx_eval / kmx_eval_meta / kmx_eval_packed: exactly one of row_spatial (fp32 NHWC rows) and row_packed (bit planes)
static int evalHostEntry(kmx_handle* handle, int n_rows, const float* const* row_spatial, const unsigned char* const* row_packed,
                         const float* const* row_global, const float* const* row_meta, const int* symmetry,
                         const float* policy_optimism, float* const* out_policy, float* out_value, float* out_score,
                         float* const* out_ownership) {
  return guarded([&] {
    if(!handle || (!row_spatial && !row_packed) || !row_global || !out_policy || !out_value || !out_score)
      throw Error(KMX_ERR_INVALID_ARG, "kmx_eval: null argument");
    if(n_rows < 1 || n_rows > handle->maxBatch) throw Error(KMX_ERR_INVALID_ARG, "batch size out of range for this handle");
    for(int i = 0; i < n_rows; i++)
      if((row_packed ? (const void*)row_packed[i] : (const void*)row_spatial[i]) == nullptr || !row_global[i] || !out_policy[i])
        throw Error(KMX_ERR_INVALID_ARG, "kmx_eval: null row pointer");
    if(symmetry)
      for(int i = 0; i < n_rows; i++)
        if(symmetry[i] < 0 || symmetry[i] > 7) throw Error(KMX_ERR_INVALID_ARG, "kmx_eval: symmetry must be in 0..7");
    handle->batches++;
    if(!handle->splits(n_rows)) {
      handle->e0().setConcurrency(1);
      handle->e0().evalHostBegin(n_rows, row_spatial, row_packed, row_global, row_meta, symmetry, policy_optimism, out_ownership);
      handle->e0().evalHostFinish(n_rows, out_policy, out_value, out_score, out_ownership);
      return;
    }
    const int K = handle->ways();
    try {
      for(int i = 0; i < K; i++) {
        int b, c;
        handle->part(n_rows, i, &b, &c);
        Engine& e = *handle->engines[i];
        e.setConcurrency(K);
        e.evalHostBegin(c, row_spatial ? row_spatial + b : nullptr, row_packed ? row_packed + b : nullptr, row_global + b,
                        row_meta ? row_meta + b : nullptr, symmetry ? symmetry + b : nullptr,
                        policy_optimism ? policy_optimism + b : nullptr, out_ownership ? out_ownership + b : nullptr);
      }
      for(int i = 0; i < K; i++) {
        int b, c;
        handle->part(n_rows, i, &b, &c);
        handle->engines[i]->evalHostFinish(c, out_policy + b, out_value + (size_t)b * 3, out_score + (size_t)b * 6,
                                           out_ownership ? out_ownership + b : nullptr);
      }
    }
    catch(...) {
      // one part failed while others may still be in flight: the call is synchronous, so nothing of it may be running
      // when the error is reported
      for(auto& e : handle->engines) {
        try { e->sync(); } catch(...) {}
      }
      throw;
    }
  });
}

int kmx_eval_meta(kmx_handle* handle, int n_rows, const float* const* row_spatial, const float* const* row_global,
                  const float* const* row_meta, const int* symmetry, const float* policy_optimism, float* const* out_policy,
                  float* out_value, float* out_score, float* const* out_ownership) {
  if(!row_spatial) return guarded([] { throw Error(KMX_ERR_INVALID_ARG, "kmx_eval: null argument"); });
  return evalHostEntry(handle, n_rows, row_spatial, nullptr, row_global, row_meta, symmetry, policy_optimism, out_policy, out_value,
                       out_score, out_ownership);
}

int kmx_eval_packed(kmx_handle* handle, int n_rows, const uint8_t* const* row_packed, const float* const* row_global,
                    const float* const* row_meta, const int* symmetry, const float* policy_optimism, float* const* out_policy,
                    float* out_value, float* out_score, float* const* out_ownership) {
  if(!row_packed) return guarded([] { throw Error(KMX_ERR_INVALID_ARG, "kmx_eval_packed: null argument"); });
  return evalHostEntry(handle, n_rows, nullptr, row_packed, row_global, row_meta, symmetry, policy_optimism, out_policy, out_value,
                       out_score, out_ownership);
}

// packBits of the reference (dataio/trainingwrite.cpp:314-337) applied plane by plane to an NHWC row
int kmx_pack_row(const float* row_spatial_nhwc, int nn_x_len, int nn_y_len, int num_channels, uint8_t* out_packed) {
  return guarded([&] {
    if(!row_spatial_nhwc || !out_packed || nn_x_len < 1 || nn_y_len < 1 || num_channels < 1)
      throw Error(KMX_ERR_INVALID_ARG, "kmx_pack_row: bad argument");
    const int S = nn_x_len * nn_y_len, PB = (S + 7) / 8;
    memset(out_packed, 0, (size_t)num_channels * PB);
    for(int p = 0; p < S; p++) {
      const float* cell = row_spatial_nhwc + (size_t)p * num_channels;
      const int byte = p >> 3;
      const uint8_t bit = (uint8_t)(1u << (7 - (p & 7)));
      for(int c = 0; c < num_channels; c++)
        if(cell[c] != 0.0f) out_packed[(size_t)c * PB + byte] |= bit;
    }
  });
}

int kmx_eval_device(kmx_handle* handle, int n_rows, const float* d_spatial, const float* d_global, const int* symmetry,
                    const float* policy_optimism, float* d_policy, float* d_value, float* d_score, float* d_ownership, int sync) {
  return kmx_eval_device_meta(handle, n_rows, d_spatial, d_global, nullptr, symmetry, policy_optimism, d_policy, d_value, d_score,
                              d_ownership, sync);
}

int kmx_eval_device_meta(kmx_handle* handle, int n_rows, const float* d_spatial, const float* d_global, const float* d_meta,
                         const int* symmetry, const float* policy_optimism, float* d_policy, float* d_value, float* d_score,
                         float* d_ownership, int sync) {
  return guarded([&] {
    if(!handle || !d_spatial || !d_global || !d_policy || !d_value || !d_score)
      throw Error(KMX_ERR_INVALID_ARG, "kmx_eval_device: null argument");
    if(symmetry)
      for(int i = 0; i < n_rows; i++)
        if(symmetry[i] < 0 || symmetry[i] > 7) throw Error(KMX_ERR_INVALID_ARG, "kmx_eval_device: symmetry must be in 0..7");
    if(n_rows < 1 || n_rows > handle->maxBatch) throw Error(KMX_ERR_INVALID_ARG, "batch size out of range for this handle");
    handle->batches++;
    if(!handle->splits(n_rows)) {
      handle->e0().setConcurrency(1);
      handle->e0().evalDevice(n_rows, d_spatial, d_global, d_meta, symmetry, policy_optimism, d_policy, d_value, d_score, d_ownership,
                              sync != 0);
      return;
    }
    Engine& e0 = handle->e0();
    const size_t S = (size_t)e0.nnXLen() * e0.nnYLen();
    const size_t spRow = S * e0.numInputChannels(), glRow = e0.numInputGlobalChannels(), mtRow = e0.numInputMetaChannels();
    const int K = handle->ways();
    // Fork: the other engines' streams start after the point of engines[0]'s stream where the fork event is recorded -
    // at entry (so that inputs the caller enqueued on kmx_handle_stream() are complete for every part), or, with a
    // stagger, some launches into part 0, which keeps the parts out of phase even when calls are queued back to back.
    // Join: engines[0]'s stream then waits for every other part's last launch, so the handle stream orders all outputs.
    try {
      for(int i = 0; i < K; i++) {
        int b, c;
        handle->part(n_rows, i, &b, &c);
        Engine& e = *handle->engines[i];
        e.setConcurrency(K);
        if(i == 0) e.setForkPoint(handle->staggerOp, handle->fork);
        else hipCheck(hipStreamWaitEvent(e.stream(), handle->fork, 0), "hipStreamWaitEvent");
        e.evalDevice(c, d_spatial + b * spRow, d_global + b * glRow, d_meta ? d_meta + b * mtRow : nullptr,
                     symmetry ? symmetry + b : nullptr, policy_optimism ? policy_optimism + b : nullptr, d_policy + b * (S + 1),
                     d_value + (size_t)b * 3, d_score + (size_t)b * 6, d_ownership ? d_ownership + b * S : nullptr, false);
        if(i == 0) e.setForkPoint(0, nullptr);
        else hipCheck(hipEventRecord(handle->join[i - 1], e.stream()), "hipEventRecord");
      }
      for(int i = 1; i < K; i++) hipCheck(hipStreamWaitEvent(e0.stream(), handle->join[i - 1], 0), "hipStreamWaitEvent");
      if(sync != 0) e0.sync();  // covers the other parts through the joins
    }
    catch(...) {
      // one part failed to launch while others may be in flight: drain everything before the error is reported
      e0.setForkPoint(0, nullptr);
      for(auto& e : handle->engines) {
        try { e->sync(); } catch(...) {}
      }
      throw;
    }
  });
}
void* kmx_handle_stream(kmx_handle* handle) { return handle ? (void*)handle->e0().stream() : nullptr; }
int kmx_handle_sync(kmx_handle* handle) {
  return guarded([&] {
    if(!handle) throw Error(KMX_ERR_INVALID_ARG, "kmx_handle_sync: null handle");
    handle->syncAll();
  });
}
int kmx_handle_stats(const kmx_handle* handle, uint64_t* rows, uint64_t* batches) {
  if(!handle) return setError(KMX_ERR_INVALID_ARG, "kmx_handle_stats: null handle");
  if(rows) {
    *rows = 0;
    for(const auto& e : handle->engines) *rows += e->rowsProcessed();
  }
  if(batches) *batches = handle->batches;
  return KMX_OK;
}

int kmx_handle_set_split_min(kmx_handle* handle, int min_rows) {
  return guarded([&] {
    if(!handle) throw Error(KMX_ERR_INVALID_ARG, "kmx_handle_set_split_min: null handle");
    handle->syncAll();
    // negative: back to the value the handle was created with
    handle->splitMin = min_rows < 0 ? handle->splitMinAtCreate : min_rows == 0 ? 0x7fffffff : (min_rows < 2 ? 2 : min_rows);
  });
}

int kmx_handle_set_profiling(kmx_handle* handle, int enabled) {
  return guarded([&] {
    if(!handle) throw Error(KMX_ERR_INVALID_ARG, "kmx_handle_set_profiling: null handle");
    for(auto& e : handle->engines) e->setProfiling(enabled != 0);
  });
}
int kmx_handle_set_graphs(kmx_handle* handle, int enabled) {
  return guarded([&] {
    if(!handle) throw Error(KMX_ERR_INVALID_ARG, "kmx_handle_set_graphs: null handle");
    handle->syncAll();
    for(auto& e : handle->engines) e->setGraphs(enabled != 0);
  });
}
int kmx_handle_graph_stats(const kmx_handle* handle, uint64_t* graph_launches) {
  if(!handle || !graph_launches) return setError(KMX_ERR_INVALID_ARG, "kmx_handle_graph_stats: null argument");
  *graph_launches = 0;
  for(const auto& e : handle->engines) *graph_launches += e->graphLaunches();
  return KMX_OK;
}
int kmx_handle_get_profile(kmx_handle* handle, kmx_profile_entry* entries, int max_entries, int* n_entries) {
  return guarded([&] {
    if(!handle || !n_entries || (max_entries > 0 && !entries)) throw Error(KMX_ERR_INVALID_ARG, "kmx_handle_get_profile: null argument");
    std::vector<Engine::ProfileEntry> prof = handle->e0().getProfile();
    for(size_t k = 1; k < handle->engines.size(); k++)
      for(const Engine::ProfileEntry& e2 : handle->engines[k]->getProfile()) {  // same classes: add the other streams' launches
        bool found = false;
        for(Engine::ProfileEntry& e : prof)
          if(e.name == e2.name) {
            e.launches += e2.launches; e.ms += e2.ms; e.flops += e2.flops; e.bytes += e2.bytes;
            found = true;
          }
        if(!found) prof.push_back(e2);
      }
    *n_entries = (int)prof.size();
    for(int i = 0; i < (int)prof.size() && i < max_entries; i++) {
      memset(&entries[i], 0, sizeof(entries[i]));
      snprintf(entries[i].name, sizeof(entries[i].name), "%s", prof[i].name.c_str());
      entries[i].launches = prof[i].launches;
      entries[i].total_ms = prof[i].ms;
      entries[i].flops = prof[i].flops;
      entries[i].bytes = prof[i].bytes;
    }
  });
}

int kmx_bench_conv(int ks, int wn, int variant, int cin, int cout, int batch, int nn_x_len, int nn_y_len, int epilogue_mode,
                   int iters, double* avg_ms) {
  return guarded([&] {
    if(!avg_ms || iters < 1 || batch < 1 || cin < 1 || cout < 1) throw Error(KMX_ERR_INVALID_ARG, "kmx_bench_conv: bad argument");
    (void)deviceCountOrThrow();
    *avg_ms = benchConv(ks, wn, variant, cin, cout, batch, nn_x_len, nn_y_len, epilogue_mode, iters);
  });
}

int kmx_bench_conv_streams(int ks, int cfg, int cin, int cout, int batch, int n_streams, double delay_us, int launches, int epilogue_mode,
                           double* total_ms) {
  return guarded([&] {
    if(!total_ms || launches < 1 || batch < 1 || cin < 1 || cout < 1) throw Error(KMX_ERR_INVALID_ARG, "kmx_bench_conv_streams: bad argument");
    (void)deviceCountOrThrow();
    *total_ms = benchConvStreams(ks, cfg, cin, cout, batch, n_streams, delay_us, launches, epilogue_mode);
  });
}

int kmx_bench_conv_chain(int batch, int n_conv, int chained, int iters, int timing, double* avg_ms) {
  return guarded([&] {
    if(!avg_ms || iters < 1 || batch < 1 || (n_conv != 2 && n_conv != 4) || (chained != 0 && chained != 2 && chained != 4) || chained > n_conv)
      throw Error(KMX_ERR_INVALID_ARG, "kmx_bench_conv_chain: bad argument");
    *avg_ms = benchConvChain(batch, n_conv, chained, iters, timing);
  });
}
int kmx_bench_seam(int batch, int iters, int timing, double* avg_ms) {
  return guarded([&] {
    if(!avg_ms || iters < 1 || batch < 1) throw Error(KMX_ERR_INVALID_ARG, "kmx_bench_seam: bad argument");
    (void)deviceCountOrThrow();
    *avg_ms = benchSeam(batch, iters, timing);
  });
}

int kmx_debug_conv_cfg(int ks, int cout_pad, int batch, int* cfg, int* instantiated) {
  return guarded([&] {
    if(!cfg || !instantiated || cout_pad < 64 || cout_pad % 64 != 0 || batch < 1 || (ks != 1 && ks != 3 && ks != 5))
      throw Error(KMX_ERR_INVALID_ARG, "kmx_debug_conv_cfg: bad argument");
    if(const char* tuneError = convTuneError()) throw Error(KMX_ERR_INVALID_ARG, tuneError);  // what engine construction reports, too
    *cfg = chooseConvCfg(ks, cout_pad, batch);
    // the special shapes of conv_mfma.hip: 113 / 114 (1x1, deep ring), 117 (3x3, split) and 118 / 119 (3x3, fetching waves) tile 32 channels, 124 tiles 64
    const int ntile = (*cfg == 124 || *cfg == 126) ? 64 : *cfg >= 111 ? 32 : 32 * (*cfg / 10) * (*cfg % 10);
    *instantiated = (convCfgInstantiated(ks, *cfg) && cout_pad % ntile == 0) ? 1 : 0;
  });
}

int kmx_bench_mfma(int waves_per_wg, int wgs, int mode, int steps, int iters, double* avg_ms, double* tflops, double* core_mhz) {
  return guarded([&] {
    if(!avg_ms || iters < 1 || steps < 1 || wgs < 1 || waves_per_wg < 1 || waves_per_wg > 8)
      throw Error(KMX_ERR_INVALID_ARG, "kmx_bench_mfma: bad argument");
    (void)deviceCountOrThrow();
    *avg_ms = benchMfma(waves_per_wg, wgs, mode, steps, iters, tflops, core_mhz);
  });
}

int kmx_bench_mfma_sustained(int wgs, int shape, int data_kind, int precision_mode, double seconds, double* tflops, double* core_mhz) {
  return guarded([&] {
    if(!tflops || wgs < 1 || wgs > 65536 || shape < 0 || shape > 5 || data_kind < 0 || data_kind > 3 || !(seconds > 0.0) || seconds > 30.0 ||
       (precision_mode != KMX_PREC_FP16 && precision_mode != KMX_PREC_BF16))
      throw Error(KMX_ERR_INVALID_ARG, "kmx_bench_mfma_sustained: bad argument");
    (void)deviceCountOrThrow();
    (void)benchMfmaSustained(wgs, shape, data_kind, dtypeForPrecision(precision_mode), seconds, tflops, core_mhz);
  });
}

int kmx_bench_launch_floor(int wgs, int lds_bytes, int mode, int launches, int iters, double* us_per_launch) {
  return guarded([&] {
    if(!us_per_launch || wgs < 1 || wgs > 65536 || lds_bytes < 8192 || lds_bytes > 160 * 1024 || mode < 0 || mode > 2 || launches < 1 || launches > 4096 ||
       iters < 1)
      throw Error(KMX_ERR_INVALID_ARG, "kmx_bench_launch_floor: bad argument");
    (void)deviceCountOrThrow();
    *us_per_launch = benchLaunchFloor(wgs, lds_bytes, mode, launches, iters);
  });
}

// the layer hooks run in any precision; the hooks of kernels that only exist for 16-bit storage (the fused seam, chained convolutions,
// transformer layers) refuse fp32
static int hookDtype(int precision_mode, bool sixteenBitOnly = false) {
  if(sixteenBitOnly && precision_mode == KMX_PREC_FP32) throw Error(KMX_ERR_UNSUPPORTED, "this kernel exists for 16-bit storage only");
  int dt = dtypeForPrecision(precision_mode);
  if(dt < 0) throw Error(KMX_ERR_INVALID_ARG, "unknown precision mode");
  (void)deviceCountOrThrow();
  return dt;
}
int kmx_test_conv(const kmx_conv_desc* desc, int batch, int nn_x_len, int nn_y_len, int precision_mode, const float* in_nhwc,
                  float* out_nhwc) {
  return guarded([&] {
    if(!desc || !in_nhwc || !out_nhwc) throw Error(KMX_ERR_INVALID_ARG, "kmx_test_conv: null argument");
    testConv(hookDtype(precision_mode), desc, batch, nn_x_len, nn_y_len, in_nhwc, out_nhwc);
  });
}
int kmx_test_bnact(const kmx_bnact_desc* desc, int batch, int nn_x_len, int nn_y_len, int precision_mode, const float* in_nhwc,
                   const float* mask_nhw, float* out_nhwc) {
  return guarded([&] {
    if(!desc || !in_nhwc || !out_nhwc) throw Error(KMX_ERR_INVALID_ARG, "kmx_test_bnact: null argument");
    testBnAct(hookDtype(precision_mode), desc, batch, nn_x_len, nn_y_len, in_nhwc, mask_nhw, out_nhwc);
  });
}
int kmx_test_resblock(const kmx_resblock_desc* desc, int batch, int nn_x_len, int nn_y_len, int precision_mode,
                      const float* in_nhwc, const float* mask_nhw, float* out_nhwc) {
  return guarded([&] {
    if(!desc || !in_nhwc || !out_nhwc) throw Error(KMX_ERR_INVALID_ARG, "kmx_test_resblock: null argument");
    testResBlock(hookDtype(precision_mode), desc, batch, nn_x_len, nn_y_len, in_nhwc, mask_nhw, out_nhwc);
  });
}
int kmx_test_gpoolblock(const kmx_gpoolblock_desc* desc, int batch, int nn_x_len, int nn_y_len, int precision_mode,
                        const float* in_nhwc, const float* mask_nhw, float* out_nhwc) {
  return guarded([&] {
    if(!desc || !in_nhwc || !out_nhwc) throw Error(KMX_ERR_INVALID_ARG, "kmx_test_gpoolblock: null argument");
    testGPoolBlock(hookDtype(precision_mode), desc, batch, nn_x_len, nn_y_len, in_nhwc, mask_nhw, out_nhwc);
  });
}

int kmx_test_pointwise_pair(int batch, int nn_x_len, int nn_y_len, int precision_mode, int c1, int c2, int c3, const float* in_nhwc,
                            const float* resid_nhwc, const float* w1_oi, const float* scale1, const float* bias1, int act1,
                            const float* w2_oi, const float* scale2, const float* bias2, int act2, const float* mask_nhw, int fused,
                            float* out_trunk_raw, float* out_mid_raw, float* out_mid_act) {
  return guarded([&] {
    testPointwisePair(hookDtype(precision_mode), batch, nn_x_len, nn_y_len, c1, c2, c3, in_nhwc, resid_nhwc, w1_oi, scale1, bias1, act1,
                      w2_oi, scale2, bias2, act2, mask_nhw, fused != 0, out_trunk_raw, out_mid_raw, out_mid_act);
  });
}
int kmx_test_conv_chain(int batch, int nn_x_len, int nn_y_len, int precision_mode, int n_conv, const float* x_nhwc, const float* r_nhwc,
                        const float* w_oihw, const float* scale, const float* bias, int activation, const float* mask_nhw, int chained,
                        float* out_r, float* out_x) {
  return guarded([&] {
    testConvChain(hookDtype(precision_mode), batch, nn_x_len, nn_y_len, n_conv, x_nhwc, r_nhwc, w_oihw, scale, bias, activation, mask_nhw,
                  chained, out_r, out_x);
  });
}
int kmx_test_rmsnorm(int batch, int nn_x_len, int nn_y_len, int precision_mode, int num_channels, float epsilon, const float* weight,
                     const float* beta, int activation, int per_board, const float* in_nhwc, const float* mask_nhw, float* out_nhwc) {
  return guarded([&] {
    testRmsNorm(hookDtype(precision_mode, true), batch, nn_x_len, nn_y_len, num_channels, epsilon, weight, beta, activation, per_board != 0,
                in_nhwc, mask_nhw, out_nhwc);
  });
}
int kmx_test_attention(int batch, int nn_x_len, int nn_y_len, int precision_mode, int num_heads, int num_kv_heads, int q_head_dim,
                       int v_head_dim, const float* rope_cos, const float* rope_sin, int rope_heads, const float* q, const float* k,
                       const float* v, const float* mask_nhw, float* out) {
  return guarded([&] {
    testAttention(hookDtype(precision_mode, true), batch, nn_x_len, nn_y_len, num_heads, num_kv_heads, q_head_dim, v_head_dim, rope_cos,
                  rope_sin, rope_heads, q, k, v, mask_nhw, out);
  });
}
int kmx_test_swiglu(int batch, int nn_x_len, int nn_y_len, int precision_mode, int ffn_channels, const float* a, const float* gate,
                    float* out) {
  return guarded([&] { testSwiGlu(hookDtype(precision_mode, true), batch, nn_x_len, nn_y_len, ffn_channels, a, gate, out); });
}

}  // extern "C"

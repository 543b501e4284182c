// katamxbackend.cpp — the reference-side binding of the katamx C ABI.
//
// This is the backend translation unit a KataGo maintainer adds next to
// cpp/neuralnet/{cuda,opencl,eigen,...}backend.cpp: it defines the 13 functions of
// `namespace NeuralNet` (cpp/neuralnet/nninterface.h:32-182) and the four opaque structs
// (nninterface.h:15-28) on top of include/katamx.h, so that the UNMODIFIED reference host code
// (nneval.cpp, search/, command/benchmark.cpp, gputest.cpp, tests/*) runs on the MI355X backend.
// It is compiled against the reference headers with -I<reference>/cpp (see INTEGRATION.md and
// oracle/Makefile); it contains no arithmetic and knows nothing but the C ABI.
// (The test builds oracle/_ref/katago_oracle{,x} link this same object file against an implementation of the ABI on the CPU
// oracle, oracle/kmx_abi_on_oracle.cpp, instead of libkatamx.so - that lives under oracle/, not here.)

#include "neuralnet/nninterface.h"
#include "neuralnet/nneval.h"
#include "neuralnet/nninputs.h"
#include "neuralnet/modelversion.h"
#include "neuralnet/desc.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "katamx.h"
#include "katamx_leaf.h"

using namespace std;

static string lastError() { return string(kmx_last_error()); }
// what the library says it runs on (the device's marketing name)
static string deviceLabel(int gpuIdx) {
  char name[256];
  return kmx_device_name(gpuIdx < 0 ? 0 : gpuIdx, name, sizeof(name)) == KMX_OK ? string(name) : string("?");
}
static const char* precisionName(int prec) {
  return prec == KMX_PREC_FP32 ? "fp32" : prec == KMX_PREC_FP16 ? "fp16" : "bf16";
}
static void check(int status, const char* what) {
  if(status != KMX_OK)
    throw StringError(string("katamx backend: ") + what + ": " + lastError());
}

// ---------------------------------------------------------------------------------------------
struct LoadedModel {
  ModelDesc modelDesc;  // parsed by the reference loader: NNEvaluator reads name/version/postprocess
  kmx_model* model = NULL;
  LoadedModel(const string& file, const string& expectedSha256) {
    ModelDesc::loadFromFileMaybeGZipped(file, modelDesc, expectedSha256);
    // The backend re-parses the file itself; it does not depend on the host's ModelDesc layout.
    check(kmx_model_load(file.c_str(), expectedSha256.c_str(), &model), "loading model");
    // This backend never applies the scale-8 transform (desc.cpp:2718-2736): outputScaleMultiplier stays 1.
    modelDesc.releaseWeights();
  }
  ~LoadedModel() {
    kmx_model_free(model);
  }
  LoadedModel() = delete;
  LoadedModel(const LoadedModel&) = delete;
  LoadedModel& operator=(const LoadedModel&) = delete;
};

struct ComputeContext {
  int nnXLen;
  int nnYLen;
  int precisionMode;
  kmx_context* ctx = NULL;
  // useFP16 = false on a net the fp32 verification mode does not cover (transformer blocks, RMSNorm tips: csrc/engine.cpp) falls back
  // to the backend's 16-bit default with a logged warning instead of failing at handle creation: a second context,
  // created on first need (under batcherMutex), frees with this one.
  kmx_context* fallbackCtx = NULL;
  std::vector<int> gpuIdxs;
  // katamxBatcher = true: every server thread of this context that serves the same (model, device) feeds ONE persistent leaf
  // batcher (kmx_batcher_*) instead of owning a handle: its getOutput submits the rows NNEvaluator::serve popped and waits
  // for their tickets. Rows of several server threads then share device batches (up to katamxBatcherInFlight of them
  // between H2D and D2H), where separate handles would run small batches side by side.
  bool useBatcher = false;
  int batcherInFlight = 2;
  struct SharedBatcher {
    kmx_batcher* batcher = NULL;
    int users = 0;
    int maxBatchSize = 0;
  };
  std::mutex batcherMutex;
  std::mutex fallbackMutex;
  std::map<std::pair<const LoadedModel*, int>, SharedBatcher> batchers;
};

struct ComputeHandle {
  const ComputeContext* context;
  const LoadedModel* loadedModel;
  bool inputsUseNHWC;
  int modelVersion;
  int numInputChannels;
  int numInputGlobalChannels;
  int numInputMetaChannels;
  kmx_handle* handle = NULL;
  kmx_batcher* batcher = NULL;  // shared (ComputeContext::batchers); NULL when katamxBatcher is off
  int gpuIdx = 0;
};

struct InputBuffers {
  int maxBatchSize;
  size_t singleInputElts;
  size_t singlePolicyElts;
  size_t singleOwnershipElts;
  // Per-call pointer tables and output staging, reused across calls.
  vector<const float*> rowSpatial;
  vector<const float*> rowGlobal;
  vector<const float*> rowMeta;  // only for nets with an sgf-metadata encoder
  vector<int> symmetry;
  vector<float> policyOptimism;
  vector<float*> outPolicy;
  vector<float*> outOwnership;
  vector<float> value;
  vector<float> score;
  vector<float> ownershipScratch;
  vector<float> nhwcScratch;  // only used when the host hands over NCHW rows
  vector<uint64_t> tickets;   // katamxBatcher
};

// ---------------------------------------------------------------------------------------------
void NeuralNet::globalInitialize() {
  check(kmx_global_init(), "global init");
}
void NeuralNet::globalCleanup() {
  kmx_global_cleanup();
}
void NeuralNet::printDevices() {
  int n = kmx_device_count();
  for(int i = 0; i < n; i++) {
    char name[256];
    if(kmx_device_name(i, name, sizeof(name)) == KMX_OK)
      cout << "Found HIP device " << i << ": " << name << endl;
  }
}

LoadedModel* NeuralNet::loadModelFile(const string& file, const string& expectedSha256) {
  return new LoadedModel(file, expectedSha256);
}
void NeuralNet::freeLoadedModel(LoadedModel* loadedModel) {
  delete loadedModel;
}
const ModelDesc& NeuralNet::getModelDesc(const LoadedModel* loadedModel) {
  return loadedModel->modelDesc;
}

ComputeContext* NeuralNet::createComputeContext(
  const vector<int>& gpuIdxs,
  Logger* logger,
  int nnXLen,
  int nnYLen,
  const string& homeDataDirOverride,
  enabled_t useFP16Mode,
  const LoadedModel* loadedModel,
  ConfigParser& cfg
) {
  (void)logger;
  (void)homeDataDirOverride;
  (void)loadedModel;
  std::unique_ptr<ComputeContext> context(new ComputeContext());  // released on success; a throwing check() must not leak it
  context->nnXLen = nnXLen;
  context->nnYLen = nnYLen;
  // useFP16 = false asks for fp32; true/auto picks the backend's 16-bit default. The private key
  // katamxPrecision = fp16|bf16|fp32 overrides (read off cfg as nninterface.h:60-62 allows).
  int precisionMode = (useFP16Mode == enabled_t::False) ? KMX_PREC_FP32 : KMX_PREC_AUTO;
  if(cfg.contains("katamxPrecision")) {
    string p = cfg.getString("katamxPrecision");
    if(p == "fp16") precisionMode = KMX_PREC_FP16;
    else if(p == "bf16") precisionMode = KMX_PREC_BF16;
    else if(p == "fp32") precisionMode = KMX_PREC_FP32;
    else if(p == "auto") precisionMode = KMX_PREC_AUTO;
    else throw StringError("katamxPrecision must be one of fp16, bf16, fp32, auto");
  }
  else if(const char* e = getenv("KATAMX_PRECISION")) {
    // for the reference commands that build their NNEvaluator without a user config (runsearchtestsv8, runtests ...)
    string p = e;
    if(p == "fp16") precisionMode = KMX_PREC_FP16;
    else if(p == "bf16") precisionMode = KMX_PREC_BF16;
    else if(p == "fp32") precisionMode = KMX_PREC_FP32;
    else if(p != "auto" && p != "") throw StringError("KATAMX_PRECISION must be one of fp16, bf16, fp32, auto");
  }
  context->precisionMode = precisionMode;
  context->gpuIdxs = gpuIdxs;
  if(cfg.contains("katamxBatcher")) context->useBatcher = cfg.getBool("katamxBatcher");
  else if(const char* e = getenv("KATAMX_BATCHER")) context->useBatcher = atoi(e) != 0;
  if(cfg.contains("katamxBatcherInFlight")) context->batcherInFlight = cfg.getInt("katamxBatcherInFlight", 1, 8);
  else if(const char* e = getenv("KATAMX_BATCHER_IN_FLIGHT")) context->batcherInFlight = std::max(1, std::min(8, atoi(e)));
  check(
    kmx_context_create(gpuIdxs.data(), (int)gpuIdxs.size(), nnXLen, nnYLen, precisionMode, &context->ctx),
    "creating compute context");
  return context.release();
}

// kmx_handle_create / kmx_batcher_create through `create(ctx)`; a context that asks for fp32 and a net that mode does not serve
// (KMX_ERR_UNSUPPORTED) retry on the context's 16-bit twin.
template <class F>
static void createWithPrecisionFallback(ComputeContext* context, Logger* logger, const string& modelName, const char* what, F create) {
  int rc = create(context->ctx);
  if(rc == KMX_ERR_UNSUPPORTED && context->precisionMode == KMX_PREC_FP32) {
    const string why = lastError();
    {
      std::lock_guard<std::mutex> lock(context->fallbackMutex);
      if(context->fallbackCtx == NULL)
        check(
          kmx_context_create(context->gpuIdxs.data(), (int)context->gpuIdxs.size(), context->nnXLen, context->nnYLen, KMX_PREC_AUTO, &context->fallbackCtx),
          "creating the 16-bit fallback context");
    }
    const string msg = "katamx backend: WARNING: useFP16 = false is not served for model " + modelName + " (" + why +
                       "); falling back to the backend's 16-bit default for it";
    if(logger != NULL) logger->write(msg);
    else cerr << msg << endl;
    rc = create(context->fallbackCtx);
  }
  check(rc, what);
}
void NeuralNet::freeComputeContext(ComputeContext* computeContext) {
  if(computeContext == NULL)
    return;
  kmx_context_free(computeContext->ctx);
  if(computeContext->fallbackCtx != NULL)
    kmx_context_free(computeContext->fallbackCtx);
  delete computeContext;
}

ComputeHandle* NeuralNet::createComputeHandle(
  ComputeContext* context,
  const LoadedModel* loadedModel,
  Logger* logger,
  int maxBatchSize,
  bool requireExactNNLen,
  bool inputsUseNHWC,
  int gpuIdxForThisThread,
  int serverThreadIdx
) {
  std::unique_ptr<ComputeHandle> handle(new ComputeHandle());
  handle->context = context;
  handle->loadedModel = loadedModel;
  handle->inputsUseNHWC = inputsUseNHWC;
  handle->modelVersion = loadedModel->modelDesc.modelVersion;
  handle->numInputChannels = loadedModel->modelDesc.numInputChannels;
  handle->numInputGlobalChannels = loadedModel->modelDesc.numInputGlobalChannels;
  handle->numInputMetaChannels = loadedModel->modelDesc.numInputMetaChannels;
  handle->gpuIdx = gpuIdxForThisThread;
  if(context->useBatcher) {
    std::lock_guard<std::mutex> lock(context->batcherMutex);
    ComputeContext::SharedBatcher& sb = context->batchers[std::make_pair(loadedModel, gpuIdxForThisThread)];
    if(sb.batcher == NULL) {
      createWithPrecisionFallback(context, logger, loadedModel->modelDesc.name, "creating the leaf batcher", [&](kmx_context* c) {
        return kmx_batcher_create(c, loadedModel->model, maxBatchSize, context->batcherInFlight, gpuIdxForThisThread, &sb.batcher);
      });
      sb.maxBatchSize = maxBatchSize;
    }
    sb.users++;
    handle->batcher = sb.batcher;
  }
  else
    createWithPrecisionFallback(context, logger, loadedModel->modelDesc.name, "creating compute handle", [&](kmx_context* c) {
      return kmx_handle_create(c, loadedModel->model, maxBatchSize, requireExactNNLen ? 1 : 0, gpuIdxForThisThread, &handle->handle);
    });
  if(logger != NULL) {
    int prec = handle->batcher != NULL ? kmx_batcher_precision(handle->batcher) : kmx_handle_precision(handle->handle);
    logger->write(
      "katamx (HIP/gfx950) backend thread " + Global::intToString(serverThreadIdx) + ": device " +
      Global::intToString(gpuIdxForThisThread) + " [" + deviceLabel(gpuIdxForThisThread) + "] precision " + precisionName(prec) +
      (handle->batcher != NULL ? " (shared leaf batcher)" : "") + " model " + loadedModel->modelDesc.name);
  }
  return handle.release();
}
void NeuralNet::freeComputeHandle(ComputeHandle* handle) {
  if(handle == NULL)
    return;
  if(handle->batcher != NULL) {
    ComputeContext* context = const_cast<ComputeContext*>(handle->context);
    std::lock_guard<std::mutex> lock(context->batcherMutex);
    auto it = context->batchers.find(std::make_pair(handle->loadedModel, handle->gpuIdx));
    if(it != context->batchers.end() && --it->second.users == 0) {
      kmx_batcher_free(it->second.batcher);
      context->batchers.erase(it);
    }
  }
  else
    kmx_handle_free(handle->handle);
  delete handle;
}

bool NeuralNet::isUsingFP16(const ComputeHandle* handle) {
  return (handle->batcher != NULL ? kmx_batcher_precision(handle->batcher) : kmx_handle_precision(handle->handle)) != KMX_PREC_FP32;
}
bool NeuralNet::setIsWarmup(const ComputeHandle* handle, bool isWarmup) {
  (void)handle;
  (void)isWarmup;
  return false;
}

InputBuffers* NeuralNet::createInputBuffers(const LoadedModel* loadedModel, int maxBatchSize, int nnXLen, int nnYLen) {
  const ModelDesc& m = loadedModel->modelDesc;
  InputBuffers* buffers = new InputBuffers();
  buffers->maxBatchSize = maxBatchSize;
  buffers->singleInputElts = (size_t)m.numInputChannels * nnXLen * nnYLen;
  buffers->singlePolicyElts = (size_t)nnXLen * nnYLen + 1;
  buffers->singleOwnershipElts = (size_t)nnXLen * nnYLen;
  buffers->rowSpatial.resize(maxBatchSize);
  buffers->rowGlobal.resize(maxBatchSize);
  buffers->rowMeta.resize(maxBatchSize);
  buffers->symmetry.resize(maxBatchSize);
  buffers->policyOptimism.resize(maxBatchSize);
  buffers->outPolicy.resize(maxBatchSize);
  buffers->outOwnership.resize(maxBatchSize);
  buffers->value.resize((size_t)maxBatchSize * 3);
  buffers->score.resize((size_t)maxBatchSize * 6);
  return buffers;
}
void NeuralNet::freeInputBuffers(InputBuffers* buffers) {
  delete buffers;
}

void NeuralNet::getOutput(
  ComputeHandle* handle,
  InputBuffers* buffers,
  int numBatchEltsFilled,
  NNResultBuf** inputBufs,
  vector<NNOutput*>& outputs
) {
  const int batchSize = numBatchEltsFilled;
  testAssert(batchSize > 0 && batchSize <= buffers->maxBatchSize);
  testAssert((int)outputs.size() == batchSize);
  const int nnXLen = handle->context->nnXLen;
  const int nnYLen = handle->context->nnYLen;
  const int cIn = handle->numInputChannels;
  const int area = nnXLen * nnYLen;

  if(!handle->inputsUseNHWC)
    buffers->nhwcScratch.resize((size_t)batchSize * buffers->singleInputElts);

  for(int row = 0; row < batchSize; row++) {
    const NNResultBuf* in = inputBufs[row];
    // same pairing as eigenbackend.cpp:2474-2484: metadata rows exactly for nets with an sgf-metadata encoder
    testAssert(in->hasRowMeta == (handle->numInputMetaChannels > 0));
    buffers->rowMeta[row] = in->hasRowMeta ? in->rowMetaBuf.data() : NULL;
    NNOutput* out = outputs[row];
    testAssert(out->nnXLen == nnXLen && out->nnYLen == nnYLen);
    if(handle->inputsUseNHWC)
      buffers->rowSpatial[row] = in->rowSpatialBuf.data();
    else {
      // NCHW hand-over (only if a config forces inputsUseNHWC=false): transpose on the host.
      float* dst = buffers->nhwcScratch.data() + (size_t)row * buffers->singleInputElts;
      const float* src = in->rowSpatialBuf.data();
      for(int c = 0; c < cIn; c++)
        for(int p = 0; p < area; p++)
          dst[(size_t)p * cIn + c] = src[(size_t)c * area + p];
      buffers->rowSpatial[row] = dst;
    }
    buffers->rowGlobal[row] = in->rowGlobalBuf.data();
    buffers->symmetry[row] = in->symmetry;
    buffers->policyOptimism[row] = (float)in->policyOptimism;
    buffers->outPolicy[row] = out->policyProbs;
    buffers->outOwnership[row] = out->whiteOwnerMap;  // NULL => skipped
  }

  if(handle->batcher != NULL) {
    // the rows join whatever batch is filling (other server threads' rows included); every ticket is waited for even after a
    // failure, so that no staging slot stays occupied
    buffers->tickets.resize(batchSize);
    int firstError = KMX_OK;
    string firstMessage;
    int submitted = 0;
    for(; submitted < batchSize; submitted++) {
      int rc = kmx_batcher_submit(
        handle->batcher, buffers->rowSpatial[submitted], buffers->rowGlobal[submitted],
        handle->numInputMetaChannels > 0 ? buffers->rowMeta[submitted] : NULL, buffers->symmetry[submitted],
        buffers->policyOptimism[submitted], buffers->outPolicy[submitted], buffers->value.data() + (size_t)submitted * 3,
        buffers->score.data() + (size_t)submitted * 6, buffers->outOwnership[submitted], &buffers->tickets[submitted]);
      if(rc != KMX_OK) {
        firstError = rc;
        firstMessage = kmx_last_error();
        break;
      }
    }
    for(int row = 0; row < submitted; row++) {
      int rc = kmx_batcher_wait(handle->batcher, buffers->tickets[row]);
      if(rc != KMX_OK && firstError == KMX_OK) {
        firstError = rc;
        firstMessage = kmx_last_error();
      }
    }
    if(firstError != KMX_OK)
      throw StringError("katamx backend: evaluating batch through the leaf batcher: " + firstMessage);
  }
  else
    check(
      kmx_eval_meta(
        handle->handle, batchSize, buffers->rowSpatial.data(), buffers->rowGlobal.data(),
        handle->numInputMetaChannels > 0 ? buffers->rowMeta.data() : NULL, buffers->symmetry.data(),
        buffers->policyOptimism.data(), buffers->outPolicy.data(), buffers->value.data(), buffers->score.data(),
        buffers->outOwnership.data()),
      "evaluating batch");

  // Scalars -> NNOutput, exactly the field mapping of eigenbackend.cpp:2569-2626.
  const int modelVersion = handle->modelVersion;
  for(int row = 0; row < batchSize; row++) {
    NNOutput* out = outputs[row];
    const float* v = buffers->value.data() + (size_t)row * 3;
    const float* s = buffers->score.data() + (size_t)row * 6;
    out->whiteWinProb = v[0];
    out->whiteLossProb = v[1];
    out->whiteNoResultProb = v[2];
    out->whiteScoreMean = s[0];
    out->whiteScoreMeanSq = s[1];
    out->whiteLead = s[2];
    out->varTimeLeft = s[3];
    if(modelVersion >= 9) {
      out->shorttermWinlossError = s[4];
      out->shorttermScoreError = s[5];
    }
    else {
      out->shorttermWinlossError = 0;
      out->shorttermScoreError = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Leaf ports (katamx_leaf.h): rows handed over by the thread that owns the leaf, results collected by ticket.
struct KatamxLeaf::Port {
  const ComputeContext* context = NULL;
  const LoadedModel* loadedModel = NULL;
  int numInputMetaChannels = 0;
  kmx_batcher* batcher = NULL;
};

KatamxLeaf::Port* KatamxLeaf::openPort(
  ComputeContext* context, const LoadedModel* loadedModel, Logger* logger, int maxBatchSize, int batchesInFlight, int gpuIdx
) {
  std::unique_ptr<Port> port(new Port());
  port->context = context;
  port->loadedModel = loadedModel;
  port->numInputMetaChannels = loadedModel->modelDesc.numInputMetaChannels;
  createWithPrecisionFallback(context, logger, loadedModel->modelDesc.name, "creating the leaf batcher", [&](kmx_context* c) {
    return kmx_batcher_create(c, loadedModel->model, maxBatchSize, batchesInFlight, gpuIdx, &port->batcher);
  });
  if(logger != NULL) {
    logger->write(
      "katamx (HIP/gfx950) leaf port: device " + Global::intToString(gpuIdx < 0 ? 0 : gpuIdx) + " [" + deviceLabel(gpuIdx) + "] precision " +
      precisionName(kmx_batcher_precision(port->batcher)) + " batch " + Global::intToString(maxBatchSize) + " model " + loadedModel->modelDesc.name);
  }
  return port.release();
}
void KatamxLeaf::closePort(Port* port) {
  if(port == NULL)
    return;
  kmx_batcher_free(port->batcher);
  delete port;
}
bool KatamxLeaf::isUsingFP16(const Port* port) {
  return kmx_batcher_precision(port->batcher) != KMX_PREC_FP32;
}
uint64_t KatamxLeaf::submit(
  Port* port, const float* rowSpatial, const float* rowGlobal, const float* rowMeta, int symmetry, float policyOptimism,
  float* outPolicy, float* outValue, float* outScore, float* outOwnership
) {
  testAssert((rowMeta != NULL) == (port->numInputMetaChannels > 0));
  uint64_t ticket = 0;
  check(
    kmx_batcher_submit(port->batcher, rowSpatial, rowGlobal, rowMeta, symmetry, policyOptimism, outPolicy, outValue, outScore, outOwnership, &ticket),
    "submitting a leaf");
  return ticket;
}
uint64_t KatamxLeaf::submitPacked(
  Port* port, const uint8_t* rowPacked, int numPlanes, const float* rowGlobal, const float* rowMeta, int symmetry, float policyOptimism,
  float* outPolicy, float* outValue, float* outScore, float* outOwnership
) {
  testAssert((rowMeta != NULL) == (port->numInputMetaChannels > 0));
  testAssert(numPlanes == port->loadedModel->modelDesc.numInputChannels);
  uint64_t ticket = 0;
  check(
    kmx_batcher_submit_packed(port->batcher, rowPacked, rowGlobal, rowMeta, symmetry, policyOptimism, outPolicy, outValue, outScore, outOwnership, &ticket),
    "submitting a leaf");
  return ticket;
}
void KatamxLeaf::wait(Port* port, uint64_t ticket) {
  check(kmx_batcher_wait(port->batcher, ticket), "evaluating a leaf");
}
void KatamxLeaf::stats(Port* port, uint64_t& rows, uint64_t& batches) {
  check(kmx_batcher_stats(port->batcher, &rows, &batches), "reading the leaf batcher's counters");
}

// ---------------------------------------------------------------------------------------------
// Test hooks. The reference hands over NCHW or NHWC fp32 vectors; the ABI is NHWC only.

static vector<float> toNHWC(const vector<float>& v, int n, int c, int h, int w, bool isNHWC) {
  if(isNHWC)
    return v;
  vector<float> r(v.size());
  for(int b = 0; b < n; b++)
    for(int ch = 0; ch < c; ch++)
      for(int p = 0; p < h * w; p++)
        r[((size_t)b * h * w + p) * c + ch] = v[((size_t)b * c + ch) * h * w + p];
  return r;
}
static vector<float> fromNHWC(const vector<float>& v, int n, int c, int h, int w, bool wantNHWC) {
  if(wantNHWC)
    return v;
  vector<float> r(v.size());
  for(int b = 0; b < n; b++)
    for(int ch = 0; ch < c; ch++)
      for(int p = 0; p < h * w; p++)
        r[((size_t)b * c + ch) * h * w + p] = v[((size_t)b * h * w + p) * c + ch];
  return r;
}
static kmx_conv_desc convDesc(const ConvLayerDesc& d) {
  kmx_conv_desc c;
  c.conv_y_size = d.convYSize;
  c.conv_x_size = d.convXSize;
  c.in_channels = d.inChannels;
  c.out_channels = d.outChannels;
  c.weights = d.weights.data();
  return c;
}
static kmx_bnact_desc bnDesc(const BatchNormLayerDesc& d, int activation) {
  kmx_bnact_desc b;
  b.num_channels = d.numChannels;
  b.activation = activation;
  b.merged_scale = d.mergedScale.data();
  b.merged_bias = d.mergedBias.data();
  return b;
}
static kmx_matmul_desc matmulDesc(const MatMulLayerDesc& d) {
  kmx_matmul_desc m;
  m.in_channels = d.inChannels;
  m.out_channels = d.outChannels;
  m.weights = d.weights.data();
  return m;
}
static bool precisionSupported(bool useFP16, int& mode) { mode = useFP16 ? KMX_PREC_AUTO : KMX_PREC_FP32; return true; }
static bool hookResult(int status, const char* what) {
  if(status == KMX_ERR_UNSUPPORTED)
    return false;
  check(status, what);
  return true;
}

bool NeuralNet::testEvaluateConv(
  const ConvLayerDesc* desc, int batchSize, int nnXLen, int nnYLen, bool useFP16, bool useNHWC,
  const vector<float>& inputBuffer, vector<float>& outputBuffer
) {
  int mode;
  if(!precisionSupported(useFP16, mode))
    return false;
  kmx_conv_desc c = convDesc(*desc);
  vector<float> in = toNHWC(inputBuffer, batchSize, desc->inChannels, nnYLen, nnXLen, useNHWC);
  vector<float> out((size_t)batchSize * nnXLen * nnYLen * desc->outChannels);
  int status = kmx_test_conv(&c, batchSize, nnXLen, nnYLen, mode, in.data(), out.data());
  if(!hookResult(status, "testEvaluateConv"))
    return false;
  outputBuffer = fromNHWC(out, batchSize, desc->outChannels, nnYLen, nnXLen, useNHWC);
  return true;
}

bool NeuralNet::testEvaluateBatchNorm(
  const BatchNormLayerDesc* desc, int batchSize, int nnXLen, int nnYLen, bool useFP16, bool useNHWC,
  const vector<float>& inputBuffer, const vector<float>& maskBuffer, vector<float>& outputBuffer
) {
  int mode;
  if(!precisionSupported(useFP16, mode))
    return false;
  kmx_bnact_desc b = bnDesc(*desc, ACTIVATION_IDENTITY);
  vector<float> in = toNHWC(inputBuffer, batchSize, desc->numChannels, nnYLen, nnXLen, useNHWC);
  vector<float> out(in.size());
  int status = kmx_test_bnact(&b, batchSize, nnXLen, nnYLen, mode, in.data(), maskBuffer.data(), out.data());
  if(!hookResult(status, "testEvaluateBatchNorm"))
    return false;
  outputBuffer = fromNHWC(out, batchSize, desc->numChannels, nnYLen, nnXLen, useNHWC);
  return true;
}

bool NeuralNet::testEvaluateResidualBlock(
  const ResidualBlockDesc* desc, int batchSize, int nnXLen, int nnYLen, bool useFP16, bool useNHWC,
  const vector<float>& inputBuffer, const vector<float>& maskBuffer, vector<float>& outputBuffer
) {
  int mode;
  if(!precisionSupported(useFP16, mode))
    return false;
  kmx_resblock_desc r;
  r.pre_bn = bnDesc(desc->preBN, desc->preActivation.activation);
  r.regular_conv = convDesc(desc->regularConv);
  r.mid_bn = bnDesc(desc->midBN, desc->midActivation.activation);
  r.final_conv = convDesc(desc->finalConv);
  const int c = desc->preBN.numChannels;
  vector<float> in = toNHWC(inputBuffer, batchSize, c, nnYLen, nnXLen, useNHWC);
  vector<float> out(in.size());
  int status = kmx_test_resblock(&r, batchSize, nnXLen, nnYLen, mode, in.data(), maskBuffer.data(), out.data());
  if(!hookResult(status, "testEvaluateResidualBlock"))
    return false;
  outputBuffer = fromNHWC(out, batchSize, c, nnYLen, nnXLen, useNHWC);
  return true;
}

bool NeuralNet::testEvaluateGlobalPoolingResidualBlock(
  const GlobalPoolingResidualBlockDesc* desc, int batchSize, int nnXLen, int nnYLen, bool useFP16, bool useNHWC,
  const vector<float>& inputBuffer, const vector<float>& maskBuffer, vector<float>& outputBuffer
) {
  int mode;
  if(!precisionSupported(useFP16, mode))
    return false;
  kmx_gpoolblock_desc g;
  g.pre_bn = bnDesc(desc->preBN, desc->preActivation.activation);
  g.regular_conv = convDesc(desc->regularConv);
  g.gpool_conv = convDesc(desc->gpoolConv);
  g.gpool_bn = bnDesc(desc->gpoolBN, desc->gpoolActivation.activation);
  g.gpool_to_bias_mul = matmulDesc(desc->gpoolToBiasMul);
  g.mid_bn = bnDesc(desc->midBN, desc->midActivation.activation);
  g.final_conv = convDesc(desc->finalConv);
  const int c = desc->preBN.numChannels;
  vector<float> in = toNHWC(inputBuffer, batchSize, c, nnYLen, nnXLen, useNHWC);
  vector<float> out(in.size());
  int status = kmx_test_gpoolblock(&g, batchSize, nnXLen, nnYLen, mode, in.data(), maskBuffer.data(), out.data());
  if(!hookResult(status, "testEvaluateGlobalPoolingResidualBlock"))
    return false;
  outputBuffer = fromNHWC(out, batchSize, c, nnYLen, nnXLen, useNHWC);
  return true;
}

// engine.h — per-GPU executor of one KataGo net: owns the device copy of the weights (re-tiled for the
// MFMA kernel), the activation buffers for maxBatch boards, and the static launch schedule.
//
// One Engine == one reference ComputeHandle (+ its InputBuffers): cpp/neuralnet/nninterface.h:76-99.
// The schedule it runs is Model::apply of the reference (eigenbackend.cpp:2162-2216) with every
// BatchNorm/activation/mask/bias/residual fused into the epilogue of the producing convolution.
#ifndef KMX_ENGINE_H_
#define KMX_ENGINE_H_

#include <hip/hip_runtime.h>

#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/katamx.h"
#include "kernels.h"
#include "model_desc.h"

namespace kmx {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
void hipCheck(hipError_t e, const char* what);

constexpr size_t DEVBUF_TAIL = DEVBUF_TAIL_BYTES;  // kernels.h: what the kernels' run-ahead is checked against
constexpr size_t MAX_GRAPHS = 48;  // captured schedules kept per engine (least recently used evicted)

// RAII device allocation
class DevBuf {
 public:
  DevBuf() : p_(nullptr), bytes_(0) {}
  explicit DevBuf(size_t bytes, bool zero = true);
  ~DevBuf();
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept;
  void* get() const { return p_; }
  template <class T> T* as() const { return (T*)p_; }
  size_t bytes() const { return bytes_; }
  void upload(const void* src, size_t bytes);

 private:
  void* p_;
  size_t bytes_;
};

inline int roundUp(int x, int m) { return (x + m - 1) / m * m; }

// Device-resident weights of one fused convolution (several reference convs concatenated along Cout).
struct FusedConv {
  int ks = 1;
  int cin = 0;       // real input channels
  int nChunks = 0;   // ceil(cin/32)
  int cout = 0;      // channels in the fused output space (segments padded to multiples of 4)
  int coutPad = 0;   // multiple of 64
  DevBuf w;          // T[nChunks][ks*ks][coutPad][32], the four 8-value slots of a row XOR-swizzled (kernels.h)
  DevBuf wFrag;      // 3x3, 16-bit: the same weights in MFMA-fragment order (ConvArgs::wFrag); empty otherwise
  DevBuf scale, bias;  // float[coutPad]; zero outside act segments
  double macPerCell = 0;  // real MACs per board cell (for flop accounting)
};
struct ConvSegment {
  const ConvDesc* conv;
  const BnDesc* bn;  // null => raw segment
};
FusedConv buildFusedConv(int dtype, const std::vector<ConvSegment>& segs, std::vector<int>* segOffsets);

class Engine {
 public:
  Engine(const ModelDesc& model, int nnXLen, int nnYLen, int maxBatch, int dtype, int device);
  bool usesScale8() const { return scale8_; }
  ~Engine();

  int dtype() const { return dtype_; }
  int maxBatch() const { return maxBatch_; }
  int nnXLen() const { return X_; }
  int nnYLen() const { return Y_; }
  int numInputChannels() const { return cin_; }
  int numInputGlobalChannels() const { return gin_; }
  hipStream_t stream() const { return stream_; }
  int device() const { return device_; }

  // Host-buffer entry (kmx_eval). Synchronous. = evalHostBegin + evalHostFinish; the two halves exist so that a handle
  // that splits a batch over two engines can have both halves in flight.
  void evalHost(int n, const float* const* rowSpatial, const float* const* rowGlobal, const float* const* rowMeta,
                const int* symmetry, const float* policyOptimism, float* const* outPolicy, float* outValue, float* outScore,
                float* const* outOwnership);
  // rowPacked (bit planes, kernels.h InputArgs::packed) replaces rowSpatial when it is given
  void evalHostBegin(int n, const float* const* rowSpatial, const unsigned char* const* rowPacked, const float* const* rowGlobal,
                     const float* const* rowMeta, const int* symmetry, const float* policyOptimism, float* const* outOwnership);
  int packedRowBytes() const { return cin_ * ((S_ + 7) / 8); }
  void evalHostFinish(int n, float* const* outPolicy, float* outValue, float* outScore, float* const* outOwnership);
  // Device-buffer entry (kmx_eval_device).
  void evalDevice(int n, const float* dSpatial, const float* dGlobal, const float* dMeta, const int* symmetry,
                  const float* policyOptimism, float* dPolicy, float* dValue, float* dScore, float* dOwnership, bool sync);
  int numInputMetaChannels() const { return min_; }
  void sync();
  bool idle();  // has everything queued on this engine's stream completed? (never blocks)
  // Staged entry (the batcher, batcher.cpp): rows are written by their submitters straight into this engine's pinned staging
  // - bit-packed planes, globals, metadata - then ONE call enqueues H2D, the pass and D2H without waiting; after sync() the
  // results sit in pinned memory. The engine must be idle when rows are staged (the batcher's slot state machine sees to it).
  unsigned char* stagedPackedRow(int i) { return hPacked_ + (size_t)i * packedRowBytes(); }
  float* stagedGlobalRow(int i) { return hGlobal_ + (size_t)i * gin_; }
  float* stagedMetaRow(int i) { return hMeta_ + (size_t)i * min_; }
  void launchStagedPacked(int n, const int* symmetry, const float* policyOptimism, bool anyOwner);
  const float* stagedPolicy(int i) const { return hPolicy_ + (size_t)i * (S_ + 1); }
  const float* stagedValue(int i) const { return hValue_ + (size_t)i * 3; }
  const float* stagedScore(int i) const { return hScore_ + (size_t)i * 6; }
  const float* stagedOwnership(int i) const { return hOwnership_ + (size_t)i * S_; }

  void setProfiling(bool enabled);
  struct ProfileEntry { std::string name; uint64_t launches = 0; double ms = 0, flops = 0, bytes = 0; };
  std::vector<ProfileEntry> getProfile();  // synchronises

  // The convolution work-group shape is chosen for `rows * scale` boards: a handle that runs two engines side by side
  // on two streams sets 2, so that each half still uses the 8-wave shape (the other half fills the rest of the chip).
  void setConcurrency(int scale) { cfgScale_ = scale < 1 ? 1 : scale; }
  int shapeRows(int n) const { return n * cfgScale_; }  // what the work-group shapes are chosen for: the boards of all concurrent parts
  // Other engines' passes run on the same device at the same time (the batcher's batches in flight): kernels that exist in a
  // "chip to itself" and a "side by side" form (the seam, kernels.h PwPairArgs::alone) take the latter.
  void setSharesDevice(bool shares) { sharesDevice_ = shares; }
  // Record `ev` on this engine's stream after the first `afterOps` launches of the next pass (0: at the top of runSchedule, i.e.
  // after the row parameters were staged and before the first launch); null clears it. A second engine's stream waits for it (kmx_api.cpp, split handle).
  void setForkPoint(int afterOps, hipEvent_t ev) { forkOps_ = afterOps < 0 ? 0 : afterOps; forkEv_ = ev; }
  // hipGraph replay of the launch schedule: a pass with the same row count, work-group shapes and buffer
  // pointers as an earlier one is captured once (on its second occurrence: the first runs directly and sets the kernels'
  // LDS attributes) and then replayed with one hipGraphLaunch instead of ~130 kernel launches. Same kernels, same
  // arguments, same order: bit-identical results. Off while profiling (per-launch events) or staggering.
  void setGraphs(bool enabled) { useGraphs_ = enabled; }
  bool graphs() const { return useGraphs_; }
  uint64_t graphLaunches() const { return graphLaunches_; }
  uint64_t rowsProcessed() const { return rows_; }
  uint64_t batchesProcessed() const { return batches_; }
  int numLaunchesPerEval() const { return (int)ops_.size(); }

 private:
  struct Stream {  // a residual stream: raw values and their next-BN-activated image
    void* raw;
    void* act;
    int stride;
  };
  struct Op {
    std::function<void(int, hipStream_t)> fn;
    int cls;               // index into opClasses_
    double flopsPerRow;    // algorithmic flops per evaluated position
    double bytesPerRow;    // algorithmic HBM bytes per evaluated position
    // an op that takes another form below a batch size (the seam: two plain convolution launches below fuseMinRows_) is accounted
    // as that form there: its own class, its own byte model, its number of launches
    int smallBelow = 0;    // rows; 0 = one form only
    int clsSmall = -1;
    double bytesPerRowSmall = 0.0;
    int launchesSmall = 1;
    int launches = 1;      // what one execution of the main form counts as in the profile (a chain of k convolutions counts k)
  };
  void construct(const ModelDesc& model);  // the body of the constructor
  bool scale8_ = false;  // the net runs at 1/8 of its values (fp16 range transform)
  void destroy() noexcept;                 // everything the destructor releases; also run when construct() throws
  int opClass(const std::string& name);
  void addOp(const std::string& cls, double flopsPerRow, double bytesPerRow, std::function<void(int, hipStream_t)> fn);
  void collectProfile();

  void buildSchedule(const ModelDesc& m);
  void buildStack(const std::vector<BlockDesc>& blocks, const Stream& s, const BnDesc* bnAfter, int depth);
  void addConv(const FusedConv* fc, const void* in, int inStride, const float* ncBias, int ncBiasStride,
               const void* resid, int residStride, void* rawOut, int rawStride, int rawBegin, int rawEnd, void* actOut,
               int actStride, int actBegin, int actEnd, int actKind);
  ConvArgs makeConvArgs(const FusedConv* fc, const void* in, int inStride, const float* ncBias, int ncBiasStride, const void* resid,
                        int residStride, void* rawOut, int rawStride, int rawBegin, int rawEnd, void* actOut, int actStride,
                        int actBegin, int actEnd, int actKind, double* bytesPerRow);
  void launchConvOp(const ConvArgs& a, int ks, int n, hipStream_t st);
  void addSeam(const ConvDesc& post, const void* in, int inStride, const Stream& s, const BnDesc& nextBN, const ConvDesc& pre,
               const Stream& mid, const BnDesc& innerBN);
  const FusedConv* newConv(const std::vector<ConvSegment>& segs, std::vector<int>* offs = nullptr);
  void addResidualConv(const ConvDesc& conv, const void* in, int inStride, const Stream& s, const BnDesc* nextBN);
  // blocks[i] (and blocks[i + 1]) as one chained launch where the shape allows it (conv_chain_kernel.h); returns the number of blocks consumed (0: none)
  int addOrdinaryChain(const std::vector<BlockDesc>& blocks, size_t i, const Stream& s, const BnDesc* bnAfter);
  void addRmsNorm(const void* in, int inStride, void* out, int outStride, int C, float eps, const std::vector<float>& w,
                  const std::vector<float>* beta, int actKind, bool perBoard);
  float* uploadFloats(const std::vector<float>& v);
  void stageRowParams(int n, const int* symmetry, const float* policyOptimism);
  void runSchedule(int n, const float* dSpatial, const unsigned char* dPacked, const float* dGlobal, const float* dMeta,
                   float* dPolicy, float* dValue, float* dScore, float* dOwnership);

  int dtype_, device_, X_, Y_, S_, maxBatch_;
  hipStream_t stream_;
  std::vector<std::unique_ptr<FusedConv>> convs_;
  std::vector<std::unique_ptr<DevBuf>> params_;  // small fp32 parameter arrays
  std::vector<std::unique_ptr<DevBuf>> acts_;    // activation buffers
  std::vector<Op> ops_;

  DevBuf zeroPage_;
  DevBuf inputT_, mask_, maskSum_, ncBias_;
  DevBuf dSymmetry_, dOptimism_;
  DevBuf dSpatialIn_, dGlobalIn_, dMetaIn_, dPackedIn_;  // staging for the host entry
  DevBuf dPolicy_, dValue_, dScore_, dOwnership_, polFeat_;
  DevBuf boardRms_;  // [maxBatch] per-board 1/rms of a spatial RMSNorm trunk tip
  // pinned host staging
  float* hSpatial_ = nullptr;
  float* hGlobal_ = nullptr;
  float* hMeta_ = nullptr;
  unsigned char* hPacked_ = nullptr;
  float* hPolicy_ = nullptr;
  float* hValue_ = nullptr;
  float* hScore_ = nullptr;
  float* hOwnership_ = nullptr;
  int* hSymmetry_ = nullptr;    // two slots of maxBatch: a call never waits for the previous call's copies
  float* hOptimism_ = nullptr;
  hipEvent_t stagingDone_[2] = {nullptr, nullptr};
  int stagingSlot_ = 0;
  bool hostAnyOwner_ = false;
  int cfgScale_ = 1;
  bool sharesDevice_ = false;
  bool fuseSeams_ = true;   // KMX_FUSE_SEAMS=0: always the two convolution launches
  bool packInputs_ = false;  // KMX_PACK_INPUTS=1: kmx_eval bit-packs 0/1 planes while staging. Off: measured on MI355X (b18c384nbt, batch 256,
                            // synchronous host entry) 32.9 k evals/s with it against 39.2 k without - the caller's thread packs while the GPU idles,
                            // and that costs more than the 0.3 ms of PCIe it saves; the batcher packs on the submitters' threads instead
  int fuseMinRows_ = 24;    // KMX_FUSE_MIN_ROWS: smallest batch that takes the fused seam kernel
  // KMX_CONV_CHAIN = 0 | 2 | 4: the longest chain of 3x3 192 -> 192 convolutions that runs as one launch with the activated image handed
  // over in LDS (conv_chain_kernel.h) when the batch takes the one-work-group-per-board shape; 0 = always separate launches
  int maxChain_ = 4;
  // (Round 4 measured the seam below that threshold as the 4-wave x 64-cell one-tile seam kernel instead of two convolution launches:
  // within the noise of the scan, -4 % per pass at batch 1, +3 % at batch 8 - removed; profiles/r04_steps/call1/small_batch_scan.txt.)
  int forkOps_ = 0;
  hipEvent_t forkEv_ = nullptr;

  // captured schedules
  struct GraphKey {
    int n, scale;
    const void* p[8];
    bool operator<(const GraphKey& o) const {
      if(n != o.n) return n < o.n;
      if(scale != o.scale) return scale < o.scale;
      for(int i = 0; i < 8; i++)
        if(p[i] != o.p[i]) return p[i] < o.p[i];
      return false;
    }
  };
  struct GraphEntry {
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    uint64_t lastUse = 0;
  };
  std::map<GraphKey, GraphEntry> graphCache_;
  bool useGraphs_ = false;  // opt-in (KMX_GRAPHS=1 / kmx_handle_set_graphs): measured on MI355X, replay is not faster than direct launches
  uint64_t graphLaunches_ = 0, graphClock_ = 0;
  void launchOps(int n);   // the ops of one pass, directly on stream_
  void dropGraphs() noexcept;
  int cin_ = 0, gin_ = 0, min_ = 0;  // spatial, global, sgf-metadata input channels

  // pointers the ops read at run time (set by runSchedule)
  const float* curSpatial_ = nullptr;
  const float* curGlobal_ = nullptr;
  const float* curMeta_ = nullptr;
  const unsigned char* curPacked_ = nullptr;
  float* curPolicy_ = nullptr;
  float* curValue_ = nullptr;
  float* curScore_ = nullptr;
  float* curOwnership_ = nullptr;

  uint64_t rows_ = 0, batches_ = 0;

  // profiling state
  bool profiling_ = false;
  std::vector<std::string> opClasses_;
  std::vector<ProfileEntry> profile_;
  struct Pending { hipEvent_t a, b; int cls; int launches; double flops, bytes; };
  std::vector<Pending> pending_;
  std::vector<hipEvent_t> eventPool_;
};

// Layer test hooks (nninterface.h:134-180) executed with the same kernels as the full net.
void testConv(int dtype, const kmx_conv_desc* d, int batch, int X, int Y, const float* in, float* out);
void testBnAct(int dtype, const kmx_bnact_desc* d, int batch, int X, int Y, const float* in, const float* mask, float* out);
void testResBlock(int dtype, const kmx_resblock_desc* d, int batch, int X, int Y, const float* in, const float* mask, float* out);
void testGPoolBlock(int dtype, const kmx_gpoolblock_desc* d, int batch, int X, int Y, const float* in, const float* mask,
                    float* out);

// One row of fp32 NHWC binary feature planes -> bit planes [C][ceil(S/8)], MSB first (packBits of dataio/trainingwrite.cpp:314-337,
// plane by plane): the host half of row f1. Returns false if a value is neither 0 nor 1 (the row is then packed as garbage).
bool packRowNHWC(const float* row, int S, int C, unsigned char* out);

void testPointwisePair(int dtype, int batch, int X, int Y, int c1, int c2, int c3, const float* in, const float* resid, const float* w1,
                       const float* scale1, const float* bias1, int act1, const float* w2, const float* scale2, const float* bias2,
                       int act2, const float* mask, bool fused, float* outTrunkRaw, float* outMidRaw, float* outMidAct);
void testConvChain(int dtype, int batch, int X, int Y, int nConv, const float* x, const float* r, const float* w, const float* scale,
                   const float* bias, int act, const float* mask, int chained, float* outR, float* outX);
// Unit hooks for the transformer kernels (experimental): fp32 NHWC in/out like the hooks above.
void testRmsNorm(int dtype, int batch, int X, int Y, int C, float eps, const float* w, const float* beta, int actKind, bool perBoard,
                 const float* in, const float* mask, float* out);
void testAttention(int dtype, int batch, int X, int Y, int H, int KVH, int QD, int VD, const float* ropeCos, const float* ropeSin,
                   int ropeHeads, const float* q, const float* k, const float* v, const float* mask, float* out);
void testSwiGlu(int dtype, int batch, int X, int Y, int F, const float* a, const float* g, float* out);

}  // namespace kmx
#endif

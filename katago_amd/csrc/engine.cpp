// engine.cpp — see engine.h. Compiled with hipcc (host code only; kernels live in the .hip files).
#include "engine.h"
#include "numa.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace kmx {

void hipCheck(hipError_t e, const char* what) {
  if(e != hipSuccess) throw Error(KMX_ERR_DEVICE, std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

// KMX_DEBUG_GUARD=1 (fault triage): every DevBuf is placed with the HIP virtual-memory API so that the first byte BEHIND
// its readable tail is unmapped - the allocation ends (up to 255 bytes of alignment slack: device pointers stay 256-byte aligned as
// hipMalloc's are) where its physical backing ends, and one more granule of address space behind it is reserved and never mapped. A
// kernel that reads or writes past DEVBUF_TAIL then takes a memory-access fault in the parity test that first runs it, instead of
// reading a neighbouring allocation unnoticed (hipMalloc hands out pieces of large mapped pools: an overrun of kilobytes is silent
// there). KMX_DEBUG_GUARD=2 guards the FRONT instead (the allocation starts where its backing starts, the granule before it is unmapped).
namespace {
int guardMode() {
  static const int m = [] { const char* e = getenv("KMX_DEBUG_GUARD"); return e ? atoi(e) : 0; }();
  return m;
}
#ifndef KMX_EMULATED_HIP
struct GuardedAlloc {
  void* base = nullptr;       // start of the reserved address range
  size_t reserved = 0;        // its size
  void* mapped = nullptr;     // start of the mapped part
  size_t mappedBytes = 0;
  hipMemGenericAllocationHandle_t handle{};
};
std::mutex guardMu;
std::map<void*, GuardedAlloc>& guardTable() { static std::map<void*, GuardedAlloc> t; return t; }

void* guardedMalloc(size_t bytes) {
  int dev = 0;
  hipCheck(hipGetDevice(&dev), "hipGetDevice");
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  hipCheck(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum), "hipMemGetAllocationGranularity");
  if(gran == 0) gran = 2u << 20;
  GuardedAlloc g;
  g.mappedBytes = (bytes + gran - 1) / gran * gran;
  g.reserved = g.mappedBytes + 2 * gran;  // an unmapped granule on either side
  hipCheck(hipMemAddressReserve(&g.base, g.reserved, gran, nullptr, 0), "hipMemAddressReserve");
  g.mapped = (char*)g.base + gran;
  hipCheck(hipMemCreate(&g.handle, g.mappedBytes, &prop, 0), "hipMemCreate");
  hipCheck(hipMemMap(g.mapped, g.mappedBytes, 0, g.handle, 0), "hipMemMap");
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  hipCheck(hipMemSetAccess(g.mapped, g.mappedBytes, &acc, 1), "hipMemSetAccess");
  // end guard: the buffer's last byte is within 255 bytes of the last mapped byte; front guard: the buffer starts the mapping
  void* p = guardMode() == 2 ? g.mapped : (void*)((char*)g.mapped + (g.mappedBytes - bytes) / 256 * 256);
  std::lock_guard<std::mutex> l(guardMu);
  guardTable()[p] = g;
  return p;
}
bool guardedFree(void* p) {
  GuardedAlloc g;
  {
    std::lock_guard<std::mutex> l(guardMu);
    auto it = guardTable().find(p);
    if(it == guardTable().end()) return false;
    g = it->second;
    guardTable().erase(it);
  }
  (void)hipDeviceSynchronize();  // hipFree synchronises; unmapping does not
  (void)hipMemUnmap(g.mapped, g.mappedBytes);
  (void)hipMemRelease(g.handle);
  (void)hipMemAddressFree(g.base, g.reserved);
  return true;
}
#else
void* guardedMalloc(size_t) { throw Error(KMX_ERR_UNSUPPORTED, "KMX_DEBUG_GUARD needs the device's virtual-memory API"); }
bool guardedFree(void*) { return false; }
#endif
void devFree(void* p) {
  if(p == nullptr) return;
  if(guardMode() != 0 && guardedFree(p)) return;
  (void)hipFree(p);
}
}  // namespace

DevBuf::DevBuf(size_t bytes, bool zero) : p_(nullptr), bytes_(bytes) {
  if(bytes == 0) return;
  // DEVBUF_TAIL readable bytes follow every allocation: the convolution's DMA pointers run up to D chunks (64 bytes
  // each) past the last channel chunk of the last cell; those requests land in a scratch area and are never used.
  if(guardMode() != 0) p_ = guardedMalloc(bytes + DEVBUF_TAIL);
  else hipCheck(hipMalloc(&p_, bytes + DEVBUF_TAIL), "hipMalloc");
  // The zero fill runs on the null stream; the engine's work runs on hipStreamNonBlocking streams, which do not
  // synchronise with it: Engine::construct ends with hipDeviceSynchronize() so that no launch can overtake a fill.
  if(zero) hipCheck(hipMemsetAsync(p_, 0, bytes + DEVBUF_TAIL, nullptr), "hipMemset");
  static const bool debugAlloc = getenv("KMX_DEBUG_ALLOC") != nullptr;  // fault triage: which buffer does an address belong to
  if(debugAlloc) fprintf(stderr, "[kmx alloc] %p .. %p (%zu + %zu bytes)\n", p_, (char*)p_ + bytes + DEVBUF_TAIL, bytes, DEVBUF_TAIL);
}
DevBuf::~DevBuf() { devFree(p_); }
DevBuf& DevBuf::operator=(DevBuf&& o) noexcept {
  if(this != &o) {
    devFree(p_);
    p_ = o.p_;
    bytes_ = o.bytes_;
    o.p_ = nullptr;
    o.bytes_ = 0;
  }
  return *this;
}
void DevBuf::upload(const void* src, size_t bytes) {
  if(bytes > bytes_) throw Error(KMX_ERR_INTERNAL, "DevBuf::upload overflow");
  if(bytes) hipCheck(hipMemcpy(p_, src, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
}

// ------------------------------------------------------------------------------------------------
// Weight re-tiling. Reference layouts: conv weights in the file are [ky][kx][ic][oc] (desc.cpp:130);
// the kernel wants, per (chunk of 32 ic, tap), rows of one output channel: T[chunk][tap][oc][32], slots swizzled.
FusedConv buildFusedConv(int dtype, const std::vector<ConvSegment>& segs, std::vector<int>* segOffsets) {
  if(segs.empty()) throw Error(KMX_ERR_INTERNAL, "buildFusedConv: no segments");
  FusedConv fc;
  const ConvDesc& first = *segs[0].conv;
  fc.cin = first.inC;
  // Segments may have different (odd, square) kernel sizes: a smaller kernel is embedded, centred, in the
  // largest one with zero taps around it - exactly the same sums (the reference's layer tests pair a 1x1
  // regular conv with a 3x3 gpool conv, cpp/tests/testnn.cpp:769-791).
  fc.ks = 1;
  for(const ConvSegment& s : segs) {
    if(s.conv->ky != s.conv->kx) throw Error(KMX_ERR_UNSUPPORTED, s.conv->name + ": non-square convolution kernels are not supported");
    fc.ks = std::max(fc.ks, s.conv->ky);
  }
  if(fc.ks != 1 && fc.ks != 3 && fc.ks != 5)
    throw Error(KMX_ERR_UNSUPPORTED, first.name + ": only 1x1, 3x3 and 5x5 convolutions are supported");
  std::vector<int> offs;
  int cout = 0;
  for(const ConvSegment& s : segs) {
    if(s.conv->inC != fc.cin) throw Error(KMX_ERR_INTERNAL, "buildFusedConv: segments disagree on input channels");
    offs.push_back(cout);
    cout += roundUp(s.conv->outC, 8);  // segment starts on 8-channel boundaries: the epilogue moves 16-byte pieces
  }
  fc.cout = cout;
  fc.coutPad = roundUp(cout, 64);  // every work-group shape (32..192 channels) that divides it is launchable
  fc.nChunks = (fc.cin + KCHUNK - 1) / KCHUNK;
  const int nt = fc.ks * fc.ks;
  // (DT_F32: the same layout with four-byte values)
  const bool f32 = dtype == DT_F32;
  std::vector<uint16_t> w(f32 ? 0 : (size_t)fc.nChunks * nt * fc.coutPad * WROW_HALFS, 0);
  std::vector<float> wf(f32 ? (size_t)fc.nChunks * nt * fc.coutPad * WROW_HALFS : 0, 0.0f);
  std::vector<float> scale(fc.coutPad, 0.0f), bias(fc.coutPad, 0.0f);
  for(size_t si = 0; si < segs.size(); si++) {
    const ConvDesc& c = *segs[si].conv;
    fc.macPerCell += (double)c.ky * c.kx * c.inC * c.outC;
    for(int chunk = 0; chunk < fc.nChunks; chunk++)
      for(int t = 0; t < nt; t++) {
        const int d = (fc.ks - c.ky) / 2;
        const int ky = t / fc.ks - d, kx = t % fc.ks - d;
        if(ky < 0 || ky >= c.ky || kx < 0 || kx >= c.kx) continue;  // zero tap of an embedded smaller kernel
        for(int oc = 0; oc < c.outC; oc++) {
          const int co = offs[si] + oc;
          const size_t row = (((size_t)chunk * nt + t) * fc.coutPad + co) * WROW_HALFS;
          for(int k = 0; k < KCHUNK; k++) {
            const int ic = chunk * KCHUNK + k;
            const int slot = (k >> 3) ^ ((co >> 2) & 3);  // the kernel's LDS swizzle, applied here so the DMA copy is linear
            if(ic >= c.inC) continue;
            if(f32) wf[row + slot * 8 + (k & 7)] = c.at(ky, kx, ic, oc);
            else w[row + slot * 8 + (k & 7)] = floatToTBits(dtype, c.at(ky, kx, ic, oc));
          }
        }
      }
    if(segs[si].bn != nullptr) {
      const BnDesc& bn = *segs[si].bn;
      if(bn.c != c.outC) throw Error(KMX_ERR_INTERNAL, "buildFusedConv: bn/conv channel mismatch");
      for(int oc = 0; oc < c.outC; oc++) {
        scale[offs[si] + oc] = bn.scale[oc];
        bias[offs[si] + oc] = bn.bias[oc];
      }
    }
  }
  if(f32) {
    fc.w = DevBuf(wf.size() * sizeof(float), false);
    fc.w.upload(wf.data(), wf.size() * sizeof(float));
  }
  else {
    fc.w = DevBuf(w.size() * sizeof(uint16_t), false);
    fc.w.upload(w.data(), w.size() * sizeof(uint16_t));
    if(fc.ks == 3) {
      // the second copy, in fragment order (ConvArgs::wFrag): row `co`, slot-swizzled 8-value group g of the row above -> tile co / 32,
      // k half g / 2, lane (co % 32) + 32 (g % 2)
      std::vector<uint16_t> wfr(w.size(), 0);
      const size_t rows = (size_t)fc.nChunks * nt * fc.coutPad;
      for(size_t r = 0; r < rows; r++) {
        const size_t slab = r / fc.coutPad, co = r % fc.coutPad;
        for(int g = 0; g < 4; g++) {
          const int slot = g ^ (int)((co >> 2) & 3);
          const size_t dst = ((((slab * (fc.coutPad / 32) + co / 32) * 2 + (size_t)(g >> 1)) * 64) + (co % 32) + 32 * (size_t)(g & 1)) * 8;
          memcpy(&wfr[dst], &w[r * WROW_HALFS + (size_t)slot * 8], 8 * sizeof(uint16_t));
        }
      }
      fc.wFrag = DevBuf(wfr.size() * sizeof(uint16_t), false);
      fc.wFrag.upload(wfr.data(), wfr.size() * sizeof(uint16_t));
    }
  }
  fc.scale = DevBuf(scale.size() * sizeof(float), false);
  fc.scale.upload(scale.data(), scale.size() * sizeof(float));
  fc.bias = DevBuf(bias.size() * sizeof(float), false);
  fc.bias.upload(bias.data(), bias.size() * sizeof(float));
  if(segOffsets) *segOffsets = offs;
  return fc;
}

// ------------------------------------------------------------------------------------------------
namespace {

void scanStack(const std::vector<BlockDesc>& blocks, int depth, std::vector<int>& levelStride, int& tmpStride, int& gStride) {
  for(const BlockDesc& b : blocks) {
    if(b.kind == BlockKind::Ordinary) tmpStride = std::max(tmpStride, roundUp(b.regularConv.outC, 32));
    else if(b.kind == BlockKind::Attention) {  // [0] holds q|k|v, [1] the attention output
      tmpStride = std::max(tmpStride, roundUp(roundUp(b.qProj.outC, 8) + roundUp(b.kProj.outC, 8) + roundUp(b.vProj.outC, 8), 32));
      gStride = std::max(gStride, roundUp(b.outProj.inC, 32));
    }
    else if(b.kind == BlockKind::FFN) {  // [0] holds linear1|gate, [1] the SwiGLU product
      tmpStride = std::max(tmpStride, roundUp(2 * roundUp(b.ffnChannels, 8), 32));
      gStride = std::max(gStride, roundUp(b.ffnChannels, 32));
    }
    else if(b.kind == BlockKind::GPool) {
      tmpStride = std::max(tmpStride, roundUp(b.regularConv.outC, 32));
      gStride = std::max(gStride, roundUp(b.gpoolConv.outC, 32));
    }
    else {
      if((int)levelStride.size() <= depth + 1) levelStride.resize(depth + 2, 0);
      levelStride[depth + 1] = std::max(levelStride[depth + 1], roundUp(b.regularConv.outC, 32));
      scanStack(b.inner, depth + 1, levelStride, tmpStride, gStride);
    }
  }
}

}  // namespace

Engine::Engine(const ModelDesc& model, int nnXLen, int nnYLen, int maxBatch, int dtype, int device)
  : dtype_(dtype), device_(device), X_(nnXLen), Y_(nnYLen), S_(nnXLen * nnYLen), maxBatch_(maxBatch), stream_(nullptr) {
  if(nnXLen < 2 || nnYLen < 2 || nnXLen > 19 || nnYLen > 19)
    throw Error(KMX_ERR_INVALID_ARG, "nnXLen/nnYLen must be in 2..19");
  if(maxBatch < 1 || maxBatch > 65535) throw Error(KMX_ERR_INVALID_ARG, "maxBatchSize must be in 1..65535");
  if(dtype != DT_F16 && dtype != DT_BF16 && dtype != DT_F32) throw Error(KMX_ERR_UNSUPPORTED, "unsupported device precision");
  if(dtype == DT_F32 && (model.hasTransformerBlocks || model.trunkNormKind != 0))
    throw Error(KMX_ERR_UNSUPPORTED, "fp32 device arithmetic (the verification mode) covers convolutional nets; transformer / RMSNorm nets run in fp16");
  // A constructor that throws does not run the destructor: release the stream, events and pinned buffers acquired so far
  // (an unsupported layer, or a device out of memory half-way through, must not leak them).
  try {
    // fp16 has five exponent bits: run the net at 1/8 of its values, as the reference's fp16 backends do (desc.cpp:2718-2736;
    // model_desc.cpp scaledBy8 - outputs unchanged). KMX_FP16_SCALE8=0 runs the file's own values (A/B, tests).
    bool scale8 = dtype == DT_F16 && model.scale8Applies() && !model.scale8Applied;
    if(const char* e = getenv("KMX_FP16_SCALE8")) scale8 = scale8 && atoi(e) != 0;
    if(scale8) {
      const std::unique_ptr<ModelDesc> scaled = model.scaledBy8();
      construct(*scaled);
      scale8_ = true;
    }
    else construct(model);
  }
  catch(...) {
    destroy();
    throw;
  }
}

void Engine::construct(const ModelDesc& model) {
  if(const char* tuneError = convTuneError()) throw Error(KMX_ERR_INVALID_ARG, tuneError);  // (a debug override that could not be parsed)
  hipCheck(hipSetDevice(device_), "hipSetDevice");
  hipCheck(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
  if(const char* e = getenv("KMX_GRAPHS")) useGraphs_ = atoi(e) != 0;
  if(const char* e = getenv("KMX_FUSE_SEAMS")) fuseSeams_ = atoi(e) != 0;
  if(const char* e = getenv("KMX_PACK_INPUTS")) packInputs_ = atoi(e) != 0;
  if(const char* e = getenv("KMX_FUSE_MIN_ROWS")) fuseMinRows_ = std::max(1, atoi(e));
  if(const char* e = getenv("KMX_CONV_CHAIN")) maxChain_ = atoi(e) >= 4 ? 4 : atoi(e) >= 2 ? 2 : 0;
  if(dtype_ == DT_F32) {  // the fp32 verification mode: one plain launch per convolution (kernels.h DT_F32)
    fuseSeams_ = false;
    maxChain_ = 0;
  }
  cin_ = model.numInputChannels;
  gin_ = model.numInputGlobalChannels;
  min_ = model.metaEncoderVersion > 0 ? model.numInputMetaChannels : 0;
  const size_t NS = (size_t)maxBatch_ * S_;
  zeroPage_ = DevBuf(ZERO_PAGE_ALLOC);
  inputT_ = DevBuf(NS * KCHUNK * dtSize(dtype_));
  mask_ = DevBuf(NS * sizeof(float));
  maskSum_ = DevBuf((size_t)maxBatch_ * sizeof(float));
  ncBias_ = DevBuf((size_t)maxBatch_ * roundUp(model.trunkC, 64) * sizeof(float));
  dSymmetry_ = DevBuf((size_t)maxBatch_ * sizeof(int));
  dOptimism_ = DevBuf((size_t)maxBatch_ * sizeof(float));
  dSpatialIn_ = DevBuf(NS * cin_ * sizeof(float));
  dGlobalIn_ = DevBuf((size_t)maxBatch_ * gin_ * sizeof(float));
  dPackedIn_ = DevBuf((size_t)maxBatch_ * packedRowBytes());
  // pinned staging on the device's NUMA node (numa.h): the runtime places hipHostMalloc memory next to the current device, and the
  // allocating thread prefers that node for as long as this scope lasts
  const numa::PreferDeviceNode stagingNode(device_);
  hipCheck(hipHostMalloc((void**)&hPacked_, (size_t)maxBatch_ * packedRowBytes()), "hipHostMalloc");
  if(min_ > 0) {
    dMetaIn_ = DevBuf((size_t)maxBatch_ * min_ * sizeof(float));
    hipCheck(hipHostMalloc((void**)&hMeta_, (size_t)maxBatch_ * min_ * sizeof(float)), "hipHostMalloc");
  }
  dPolicy_ = DevBuf((size_t)maxBatch_ * (S_ + 1) * sizeof(float));
  dValue_ = DevBuf((size_t)maxBatch_ * 3 * sizeof(float));
  dScore_ = DevBuf((size_t)maxBatch_ * 6 * sizeof(float));
  dOwnership_ = DevBuf(NS * sizeof(float));
  hipCheck(hipHostMalloc((void**)&hSpatial_, NS * cin_ * sizeof(float)), "hipHostMalloc");
  hipCheck(hipHostMalloc((void**)&hGlobal_, (size_t)maxBatch_ * gin_ * sizeof(float)), "hipHostMalloc");
  hipCheck(hipHostMalloc((void**)&hPolicy_, (size_t)maxBatch_ * (S_ + 1) * sizeof(float)), "hipHostMalloc");
  hipCheck(hipHostMalloc((void**)&hValue_, (size_t)maxBatch_ * 3 * sizeof(float)), "hipHostMalloc");
  hipCheck(hipHostMalloc((void**)&hScore_, (size_t)maxBatch_ * 6 * sizeof(float)), "hipHostMalloc");
  hipCheck(hipHostMalloc((void**)&hOwnership_, NS * sizeof(float)), "hipHostMalloc");
  hipCheck(hipHostMalloc((void**)&hSymmetry_, (size_t)2 * maxBatch_ * sizeof(int)), "hipHostMalloc");
  hipCheck(hipHostMalloc((void**)&hOptimism_, (size_t)2 * maxBatch_ * sizeof(float)), "hipHostMalloc");
  for(int i = 0; i < 2; i++) {
    hipCheck(hipEventCreateWithFlags(&stagingDone_[i], hipEventDisableTiming), "hipEventCreate");
    hipCheck(hipEventRecord(stagingDone_[i], stream_), "hipEventRecord");
  }
  buildSchedule(model);
  hipCheck(hipStreamSynchronize(stream_), "sync after build");
  hipCheck(hipDeviceSynchronize(), "sync after build");  // the null-stream zero fills and uploads of every DevBuf
}

Engine::~Engine() { destroy(); }

void Engine::destroy() noexcept {
  // (the thread that frees a handle or a batcher need not be one that ever used it: with one port per GPU in one process the
  // evaluator's owner tears all of them down - found by the dry run on 8 fake devices, tests/test_schedule_dryrun.py)
  // The caller's current device is put back at the end: tear-down runs on an owner thread that goes on with its own device
  // (torch in bench.py, another backend in the same process).
  int callerDevice = -1;
  if(hipGetDevice(&callerDevice) != hipSuccess) callerDevice = -1;
  struct RestoreDevice {
    int dev, mine;
    ~RestoreDevice() {
      if(dev >= 0 && dev != mine) (void)hipSetDevice(dev);
    }
  } restore{callerDevice, device_};
  (void)hipSetDevice(device_);
  if(stream_) (void)hipStreamSynchronize(stream_);
  dropGraphs();
  for(const Pending& p : pending_) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for(hipEvent_t e : eventPool_) (void)hipEventDestroy(e);
  for(int i = 0; i < 2; i++)
    if(stagingDone_[i]) (void)hipEventDestroy(stagingDone_[i]);
  pending_.clear();
  eventPool_.clear();
  stagingDone_[0] = stagingDone_[1] = nullptr;
  void* pinned[] = {hSpatial_, hGlobal_, hMeta_, hPacked_, hPolicy_, hValue_, hScore_, hOwnership_, hSymmetry_, hOptimism_};
  for(void* p : pinned)
    if(p) (void)hipHostFree(p);
  hSpatial_ = hGlobal_ = hMeta_ = hPolicy_ = hValue_ = hScore_ = hOwnership_ = hOptimism_ = nullptr;
  hPacked_ = nullptr;
  hSymmetry_ = nullptr;
  if(stream_) (void)hipStreamDestroy(stream_);
  stream_ = nullptr;
}

float* Engine::uploadFloats(const std::vector<float>& v) {
  std::unique_ptr<DevBuf> b(new DevBuf(std::max<size_t>(v.size(), 1) * sizeof(float), true));
  b->upload(v.data(), v.size() * sizeof(float));
  float* p = b->as<float>();
  params_.push_back(std::move(b));
  return p;
}
const FusedConv* Engine::newConv(const std::vector<ConvSegment>& segs, std::vector<int>* offs) {
  convs_.emplace_back(new FusedConv(buildFusedConv(dtype_, segs, offs)));
  return convs_.back().get();
}

ConvArgs Engine::makeConvArgs(const FusedConv* fc, const void* in, int inStride, const float* ncBias, int ncBiasStride,
                              const void* resid, int residStride, void* rawOut, int rawStride, int rawBegin, int rawEnd,
                              void* actOut, int actStride, int actBegin, int actEnd, int actKind, double* bytesPerRow) {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in;
  a.w = fc->w.get();
  a.wFrag = fc->wFrag.get();
  a.zeroPage = zeroPage_.get();
  a.inC = inStride;
  a.nChunks = fc->nChunks;
  a.coutPad = fc->coutPad;
  a.X = X_;
  a.Y = Y_;
  a.ncBias = ncBias;
  a.ncBiasStride = ncBiasStride;
  a.resid = resid;
  a.residC = residStride;
  a.rawOut = rawOut;
  a.rawC = rawStride;
  a.rawBegin = rawBegin;
  a.rawEnd = std::min(rawEnd, rawBegin + rawStride);  // never write past the channel stride of the destination
  a.actOut = actOut;
  a.actC = actStride;
  a.actBegin = actBegin;
  a.actEnd = std::min(actEnd, actBegin + actStride);
  a.scale = fc->scale.as<float>();
  a.bias = fc->bias.as<float>();
  a.actKind = actKind;
  a.mask = mask_.as<float>();
  if(inStride < fc->nChunks * KCHUNK) throw Error(KMX_ERR_INTERNAL, "addConv: input stride smaller than the padded channel count");
  // algorithmic traffic: read the input once, the residual once, write each output once (16-bit elements)
  double bytes = 2.0 * S_ * (double)fc->cin;
  if(resid) bytes += 2.0 * S_ * (double)(a.rawEnd - a.rawBegin);
  if(rawOut) bytes += 2.0 * S_ * (double)(a.rawEnd - a.rawBegin);
  if(actOut) bytes += 2.0 * S_ * (double)(a.actEnd - a.actBegin);
  if(bytesPerRow) *bytesPerRow = bytes;
  return a;
}

void Engine::launchConvOp(const ConvArgs& a, int ks, int n, hipStream_t st) {
  ConvArgs b = a;
  b.N = n;
  hipCheck(launchConv(dtype_, ks, chooseConvCfg(ks, a.coutPad, shapeRows(n)), b, st), "convolution launch");
}

void Engine::addConv(const FusedConv* fc, const void* in, int inStride, const float* ncBias, int ncBiasStride,
                     const void* resid, int residStride, void* rawOut, int rawStride, int rawBegin, int rawEnd,
                     void* actOut, int actStride, int actBegin, int actEnd, int actKind) {
  double bytes = 0;
  const ConvArgs a = makeConvArgs(fc, in, inStride, ncBias, ncBiasStride, resid, residStride, rawOut, rawStride, rawBegin, rawEnd, actOut,
                                  actStride, actBegin, actEnd, actKind, &bytes);
  const int ks = fc->ks;
  addOp(ks == 1 ? "conv1x1" : ks == 3 ? "conv3x3" : "conv5x5", 2.0 * fc->macPerCell * S_, bytes,
        [this, a, ks](int n, hipStream_t st) { launchConvOp(a, ks, n, st); });
}

// The seam between two nested-bottleneck blocks (pointwise_kernel.h): block i's closing 1x1 convolution into the residual
// stream `s`, block i+1's preBN + activation, block i+1's opening 1x1 convolution into `mid` with the first inner block's
// preBN. Batches of at least fuseMinRows_ boards run it as ONE launch in which the activated trunk image never leaves the
// CU; smaller ones as the two convolution launches it replaces (same arithmetic, bit-identical, tests/test_gpu_pointwise.py).
void Engine::addSeam(const ConvDesc& post, const void* in, int inStride, const Stream& s, const BnDesc& nextBN, const ConvDesc& pre,
                     const Stream& mid, const BnDesc& innerBN) {
  const FusedConv* c1 = newConv({{&post, &nextBN}});
  const FusedConv* c2 = newConv({{&pre, &innerBN}});
  double b1 = 0, b2 = 0;
  const ConvArgs a1 = makeConvArgs(c1, in, inStride, nullptr, 0, s.raw, s.stride, s.raw, s.stride, 0, c1->coutPad, s.act, s.stride, 0,
                                   c1->coutPad, nextBN.act, &b1);
  const ConvArgs a2 = makeConvArgs(c2, s.act, s.stride, nullptr, 0, nullptr, 0, mid.raw, mid.stride, 0, c2->coutPad, mid.act, mid.stride, 0,
                                   c2->coutPad, innerBN.act, &b2);
  PwPairArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.in = in; pa.inC = inStride;
  pa.w1 = c1->w.get();
  pa.resid = s.raw; pa.rawOut = s.raw; pa.trunkC = s.stride;
  pa.actOut = nullptr;  // only block i+1's opening convolution reads the activated trunk, and that happens in LDS
  pa.scale1 = c1->scale.as<float>(); pa.bias1 = c1->bias.as<float>(); pa.actKind1 = nextBN.act;
  pa.w2 = c2->w.get();
  pa.rawOut2 = mid.raw; pa.actOut2 = mid.act; pa.midC = mid.stride;
  pa.scale2 = c2->scale.as<float>(); pa.bias2 = c2->bias.as<float>(); pa.actKind2 = innerBN.act;
  pa.mask = mask_.as<float>();
  pa.zeroPage = zeroPage_.get();
  const int C1 = post.inC, C2 = post.outC, C3 = pre.outC, S = S_;
  // traffic of the fused form: in + residual + trunk raw + mid raw + mid act
  const double fusedBytes = 2.0 * S_ * ((double)C1 + 2.0 * C2 + 2.0 * C3);
  addOp("conv1x1_pair", 2.0 * (c1->macPerCell + c2->macPerCell) * S_, fusedBytes, [this, a1, a2, pa, C1, C2, C3, S](int n, hipStream_t st) {
    if(n >= fuseMinRows_) {
      PwPairArgs x = pa;
      x.cells = (long long)n * S;
      x.alone = cfgScale_ <= 1 && !sharesDevice_;
      hipCheck(launchPointwisePair(dtype_, C1, C2, C3, x, st), "pointwise pair launch");
    }
    else {
      launchConvOp(a1, 1, n, st);
      launchConvOp(a2, 1, n, st);
    }
  });
  // the two-launch form moves the activated trunk image through HBM: its own class and byte model in the profile
  Op& op = ops_.back();
  op.smallBelow = fuseMinRows_;  // below it: two launches
  op.clsSmall = opClass("conv1x1_pair_unfused");
  op.bytesPerRowSmall = b1 + b2;
  op.launchesSmall = 2;
}

// The last convolution of a block adds into the residual stream and, when the next consumer is a convolutional block (or a
// BatchNorm tip), also writes that consumer's BN+activation image of the stream. Transformer blocks and RMSNorm tips
// normalise over all channels of a cell, which no single work-group of the convolution sees: they get no image (nextBN null).
void Engine::addResidualConv(const ConvDesc& conv, const void* in, int inStride, const Stream& s, const BnDesc* nextBN) {
  const FusedConv* c = newConv({{&conv, nextBN}});
  if(nextBN != nullptr)
    addConv(c, in, inStride, nullptr, 0, s.raw, s.stride, s.raw, s.stride, 0, c->coutPad, s.act, s.stride, 0, c->coutPad, nextBN->act);
  else
    addConv(c, in, inStride, nullptr, 0, s.raw, s.stride, s.raw, s.stride, 0, c->coutPad, nullptr, 0, 0, 0, KMX_ACT_IDENTITY);
}

// One or two consecutive ordinary residual blocks on a 192-channel stream (the inner blocks of b18c384nbt's nested-bottleneck
// blocks, eigenbackend.cpp:1103-1146) as ONE op: at batch sizes that take the one-work-group-per-board shape it is one launch of
// conv_chain_kernel.h - two or four convolutions, the activated images handed over inside the CU (with KMX_CONV_CHAIN=2: launches of
// two) - otherwise exactly the launches the blocks get one by one. Same arithmetic either way, bit for bit. Returns the number of
// blocks that went into the op (0: blocks[i] is not of that shape).
int Engine::addOrdinaryChain(const std::vector<BlockDesc>& blocks, size_t i, const Stream& s, const BnDesc* bnAfter) {
  if(maxChain_ < 2 || s.stride != CHAIN_CHANNELS) return 0;
  auto nextBnOf = [&](size_t k) -> const BnDesc* { return k + 1 < blocks.size() ? (blocks[k + 1].isTransformer() ? nullptr : &blocks[k + 1].preBN) : bnAfter; };
  auto eligible = [&](size_t k) {
    if(k >= blocks.size() || blocks[k].kind != BlockKind::Ordinary) return false;
    const BlockDesc& b = blocks[k];
    const BnDesc* nb = nextBnOf(k);
    if(nb == nullptr || nb->c != CHAIN_CHANNELS || b.midBN.c != CHAIN_CHANNELS) return false;
    for(const ConvDesc* c : {&b.regularConv, &b.finalConv})
      if(c->ky != 3 || c->kx != 3 || c->inC != CHAIN_CHANNELS || c->outC != CHAIN_CHANNELS) return false;
    return b.midBN.act == nb->act && convChainSupported(nb->act);
  };
  if(!eligible(i)) return 0;
  const int nBlocks = eligible(i + 1) && nextBnOf(i)->act == nextBnOf(i + 1)->act ? 2 : 1;
  const int nc = 2 * nBlocks;
  const ConvDesc* cd[MAX_CHAIN];
  const BnDesc* bd[MAX_CHAIN];
  for(int k = 0; k < nBlocks; k++) {
    cd[2 * k] = &blocks[i + k].regularConv;
    bd[2 * k] = &blocks[i + k].midBN;
    cd[2 * k + 1] = &blocks[i + k].finalConv;
    bd[2 * k + 1] = nextBnOf(i + k);
  }
  const FusedConv* fc[MAX_CHAIN];
  for(int k = 0; k < nc; k++) fc[k] = newConv({{cd[k], bd[k]}});
  void* tmp = acts_[0]->get();
  // the launches of the unchained form: what buildStack adds for the blocks one by one
  std::vector<ConvArgs> ca(nc);
  double bytesUnchained = 0.0;
  for(int k = 0; k < nc; k++) {
    double b = 0.0;
    if(k % 2 == 0)
      ca[k] = makeConvArgs(fc[k], s.act, s.stride, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, tmp, CHAIN_CHANNELS, 0, fc[k]->coutPad, bd[k]->act, &b);
    else
      ca[k] = makeConvArgs(fc[k], tmp, CHAIN_CHANNELS, nullptr, 0, s.raw, s.stride, s.raw, s.stride, 0, fc[k]->coutPad, s.act, s.stride, 0,
                           fc[k]->coutPad, bd[k]->act, &b);
    bytesUnchained += b;
  }
  ConvChainArgs ch;
  memset(&ch, 0, sizeof(ch));
  ch.in = s.act;
  ch.zeroPage = zeroPage_.get();
  ch.mask = mask_.as<float>();
  ch.X = X_;
  ch.Y = Y_;
  ch.actKind = bd[0]->act;
  for(int k = 0; k < nc; k++) {
    ch.conv[k].w = fc[k]->w.get();
    ch.conv[k].scale = fc[k]->scale.as<float>();
    ch.conv[k].bias = fc[k]->bias.as<float>();
    ch.conv[k].resid = k % 2 == 1 ? s.raw : nullptr;
    ch.conv[k].rawOut = k % 2 == 1 ? s.raw : nullptr;
    ch.conv[k].actOut = k % 2 == 1 ? s.act : tmp;
  }
  double macs = 0.0;
  for(int k = 0; k < nc; k++) macs += fc[k]->macPerCell;
  const int perLaunch = std::min(nc, maxChain_);  // convolutions per chained launch
  // traffic of the chained form (16-bit elements per cell): per launch the input in and the last activated image out; per residual
  // convolution the residual in and the raw stream out; a hand-over moves half an image out and in again
  const double C = CHAIN_CHANNELS;
  const double launches = (double)(nc / perLaunch), handovers = (double)(nc - nc / perLaunch);
  const double bytesChained = 2.0 * S_ * (2.0 * launches * C + (double)nBlocks * 2.0 * C + handovers * C);
  addOp("conv3x3", 2.0 * macs * S_, bytesChained, [this, ca, ch, nc, perLaunch](int n, hipStream_t st) {
    if(chooseConvCfg(3, CHAIN_CHANNELS, shapeRows(n)) == 23) {
      for(int k0 = 0; k0 < nc; k0 += perLaunch) {
        ConvChainArgs x = ch;
        x.N = n;
        x.nConv = perLaunch;
        for(int k = 0; k < perLaunch; k++) x.conv[k] = ch.conv[k0 + k];
        hipCheck(launchConvChain(dtype_, x, st), "convolution chain launch");
      }
    }
    else
      for(int k = 0; k < nc; k++) launchConvOp(ca[k], 3, n, st);
  });
  Op& op = ops_.back();
  op.launches = nc;  // the profile counts convolutions: per-convolution times stay comparable between the forms
  op.launchesSmall = nc;
  op.clsSmall = op.cls;
  op.bytesPerRowSmall = bytesUnchained;
  op.smallBelow = -1;  // decided per pass, see runSchedule
  return nBlocks;
}

namespace {
ConvDesc convOfMatMul(const MatMulDesc& m) {  // a matmul over channels at every cell = a 1x1 convolution; [ic][oc] = [1][1][ic][oc]
  ConvDesc c;
  c.name = m.name;
  c.ky = c.kx = 1;
  c.inC = m.inC;
  c.outC = m.outC;
  c.w = m.w;
  return c;
}
}  // namespace

void Engine::addRmsNorm(const void* in, int inStride, void* out, int outStride, int C, float eps, const std::vector<float>& w,
                        const std::vector<float>* beta, int actKind, bool perBoard) {
  RmsNormArgs ra;
  memset(&ra, 0, sizeof(ra));
  ra.in = in; ra.inStride = inStride; ra.out = out; ra.outStride = outStride; ra.C = C; ra.eps = eps;
  ra.w = uploadFloats(w);
  ra.beta = beta != nullptr ? uploadFloats(*beta) : nullptr;
  ra.actKind = actKind;
  ra.mask = mask_.as<float>();
  ra.S = S_;
  const int dtype = dtype_;
  if(perBoard) {
    if(boardRms_.get() == nullptr) boardRms_ = DevBuf((size_t)maxBatch_ * sizeof(float));
    float* rms = boardRms_.as<float>();
    const float* mask = mask_.as<float>();
    const float* maskSum = maskSum_.as<float>();
    const int S = S_;
    ra.boardRms = rms;
    addOp("rmsnorm", 2.0 * S_ * C, 2.0 * S_ * C, [=](int n, hipStream_t st) {
      hipCheck(launchBoardRms(dtype, in, inStride, C, mask, maskSum, n, S, eps, rms, st), "board rms launch");
    });
  }
  addOp("rmsnorm", 3.0 * S_ * C, 2.0 * S_ * (C + outStride), [=](int n, hipStream_t st) {
    RmsNormArgs x = ra;
    x.N = n;
    hipCheck(launchRmsNorm(dtype, x, st), "rmsnorm launch");
  });
}

// TransformerAttentionDesc::computeRopeCosSin (desc.cpp:1300-1363) for this engine's buffer: [heads][numPairs][S]
static void ropeTables(const BlockDesc& b, int X, int Y, std::vector<float>& cosT, std::vector<float>& sinT) {
  const int S = X * Y, numPairs = b.qHeadDim / 2, heads = b.learnableRope ? b.numKVHeads : 1;
  cosT.assign((size_t)heads * numPairs * S, 1.0f);
  sinT.assign((size_t)heads * numPairs * S, 0.0f);
  for(int h = 0; h < heads; h++)
    for(int p = 0; p < numPairs; p++)
      for(int y = 0; y < Y; y++)
        for(int x = 0; x < X; x++) {
          float angle;
          if(b.learnableRope)
            angle = (float)x * b.ropeFreqs[((size_t)h * numPairs + p) * 2 + 0] + (float)y * b.ropeFreqs[((size_t)h * numPairs + p) * 2 + 1];
          else {
            const int perDim = numPairs / 2, dimHalf = b.qHeadDim / 2;
            if(p < perDim) angle = (float)y * (1.0f / powf(b.ropeTheta, (float)(2 * p) / (float)dimHalf));
            else angle = (float)x * (1.0f / powf(b.ropeTheta, (float)(2 * (p - perDim)) / (float)dimHalf));
          }
          cosT[((size_t)h * numPairs + p) * S + y * X + x] = cosf(angle);
          sinT[((size_t)h * numPairs + p) * S + y * X + x] = sinf(angle);
        }
}

void Engine::buildStack(const std::vector<BlockDesc>& blocks, const Stream& s, const BnDesc* bnAfter, int depth) {
  bool openedBySeam = false;  // the previous block's seam launch already ran this block's opening convolution
  for(size_t i = 0; i < blocks.size(); i++) {
    const BlockDesc& b = blocks[i];
    const BnDesc* nextBN = i + 1 < blocks.size() ? (blocks[i + 1].isTransformer() ? nullptr : &blocks[i + 1].preBN) : bnAfter;
    if(b.kind == BlockKind::Attention) {
      // ln = rmsnorm(s.raw); q|k|v = ln W; att = softmax(rope(q) rope(k)^T / sqrt(d) + keymask) v; s.raw += att Wo  (eigenbackend.cpp:1376-1600)
      if(b.preLN.c % 8 != 0) throw Error(KMX_ERR_UNSUPPORTED, b.name + ": channel counts must be multiples of 8");
      if(!attentionDimsSupported(b.qHeadDim, b.vHeadDim)) throw Error(KMX_ERR_UNSUPPORTED, b.name + ": attention head dims above 64 are not supported");
      addRmsNorm(s.raw, s.stride, s.act, s.stride, b.preLN.c, b.preLN.eps, b.preLN.w, nullptr, KMX_ACT_IDENTITY, false);
      const ConvDesc cq = convOfMatMul(b.qProj), ck = convOfMatMul(b.kProj), cv = convOfMatMul(b.vProj), co = convOfMatMul(b.outProj);
      std::vector<int> offs;
      const FusedConv* qkv = newConv({{&cq, nullptr}, {&ck, nullptr}, {&cv, nullptr}}, &offs);
      void* tQkv = acts_[0]->get();
      void* tAtt = acts_[1]->get();
      const int qkvStride = roundUp(qkv->cout, 32), attStride = roundUp(co.inC, 32);
      addConv(qkv, s.act, s.stride, nullptr, 0, nullptr, 0, tQkv, qkvStride, 0, qkv->coutPad, nullptr, 0, 0, 0, KMX_ACT_IDENTITY);
      AttentionArgs aa;
      memset(&aa, 0, sizeof(aa));
      aa.qkv = tQkv; aa.stride = qkvStride; aa.kOff = offs[1]; aa.vOff = offs[2];
      aa.H = b.numHeads; aa.KVH = b.numKVHeads; aa.QD = b.qHeadDim; aa.VD = b.vHeadDim;
      if(b.useRope) {
        std::vector<float> cosT, sinT;
        ropeTables(b, X_, Y_, cosT, sinT);
        aa.ropeCos = uploadFloats(cosT);
        aa.ropeSin = uploadFloats(sinT);
        aa.ropeHeads = b.learnableRope ? b.numKVHeads : 1;
      }
      aa.mask = mask_.as<float>();
      aa.out = tAtt; aa.outStride = attStride;
      aa.scale = 1.0f / sqrtf((float)b.qHeadDim);
      aa.S = S_;
      const int dtype = dtype_;
      addOp("attention", 2.0 * S_ * S_ * b.numHeads * (b.qHeadDim + b.vHeadDim), 2.0 * S_ * (qkv->cout + co.inC), [=](int n, hipStream_t st) {
        AttentionArgs x = aa;
        x.N = n;
        hipCheck(launchAttention(dtype, x, st), "attention launch");
      });
      addResidualConv(co, tAtt, attStride, s, nextBN);
    }
    else if(b.kind == BlockKind::FFN) {
      // ln = rmsnorm(s.raw); h = silu(ln W1) * (ln Wg); s.raw += h W2   (eigenbackend.cpp:1645-1718)
      if(b.preLN.c % 8 != 0 || b.ffnChannels % 8 != 0) throw Error(KMX_ERR_UNSUPPORTED, b.name + ": channel counts must be multiples of 8");
      addRmsNorm(s.raw, s.stride, s.act, s.stride, b.preLN.c, b.preLN.eps, b.preLN.w, nullptr, KMX_ACT_IDENTITY, false);
      const ConvDesc c1 = convOfMatMul(b.linear1), cg = convOfMatMul(b.linearGate), c2 = convOfMatMul(b.linear2);
      std::vector<int> offs;
      const FusedConv* up = newConv({{&c1, nullptr}, {&cg, nullptr}}, &offs);
      void* tUp = acts_[0]->get();
      void* tH = acts_[1]->get();
      const int upStride = roundUp(up->cout, 32), hStride = roundUp(b.ffnChannels, 32);
      addConv(up, s.act, s.stride, nullptr, 0, nullptr, 0, tUp, upStride, 0, up->coutPad, nullptr, 0, 0, 0, KMX_ACT_IDENTITY);
      SwiGluArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.in = tUp; ga.inStride = upStride; ga.gOff = offs[1]; ga.F = b.ffnChannels;
      ga.out = tH; ga.outStride = hStride;
      const int dtype = dtype_;
      const size_t S = (size_t)S_;
      addOp("swiglu", 4.0 * S_ * b.ffnChannels, 2.0 * S_ * 3.0 * b.ffnChannels, [=](int n, hipStream_t st) {
        SwiGluArgs x = ga;
        x.cells = (size_t)n * S;
        hipCheck(launchSwiGlu(dtype, x, st), "swiglu launch");
      });
      addResidualConv(c2, tH, hStride, s, nextBN);
    }
    else if(int chained = b.kind == BlockKind::Ordinary ? addOrdinaryChain(blocks, i, s, bnAfter) : 0) {
      i += (size_t)chained - 1;  // (one or two blocks went into one op)
    }
    else if(b.kind == BlockKind::Ordinary) {
      // mid = act(midBN(conv1(s.act)));  s.raw += conv2(mid);  s.act = act(nextBN(s.raw))   (eigenbackend.cpp:1139-1145)
      const FusedConv* c1 = newConv({{&b.regularConv, &b.midBN}});
      void* tmp = acts_[0]->get();
      const int tmpStride = roundUp(b.regularConv.outC, 32);
      addConv(c1, s.act, s.stride, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, tmp, tmpStride, 0, c1->coutPad, b.midBN.act);
      addResidualConv(b.finalConv, tmp, tmpStride, s, nextBN);
    }
    else if(b.kind == BlockKind::GPool) {
      // r = convR(s.act); g = act(gpoolBN(convG(s.act))); r += W*pool(g); s.raw += conv2(act(midBN(r)))  (eigenbackend.cpp:1204-1220)
      std::vector<int> offs;
      const FusedConv* c1 = newConv({{&b.regularConv, nullptr}, {&b.gpoolConv, &b.gpoolBN}}, &offs);
      const int R = b.regularConv.outC, G = b.gpoolConv.outC;
      const int rStride = roundUp(R, 32), gStride = roundUp(G, 32);
      void* tmpR = acts_[0]->get();
      void* tmpG = acts_[1]->get();
      addConv(c1, s.act, s.stride, nullptr, 0, nullptr, 0, tmpR, rStride, 0, offs[1], tmpG, gStride, offs[1], c1->coutPad,
              b.gpoolBN.act);
      GPoolArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.g = tmpG; ga.gStride = gStride; ga.gOffset = 0; ga.G = G;
      ga.r = tmpR; ga.rStride = rStride; ga.rOffset = 0; ga.R = R;
      ga.w = uploadFloats(b.gpoolToBiasMul.w);
      ga.scale = uploadFloats(b.midBN.scale);
      ga.bias = uploadFloats(b.midBN.bias);
      ga.actKind = b.midBN.act;
      ga.mask = mask_.as<float>();
      ga.maskSum = maskSum_.as<float>();
      ga.featOut = nullptr;
      ga.S = S_;
      const int dtype = dtype_;
      addOp("gpool_bias_act", 2.0 * 3 * G * R, 2.0 * S_ * (G + 2.0 * R), [=](int n, hipStream_t st) {
        GPoolArgs x = ga;
        x.N = n;
        hipCheck(launchGPoolApply(dtype, x, st), "gpool launch");
      });
      addResidualConv(b.finalConv, tmpR, rStride, s, nextBN);
    }
    else {
      // mid = conv1x1(s.act); inner stack on mid; s.raw += conv1x1(act(postBN(mid)))   (eigenbackend.cpp:1308-1314)
      const int M = b.regularConv.outC;
      Stream mid;
      mid.raw = acts_[2 + 2 * (depth + 1)]->get();
      mid.act = acts_[3 + 2 * (depth + 1)]->get();
      mid.stride = roundUp(M, 32);
      const BnDesc* firstInnerBN = b.inner[0].isTransformer() ? nullptr : &b.inner[0].preBN;
      if(!openedBySeam) {
        const FusedConv* pre = newConv({{&b.regularConv, firstInnerBN}});
        if(firstInnerBN != nullptr)
          addConv(pre, s.act, s.stride, nullptr, 0, nullptr, 0, mid.raw, mid.stride, 0, pre->coutPad, mid.act, mid.stride, 0,
                  pre->coutPad, firstInnerBN->act);
        else
          addConv(pre, s.act, s.stride, nullptr, 0, nullptr, 0, mid.raw, mid.stride, 0, pre->coutPad, nullptr, 0, 0, 0, KMX_ACT_IDENTITY);
      }
      openedBySeam = false;
      buildStack(b.inner, mid, &b.midBN, depth + 1);
      // the closing convolution, fused with the next block's opening one when that is a nested block of a supported shape
      const BlockDesc* nx = i + 1 < blocks.size() ? &blocks[i + 1] : nullptr;
      const bool seam = fuseSeams_ && nx != nullptr && nx->kind == BlockKind::Nested && nextBN != nullptr && !nx->inner.empty() &&
                        !nx->inner[0].isTransformer() && b.finalConv.ky == 1 && b.finalConv.kx == 1 && nx->regularConv.ky == 1 &&
                        nx->regularConv.kx == 1 && nx->regularConv.inC == b.finalConv.outC &&
                        pointwisePairSupported(b.finalConv.inC, b.finalConv.outC, nx->regularConv.outC) &&
                        // the fused kernel runs IN PLACE: its input (this block's activated mid image) and its mid outputs (the
                        // next block's) are the same buffers, which is race-free only while both have the same row stride - a
                        // work-group then overwrites exactly the rows it has fetched itself. Other shapes take the two launches.
                        roundUp(nx->regularConv.outC, 32) == mid.stride;
      if(seam) {
        Stream nmid = mid;  // the next block's mid stream lives in the same buffers (same nesting depth)
        nmid.stride = roundUp(nx->regularConv.outC, 32);
        addSeam(b.finalConv, mid.act, mid.stride, s, *nextBN, nx->regularConv, nmid, nx->inner[0].preBN);
        openedBySeam = true;
      }
      else addResidualConv(b.finalConv, mid.act, mid.stride, s, nextBN);
    }
  }
}

void Engine::buildSchedule(const ModelDesc& m) {
  const size_t NS = (size_t)maxBatch_ * S_;
  // ---- activation buffers: [0] tmp, [1] gpool tmp, then (raw, act) per nesting depth ----
  std::vector<int> levelStride(1, roundUp(m.trunkC, 32));
  int tmpStride = 32, gStride = 32;
  scanStack(m.blocks, 0, levelStride, tmpStride, gStride);
  acts_.emplace_back(new DevBuf(NS * tmpStride * dtSize(dtype_)));
  acts_.emplace_back(new DevBuf(NS * gStride * dtSize(dtype_)));
  for(size_t d = 0; d < levelStride.size(); d++) {
    acts_.emplace_back(new DevBuf(NS * std::max(levelStride[d], 32) * dtSize(dtype_)));
    acts_.emplace_back(new DevBuf(NS * std::max(levelStride[d], 32) * dtSize(dtype_)));
  }
  Stream trunk;
  trunk.raw = acts_[2]->get();
  trunk.act = acts_[3]->get();
  trunk.stride = roundUp(m.trunkC, 32);

  // ---- input staging ----
  {
    InputArgs ia;
    memset(&ia, 0, sizeof(ia));
    ia.cin = cin_;
    ia.gin = gin_;
    ia.X = X_;
    ia.Y = Y_;
    ia.out = inputT_.get();
    ia.mask = mask_.as<float>();
    ia.maskSum = maskSum_.as<float>();
    ia.wGlobal = uploadFloats(m.initialMatMul.w);
    ia.ncBias = ncBias_.as<float>();
    ia.C = m.trunkC;
    ia.ncStride = roundUp(m.trunkC, 64);
    ia.symmetry = dSymmetry_.as<int>();
    if(min_ > 0) {
      ia.metaIn = min_;
      ia.metaC1 = m.metaMul1.outC;
      ia.metaC2 = m.metaMul2.outC;
      ia.metaAct1 = m.metaAct1;
      ia.metaAct2 = m.metaAct2;
      ia.mW1 = uploadFloats(m.metaMul1.w);
      ia.mB1 = uploadFloats(m.metaBias1.w);
      ia.mW2 = uploadFloats(m.metaMul2.w);
      ia.mB2 = uploadFloats(m.metaBias2.w);
      ia.mW3 = uploadFloats(m.metaMul3.w);
    }
    const int dtype = dtype_;
    addOp("input_stage", 2.0 * gin_ * m.trunkC, S_ * (4.0 * cin_ + 2.0 * KCHUNK + 4.0), [=](int n, hipStream_t st) {
      InputArgs x = ia;
      x.N = n;
      x.spatial = curSpatial_;
      x.packed = curPacked_;
      x.global = curGlobal_;
      x.meta = curMeta_;
      hipCheck(launchInputExpand(dtype, x, st), "input staging launch");
    });
  }
  // ---- trunk (Trunk::apply, eigenbackend.cpp:1909-1947) ----
  const BnDesc* firstBN = m.blocks[0].isTransformer() ? nullptr : &m.blocks[0].preBN;
  const FusedConv* stem = newConv({{&m.initialConv, firstBN}});
  if(firstBN != nullptr)
    addConv(stem, inputT_.get(), KCHUNK, ncBias_.as<float>(), roundUp(m.trunkC, 64), nullptr, 0, trunk.raw, trunk.stride, 0,
            stem->coutPad, trunk.act, trunk.stride, 0, stem->coutPad, firstBN->act);
  else
    addConv(stem, inputT_.get(), KCHUNK, ncBias_.as<float>(), roundUp(m.trunkC, 64), nullptr, 0, trunk.raw, trunk.stride, 0,
            stem->coutPad, nullptr, 0, 0, 0, KMX_ACT_IDENTITY);
  if(m.trunkNormKind == 0) buildStack(m.blocks, trunk, &m.trunkTipBN, 0);
  else {
    // RMSNorm trunk tip (RMSNormLayer::apply, eigenbackend.cpp:960-1031): gamma, beta, activation, masked; per cell or per board
    if(m.trunkC % 8 != 0) throw Error(KMX_ERR_UNSUPPORTED, "RMSNorm trunk tip: trunk channels must be a multiple of 8");
    buildStack(m.blocks, trunk, nullptr, 0);
    addRmsNorm(trunk.raw, trunk.stride, trunk.act, trunk.stride, m.trunkC, m.rmsEps, m.rmsGamma, &m.rmsBeta, m.trunkTipAct, m.rmsSpatial);
  }

  // ---- heads: one 1x1 conv for p1 (raw), g1 (BN+act), v1 (BN+act) ----
  const bool fuseV = m.g1BN.act == m.v1BN.act;
  std::vector<int> offs;
  std::vector<ConvSegment> segs = {{&m.p1Conv, nullptr}, {&m.g1Conv, &m.g1BN}};
  if(fuseV) segs.push_back({&m.v1Conv, &m.v1BN});
  const FusedConv* heads = newConv(segs, &offs);
  const int P1 = m.p1Conv.outC, G1 = m.g1Conv.outC, V1 = m.v1Conv.outC;
  const int headRawStride = roundUp(P1, 32);
  const int headActC = heads->cout - offs[1];
  const int headActStride = roundUp(headActC, 32);
  acts_.emplace_back(new DevBuf(NS * headRawStride * dtSize(dtype_)));
  void* headRaw = acts_.back()->get();
  acts_.emplace_back(new DevBuf(NS * headActStride * dtSize(dtype_)));
  void* headAct = acts_.back()->get();
  addConv(heads, trunk.act, trunk.stride, nullptr, 0, nullptr, 0, headRaw, headRawStride, 0, offs[1], headAct, headActStride,
          offs[1], heads->cout, m.g1BN.act);
  void* vAct = headAct;
  int vStride = headActStride, vOffset = fuseV ? offs[2] - offs[1] : 0;
  if(!fuseV) {
    const FusedConv* vconv = newConv({{&m.v1Conv, &m.v1BN}});
    vStride = roundUp(V1, 32);
    acts_.emplace_back(new DevBuf(NS * vStride * dtSize(dtype_)));
    vAct = acts_.back()->get();
    addConv(vconv, trunk.act, trunk.stride, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, vAct, vStride, 0, vconv->coutPad, m.v1BN.act);
  }
  polFeat_ = DevBuf((size_t)maxBatch_ * 3 * G1 * sizeof(float));
  {
    GPoolArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.g = headAct; ga.gStride = headActStride; ga.gOffset = 0; ga.G = G1;
    ga.r = headRaw; ga.rStride = headRawStride; ga.rOffset = 0; ga.R = P1;
    ga.w = uploadFloats(m.gpoolToBiasMul.w);
    ga.scale = uploadFloats(m.p1BN.scale);
    ga.bias = uploadFloats(m.p1BN.bias);
    ga.actKind = m.p1BN.act;
    ga.mask = mask_.as<float>();
    ga.maskSum = maskSum_.as<float>();
    ga.featOut = polFeat_.as<float>();
    ga.S = S_;
    const int dtype = dtype_;
    addOp("gpool_bias_act", 2.0 * 3 * G1 * P1, 2.0 * S_ * (G1 + 2.0 * P1), [=](int n, hipStream_t st) {
      GPoolArgs x = ga;
      x.N = n;
      hipCheck(launchGPoolApply(dtype, x, st), "policy gpool launch");
    });
  }
  {
    PolicyArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.p = headRaw; pa.pStride = headRawStride; pa.pOffset = 0; pa.P = P1;
    pa.w2 = uploadFloats(m.p2Conv.w);
    pa.NP = m.numPolicyChannels;
    pa.feat = polFeat_.as<float>();
    pa.G3 = 3 * G1;
    pa.wPass = uploadFloats(m.gpoolToPassMul.w);
    if(m.hasPassMLP) {
      pa.bPass = uploadFloats(m.gpoolToPassBias.w);
      pa.wPass2 = uploadFloats(m.gpoolToPassMul2.w);
      pa.passHidden = m.gpoolToPassMul.outC;
      pa.passAct = m.passAct;
    }
    pa.symmetry = dSymmetry_.as<int>();
    pa.optimism = dOptimism_.as<float>();
    pa.X = X_;
    pa.Y = Y_;
    const int dtype = dtype_;
    addOp("policy_tail", 2.0 * S_ * P1 * m.numPolicyChannels, S_ * (2.0 * P1 + 4.0), [=](int n, hipStream_t st) {
      PolicyArgs x = pa;
      x.N = n;
      x.out = curPolicy_;
      hipCheck(launchPolicyFinal(dtype, x, st), "policy tail launch");
    });
  }
  {
    ValueArgs va;
    memset(&va, 0, sizeof(va));
    va.v = vAct; va.vStride = vStride; va.vOffset = vOffset; va.V1 = V1;
    va.w2 = uploadFloats(m.v2Mul.w);
    va.b2 = uploadFloats(m.v2Bias.w);
    va.V2 = m.v2Mul.outC;
    va.v2Act = m.v2Act;
    va.w3 = uploadFloats(m.v3Mul.w);
    va.b3 = uploadFloats(m.v3Bias.w);
    va.wsv = uploadFloats(m.sv3Mul.w);
    va.bsv = uploadFloats(m.sv3Bias.w);
    va.NSV = m.numScoreValueChannels;
    va.wOwn = uploadFloats(m.vOwnershipConv.w);
    va.maskSum = maskSum_.as<float>();
    va.symmetry = dSymmetry_.as<int>();
    va.X = X_;
    va.Y = Y_;
    const int dtype = dtype_;
    addOp("value_tail", 2.0 * (S_ * V1 + 3.0 * V1 * m.v2Mul.outC), S_ * (2.0 * V1 + 4.0), [=](int n, hipStream_t st) {
      ValueArgs x = va;
      x.N = n;
      x.value = curValue_;
      x.score = curScore_;
      x.ownership = curOwnership_;
      hipCheck(launchValueFinal(dtype, x, st), "value tail launch");
    });
  }
}

void Engine::runSchedule(int n, const float* dSpatial, const unsigned char* dPacked, const float* dGlobal, const float* dMeta,
                         float* dPolicy, float* dValue, float* dScore, float* dOwnership) {
  // the reference asserts the same pairing (eigenbackend.cpp:1929-1936)
  if(min_ > 0 && dMeta == nullptr) throw Error(KMX_ERR_INVALID_ARG, "this net has an sgf-metadata encoder: rows need the metadata input (kmx_eval_meta)");
  if(min_ == 0 && dMeta != nullptr) throw Error(KMX_ERR_INVALID_ARG, "this net has no sgf-metadata encoder: the metadata input must be NULL");
  curSpatial_ = dSpatial;
  curPacked_ = dPacked;
  curGlobal_ = dGlobal;
  curMeta_ = dMeta;
  curPolicy_ = dPolicy;
  curValue_ = dValue;
  curScore_ = dScore;
  curOwnership_ = dOwnership;
  const int forkAt = forkEv_ == nullptr ? -1 : std::min(forkOps_, (int)ops_.size());
  if(forkAt == 0) hipCheck(hipEventRecord(forkEv_, stream_), "hipEventRecord");
  if(!profiling_) {
    if(useGraphs_ && forkAt <= 0) {
      GraphKey key;
      key.n = n;
      key.scale = cfgScale_;
      const void* ptrs[8] = {dSpatial, dPacked, dGlobal, dMeta, dPolicy, dValue, dScore, dOwnership};
      for(int i = 0; i < 8; i++) key.p[i] = ptrs[i];
      auto it = graphCache_.find(key);
      if(it == graphCache_.end()) {
        // first sight: run directly (also sets the >64 KiB LDS attribute of every kernel shape this pass uses, which
        // must not happen inside a capture) and remember the key
        if(graphCache_.size() >= MAX_GRAPHS) {  // evict the least recently used entry
          auto victim = graphCache_.begin();
          for(auto j = graphCache_.begin(); j != graphCache_.end(); ++j)
            if(j->second.lastUse < victim->second.lastUse) victim = j;
          if(victim->second.exec) (void)hipGraphExecDestroy(victim->second.exec);
          if(victim->second.graph) (void)hipGraphDestroy(victim->second.graph);
          graphCache_.erase(victim);
        }
        graphCache_[key].lastUse = ++graphClock_;
        launchOps(n);
        return;
      }
      GraphEntry& g = it->second;
      g.lastUse = ++graphClock_;
      if(g.exec == nullptr) {
        if(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal) != hipSuccess) {
          // the runtime cannot capture here (e.g. this thread is inside someone else's capture): launch directly from now on
          (void)hipGetLastError();
          useGraphs_ = false;
          launchOps(n);
          return;
        }
        try {
          launchOps(n);
        }
        catch(...) {
          hipGraph_t dead = nullptr;
          (void)hipStreamEndCapture(stream_, &dead);
          if(dead) (void)hipGraphDestroy(dead);
          graphCache_.erase(it);
          throw;
        }
        hipCheck(hipStreamEndCapture(stream_, &g.graph), "hipStreamEndCapture");
        hipCheck(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0), "hipGraphInstantiate");
      }
      hipCheck(hipGraphLaunch(g.exec, stream_), "hipGraphLaunch");
      graphLaunches_++;
      return;
    }
    // KMX_DEBUG_SYNC=1 (fault triage, with KMX_GRAPHS=0): every op is named on stderr before its launch and waited for after it - the
    // last line before a device fault names the op that faulted (round 6). KMX_DEBUG_SQUAT=<bytes> on top of it: before
    // every op a squatter launch (kernels.h launchLdsSquatter: four one-wave work-groups per CU holding <bytes> of LDS each for 200 us)
    // goes out on a second stream, so that the op's work-groups start at a nonzero LDS base wherever they fit beside it; LDS words of
    // the squatters that the op overwrote are reported.
    static const bool debugSync = getenv("KMX_DEBUG_SYNC") != nullptr;
    static const int debugSquat = getenv("KMX_DEBUG_SQUAT") ? atoi(getenv("KMX_DEBUG_SQUAT")) : 0;
    static hipStream_t squatStream = nullptr;
    static unsigned* squatCorrupt = nullptr;
    if(debugSync && debugSquat > 0 && squatStream == nullptr) {
      hipCheck(hipStreamCreateWithFlags(&squatStream, hipStreamNonBlocking), "hipStreamCreate");
      hipCheck(hipMalloc((void**)&squatCorrupt, sizeof(unsigned)), "hipMalloc");
      hipCheck(hipMemset(squatCorrupt, 0, sizeof(unsigned)), "hipMemset");
    }
    for(size_t i = 0; i < ops_.size(); i++) {
      if(debugSync) {
        fprintf(stderr, "[kmx op] %zu %s rows %d\n", i, opClasses_[ops_[i].cls].c_str(), n);
        fflush(stderr);
        if(debugSquat > 0) hipCheck(launchLdsSquatter(1024, debugSquat, 200, squatCorrupt, squatStream), "squatter launch");
      }
      ops_[i].fn(n, stream_);
      if(debugSync) {
        hipCheck(hipStreamSynchronize(stream_), "hipStreamSynchronize (KMX_DEBUG_SYNC)");
        if(debugSquat > 0) {
          hipCheck(hipStreamSynchronize(squatStream), "hipStreamSynchronize (KMX_DEBUG_SQUAT)");
          unsigned bad = 0;
          hipCheck(hipMemcpy(&bad, squatCorrupt, sizeof(bad), hipMemcpyDeviceToHost), "hipMemcpy");
          if(bad != 0) {
            fprintf(stderr, "[kmx squat] op %zu %s rows %d overwrote %u LDS words outside its allocation\n", i, opClasses_[ops_[i].cls].c_str(), n, bad);
            fflush(stderr);
            hipCheck(hipMemset(squatCorrupt, 0, sizeof(unsigned)), "hipMemset");
          }
        }
      }
      if((int)i + 1 == forkAt) hipCheck(hipEventRecord(forkEv_, stream_), "hipEventRecord");
    }
    return;
  }
  for(size_t i = 0; i < ops_.size(); i++) {
    const Op& op = ops_[i];
    Pending p = {};
    for(hipEvent_t* e : {&p.a, &p.b}) {
      if(eventPool_.empty()) hipCheck(hipEventCreate(e), "hipEventCreate");
      else {
        *e = eventPool_.back();
        eventPool_.pop_back();
      }
    }
    // (a chain op - smallBelow < 0 - takes its separate-launch form whenever the batch does not take the one-work-group-per-board shape)
    const bool smallForm = op.smallBelow < 0 ? chooseConvCfg(3, CHAIN_CHANNELS, shapeRows(n)) != 23 : n < op.smallBelow;
    p.cls = smallForm ? op.clsSmall : op.cls;
    p.launches = smallForm ? op.launchesSmall : op.launches;
    p.flops = op.flopsPerRow * n;
    p.bytes = (smallForm ? op.bytesPerRowSmall : op.bytesPerRow) * n;
    hipCheck(hipEventRecord(p.a, stream_), "hipEventRecord");
    op.fn(n, stream_);
    hipCheck(hipEventRecord(p.b, stream_), "hipEventRecord");
    pending_.push_back(p);
    if((int)i + 1 == forkAt) hipCheck(hipEventRecord(forkEv_, stream_), "hipEventRecord");
  }
}

void Engine::launchOps(int n) {
  for(const Op& op : ops_) op.fn(n, stream_);
}

void Engine::dropGraphs() noexcept {
  for(auto& kv : graphCache_) {
    if(kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if(kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
  }
  graphCache_.clear();
}

int Engine::opClass(const std::string& name) {
  for(size_t i = 0; i < opClasses_.size(); i++)
    if(opClasses_[i] == name) return (int)i;
  opClasses_.push_back(name);
  return (int)opClasses_.size() - 1;
}
void Engine::addOp(const std::string& cls, double flopsPerRow, double bytesPerRow, std::function<void(int, hipStream_t)> fn) {
  Op op;
  op.fn = std::move(fn);
  op.cls = opClass(cls);
  op.flopsPerRow = flopsPerRow;
  op.bytesPerRow = bytesPerRow;
  ops_.push_back(std::move(op));
}
void Engine::setProfiling(bool enabled) {
  sync();
  collectProfile();
  profiling_ = enabled;
  profile_.clear();
}
void Engine::collectProfile() {
  if(pending_.empty()) return;
  if(profile_.size() < opClasses_.size()) profile_.resize(opClasses_.size());
  for(const Pending& p : pending_) {
    float ms = 0.0f;
    hipCheck(hipEventElapsedTime(&ms, p.a, p.b), "hipEventElapsedTime");
    ProfileEntry& e = profile_[p.cls];
    e.name = opClasses_[p.cls];
    e.launches += p.launches;
    e.ms += ms;
    e.flops += p.flops;
    e.bytes += p.bytes;
    eventPool_.push_back(p.a);
    eventPool_.push_back(p.b);
  }
  pending_.clear();
}
std::vector<Engine::ProfileEntry> Engine::getProfile() {
  sync();
  collectProfile();
  std::vector<ProfileEntry> out;
  for(const ProfileEntry& e : profile_)
    if(e.launches > 0) out.push_back(e);
  return out;
}

// (callable from any thread - the batcher's completion thread never made this engine's device current - and one process may
// hold engines on several GPUs: nneval.cpp:399-407 runs one server thread per GPU)
void Engine::sync() {
  hipCheck(hipSetDevice(device_), "hipSetDevice");
  hipCheck(hipStreamSynchronize(stream_), "stream synchronize");
}
bool Engine::idle() {
  hipCheck(hipSetDevice(device_), "hipSetDevice");
  const hipError_t e = hipStreamQuery(stream_);
  if(e == hipSuccess) return true;
  if(e == hipErrorNotReady) return false;
  hipCheck(e, "hipStreamQuery");
  return false;
}

// symmetry / optimism of the rows -> device. Pinned staging is double-buffered: the slot used two calls ago is free as
// soon as ITS copies have run, so back-to-back asynchronous calls queue up without draining the stream in between.
void Engine::stageRowParams(int n, const int* symmetry, const float* policyOptimism) {
  const int slot = stagingSlot_;
  stagingSlot_ ^= 1;
  hipCheck(hipEventSynchronize(stagingDone_[slot]), "staging slot");
  int* hs = hSymmetry_ + (size_t)slot * maxBatch_;
  float* ho = hOptimism_ + (size_t)slot * maxBatch_;
  for(int i = 0; i < n; i++) {
    hs[i] = symmetry ? symmetry[i] : 0;
    ho[i] = policyOptimism ? policyOptimism[i] : 0.0f;
  }
  hipCheck(hipMemcpyAsync(dSymmetry_.get(), hs, n * sizeof(int), hipMemcpyHostToDevice, stream_), "H2D symmetry");
  hipCheck(hipMemcpyAsync(dOptimism_.get(), ho, n * sizeof(float), hipMemcpyHostToDevice, stream_), "H2D optimism");
  hipCheck(hipEventRecord(stagingDone_[slot], stream_), "hipEventRecord");
}

void Engine::evalDevice(int n, const float* dSpatial, const float* dGlobal, const float* dMeta, const int* symmetry,
                        const float* policyOptimism, float* dPolicy, float* dValue, float* dScore, float* dOwnership, bool doSync) {
  if(n < 1 || n > maxBatch_) throw Error(KMX_ERR_INVALID_ARG, "batch size out of range for this handle");
  hipCheck(hipSetDevice(device_), "hipSetDevice");
  stageRowParams(n, symmetry, policyOptimism);
  runSchedule(n, dSpatial, nullptr, dGlobal, dMeta, dPolicy, dValue, dScore, dOwnership);
  rows_ += (uint64_t)n;
  batches_ += 1;
  if(doSync) sync();
}

bool packRowNHWC(const float* row, int S, int C, unsigned char* out) {
  const int PB = (S + 7) / 8;
  memset(out, 0, (size_t)C * PB);
  bool binary = true;
  for(int p = 0; p < S; p++) {
    const float* cell = row + (size_t)p * C;
    unsigned m = 0, bad = 0;
    for(int c = 0; c < C; c++) {  // branch-free: which channels are set in this cell
      uint32_t u;
      memcpy(&u, cell + c, 4);
      const bool zero = (u & 0x7fffffffu) == 0u;  // +0.0f and -0.0f
      m |= (unsigned)(!zero) << c;
      bad |= (unsigned)(!zero && u != 0x3f800000u);
    }
    binary = binary && bad == 0;
    const int byte = p >> 3;
    const unsigned char bit = (unsigned char)(1u << (7 - (p & 7)));
    while(m) {
      const int c = __builtin_ctz(m);
      m &= m - 1;
      out[(size_t)c * PB + byte] |= bit;
    }
  }
  return binary;
}

void Engine::evalHostBegin(int n, const float* const* rowSpatial, const unsigned char* const* rowPacked,
                           const float* const* rowGlobal, const float* const* rowMeta, const int* symmetry,
                           const float* policyOptimism, float* const* outOwnership) {
  if(n < 1 || n > maxBatch_) throw Error(KMX_ERR_INVALID_ARG, "batch size out of range for this handle");
  hipCheck(hipSetDevice(device_), "hipSetDevice");
  hipCheck(hipStreamSynchronize(stream_), "stream synchronize");  // the single-buffered row staging below
  const size_t rowElts = (size_t)S_ * cin_;
  const size_t rowBytes = (size_t)packedRowBytes();
  // Opt-in (KMX_PACK_INPUTS=1, see engine.h for the measurement): fp32 rows of 0/1 feature planes (all V7 inputs are;
  // nninputs.cpp:2288-2731) are bit-packed WHILE they are staged and cross PCIe as 1012 instead of 31768 bytes per 19x19 row
  // (SURVEY 8f1). The device expands them to the same 16-bit image as fp32 rows (bit-identical, tested). A batch with any
  // other value in a plane is staged as fp32 rows.
  bool packedStaging = rowPacked != nullptr;
  if(!rowPacked && cin_ <= 32 && packInputs_) {
    packedStaging = true;
    for(int i = 0; i < n && packedStaging; i++) packedStaging = packRowNHWC(rowSpatial[i], S_, cin_, hPacked_ + i * rowBytes);
  }
  for(int i = 0; i < n; i++) {
    if(rowPacked) memcpy(hPacked_ + i * rowBytes, rowPacked[i], rowBytes);
    else if(!packedStaging) memcpy(hSpatial_ + i * rowElts, rowSpatial[i], rowElts * sizeof(float));
    memcpy(hGlobal_ + (size_t)i * gin_, rowGlobal[i], gin_ * sizeof(float));
  }
  if(packedStaging) hipCheck(hipMemcpyAsync(dPackedIn_.get(), hPacked_, n * rowBytes, hipMemcpyHostToDevice, stream_), "H2D packed planes");
  else hipCheck(hipMemcpyAsync(dSpatialIn_.get(), hSpatial_, n * rowElts * sizeof(float), hipMemcpyHostToDevice, stream_), "H2D spatial");
  hipCheck(hipMemcpyAsync(dGlobalIn_.get(), hGlobal_, (size_t)n * gin_ * sizeof(float), hipMemcpyHostToDevice, stream_), "H2D global");
  if(min_ > 0 && rowMeta != nullptr) {
    for(int i = 0; i < n; i++) {
      if(!rowMeta[i]) throw Error(KMX_ERR_INVALID_ARG, "null metadata row pointer");
      memcpy(hMeta_ + (size_t)i * min_, rowMeta[i], min_ * sizeof(float));
    }
    hipCheck(hipMemcpyAsync(dMetaIn_.get(), hMeta_, (size_t)n * min_ * sizeof(float), hipMemcpyHostToDevice, stream_), "H2D meta");
  }
  if(min_ == 0 && rowMeta != nullptr) throw Error(KMX_ERR_INVALID_ARG, "this net has no sgf-metadata encoder: the metadata input must be NULL");
  bool anyOwner = false;
  if(outOwnership)
    for(int i = 0; i < n; i++) anyOwner = anyOwner || outOwnership[i] != nullptr;
  hostAnyOwner_ = anyOwner;
  stageRowParams(n, symmetry, policyOptimism);
  runSchedule(n, packedStaging ? nullptr : dSpatialIn_.as<float>(), packedStaging ? dPackedIn_.as<unsigned char>() : nullptr,
              dGlobalIn_.as<float>(), (min_ > 0 && rowMeta) ? dMetaIn_.as<float>() : nullptr,
              dPolicy_.as<float>(), dValue_.as<float>(), dScore_.as<float>(), anyOwner ? dOwnership_.as<float>() : nullptr);
  hipCheck(hipMemcpyAsync(hPolicy_, dPolicy_.get(), (size_t)n * (S_ + 1) * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H policy");
  hipCheck(hipMemcpyAsync(hValue_, dValue_.get(), (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H value");
  hipCheck(hipMemcpyAsync(hScore_, dScore_.get(), (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H score");
  if(anyOwner)
    hipCheck(hipMemcpyAsync(hOwnership_, dOwnership_.get(), (size_t)n * S_ * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H ownership");
}

void Engine::launchStagedPacked(int n, const int* symmetry, const float* policyOptimism, bool anyOwner) {
  if(n < 1 || n > maxBatch_) throw Error(KMX_ERR_INVALID_ARG, "batch size out of range for this handle");
  hipCheck(hipSetDevice(device_), "hipSetDevice");
  const size_t rowBytes = (size_t)packedRowBytes();
  hipCheck(hipMemcpyAsync(dPackedIn_.get(), hPacked_, n * rowBytes, hipMemcpyHostToDevice, stream_), "H2D packed planes");
  hipCheck(hipMemcpyAsync(dGlobalIn_.get(), hGlobal_, (size_t)n * gin_ * sizeof(float), hipMemcpyHostToDevice, stream_), "H2D global");
  if(min_ > 0) hipCheck(hipMemcpyAsync(dMetaIn_.get(), hMeta_, (size_t)n * min_ * sizeof(float), hipMemcpyHostToDevice, stream_), "H2D meta");
  hostAnyOwner_ = anyOwner;
  stageRowParams(n, symmetry, policyOptimism);
  runSchedule(n, nullptr, dPackedIn_.as<unsigned char>(), dGlobalIn_.as<float>(), min_ > 0 ? dMetaIn_.as<float>() : nullptr,
              dPolicy_.as<float>(), dValue_.as<float>(), dScore_.as<float>(), anyOwner ? dOwnership_.as<float>() : nullptr);
  hipCheck(hipMemcpyAsync(hPolicy_, dPolicy_.get(), (size_t)n * (S_ + 1) * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H policy");
  hipCheck(hipMemcpyAsync(hValue_, dValue_.get(), (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H value");
  hipCheck(hipMemcpyAsync(hScore_, dScore_.get(), (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H score");
  if(anyOwner)
    hipCheck(hipMemcpyAsync(hOwnership_, dOwnership_.get(), (size_t)n * S_ * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H ownership");
  rows_ += (uint64_t)n;
  batches_ += 1;
}

void Engine::evalHostFinish(int n, float* const* outPolicy, float* outValue, float* outScore, float* const* outOwnership) {
  sync();
  for(int i = 0; i < n; i++) {
    memcpy(outPolicy[i], hPolicy_ + (size_t)i * (S_ + 1), (S_ + 1) * sizeof(float));
    if(hostAnyOwner_ && outOwnership[i]) memcpy(outOwnership[i], hOwnership_ + (size_t)i * S_, S_ * sizeof(float));
  }
  memcpy(outValue, hValue_, (size_t)n * 3 * sizeof(float));
  memcpy(outScore, hScore_, (size_t)n * 6 * sizeof(float));
  rows_ += (uint64_t)n;
  batches_ += 1;
}

void Engine::evalHost(int n, const float* const* rowSpatial, const float* const* rowGlobal, const float* const* rowMeta,
                      const int* symmetry, const float* policyOptimism, float* const* outPolicy, float* outValue, float* outScore,
                      float* const* outOwnership) {
  evalHostBegin(n, rowSpatial, nullptr, rowGlobal, rowMeta, symmetry, policyOptimism, outOwnership);
  evalHostFinish(n, outPolicy, outValue, outScore, outOwnership);
}

// ------------------------------------------------------------------------------------------------
// Layer test hooks. Small one-shot runs on the default device with the production kernels.
namespace {

struct HookCtx {
  int dtype, N, X, Y, S;
  hipStream_t st;
  DevBuf zero, mask;
  HookCtx(int dt, int n, int x, int y, const float* hostMask) : dtype(dt), N(n), X(x), Y(y), S(x * y), st(nullptr) {
    if(x < 2 || y < 2 || x > 19 || y > 19 || n < 1) throw Error(KMX_ERR_INVALID_ARG, "test hook: bad sizes");
    zero = DevBuf(ZERO_PAGE_ALLOC);
    std::vector<float> ones((size_t)n * S, 1.0f);
    mask = DevBuf((size_t)n * S * sizeof(float), false);
    mask.upload(hostMask ? hostMask : ones.data(), (size_t)n * S * sizeof(float));
  }
  // fp32 host NHWC [N][S][C] -> device T [N][S][round32(C)]
  DevBuf toDevice(const float* host, int C, int* stride) {
    const size_t cells = (size_t)N * S;
    DevBuf f(cells * C * sizeof(float), false);
    f.upload(host, cells * C * sizeof(float));
    *stride = roundUp(C, 32);
    DevBuf t(cells * (*stride) * dtSize(dtype));
    hipCheck(launchFloatToT(dtype, f.as<float>(), C, t.get(), *stride, cells, st), "floatToT");
    hipCheck(hipStreamSynchronize(st), "sync");
    return t;
  }
  void toHost(const DevBuf& t, int stride, int C, float* host) {
    const size_t cells = (size_t)N * S;
    DevBuf f(cells * C * sizeof(float), false);
    hipCheck(launchTToFloat(dtype, t.get(), stride, 0, f.as<float>(), C, cells, st), "tToFloat");
    hipCheck(hipStreamSynchronize(st), "sync");
    hipCheck(hipMemcpy(host, f.get(), cells * C * sizeof(float), hipMemcpyDeviceToHost), "D2H");
  }
  void conv(const FusedConv& fc, const void* in, int inStride, const void* resid, int residStride, void* rawOut, int rawStride,
            int rawBegin, int rawEnd, void* actOut, int actStride, int actBegin, int actEnd, int actKind) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.w = fc.w.get(); a.wFrag = fc.wFrag.get(); a.zeroPage = zero.get(); a.inC = inStride; a.nChunks = fc.nChunks; a.coutPad = fc.coutPad;
    a.N = N; a.X = X; a.Y = Y;
    a.resid = resid; a.residC = residStride;
    a.rawOut = rawOut; a.rawC = rawStride; a.rawBegin = rawBegin; a.rawEnd = std::min(rawEnd, rawBegin + rawStride);
    a.actOut = actOut; a.actC = actStride; a.actBegin = actBegin; a.actEnd = std::min(actEnd, actBegin + actStride);
    a.scale = fc.scale.as<float>(); a.bias = fc.bias.as<float>(); a.actKind = actKind;
    a.mask = mask.as<float>();
    hipCheck(launchConv(dtype, fc.ks, chooseConvCfg(fc.ks, fc.coutPad, N), a, st), "test conv launch");
  }
};

ConvDesc convFromAbi(const kmx_conv_desc* d) {  // [oc][ic][ky][kx] -> file order [ky][kx][ic][oc]
  if(!d || !d->weights) throw Error(KMX_ERR_INVALID_ARG, "null conv desc");
  ConvDesc c;
  c.name = "testconv";
  c.ky = d->conv_y_size; c.kx = d->conv_x_size; c.inC = d->in_channels; c.outC = d->out_channels;
  if(c.ky < 1 || c.kx < 1 || c.inC < 1 || c.outC < 1) throw Error(KMX_ERR_INVALID_ARG, "bad conv desc");
  c.w.resize((size_t)c.ky * c.kx * c.inC * c.outC);
  for(int o = 0; o < c.outC; o++)
    for(int i = 0; i < c.inC; i++)
      for(int y = 0; y < c.ky; y++)
        for(int x = 0; x < c.kx; x++)
          c.w[(((size_t)y * c.kx + x) * c.inC + i) * c.outC + o] = d->weights[(((size_t)o * c.inC + i) * c.ky + y) * c.kx + x];
  return c;
}
BnDesc bnFromAbi(const kmx_bnact_desc* d) {
  if(!d || !d->merged_scale || !d->merged_bias) throw Error(KMX_ERR_INVALID_ARG, "null bn desc");
  BnDesc b;
  b.name = "testbn";
  b.c = d->num_channels;
  b.act = d->activation;
  b.scale.assign(d->merged_scale, d->merged_scale + b.c);
  b.bias.assign(d->merged_bias, d->merged_bias + b.c);
  return b;
}
DevBuf uploadVec(const std::vector<float>& v) {
  DevBuf b(std::max<size_t>(v.size(), 1) * sizeof(float));
  b.upload(v.data(), v.size() * sizeof(float));
  return b;
}
void runBnAct(HookCtx& h, const BnDesc& bn, const DevBuf& in, DevBuf& out, int stride) {
  DevBuf sc = uploadVec(bn.scale), bi = uploadVec(bn.bias);
  BnActArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in.get(); a.out = out.get(); a.stride = stride; a.C = bn.c;
  a.scale = sc.as<float>(); a.bias = bi.as<float>(); a.actKind = bn.act; a.mask = h.mask.as<float>();
  a.N = h.N; a.S = h.S;
  hipCheck(launchBnAct(h.dtype, a, h.st), "bnact launch");
  hipCheck(hipStreamSynchronize(h.st), "sync");
}

}  // namespace

void testConv(int dtype, const kmx_conv_desc* d, int batch, int X, int Y, const float* in, float* out) {
  HookCtx h(dtype, batch, X, Y, nullptr);
  ConvDesc c = convFromAbi(d);
  FusedConv fc = buildFusedConv(dtype, {{&c, nullptr}}, nullptr);
  int inStride, outStride = roundUp(c.outC, 32);
  DevBuf x = h.toDevice(in, c.inC, &inStride);
  DevBuf y((size_t)batch * h.S * outStride * dtSize(h.dtype));
  h.conv(fc, x.get(), inStride, nullptr, 0, y.get(), outStride, 0, roundUp(c.outC, 8), nullptr, 0, 0, 0, KMX_ACT_IDENTITY);
  h.toHost(y, outStride, c.outC, out);
}

void testBnAct(int dtype, const kmx_bnact_desc* d, int batch, int X, int Y, const float* in, const float* mask, float* out) {
  HookCtx h(dtype, batch, X, Y, mask);
  BnDesc bn = bnFromAbi(d);
  int stride;
  DevBuf x = h.toDevice(in, bn.c, &stride);
  runBnAct(h, bn, x, x, stride);
  h.toHost(x, stride, bn.c, out);
}

void testResBlock(int dtype, const kmx_resblock_desc* d, int batch, int X, int Y, const float* in, const float* mask, float* out) {
  if(!d) throw Error(KMX_ERR_INVALID_ARG, "null resblock desc");
  HookCtx h(dtype, batch, X, Y, mask);
  BnDesc preBN = bnFromAbi(&d->pre_bn), midBN = bnFromAbi(&d->mid_bn);
  ConvDesc c1 = convFromAbi(&d->regular_conv), c2 = convFromAbi(&d->final_conv);
  FusedConv f1 = buildFusedConv(dtype, {{&c1, &midBN}}, nullptr);
  FusedConv f2 = buildFusedConv(dtype, {{&c2, nullptr}}, nullptr);
  const int C = preBN.c;
  int stride;
  DevBuf raw = h.toDevice(in, C, &stride);
  DevBuf act((size_t)batch * h.S * stride * dtSize(h.dtype));
  runBnAct(h, preBN, raw, act, stride);
  const int midStride = roundUp(c1.outC, 32);
  DevBuf mid((size_t)batch * h.S * midStride * dtSize(h.dtype));
  h.conv(f1, act.get(), stride, nullptr, 0, nullptr, 0, 0, 0, mid.get(), midStride, 0, f1.coutPad, midBN.act);
  h.conv(f2, mid.get(), midStride, raw.get(), stride, raw.get(), stride, 0, roundUp(C, 8), nullptr, 0, 0, 0, KMX_ACT_IDENTITY);
  h.toHost(raw, stride, C, out);
}

void testGPoolBlock(int dtype, const kmx_gpoolblock_desc* d, int batch, int X, int Y, const float* in, const float* mask,
                    float* out) {
  if(!d) throw Error(KMX_ERR_INVALID_ARG, "null gpoolblock desc");
  if(!d->gpool_to_bias_mul.weights) throw Error(KMX_ERR_INVALID_ARG, "null matmul desc");
  HookCtx h(dtype, batch, X, Y, mask);
  BnDesc preBN = bnFromAbi(&d->pre_bn), gBN = bnFromAbi(&d->gpool_bn), midBN = bnFromAbi(&d->mid_bn);
  ConvDesc cr = convFromAbi(&d->regular_conv), cg = convFromAbi(&d->gpool_conv), c2 = convFromAbi(&d->final_conv);
  std::vector<int> offs;
  FusedConv f1 = buildFusedConv(dtype, {{&cr, nullptr}, {&cg, &gBN}}, &offs);
  FusedConv f2 = buildFusedConv(dtype, {{&c2, nullptr}}, nullptr);
  const int C = preBN.c, R = cr.outC, G = cg.outC;
  int stride;
  DevBuf raw = h.toDevice(in, C, &stride);
  DevBuf act((size_t)batch * h.S * stride * dtSize(h.dtype));
  runBnAct(h, preBN, raw, act, stride);
  const int rStride = roundUp(R, 32), gStride = roundUp(G, 32);
  DevBuf r((size_t)batch * h.S * rStride * dtSize(h.dtype)), g((size_t)batch * h.S * gStride * dtSize(h.dtype));
  h.conv(f1, act.get(), stride, nullptr, 0, r.get(), rStride, 0, offs[1], g.get(), gStride, offs[1], f1.coutPad, gBN.act);
  // mask sums on the host (computeMaskSum, eigenbackend.cpp:124-134)
  std::vector<float> ms(batch, 0.0f);
  for(int b = 0; b < batch; b++)
    for(int p = 0; p < h.S; p++) ms[b] += mask ? mask[(size_t)b * h.S + p] : 1.0f;
  DevBuf dms = uploadVec(ms);
  std::vector<float> wv(d->gpool_to_bias_mul.weights,
                        d->gpool_to_bias_mul.weights + (size_t)d->gpool_to_bias_mul.in_channels * d->gpool_to_bias_mul.out_channels);
  DevBuf w = uploadVec(wv), sc = uploadVec(midBN.scale), bi = uploadVec(midBN.bias);
  GPoolArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.g = g.get(); ga.gStride = gStride; ga.G = G;
  ga.r = r.get(); ga.rStride = rStride; ga.R = R;
  ga.w = w.as<float>(); ga.scale = sc.as<float>(); ga.bias = bi.as<float>(); ga.actKind = midBN.act;
  ga.mask = h.mask.as<float>(); ga.maskSum = dms.as<float>(); ga.N = batch; ga.S = h.S;
  hipCheck(launchGPoolApply(dtype, ga, h.st), "gpool launch");
  h.conv(f2, r.get(), rStride, raw.get(), stride, raw.get(), stride, 0, roundUp(C, 8), nullptr, 0, 0, 0, KMX_ACT_IDENTITY);
  h.toHost(raw, stride, C, out);
}

// The seam of two 1x1 convolutions (pointwise_kernel.h) on its own: fused = 1 runs the one-launch kernel, fused = 0 the two
// convolution launches it replaces; both from the same re-tiled weights.
void testPointwisePair(int dtype, int batch, int X, int Y, int c1, int c2, int c3, const float* in, const float* resid, const float* w1,
                       const float* scale1, const float* bias1, int act1, const float* w2, const float* scale2, const float* bias2,
                       int act2, const float* mask, bool fused, float* outTrunkRaw, float* outMidRaw, float* outMidAct) {
  if(!in || !resid || !w1 || !w2 || !scale1 || !bias1 || !scale2 || !bias2 || !outTrunkRaw || !outMidRaw || !outMidAct)
    throw Error(KMX_ERR_INVALID_ARG, "test pointwise pair: null argument");
  if(fused && (dtype == DT_F32 || !pointwisePairSupported(c1, c2, c3)))
    throw Error(KMX_ERR_UNSUPPORTED, "test pointwise pair: no fused kernel for these channel counts / this precision");
  HookCtx h(dtype, batch, X, Y, mask);
  auto conv1x1 = [](const char* name, int ic, int oc, const float* w) {  // [oc][ic] -> file order [1][1][ic][oc]
    ConvDesc c;
    c.name = name; c.ky = c.kx = 1; c.inC = ic; c.outC = oc;
    c.w.resize((size_t)ic * oc);
    for(int o = 0; o < oc; o++)
      for(int i = 0; i < ic; i++) c.w[(size_t)i * oc + o] = w[(size_t)o * ic + i];
    return c;
  };
  auto bn = [](int c, const float* sc, const float* bi, int act) {
    BnDesc b;
    b.name = "testbn"; b.c = c; b.act = act;
    b.scale.assign(sc, sc + c);
    b.bias.assign(bi, bi + c);
    return b;
  };
  const ConvDesc cd1 = conv1x1("post", c1, c2, w1), cd2 = conv1x1("pre", c2, c3, w2);
  const BnDesc bn1 = bn(c2, scale1, bias1, act1), bn2 = bn(c3, scale2, bias2, act2);
  FusedConv f1 = buildFusedConv(dtype, {{&cd1, &bn1}}, nullptr), f2 = buildFusedConv(dtype, {{&cd2, &bn2}}, nullptr);
  int inStride, trunkStride;
  DevBuf x = h.toDevice(in, c1, &inStride);
  DevBuf trunkRaw = h.toDevice(resid, c2, &trunkStride);
  const int midStride = roundUp(c3, 32);
  const size_t cells = (size_t)batch * h.S;
  DevBuf trunkAct(cells * trunkStride * dtSize(h.dtype)), midRaw(cells * midStride * dtSize(h.dtype)), midAct(cells * midStride * dtSize(h.dtype));
  if(fused) {
    PwPairArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.in = x.get(); pa.inC = inStride; pa.w1 = f1.w.get();
    pa.resid = trunkRaw.get(); pa.rawOut = trunkRaw.get(); pa.trunkC = trunkStride; pa.actOut = nullptr;
    pa.scale1 = f1.scale.as<float>(); pa.bias1 = f1.bias.as<float>(); pa.actKind1 = act1;
    pa.w2 = f2.w.get(); pa.rawOut2 = midRaw.get(); pa.actOut2 = midAct.get(); pa.midC = midStride;
    pa.scale2 = f2.scale.as<float>(); pa.bias2 = f2.bias.as<float>(); pa.actKind2 = act2;
    pa.mask = h.mask.as<float>(); pa.cells = (long long)cells; pa.zeroPage = h.zero.get();
    pa.alone = 1;  // the unit hook tests the persistent kernel where it exists (KMX_PW_V2=0: the one-tile-per-group kernel)
    hipCheck(launchPointwisePair(dtype, c1, c2, c3, pa, h.st), "pointwise pair launch");
  }
  else {
    h.conv(f1, x.get(), inStride, trunkRaw.get(), trunkStride, trunkRaw.get(), trunkStride, 0, f1.coutPad, trunkAct.get(), trunkStride, 0,
           f1.coutPad, act1);
    h.conv(f2, trunkAct.get(), trunkStride, nullptr, 0, midRaw.get(), midStride, 0, f2.coutPad, midAct.get(), midStride, 0, f2.coutPad, act2);
  }
  hipCheck(hipStreamSynchronize(h.st), "sync");
  h.toHost(trunkRaw, trunkStride, c2, outTrunkRaw);
  h.toHost(midRaw, midStride, c3, outMidRaw);
  h.toHost(midAct, midStride, c3, outMidAct);
}

// ---- unit hooks for the transformer kernels (experimental; tests/test_gpu_transformer.py) ----
void testConvChain(int dtype, int batch, int X, int Y, int nConv, const float* xIn, const float* rIn, const float* w, const float* scale,
                   const float* bias, int act, const float* mask, int chained, float* outR, float* outX) {
  if(!xIn || !rIn || !w || !scale || !bias || !outR || !outX) throw Error(KMX_ERR_INVALID_ARG, "test conv chain: null argument");
  if(nConv != 2 && nConv != 4) throw Error(KMX_ERR_INVALID_ARG, "test conv chain: n_conv must be 2 or 4");
  if(chained != 0 && chained != 2 && chained != 4) throw Error(KMX_ERR_INVALID_ARG, "test conv chain: chained must be 0, 2 or 4");
  if(chained > nConv) chained = nConv;
  if(chained != 0 && (dtype == DT_F32 || !convChainSupported(act)))
    throw Error(KMX_ERR_UNSUPPORTED, "test conv chain: no chained kernel for this activation / this precision");
  const int C = CHAIN_CHANNELS;
  HookCtx h(dtype, batch, X, Y, mask);
  std::vector<FusedConv> fc;
  for(int k = 0; k < nConv; k++) {
    kmx_conv_desc d;
    d.conv_y_size = d.conv_x_size = 3;
    d.in_channels = d.out_channels = C;
    d.weights = w + (size_t)k * C * C * 9;
    const ConvDesc cd = convFromAbi(&d);
    BnDesc bn;
    bn.name = "testbn"; bn.c = C; bn.act = act;
    bn.scale.assign(scale + (size_t)k * C, scale + (size_t)(k + 1) * C);
    bn.bias.assign(bias + (size_t)k * C, bias + (size_t)(k + 1) * C);
    fc.push_back(buildFusedConv(dtype, {{&cd, &bn}}, nullptr));
  }
  int stride;
  DevBuf x = h.toDevice(xIn, C, &stride);
  DevBuf r = h.toDevice(rIn, C, &stride);
  const size_t cells = (size_t)batch * h.S;
  DevBuf tmp(cells * C * dtSize(h.dtype));
  if(chained == 0) {
    for(int k = 0; k < nConv; k++) {
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      a.w = fc[k].w.get(); a.wFrag = fc[k].wFrag.get(); a.zeroPage = h.zero.get(); a.inC = C; a.nChunks = fc[k].nChunks; a.coutPad = fc[k].coutPad;
      a.N = batch; a.X = X; a.Y = Y;
      a.scale = fc[k].scale.as<float>(); a.bias = fc[k].bias.as<float>(); a.actKind = act; a.mask = h.mask.as<float>();
      a.actC = C; a.actBegin = 0; a.actEnd = C;
      if(k % 2 == 0) { a.in = x.get(); a.actOut = tmp.get(); }
      else {
        a.in = tmp.get(); a.actOut = x.get();
        a.resid = r.get(); a.residC = C; a.rawOut = r.get(); a.rawC = C; a.rawBegin = 0; a.rawEnd = C;
      }
      hipCheck(launchConv(dtype, 3, 23, a, h.st), "test conv chain: convolution launch");
    }
  }
  else {
    for(int k0 = 0; k0 < nConv; k0 += chained) {
      ConvChainArgs ch;
      memset(&ch, 0, sizeof(ch));
      ch.in = x.get(); ch.zeroPage = h.zero.get(); ch.mask = h.mask.as<float>();
      ch.N = batch; ch.X = X; ch.Y = Y; ch.nConv = chained; ch.actKind = act;
      for(int k = 0; k < chained; k++) {
        ChainConv& c = ch.conv[k];
        c.w = fc[k0 + k].w.get(); c.scale = fc[k0 + k].scale.as<float>(); c.bias = fc[k0 + k].bias.as<float>();
        c.resid = k % 2 == 1 ? r.get() : nullptr;
        c.rawOut = k % 2 == 1 ? r.get() : nullptr;
        c.actOut = k % 2 == 1 ? x.get() : tmp.get();
      }
      hipCheck(launchConvChain(dtype, ch, h.st), "test conv chain: chain launch");
    }
  }
  hipCheck(hipStreamSynchronize(h.st), "sync");
  h.toHost(r, C, C, outR);
  h.toHost(x, C, C, outX);
}

void testRmsNorm(int dtype, int batch, int X, int Y, int C, float eps, const float* w, const float* beta, int actKind, bool perBoard,
                 const float* in, const float* mask, float* out) {
  if(!w || !in || !out || C < 1) throw Error(KMX_ERR_INVALID_ARG, "test rmsnorm: bad argument");
  HookCtx h(dtype, batch, X, Y, mask);
  int stride;
  DevBuf x = h.toDevice(in, C, &stride);
  DevBuf y((size_t)batch * h.S * stride * dtSize(h.dtype));
  DevBuf dw = uploadVec(std::vector<float>(w, w + C));
  DevBuf db = uploadVec(beta ? std::vector<float>(beta, beta + C) : std::vector<float>(1, 0.0f));
  DevBuf rms((size_t)batch * sizeof(float));
  RmsNormArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x.get(); a.inStride = stride; a.out = y.get(); a.outStride = stride; a.C = C; a.eps = eps;
  a.w = dw.as<float>(); a.beta = beta ? db.as<float>() : nullptr; a.actKind = actKind;
  a.mask = h.mask.as<float>(); a.N = batch; a.S = h.S;
  if(perBoard) {
    std::vector<float> ms(batch, 0.0f);
    for(int b = 0; b < batch; b++)
      for(int p = 0; p < h.S; p++) ms[b] += mask ? mask[(size_t)b * h.S + p] : 1.0f;
    DevBuf dms = uploadVec(ms);
    hipCheck(launchBoardRms(dtype, x.get(), stride, C, h.mask.as<float>(), dms.as<float>(), batch, h.S, eps, rms.as<float>(), h.st), "board rms launch");
    hipCheck(hipStreamSynchronize(h.st), "sync");
    a.boardRms = rms.as<float>();
  }
  hipCheck(launchRmsNorm(dtype, a, h.st), "rmsnorm launch");
  hipCheck(hipStreamSynchronize(h.st), "sync");
  h.toHost(y, stride, C, out);
}

void testAttention(int dtype, int batch, int X, int Y, int H, int KVH, int QD, int VD, const float* ropeCos, const float* ropeSin,
                   int ropeHeads, const float* q, const float* k, const float* v, const float* mask, float* out) {
  if(!q || !k || !v || !out || H < 1 || KVH < 1 || QD < 1 || VD < 1 || H % KVH != 0 || (ropeCos != nullptr) != (ropeSin != nullptr))
    throw Error(KMX_ERR_INVALID_ARG, "test attention: bad argument");
  if(!attentionDimsSupported(QD, VD)) throw Error(KMX_ERR_UNSUPPORTED, "test attention: head dims above 64");
  HookCtx h(dtype, batch, X, Y, mask);
  const int kOff = roundUp(H * QD, 8), vOff = kOff + roundUp(KVH * QD, 8), ctot = vOff + roundUp(KVH * VD, 8);
  const size_t cells = (size_t)batch * h.S;
  std::vector<float> cat(cells * ctot, 0.0f);
  for(size_t c = 0; c < cells; c++) {
    std::copy(q + c * H * QD, q + (c + 1) * H * QD, cat.begin() + c * ctot);
    std::copy(k + c * KVH * QD, k + (c + 1) * KVH * QD, cat.begin() + c * ctot + kOff);
    std::copy(v + c * KVH * VD, v + (c + 1) * KVH * VD, cat.begin() + c * ctot + vOff);
  }
  int stride;
  DevBuf qkv = h.toDevice(cat.data(), ctot, &stride);
  const int outStride = roundUp(H * VD, 32);
  DevBuf y(cells * outStride * dtSize(h.dtype));
  const size_t tbl = ropeCos ? (size_t)(ropeHeads > 1 ? KVH : 1) * (QD / 2) * h.S : 1;
  DevBuf dc = uploadVec(ropeCos ? std::vector<float>(ropeCos, ropeCos + tbl) : std::vector<float>(1, 1.0f));
  DevBuf ds = uploadVec(ropeSin ? std::vector<float>(ropeSin, ropeSin + tbl) : std::vector<float>(1, 0.0f));
  AttentionArgs a;
  memset(&a, 0, sizeof(a));
  a.qkv = qkv.get(); a.stride = stride; a.kOff = kOff; a.vOff = vOff;
  a.H = H; a.KVH = KVH; a.QD = QD; a.VD = VD;
  a.ropeCos = ropeCos ? dc.as<float>() : nullptr;
  a.ropeSin = ropeSin ? ds.as<float>() : nullptr;
  a.ropeHeads = ropeHeads > 1 ? KVH : 1;
  a.mask = h.mask.as<float>(); a.out = y.get(); a.outStride = outStride;
  a.scale = 1.0f / sqrtf((float)QD);
  a.N = batch; a.S = h.S;
  hipCheck(launchAttention(dtype, a, h.st), "attention launch");
  hipCheck(hipStreamSynchronize(h.st), "sync");
  h.toHost(y, outStride, H * VD, out);
}

void testSwiGlu(int dtype, int batch, int X, int Y, int F, const float* a1, const float* g, float* out) {
  if(!a1 || !g || !out || F < 1 || F % 8 != 0) throw Error(KMX_ERR_INVALID_ARG, "test swiglu: bad argument (ffn channels must be a multiple of 8)");
  HookCtx h(dtype, batch, X, Y, nullptr);
  const size_t cells = (size_t)batch * h.S;
  std::vector<float> cat(cells * 2 * F);
  for(size_t c = 0; c < cells; c++) {
    std::copy(a1 + c * F, a1 + (c + 1) * F, cat.begin() + c * 2 * F);
    std::copy(g + c * F, g + (c + 1) * F, cat.begin() + c * 2 * F + F);
  }
  int stride;
  DevBuf x = h.toDevice(cat.data(), 2 * F, &stride);
  const int outStride = roundUp(F, 32);
  DevBuf y(cells * outStride * dtSize(h.dtype));
  SwiGluArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x.get(); a.inStride = stride; a.gOff = F; a.F = F; a.out = y.get(); a.outStride = outStride; a.cells = cells;
  hipCheck(launchSwiGlu(dtype, a, h.st), "swiglu launch");
  hipCheck(hipStreamSynchronize(h.st), "sync");
  h.toHost(y, outStride, F, out);
}

}  // namespace kmx

// pointwise.hip — instantiations and dispatch of the fused 1x1 -> 1x1 seam kernel (pointwise_kernel.h).
#include <atomic>
#include <cstdlib>

#include "pointwise_kernel.h"
#include "pointwise2_kernel.h"
#include "pointwise3_kernel.h"

namespace kmx {

namespace {
using namespace pwk;

// (K1 = C1/32, WN1 = C2/128, WN2 = C3/64): b18c384nbt is (6, 3, 3): 192 -> 384 -> 192.
#define KMX_PW_LIST(X) X(6, 3, 3)

// Default: 8 waves x 128 cells, one work-group per CU. KMX_PW_WAVES=4 selects 4 waves x 64 cells, two work-groups per CU
// (pointwise_kernel.h): measured on MI355X (b18c384nbt, batch 256, A/B on one box) 41.0 k evals/s against 41.9 k, the
// seam's share of the step 21.2 % against 20.3 % - co-residence did not overlap the groups' memory and compute phases (neither
// did staggering the two half-batch streams), and each group multiplies with half the tile per weight slab.
int pwWaves() {
  static const int w = [] {
    const char* e = getenv("KMX_PW_WAVES");
    return e && atoi(e) == 4 ? 4 : 8;
  }();
  return w;
}

// The persistent, software-pipelined form (pointwise2_kernel.h) exists for the shape and activation of b18c384nbt
// (192 -> 384 -> 192, mish) and needs the activated trunk image to stay in LDS (actOut null, as the engine passes it);
// everything else, and KMX_PW_V2=0, runs the one-tile-per-work-group kernel above.
bool persistentWanted() {
  static const bool on = [] {
    const char* e = getenv("KMX_PW_V2");
    return e == nullptr || atoi(e) != 0;
  }();
  return on;
}
bool persistentForced() {  // KMX_PW_V2=2: the persistent kernel whatever runs beside it (A/B)
  static const bool on = [] {
    const char* e = getenv("KMX_PW_V2");
    return e != nullptr && atoi(e) == 2;
  }();
  return on;
}
int numComputeUnits() {  // per device: one persistent work-group per CU (KMX_PW_GRID overrides: tests walk several tiles per group)
  static const int forced = [] {
    const char* e = getenv("KMX_PW_GRID");
    return e ? atoi(e) : 0;
  }();
  if(forced > 0) return forced;
  constexpr int MAX_DEVICES = 64;
  static std::atomic<int> cus[MAX_DEVICES];
  int dev = 0;
  if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
  int n = cus[dev].load(std::memory_order_acquire);
  if(n == 0) {
    hipDeviceProp_t prop;
    n = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    cus[dev].store(n, std::memory_order_release);
  }
  return n;
}

// Which seam kernel (KMX_PW_KERNEL = 3 | 2 | 1, default 3):
//   3  pointwise3_kernel.h (round 5): weights resident on the CU (AGPRs + LDS), one wave per SIMD, 64-cell tiles;
//   2  pointwise2_kernel.h (round 3): persistent, software-pipelined, weights re-fetched per 128-cell tile;
//   1  pointwise_kernel.h: one tile per work-group.
// 3 and 2 are persistent work-groups that own their CU (150 KB of LDS): taken when the launch has the chip to itself or holds at least
// two tiles of 128 cells per CU (PwPairArgs::alone, kernels.h); a smaller launch beside another stream's kernels takes kernel 1.
// (KMX_PW_V2 = 0 | 2 of rounds 3-4 still reads as "never" / "always" persistent.)
int pwKernel() {
  static const int k = [] {
    const char* e = getenv("KMX_PW_KERNEL");
    const int v = e ? atoi(e) : 3;
    return v >= 1 && v <= 3 ? v : 3;
  }();
  return k;
}

template <class TR>
hipError_t launchT(int c1, int c2, int c3, const PwPairArgs& a, hipStream_t stream) {
  if(c1 == 192 && c2 == 384 && c3 == 192 && a.actOut == nullptr && a.actKind1 == a.actKind2 && pwWaves() == 8 && persistentWanted() && pwKernel() >= 2 &&
     (a.alone != 0 || (a.cells + pw2::TM - 1) / pw2::TM >= 2LL * numComputeUnits() || persistentForced()))
  {
    if(pwKernel() == 3) {
      if(a.actKind1 == KMX_ACT_MISH) {
        if(a.dbg != nullptr) return pw3::launchResident<TR, 6, 12, 6, KMX_ACT_MISH, KMX_ACT_MISH, true>(a, numComputeUnits(), stream);
        return pw3::launchResident<TR, 6, 12, 6, KMX_ACT_MISH, KMX_ACT_MISH>(a, numComputeUnits(), stream);
      }
      if(a.actKind1 == KMX_ACT_MISH_SCALE8 && a.dbg == nullptr)
        return pw3::launchResident<TR, 6, 12, 6, KMX_ACT_MISH_SCALE8, KMX_ACT_MISH_SCALE8>(a, numComputeUnits(), stream);
    }
    if(a.actKind1 == KMX_ACT_MISH) {
      if(a.dbg != nullptr) return pw2::launchPersistent<TR, 6, 12, 3, KMX_ACT_MISH, KMX_ACT_MISH, true>(a, numComputeUnits(), stream);
      return pw2::launchPersistent<TR, 6, 12, 3, KMX_ACT_MISH, KMX_ACT_MISH>(a, numComputeUnits(), stream);
    }
    // the same net under the fp16 range transform (model_desc.cpp scaledBy8)
    if(a.actKind1 == KMX_ACT_MISH_SCALE8 && a.dbg == nullptr)
      return pw2::launchPersistent<TR, 6, 12, 3, KMX_ACT_MISH_SCALE8, KMX_ACT_MISH_SCALE8>(a, numComputeUnits(), stream);
  }
  // the 4-wave shape: a wave of GEMM 1 owns C2/2 channels (WN1 doubles), of GEMM 2 C3/2 (WN2 as is)
#define KMX_PW(K1_, WN1_, WN2_) \
  if(c1 == 32 * K1_ && c2 == 128 * WN1_ && c3 == 64 * WN2_) \
    return pwWaves() == 8 ? launchPair<TR, K1_, WN1_, WN2_, 128, 8>(a, stream) : launchPair<TR, K1_, 2 * WN1_, WN2_, 64, 4>(a, stream);
  KMX_PW_LIST(KMX_PW)
#undef KMX_PW
  return hipErrorInvalidValue;
}
}  // namespace

bool pointwisePairSupported(int c1, int c2, int c3) {
#define KMX_PW(K1_, WN1_, WN2_) \
  if(c1 == 32 * K1_ && c2 == 128 * WN1_ && c3 == 64 * WN2_) return true;
  KMX_PW_LIST(KMX_PW)
#undef KMX_PW
  return false;
}

hipError_t launchPointwisePair(int dtype, int c1, int c2, int c3, const PwPairArgs& a, hipStream_t stream) {
  if(a.inC < c1 || a.trunkC < c2 || a.midC < c3 || a.inC % 8 != 0 || a.trunkC % 8 != 0 || a.midC % 8 != 0) return hipErrorInvalidValue;
  if(dtype == DT_F16) return launchT<TraitsF16>(c1, c2, c3, a, stream);
  if(dtype == DT_BF16) return launchT<TraitsBF16>(c1, c2, c3, a, stream);
  return hipErrorInvalidValue;
}

}  // namespace kmx

// conv_mfma.hip — product instantiations and dispatch of the MFMA convolution (kernel: conv_kernel.h).
#include <cstdlib>

#include "conv_kernel.h"
#include "conv_small_kernel.h"

namespace kmx {

namespace {
using namespace convk;

// cfg = 10*WNW + WN : WNW = waves along the channel dimension (1: 4-wave work-group, 2: 8-wave), WN = 32-channel
// tiles per wave; a work-group covers 32*WN*WNW output channels of one board.
// The instantiated (kernel size, WNW, WN, ring depth) combinations: one list for the dispatcher and for
// convCfgInstantiated(), which the chooser and tests/test_conv_chooser.py check against.
#define KMX_CFG_LIST(X)                      \
  X(3, 1, 1, 2) X(3, 1, 2, 2) X(3, 1, 3, 2)  \
  X(3, 2, 2, 3) X(3, 2, 3, 3)                \
  X(1, 1, 1, 2) X(1, 1, 2, 2) X(1, 1, 3, 2)  \
  X(1, 2, 2, 3) X(1, 2, 3, 3)                \
  X(5, 1, 1, 2) X(5, 1, 3, 2)                \
  X(5, 2, 2, 2)
// (no 5x5 4-wave x 64-channel shape: it spills 50 registers to scratch - found in round 3 with -Rpass-analysis=kernel-resource-usage,
// the same cause as the two "unexplained" 2x cliffs of round 2, ring depth 4 and the even-tap barrier variant)

// The small-batch 3x3 shape, cfg 111 = 12 cell waves x 32 channels (conv_kernel.h, Geom<.., CW = 12>): a work-group still owns a board
// and 32 output channels, but twelve waves of ONE 32-cell tile each instead of four waves of three. While the chip is not full a
// layer takes as long as ONE work-group does (they all run side by side on idle CUs), and a 4-wave work-group's step is 6 MFMAs plus
// two LDS-DMA requests per wave (a third of them padding) at ~160 cycles of issue each: 697 cycles per step, 320 of them requests
// (profiles/r03_steps/small_batch_conv_timing.txt). With twelve waves a step is 2 MFMAs per wave and at most ONE request (two
// waves fetch the 2 KB slab, ten share the board image, none padded). Same MFMAs per output in the same K order: bit-identical
// results (on the MI355X: the digest of 64 rows' outputs is the same with either shape and at every batch size,
// profiles/r03_steps/small_batch_cw12/). Measured there, b18c384nbt through kmx_eval: 26.9 -> 23.4 us per 3x3 launch at batch 1,
// 28.1 -> 23.8 at batch 8; a pass 2.59 -> 2.35 ms (batch 1), 2.94 -> 2.60 (8), 3.36 -> 3.10 (32), 3.58 -> 3.30 (42). Less than the
// request count suggests: with two MFMAs per wave and step nothing hides the LDS read latency of the fragments any more, and a
// launch costs ~14 us whatever it does (the 1x1 layers and the small kernels of the same pass take 14-37 us each).
// KMX_CONV_CW12=0 / 1 overrides the default (on).
// (Round 4 measured the variant that reads a step's fragments one step ahead, behind the current MFMAs: 25.7 against 23.4 us per launch,
// 2.47 against 2.31 ms per pass at batch 1 - slower, removed; profiles/r04_steps/call1/small_batch_scan.txt.)
constexpr bool kCw12Default = true;
bool cw12Enabled() {
  static const bool on = [] {
    const char* e = getenv("KMX_CONV_CW12");
    return e != nullptr ? e[0] == '1' : kCw12Default;
  }();
  return on;
}
constexpr int CFG_CW12 = 111;
// The small-batch 3x3 shape with dedicated fetching waves, cfg 118 (conv_small_kernel.h, round 4): a board x 32 channels like cfg 11 and
// cfg 111, four multiplying waves of three cell tiles each and four waves that issue every LDS-DMA request. KMX_CONV_LOADERS=0 / 1
// overrides the default.
constexpr int CFG_LOADERS = 118;
constexpr int CFG_LOADERS_PACKED = 119;  // the same with two work-groups per CU (conv_small_kernel.h PACK)
constexpr bool kLoadersDefault = true;
bool loadersEnabled() {
  static const bool on = [] {
    const char* e = getenv("KMX_CONV_LOADERS");
    return e != nullptr ? e[0] == '1' : kLoadersDefault;
  }();
  return on;
}
template <class TR>
hipError_t launchT(int ks, int cfg, const ConvArgs& a, hipStream_t stream) {
  if(ks == 3 && cfg == CFG_CW12) return launchOne<TR, 3, 1, 1, 2, 0, 12>(a, stream);
  if(ks == 3 && cfg == CFG_LOADERS) return smallk::launchSmall<TR, false>(a, stream);
  if(ks == 3 && cfg == CFG_LOADERS_PACKED) return smallk::launchSmall<TR, true>(a, stream);
#define KMX_CFG(KS_, WNW_, WN_, D_) \
  if(ks == KS_ && cfg == 10 * WNW_ + WN_) return launchOne<TR, KS_, WN_, WNW_, D_, 0>(a, stream);
  KMX_CFG_LIST(KMX_CFG)
#undef KMX_CFG
  return hipErrorInvalidValue;
}

}  // namespace

hipError_t launchConv(int dtype, int ks, int cfg, const ConvArgs& a, hipStream_t stream) {
  if(a.X < 2 || a.Y < 2 || a.X > MAXLEN || a.Y > MAXLEN || a.N <= 0) return hipErrorInvalidValue;
  if(a.inC % 8 != 0 || a.coutPad % 32 != 0 || (a.nChunks + 4) * 64 > ZERO_PAGE_BYTES) return hipErrorInvalidValue;
  if(dtype == DT_F16) return launchT<TraitsF16>(ks, cfg, a, stream);
  if(dtype == DT_BF16) return launchT<TraitsBF16>(ks, cfg, a, stream);
  return hipErrorInvalidValue;
}

bool convCfgInstantiated(int ks, int cfg) {
  if(ks == 3 && (cfg == CFG_CW12 || cfg == CFG_LOADERS || cfg == CFG_LOADERS_PACKED)) return true;
#define KMX_CFG(KS_, WNW_, WN_, D_) \
  if(ks == KS_ && cfg == 10 * WNW_ + WN_) return true;
  KMX_CFG_LIST(KMX_CFG)
#undef KMX_CFG
  return false;
}

// Work-group shape for a convolution with coutPad (a multiple of 32) output channels on `batch` boards (the caller
// passes boards x concurrent streams). Rules read off tools/small_batch_sweep.py on MI355X
// (profiles/r01_final/small_batch_sweep.log): a work-group's time is set by its 6..12 x taps steps almost regardless of
// how many channels it covers, so while the chip is not full the NARROWEST shape (most work-groups) wins: one board
// of a 3x3 192->192 takes 21.6 us as 6 work-groups of 32 channels, 42.6 us as 2 of 96, 61.7 us as 1 of 192. Once the
// narrow shape would need more than one round of work-groups, the next wider one takes over; the 8-wave shapes
// (one image fetch for 128/192 channels) win when they alone fill most of the 256 CUs.
int chooseConvCfg(int ks, int coutPad, int batch) {
  const int tiles = coutPad / 32;
  static const int minWgs8 = [] {  // experiments only: KMX_MIN_WGS8 moves the switch to the 8-wave x 192 shape
    const char* e = getenv("KMX_MIN_WGS8");
    return e ? atoi(e) : 150;
  }();
  // a shape is a candidate if it tiles the channels AND exists for this kernel size (5x5 has no 8-wave x 192 shape: its
  // ring would not fit the LDS); 1x1 never uses the 4-wave x 96 shape (measured slower than 4-wave x 64)
  auto fits = [&](int cfg) { return tiles % ((cfg / 10) * (cfg % 10)) == 0 && convCfgInstantiated(ks, cfg) && !(ks == 1 && cfg == 13); };
  auto wgs = [&](int cfg) { return batch * (tiles / ((cfg / 10) * (cfg % 10))); };
  const int widest8 = fits(23) ? 23 : fits(22) ? 22 : 0;
  if(widest8 && wgs(widest8) >= minWgs8) return widest8;
  // twelve waves per 32-channel work-group while that is at most one work-group per CU (it is the only one on its CU; measured up to
  // there - KMX_CONV_CW12_MAX_WGS moves the limit for scans beyond it)
  static const int cw12MaxWgs = [] {
    const char* e = getenv("KMX_CONV_CW12_MAX_WGS");
    return e ? atoi(e) : 256;
  }();
  // the fetching-waves shape while it is the only work-group on its CU (134 VGPRs x 8 waves: one work-group per CU;
  // KMX_CONV_LOADERS_MAX_WGS moves the limit for scans beyond it)
  static const int loadersMaxWgs = [] {
    const char* e = getenv("KMX_CONV_LOADERS_MAX_WGS");
    return e ? atoi(e) : 256;
  }();
  if(ks == 3 && loadersEnabled() && batch * tiles <= loadersMaxWgs) return CFG_LOADERS;
  // ... and two per CU up to twice that (KMX_CONV_LOADERS_PACKED_MAX_WGS; 0 = off)
  static const int packedMaxWgs = [] {
    const char* e = getenv("KMX_CONV_LOADERS_PACKED_MAX_WGS");
    return e ? atoi(e) : 0;
  }();
  if(ks == 3 && loadersEnabled() && batch * tiles <= packedMaxWgs) return CFG_LOADERS_PACKED;
  if(ks == 3 && cw12Enabled() && batch * tiles <= cw12MaxWgs) return CFG_CW12;
  // 3x3/5x5 narrow shapes keep two work-groups per CU (LDS), 1x1 shapes one
  const int round = ks == 1 ? 200 : 420;
  if(fits(11) && wgs(11) <= round) return 11;
  if(fits(12) && wgs(12) <= round) return 12;
  if(ks == 1) {
    if(fits(22) && wgs(22) <= round) return 22;
    if(widest8) return widest8;
  }
  if(fits(13)) return 13;
  if(fits(12)) return 12;
  return 11;
}

}  // namespace kmx

// conv_mfma.hip — product instantiations and dispatch of the MFMA convolution (kernel: conv_kernel.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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

// The small-batch 3x3 shape with dedicated fetching waves, cfg 118 (conv_small_kernel.h, round 4): a board x 32 channels like cfg 11,
// four multiplying waves of three cell tiles each and four waves that issue every LDS-DMA request: 22.0 -> 17.3 us per 3x3 launch at
// batch 1 against round 3's shape of twelve cell waves (cfg 111, removed), a pass 2.30 -> 2.00 ms at batch 1, 2.51 -> 2.34 at batch 8
// (profiles/r04_steps/small_batch). KMX_CONV_TUNE (below) switches it off for A/B.
// 1x1 at small batch, cfg 114 (round 4): the 4-wave x 32-channel shape of conv_kernel.h with a ring of FOUR steps instead of two. A 1x1
// step is a whole 32-channel image chunk (23 KB) and two MFMAs per wave; with one step of cover every step waited out a memory round
// trip - 14.4 us of kernel time for the 12 steps of a 384 -> 192 layer at batch 8 (profiles/r04_steps/call1/trace_pass8_kernel_stats.csv),
// 37 such launches per pass. One work-group per CU either way (LDS 81 -> 132 KB). (KMX_CONV_TUNE deep1x1=0: the two-step ring.) A ring of
// five (157 KB) measured the same as four and is gone (profiles/r04_steps/small_batch/deep1x1_scan.txt).
constexpr int CFG_DEEP1X1_4 = 114;
// cfg 114 with a board's cell tiles over three work-groups (conv_kernel.h ABL_SPLIT) while batch x tiles x 3 <= 256: 1x1 launch 14.3 ->
// 11.5 us at batch 1, a pass 1.53 -> 1.44 ms (1.72 -> 1.67 at batch 8); profiles/r04_steps/small_batch/split1x1_scan.txt
constexpr int CFG_DEEP1X1_SPLIT = 113;
constexpr int CFG_DEEP1X1_64 = 124;  // the 4-wave x 64-channel shape with a ring of four (LDS 86 -> 141 KB), for the next 256 work-groups
constexpr int CFG_LOADERS = 118;
// ... with a board's cell tiles over three work-groups (conv_small_kernel.h MTW = 1) while batch x channel tiles x 3 <= 256: same box,
// 3x3 launch 16.5 -> 12.4 us at batch 1, 17.6 -> 13.5 at 8; a pass 1.90 -> 1.60 ms / 2.07 -> 1.77 (profiles/r04_steps/small_batch/split_scan.txt)
constexpr int CFG_LOADERS_SPLIT = 117;
constexpr int CFG_LOADERS_PACKED = 119;  // the same with two work-groups per CU (conv_small_kernel.h PACK)
// Round 5: the fetching-waves shape with the WEIGHTS IN REGISTERS (conv_small_kernel.h REGW): every multiplying wave loads its weight
// fragments straight from global memory a chunk ahead, LDS holds only the board image, one barrier per chunk instead of nine.
constexpr int CFG_REGW = 128;        // a board x 32 channels in one work-group (in place of cfg 118)
constexpr int CFG_REGW_SPLIT = 127;  // its cell tiles over three work-groups (in place of cfg 117)
constexpr int CFG_REGW64 = 126;      // a board x 64 channels (two channel tiles per wave), one per CU, for the next 256 work-groups (in place of cfg 119)
constexpr int CFG_REGW_HALF = 125;   // cfg 128's cell tiles over TWO work-groups (two tiles | one), between the three-way split and cfg 128
// ---- ONE debug override for every switch of the shape choice: KMX_CONV_TUNE="key=value,key=value" (read once) ----
// Defaults are the product; the keys exist for A/B scans (tools/small_batch_scan.py) and for tests that force a shape at a size the
// chooser would not pick it for (tests/test_kernels_latest_completion.py, tests/test_engine_emulated.py). Unknown keys abort: a typo must
// not silently measure the default - reported through the normal error path since round 6 (convTuneError(): engine construction fails with
// KMX_ERR_INVALID_ARG, launchConv returns hipErrorInvalidValue) instead of abort() inside the embedder's process. (Until round 4 these were
// nine separate environment variables: a leftover one is named on stderr once.)
//   min_wgs8        129  work-groups from which the 8-wave x 192 / x 128 shapes are taken. Round 6 (150 until then): from 129 boards on the
//                        4-wave x 96 shape is more than one work-group per CU (a pass over 144 rows: 5.11 -> 4.32 ms; 130: 4.19 -> 4.12; at 96-112
//                        the 4-wave shapes stay 3-4 % ahead), and where two batches of ~110-135 rows share the chip - `benchmark -v 1600 -t
//                        256` 25.2 -> 26.7 k nnEvals/s, self-play at 32 games x 8 leaves 25.9 -> 27.2 k NN rows/s, two runs each
//                        (profiles/r06_steps/midbatch). Choosing shapes for the rows that SHARE the chip (the batcher telling an engine
//                        what runs beside its batch) was measured in the same call and adds nothing: removed.
//   loaders         1    the small-batch 3x3 shape with fetching waves (0: the 4-wave shapes of conv_kernel.h)
//   loaders_depth   1    its fetch depth: slabs six steps / images two chunks ahead (0: three / one); 16.94 -> 16.47 us per 3x3 launch at
//                        batch 1, a pass 1.93 -> 1.88 ms (profiles/r04_steps/small_batch/depth_scan.txt)
//   loaders_split   1    a board's cell tiles over three work-groups while batch x channel tiles x 3 <= loaders_max_wgs
//   loaders_max_wgs 256  work-groups up to which the fetching-waves shape is alone on its CU
//   packed_max_wgs  512  ... and up to which two of them share a CU (0: off)
//   deep1x1         1    1x1 at small batch on a ring of four steps (0: two)
//   deep1x1_max_wgs 256  work-groups up to which the 32-channel deep shape is taken (tests: 0 sends even tile counts to the 64-channel one)
//   split1x1        1    a board's cell tiles over three work-groups for 1x1 layers too
//   regw            3    the fetching-waves shapes with their weights in registers: 0 - none (the slab-ring shapes of round 4); 1 - cfg 128
//                        in place of 118; 2 - also the 64-channel one (cfg 126) where the two-per-CU shape (cfg 119) was taken; 3 - also
//                        cfg 127 (cell tiles over three work-groups) in place of 117. Measured on one box (profiles/r05_steps/regw): a pass
//                        from host rows 1.38 -> 1.31 ms at batch 1, 1.59 -> 1.49 at 8, 2.11 -> 1.93 at 16, 2.12 -> 1.98 at 32, 3.24 -> 2.88 at 64,
//                        3.78 -> 3.36 at 85; self-play at 8 games x 8 leaves 20.4 -> 21.7 k NN rows/s
//   regw64_max_wgs  256  work-groups up to which cfg 126 is taken
//   regw_half       1    cfg 125 (cell tiles over two work-groups) while batch x channel tiles x 2 <= loaders_max_wgs (0: cfg 128 there;
//                        2: also where the three-way split would be taken - tests). One box, a pass from host rows: 1.92 -> 1.78 ms at
//                        batch 15, 1.95 -> 1.79-1.83 at 16, 2.03-2.07 -> 1.89-1.95 at 21; production self-play at 8 games x 8 leaves 21.0
//                        -> 22.7-23.2 k NN rows/s (profiles/r06_steps/fault/fixed_*). Round 5 switched it on in its last hour and the
//                        driver's self-play run died of a GPU exception: the register-weights shapes read image fragments past their
//                        last chunk and never waited for them, and THIS instantiation's epilogue happened to re-use those registers for
//                        a row pointer (conv_small_kernel.h chunkBody). Fixed in the kernel, guarded by
//                        tools/check_async_loads.py (tests/test_async_loads.py) and tests/test_gpu_selfplay_production.py.
struct ConvTune {
  int minWgs8 = 129, loaders = 1, loadersDepth = 1, loadersSplit = 1, loadersMaxWgs = 256, packedMaxWgs = 512, deep1x1 = 1, deep1x1MaxWgs = 256,
      split1x1 = 1, regw = 3, regw64MaxWgs = 256, regwHalf = 1;
  std::string error;  // non-empty: KMX_CONV_TUNE could not be parsed (an unknown key)
};
const ConvTune& convTune() {
  static const ConvTune t = [] {
    ConvTune t;
    for(const char* legacy : {"KMX_CONV_LOADERS", "KMX_CONV_LOADERS_DEPTH", "KMX_CONV_LOADERS_SPLIT", "KMX_CONV_LOADERS_MAX_WGS", "KMX_CONV_PACKED_MAX_WGS",
                              "KMX_CONV_DEEP1X1", "KMX_CONV_DEEP1X1_MAX_WGS", "KMX_CONV_SPLIT1X1", "KMX_MIN_WGS8"})
      if(getenv(legacy) != nullptr)
        fprintf(stderr, "katamx: %s is no longer read (round 4): use KMX_CONV_TUNE=\"key=value,...\" (csrc/conv_mfma.hip)\n", legacy);
    const char* e = getenv("KMX_CONV_TUNE");
    if(e == nullptr) return t;
    const struct { const char* key; int* v; } keys[] = {
      {"min_wgs8", &t.minWgs8}, {"loaders", &t.loaders}, {"loaders_depth", &t.loadersDepth}, {"loaders_split", &t.loadersSplit},
      {"loaders_max_wgs", &t.loadersMaxWgs}, {"packed_max_wgs", &t.packedMaxWgs}, {"deep1x1", &t.deep1x1},
      {"deep1x1_max_wgs", &t.deep1x1MaxWgs}, {"split1x1", &t.split1x1}, {"regw", &t.regw}, {"regw64_max_wgs", &t.regw64MaxWgs}, {"regw_half", &t.regwHalf}};
    std::string s(e);
    size_t i = 0;
    while(i < s.size()) {
      size_t j = s.find(',', i);
      if(j == std::string::npos) j = s.size();
      const std::string item = s.substr(i, j - i);
      i = j + 1;
      if(item.empty()) continue;
      const size_t eq = item.find('=');
      bool known = false;
      if(eq != std::string::npos)
        for(const auto& k : keys)
          if(item.compare(0, eq, k.key) == 0 && strlen(k.key) == eq) {
            *k.v = atoi(item.c_str() + eq + 1);
            known = true;
          }
      if(!known && t.error.empty()) t.error = "KMX_CONV_TUNE: unknown item '" + item + "' (keys: csrc/conv_mfma.hip)";
    }
    return t;
  }();
  return t;
}
template <class TR>
hipError_t launchT(int ks, int cfg, const ConvArgs& a, hipStream_t stream) {
  if(ks == 3 && cfg == CFG_LOADERS) {
    return convTune().loadersDepth == 0 ? smallk::launchSmall<TR, false, 0, 3>(a, stream) : smallk::launchSmall<TR, false, 1, 3>(a, stream);
  }
  if(ks == 3 && cfg == CFG_LOADERS_SPLIT) return smallk::launchSmall<TR, false, 1, 1>(a, stream);
  if(ks == 3 && cfg == CFG_LOADERS_PACKED) return smallk::launchSmall<TR, true, 0, 3>(a, stream);
  if(ks == 3 && cfg == CFG_REGW) return smallk::launchSmall<TR, false, 1, 3, true>(a, stream);
  if(ks == 3 && cfg == CFG_REGW_SPLIT) return smallk::launchSmall<TR, false, 1, 1, true>(a, stream);
  if(ks == 3 && cfg == CFG_REGW64) return smallk::launchSmall<TR, false, 1, 3, true, 2>(a, stream);
  if(ks == 3 && cfg == CFG_REGW_HALF) return smallk::launchSmall<TR, false, 1, 2, true>(a, stream);
  if(ks == 1 && cfg == CFG_DEEP1X1_4) return launchOne<TR, 1, 1, 1, 4, 0>(a, stream);
  if(ks == 1 && cfg == CFG_DEEP1X1_64) return launchOne<TR, 1, 2, 1, 4, 0>(a, stream);
  if(ks == 1 && cfg == CFG_DEEP1X1_SPLIT) return launchOne<TR, 1, 1, 1, 4, ABL_SPLIT>(a, stream);
#define KMX_CFG(KS_, WNW_, WN_, D_) \
  if(ks == KS_ && cfg == 10 * WNW_ + WN_) return launchOne<TR, KS_, WN_, WNW_, D_, 0>(a, stream);
  KMX_CFG_LIST(KMX_CFG)
#undef KMX_CFG
  return hipErrorInvalidValue;
}

}  // namespace

const char* convTuneError() {
  const ConvTune& t = convTune();
  return t.error.empty() ? nullptr : t.error.c_str();
}

hipError_t launchConv(int dtype, int ks, int cfg, const ConvArgs& a, hipStream_t stream) {
  if(convTuneError() != nullptr) return hipErrorInvalidValue;  // a typo in KMX_CONV_TUNE must not silently measure the default
  if(a.X < 2 || a.Y < 2 || a.X > MAXLEN || a.Y > MAXLEN || a.N <= 0) return hipErrorInvalidValue;
  if(a.inC % 8 != 0 || a.coutPad % 32 != 0 || (a.nChunks + 4) * 64 > ZERO_PAGE_BYTES) return hipErrorInvalidValue;
  if(dtype == DT_F32) return launchConvF32(ks, a, stream);
  if(dtype == DT_F16) return launchT<TraitsF16>(ks, cfg, a, stream);
  if(dtype == DT_BF16) return launchT<TraitsBF16>(ks, cfg, a, stream);
  return hipErrorInvalidValue;
}

bool convCfgInstantiated(int ks, int cfg) {
  if(ks == 3 && (cfg == CFG_LOADERS || cfg == CFG_LOADERS_SPLIT || cfg == CFG_LOADERS_PACKED || cfg == CFG_REGW || cfg == CFG_REGW_SPLIT || cfg == CFG_REGW64 || cfg == CFG_REGW_HALF)) return true;
  if(ks == 1 && (cfg == CFG_DEEP1X1_4 || cfg == CFG_DEEP1X1_64 || cfg == CFG_DEEP1X1_SPLIT)) return true;
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
  const ConvTune& tn = convTune();
  // a shape is a candidate if it tiles the channels AND exists for this kernel size (5x5 has no 8-wave x 192 shape: its
  // ring would not fit the LDS); 1x1 never uses the 4-wave x 96 shape (measured slower than 4-wave x 64)
  auto fits = [&](int cfg) { return tiles % ((cfg / 10) * (cfg % 10)) == 0 && convCfgInstantiated(ks, cfg) && !(ks == 1 && cfg == 13); };
  auto wgs = [&](int cfg) { return batch * (tiles / ((cfg / 10) * (cfg % 10))); };
  const int widest8 = fits(23) ? 23 : fits(22) ? 22 : 0;
  if(widest8 && wgs(widest8) >= tn.minWgs8) return widest8;
  // the fetching-waves shape while it is the only work-group on its CU (134 VGPRs x 8 waves: one work-group per CU), its cell tiles over
  // three work-groups while even that leaves CUs idle
  if(ks == 3 && tn.loaders && tn.loadersSplit && tn.regw >= 3 && tn.regwHalf >= 2 && batch * tiles * 2 <= tn.loadersMaxWgs) return CFG_REGW_HALF;
  if(ks == 3 && tn.loaders && tn.loadersSplit && batch * tiles * 3 <= tn.loadersMaxWgs) return tn.regw >= 3 ? CFG_REGW_SPLIT : CFG_LOADERS_SPLIT;
  // (split AND two work-groups per CU for the next 256 work-groups - 88 registers, 68 KB of LDS - measured within 1 % of the unsplit
  // shape at batch 16 - 28 and is not kept: profiles/r04_steps/small_batch/split1x1_scan.txt)
  if(ks == 3 && tn.loaders && tn.loadersSplit && tn.regw >= 3 && tn.regwHalf && batch * tiles * 2 <= tn.loadersMaxWgs) return CFG_REGW_HALF;
  if(ks == 3 && tn.loaders && batch * tiles <= tn.loadersMaxWgs) return tn.regw ? CFG_REGW : CFG_LOADERS;
  // ... and two per CU up to twice that. Measured on the MI355X, b18c384nbt device-resident: batch 43 2.72 -> 2.43 ms per pass,
  // 48 2.77 -> 2.51, 64 2.86 -> 2.69, 85 3.06 -> 3.05; beyond two per CU it loses (96: 3.89 -> 4.10 ms),
  // profiles/r04_steps/small_batch/mid_batch_packed.txt
  if(ks == 3 && tn.loaders && tn.regw >= 2 && tiles % 2 == 0 && batch * (tiles / 2) <= tn.regw64MaxWgs) return CFG_REGW64;
  if(ks == 3 && tn.loaders && batch * tiles <= tn.packedMaxWgs) return CFG_LOADERS_PACKED;
  // 1x1 while every work-group has a CU to itself: the deep ring
  if(ks == 1 && tn.deep1x1 && tn.split1x1 && batch * tiles * 3 <= tn.deep1x1MaxWgs) return CFG_DEEP1X1_SPLIT;
  if(ks == 1 && tn.deep1x1 && batch * tiles <= tn.deep1x1MaxWgs) return CFG_DEEP1X1_4;
  if(ks == 1 && tn.deep1x1 && tiles % 2 == 0 && batch * (tiles / 2) <= 256) return CFG_DEEP1X1_64;
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

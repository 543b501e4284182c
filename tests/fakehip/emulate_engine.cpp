// TEST INFRASTRUCTURE: the runtime half of a CPU build of libkatamx (tests/test_engine_emulated.py): the engine, the model
// loader, the C ABI, the small kernels (misc_kernels.hip) and the transformer kernels (transformer_kernels.hip) are
// compiled UNCHANGED for x86 against emul/hip/hip_runtime.h and executed on the CPU; this file supplies what cannot be
// emulated — the MFMA convolution kernel — as a plain-loop executor of the ConvArgs contract documented in
// katago_amd/csrc/kernels.h (same weight tiling and swizzle, same residual / ncBias / BN+activation / mask epilogue, same
// 16-bit rounding points), plus the shape chooser and the two benchmark entry points as stubs.
// What a whole-net run through this build checks: model parsing, weight re-tiling, buffer planning, strides and channel
// offsets of every launch, the small kernels and the transformer kernels — everything except the convolution kernel
// itself, which is verified on the MI355X. Nothing in the product links or loads this file.
#include <hip/hip_runtime.h>

#include "../../katago_amd/csrc/device_common.h"
#include "emul/emu_runtime.inc"

namespace kmx {

#ifndef KMX_EMU_REAL_CONV
int chooseConvCfg(int, int, int) { return 11; }
bool convCfgInstantiated(int, int) { return true; }
const char* convTuneError() { return nullptr; }
// no fused seam kernel in this build: the engine then schedules the two convolution launches
bool pointwisePairSupported(int, int, int) { return false; }
hipError_t launchPointwisePair(int, int, int, int, const PwPairArgs&, hipStream_t) { return 801; }
// ... and no chained convolution kernel: the engine then schedules one launch per convolution
bool convChainSupported(int) { return false; }
hipError_t launchConvChain(int, const ConvChainArgs&, hipStream_t) { return 801; }
#endif
double benchConv(int, int, int, int, int, int, int, int, int, int) { return 0.0; }
double benchConvStreams(int, int, int, int, int, int, double, int, int) { return 0.0; }
double benchMfma(int, int, int, int, int, double*, double*) { return 0.0; }
double benchLaunchFloor(int, int, int, int, int) { return 0.0; }
double benchMfmaSustained(int, int, int, int, double, double*, double*) { return 0.0; }
double benchSeam(int, int, int) { return 0.0; }
double benchConvChain(int, int, int, int, int) { return 0.0; }
hipError_t launchLdsSquatter(int, int, int, unsigned*, hipStream_t) { return 801; }  // (device-only triage tool)

#ifndef KMX_EMU_REAL_CONV  // the "real convolution" build compiles a transformed copy of conv_mfma.hip / conv_kernel.h instead
namespace {
template <class TR>
hipError_t convRef(int ks, const ConvArgs& a) {
  typedef typename TR::T T;
  const int S = a.X * a.Y, nt = ks * ks, halo = ks / 2;
  const T* in = (const T*)a.in;
  const T* w = (const T*)a.w;
  std::vector<float> acc(a.coutPad);
  for(int n = 0; n < a.N; n++)
    for(int y = 0; y < a.Y; y++)
      for(int x = 0; x < a.X; x++) {
        const int cell = y * a.X + x;
        const size_t gcell = (size_t)n * S + cell;
        for(int c = 0; c < a.coutPad; c++) {
          float v = 0.0f;
          if(a.resid != nullptr && c >= a.rawBegin && c < a.rawEnd) v = TR::toFloat(((const T*)a.resid)[gcell * a.residC + (c - a.rawBegin)]);
          acc[c] = v;
        }
        for(int chunk = 0; chunk < a.nChunks; chunk++)
          for(int t = 0; t < nt; t++) {
            const int yy = y + t / ks - halo, xx = x + t % ks - halo;
            if(yy < 0 || yy >= a.Y || xx < 0 || xx >= a.X) continue;
            const T* irow = in + ((size_t)n * S + yy * a.X + xx) * a.inC + chunk * KCHUNK;
            float iv[KCHUNK];
            for(int k = 0; k < KCHUNK; k++) iv[k] = TR::toFloat(irow[k]);
            const T* wslab = w + ((size_t)chunk * nt + t) * a.coutPad * WROW_HALFS;
            for(int c = 0; c < a.coutPad; c++) {
              const T* wr = wslab + (size_t)c * WROW_HALFS;
              float s = 0.0f;
              for(int k = 0; k < KCHUNK; k++) s += iv[k] * TR::toFloat(wr[(((k >> 3) ^ ((c >> 2) & 3)) << 3) + (k & 7)]);
              acc[c] += s;
            }
          }
        for(int c = 0; c < a.coutPad; c++) {
          const float v = acc[c] + (a.ncBias != nullptr ? a.ncBias[(size_t)n * a.ncBiasStride + c] : 0.0f);
          if(c >= a.rawBegin && c < a.rawEnd) ((T*)a.rawOut)[gcell * a.rawC + (c - a.rawBegin)] = TR::fromFloat(v);
          if(c >= a.actBegin && c < a.actEnd)
            ((T*)a.actOut)[gcell * a.actC + (c - a.actBegin)] = TR::fromFloat(actApply(v * a.scale[c] + a.bias[c], a.actKind) * a.mask[gcell]);
        }
      }
  return hipSuccess;
}
}  // namespace

hipError_t launchConv(int dtype, int ks, int cfg, const ConvArgs& a, hipStream_t) {
  (void)cfg;
  if(a.X < 2 || a.Y < 2 || a.X > 19 || a.Y > 19 || a.N <= 0) return hipErrorInvalidValue;
  if(a.inC % 8 != 0 || a.coutPad % 32 != 0 || (a.nChunks + 4) * 64 > ZERO_PAGE_BYTES || a.inC < a.nChunks * KCHUNK) return hipErrorInvalidValue;
  if(ks != 1 && ks != 3 && ks != 5) return hipErrorInvalidValue;
  if(dtype == DT_F16) return convRef<TraitsF16>(ks, a);
  if(dtype == DT_BF16) return convRef<TraitsBF16>(ks, a);
  if(dtype == DT_F32) return convRef<TraitsF32>(ks, a);  // (the product's own fp32 kernel, conv_f32.hip, runs in the "real convolution" build)
  return hipErrorInvalidValue;
}
#endif

}  // namespace kmx

// conv_mfma.hip — product instantiations and dispatch of the MFMA convolution (kernel: conv_kernel.h).
#include "conv_kernel.h"

namespace kmx {

namespace {
using namespace convk;

// cfg = 10*WNW + WN : WNW = waves along the channel dimension (1: 4-wave work-group, 2: 8-wave), WN = 32-channel
// tiles per wave; a work-group covers 32*WN*WNW output channels of one board.
template <class TR>
hipError_t launchT(int ks, int cfg, const ConvArgs& a, hipStream_t stream) {
#define KMX_CFG(KS_, WNW_, WN_, D_) \
  if(ks == KS_ && cfg == 10 * WNW_ + WN_) return launchOne<TR, KS_, WN_, WNW_, D_, 0>(a, stream);
  KMX_CFG(3, 1, 1, 2) KMX_CFG(3, 1, 2, 2) KMX_CFG(3, 1, 3, 2)
  KMX_CFG(3, 2, 2, 3) KMX_CFG(3, 2, 3, 3)
  KMX_CFG(1, 1, 1, 2) KMX_CFG(1, 1, 2, 2) KMX_CFG(1, 1, 3, 2)
  KMX_CFG(1, 2, 2, 3) KMX_CFG(1, 2, 3, 3)
  KMX_CFG(5, 1, 1, 2) KMX_CFG(5, 1, 2, 2) KMX_CFG(5, 1, 3, 2)
  KMX_CFG(5, 2, 2, 2)
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

// Work-group shape for a convolution with coutPad (a multiple of 32) output channels on `batch` boards.
// 4-wave work-groups of 96 channels leave LDS and registers for a second work-group on the same CU; 8-wave
// work-groups read the board image once for up to 192 channels. See profiles/ for the measurements behind the order.
int chooseConvCfg(int ks, int coutPad, int batch) {
  const int tiles = coutPad / 32;
  // 8-wave work-groups (up to 192 channels of a board) once they fill the 256 CUs; below that the 4-wave shape gives
  // twice the work-groups (measured, profiles/r01_v5/sweep_v5.log: batch 128 3x3 54 us vs 71 us, batch 256 1x1 108 vs 78).
  static const int order8[] = {23, 22, 13, 12, 11};
  static const int order4[] = {13, 12, 23, 22, 11};
  int wgs8 = 0;
  for(int i = 0; i < 2 && wgs8 == 0; i++)
    if(tiles % (2 * (order8[i] % 10)) == 0 && !(ks == 5 && order8[i] == 23)) wgs8 = batch * (tiles / (2 * (order8[i] % 10)));
  const int* order = wgs8 >= 200 ? order8 : order4;
  for(int i = 0; i < 5; i++) {
    const int wnw = order[i] / 10, wn = order[i] % 10;
    if(ks == 5 && order[i] == 23) continue;
    if(tiles % (wn * wnw) == 0) return order[i];
  }
  return 11;
}

}  // namespace kmx

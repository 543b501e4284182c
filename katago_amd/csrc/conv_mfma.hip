// conv_mfma.hip — product instantiations and dispatch of the MFMA convolution (kernel: conv_kernel.h).
#include "conv_kernel.h"

namespace kmx {

namespace {
using namespace convk;

constexpr int DEPTH = 2;  // software-pipeline depth used by the engine (see conv_kernel.h; tuned in profiles/)

template <class TR>
hipError_t launchT(int ks, int wn, const ConvArgs& a, hipStream_t stream) {
  if(ks == 3) {
    if(wn == 1) return launchOne<TR, 3, 1, DEPTH, 0>(a, stream);
    if(wn == 2) return launchOne<TR, 3, 2, DEPTH, 0>(a, stream);
    if(wn == 3) return launchOne<TR, 3, 3, DEPTH, 0>(a, stream);
  }
  else if(ks == 1) {
    if(wn == 1) return launchOne<TR, 1, 1, DEPTH, 0>(a, stream);
    if(wn == 2) return launchOne<TR, 1, 2, DEPTH, 0>(a, stream);
    if(wn == 3) return launchOne<TR, 1, 3, DEPTH, 0>(a, stream);
  }
  else if(ks == 5) {
    if(wn == 1) return launchOne<TR, 5, 1, DEPTH, 0>(a, stream);
    if(wn == 2) return launchOne<TR, 5, 2, DEPTH, 0>(a, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace

hipError_t launchConv(int dtype, int ks, int wn, const ConvArgs& a, hipStream_t stream) {
  if(a.X < 2 || a.Y < 2 || a.X > MAXLEN || a.Y > MAXLEN || a.N <= 0) return hipErrorInvalidValue;
  if(a.inC % 8 != 0 || a.coutPad % 64 != 0) return hipErrorInvalidValue;
  if(dtype == DT_F16) return launchT<TraitsF16>(ks, wn, a, stream);
  if(dtype == DT_BF16) return launchT<TraitsBF16>(ks, wn, a, stream);
  return hipErrorInvalidValue;
}

int chooseConvWN(int ks, int coutPad, int batch) {
  const int tiles = coutPad / 64;
  const int maxWN = ks == 5 ? 2 : 3;
  // the widest tile (fewest re-reads of the board image) that still gives every CU a work-group
  for(int wn = maxWN; wn > 1; wn--)
    if(tiles % wn == 0 && (tiles / wn) * batch >= 200) return wn;
  return 1;
}

}  // namespace kmx

// pointwise.hip — instantiations and dispatch of the fused 1x1 -> 1x1 seam kernel (pointwise_kernel.h).
#include "pointwise_kernel.h"

namespace kmx {

namespace {
using namespace pwk;

// (K1 = C1/32, WN1 = C2/128, WN2 = C3/64): b18c384nbt is (6, 3, 3): 192 -> 384 -> 192.
#define KMX_PW_LIST(X) X(6, 3, 3)

template <class TR>
hipError_t launchT(int c1, int c2, int c3, const PwPairArgs& a, hipStream_t stream) {
#define KMX_PW(K1_, WN1_, WN2_) \
  if(c1 == 32 * K1_ && c2 == 128 * WN1_ && c3 == 64 * WN2_) return launchPair<TR, K1_, WN1_, WN2_, 128>(a, stream);
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

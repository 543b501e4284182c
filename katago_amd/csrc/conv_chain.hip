// conv_chain.hip — instantiations and dispatch of the chained 3x3 convolution (kernel: conv_chain_kernel.h).
#include "conv_chain_kernel.h"

namespace kmx {

bool convChainSupported(int actKind) {
  return actKind == KMX_ACT_MISH || actKind == KMX_ACT_MISH_SCALE8 || actKind == KMX_ACT_RELU;
}

namespace {
template <class TR>
hipError_t launchT(const ConvChainArgs& a, hipStream_t stream) {
  switch(a.actKind) {
    case KMX_ACT_MISH: return chaink::launchChainOne<TR, KMX_ACT_MISH>(a, stream);
    case KMX_ACT_MISH_SCALE8: return chaink::launchChainOne<TR, KMX_ACT_MISH_SCALE8>(a, stream);
    case KMX_ACT_RELU: return chaink::launchChainOne<TR, KMX_ACT_RELU>(a, stream);
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

hipError_t launchConvChain(int dtype, const ConvChainArgs& a, hipStream_t stream) {
  if(a.X < 2 || a.Y < 2 || a.X > convk::MAXLEN || a.Y > convk::MAXLEN || a.N <= 0) return hipErrorInvalidValue;
  if(a.nConv < 2 || a.nConv > MAX_CHAIN || a.in == nullptr || a.zeroPage == nullptr || a.mask == nullptr) return hipErrorInvalidValue;
  for(int i = 0; i < a.nConv; i++) {
    const ChainConv& c = a.conv[i];
    if(c.w == nullptr || c.scale == nullptr || c.bias == nullptr || c.actOut == nullptr) return hipErrorInvalidValue;
    if(c.resid != nullptr && c.rawOut == nullptr) return hipErrorInvalidValue;
  }
  if(dtype == DT_F16) return launchT<TraitsF16>(a, stream);
  if(dtype == DT_BF16) return launchT<TraitsBF16>(a, stream);
  return hipErrorInvalidValue;
}

}  // namespace kmx

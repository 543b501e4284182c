// conv_bench.hip — instrumentation only: times single launches of the MFMA convolution, including
// ablated variants (no epilogue / no MFMA / no DMA ...) and other pipeline depths, so that the dominant cost
// of the kernel can be located on hardware before it is optimised. Not used by the evaluation path.
#include <cstring>
#include <vector>

#include "conv_kernel.h"
#include "engine.h"

namespace kmx {
namespace {
using namespace convk;

// variant = D * 1000 + ABL
hipError_t launchVariant(int ks, int wn, int variant, const ConvArgs& a, hipStream_t st) {
  typedef TraitsBF16 TR;
#define V(KS_, WN_, D_, ABL_) \
  if(ks == KS_ && wn == WN_ && variant == D_ * 1000 + ABL_) return launchOne<TR, KS_, WN_, D_, ABL_>(a, st);
  V(3, 3, 1, 0) V(3, 3, 2, 0) V(3, 3, 3, 0)
  V(3, 3, 2, 1) V(3, 3, 2, 2) V(3, 3, 2, 4) V(3, 3, 2, 5) V(3, 3, 2, 12) V(3, 3, 2, 13) V(3, 3, 2, 16) V(3, 3, 2, 32) V(3, 3, 2, 64) V(3, 3, 2, 192) V(3, 3, 2, 65) V(3, 3, 2, 193) V(3, 3, 2, 80) V(3, 3, 2, 256) V(3, 3, 2, 512) V(3, 3, 2, 1024) V(3, 3, 2, 1536)
  V(3, 2, 2, 0) V(3, 1, 2, 0) V(3, 1, 3, 0)
  V(1, 3, 1, 0) V(1, 3, 2, 0) V(1, 3, 2, 1) V(1, 3, 2, 2) V(1, 3, 2, 4) V(1, 3, 2, 5) V(1, 3, 2, 32) V(1, 3, 2, 256) V(1, 3, 2, 257)
  V(1, 1, 2, 0) V(1, 1, 3, 0) V(1, 2, 2, 0)
#undef V
  return hipErrorInvalidValue;
}
}  // namespace

// epilogueMode: 0 = BN+act output only; 1 = residual in, raw out + BN+act out
double benchConv(int ks, int wn, int variant, int cin, int cout, int batch, int X, int Y, int epilogueMode, int iters) {
  const int dtype = DT_BF16;
  const int S = X * Y;
  const size_t cells = (size_t)batch * S;
  const int inStride = roundUp(cin, 32), outStride = roundUp(cout, 32);
  ConvDesc c;
  c.name = "bench";
  c.ky = c.kx = ks;
  c.inC = cin;
  c.outC = cout;
  c.w.resize((size_t)ks * ks * cin * cout);
  uint32_t rng = 12345;
  auto rnd = [&]() {
    rng = rng * 1664525u + 1013904223u;
    return ((rng >> 8) & 0xffff) / 65536.0f - 0.5f;
  };
  for(float& v : c.w) v = rnd() * 0.1f;
  BnDesc bn;
  bn.c = cout;
  bn.act = KMX_ACT_MISH;
  bn.scale.assign(cout, 1.0f);
  bn.bias.assign(cout, 0.1f);
  FusedConv fc = buildFusedConv(dtype, {{&c, &bn}}, nullptr);
  std::vector<uint16_t> hin(cells * inStride);
  for(uint16_t& v : hin) v = floatToBf16Bits(rnd());
  DevBuf in(hin.size() * 2, false), resid(cells * outStride * 2), raw(cells * outStride * 2), act(cells * outStride * 2), zero(4096);
  in.upload(hin.data(), hin.size() * 2);
  std::vector<float> ones(cells, 1.0f);
  DevBuf mask(cells * sizeof(float), false);
  mask.upload(ones.data(), cells * sizeof(float));
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in.get(); a.w = fc.w.get(); a.zeroPage = zero.get(); a.inC = inStride; a.nChunks = fc.nChunks; a.coutPad = fc.coutPad;
  a.N = batch; a.X = X; a.Y = Y;
  if(epilogueMode == 1) {
    a.resid = resid.get(); a.residC = outStride;
    a.rawOut = raw.get(); a.rawC = outStride; a.rawBegin = 0; a.rawEnd = std::min(fc.coutPad, outStride);
  }
  a.actOut = act.get(); a.actC = outStride; a.actBegin = 0; a.actEnd = std::min(fc.coutPad, outStride);
  a.scale = fc.scale.as<float>(); a.bias = fc.bias.as<float>(); a.actKind = KMX_ACT_MISH; a.mask = mask.as<float>();
  hipStream_t st = nullptr;
  auto launch = [&]() {
    hipError_t e = variant == 0 ? launchConv(dtype, ks, wn, a, st) : launchVariant(ks, wn, variant, a, st);
    hipCheck(e, "bench conv launch");
  };
  for(int i = 0; i < 3; i++) launch();
  hipCheck(hipStreamSynchronize(st), "sync");
  hipEvent_t e0, e1;
  hipCheck(hipEventCreate(&e0), "event");
  hipCheck(hipEventCreate(&e1), "event");
  hipCheck(hipEventRecord(e0, st), "record");
  for(int i = 0; i < iters; i++) launch();
  hipCheck(hipEventRecord(e1, st), "record");
  hipCheck(hipStreamSynchronize(st), "sync");
  float ms = 0;
  hipCheck(hipEventElapsedTime(&ms, e0, e1), "elapsed");
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return (double)ms / iters;
}

}  // namespace kmx

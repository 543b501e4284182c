// conv_bench.hip — instrumentation only: times single launches of the MFMA convolution, including
// ablated variants (no epilogue / no MFMA / no DMA ...) and other pipeline depths, so that the dominant cost
// of the kernel can be located on hardware before it is optimised. Not used by the evaluation path.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "conv_chain_kernel.h"
#include "conv_kernel.h"
#include "conv_small_kernel.h"
#include "engine.h"

namespace kmx {
namespace {
using namespace convk;

// variant = D * 1000 + ABL; cfg = 10*WNW + WN as in launchConv
hipError_t launchVariant(int ks, int cfg, int variant, const ConvArgs& a, hipStream_t st) {
  typedef TraitsBF16 TR;
#define V(KS_, WNW_, WN_, D_, ABL_) \
  if(ks == KS_ && cfg == 10 * WNW_ + WN_ && variant == D_ * 1000 + ABL_) return launchOne<TR, KS_, WN_, WNW_, D_, ABL_>(a, st);
  // The instantiated variants (each costs ~10 s of compile time; round 1-2 swept ~140 of them, results in profiles/r01*, r02_steps):
  // cycle stamps of the product shapes, the ablations of the dominant shape, ring depths 2 and 4 of it.
  V(3, 2, 3, 3, 2048) V(3, 2, 3, 3, 2049) V(3, 2, 3, 3, 2052) V(3, 2, 3, 3, 2056) V(1, 2, 3, 3, 2048) V(3, 1, 3, 2, 2048) V(3, 1, 1, 2, 2048)
  V(3, 2, 3, 3, 1) V(3, 2, 3, 3, 2) V(3, 2, 3, 3, 4) V(3, 2, 3, 3, 5) V(3, 2, 3, 3, 512) V(3, 2, 3, 3, 1024)
  V(3, 2, 3, 2, 0) V(3, 2, 3, 4, 0) V(3, 2, 3, 4, 2048)
  V(3, 2, 3, 3, 131072) V(1, 2, 3, 3, 131072) V(3, 1, 3, 2, 131072)  // ABL_BATCHED: the round-2 form of a step, for A/B
  V(3, 2, 3, 3, 0)
#undef V
  return hipErrorInvalidValue;
}
}  // namespace

// epilogueMode: 0 = BN+act output only; 1 = residual in, raw out + BN+act out
double benchConv(int ks, int cfg, int variant, int cin, int cout, int batch, int X, int Y, int epilogueMode, int iters) {
  // KMX_BENCH_DTYPE=fp16 times the fp16 instantiation (product dispatch only: variant 0)
  const char* dtEnv = getenv("KMX_BENCH_DTYPE");
  const int dtype = (dtEnv && std::string(dtEnv) == "fp16" && variant == 0) ? DT_F16 : DT_BF16;
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
  for(uint16_t& v : hin) v = dtype == DT_F16 ? floatToHalfBits(rnd()) : floatToBf16Bits(rnd());
  DevBuf in(hin.size() * 2, false), resid(cells * outStride * 2), raw(cells * outStride * 2), act(cells * outStride * 2), zero(ZERO_PAGE_ALLOC);
  in.upload(hin.data(), hin.size() * 2);
  std::vector<float> ones(cells, 1.0f);
  DevBuf mask(cells * sizeof(float), false);
  mask.upload(ones.data(), cells * sizeof(float));
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in.get(); a.w = fc.w.get(); a.wFrag = fc.wFrag.get(); a.zeroPage = zero.get(); a.inC = inStride; a.nChunks = fc.nChunks; a.coutPad = fc.coutPad;
  a.N = batch; a.X = X; a.Y = Y;
  if(epilogueMode == 1) {
    a.resid = resid.get(); a.residC = outStride;
    a.rawOut = raw.get(); a.rawC = outStride; a.rawBegin = 0; a.rawEnd = std::min(fc.coutPad, outStride);
  }
  a.actOut = act.get(); a.actC = outStride; a.actBegin = 0; a.actEnd = std::min(fc.coutPad, outStride);
  a.scale = fc.scale.as<float>(); a.bias = fc.bias.as<float>(); a.actKind = KMX_ACT_MISH; a.mask = mask.as<float>();
  DevBuf dbg(8 * 8 * sizeof(unsigned long long));
  a.dbg = dbg.as<unsigned long long>();
  hipStream_t st = nullptr;
  auto launch = [&]() {
    // variant 9999: the register-weights small-batch shape cfg 127 of conv_mfma.hip with cycle stamps (conv_small_kernel.h TIMING). Round 5
    // also stamped cfg 128 and 126: with the stamps' registers those two instantiations SPILLED (452 / 940 bytes of scratch per lane, found
    // in round 6 by tools/check_async_loads.py) - a spilled fragment register is copied while its load is in flight, so their stamps
    // described a kernel that is not the product's and they are gone (DESIGN.md 4.14 keeps the numbers with that caveat).
    hipError_t e = variant == 0      ? launchConv(dtype, ks, cfg, a, st)
                   : variant == 9999 ? (cfg == 127 ? smallk::launchSmall<TraitsBF16, false, 1, 1, true, 1, true>(a, st) : hipErrorInvalidValue)
                                     : launchVariant(ks, cfg, variant, a, st);
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
  const int r2 = variant - 2000, r3 = variant - 3000;  // variant = D*1000 + ABL with D in {2,3} for the timing variants
  if((r2 >= 2048 && r2 < 2304) || (r3 >= 2048 && r3 < 2304) || variant == 4000 + 2048) {  // ABL_TIMING: cycle sums of the last launch, one line per wave
    unsigned long long h[64];
    hipCheck(hipMemcpy(h, dbg.get(), sizeof(h), hipMemcpyDeviceToHost), "copy timing");
    const int nw = (cfg / 10) * 4;
    for(int w = 0; w < nw; w++)
      fprintf(stderr, "[timing] wave %d: wait+barrier %llu | MFMA F0 + read F1 %llu | DMA issue %llu | MFMA F1 + read F0' %llu | loop %llu "
                      "| prologue %llu | epilogue %llu | kernel %llu cycles\n",
              w, h[w * 8 + 0], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5], h[w * 8 + 6], h[w * 8 + 7]);
  }
  if(variant == 9999) {  // one line per wave of the middle board's first work-group, last launch
    unsigned long long h[64];
    hipCheck(hipMemcpy(h, dbg.get(), sizeof(h), hipMemcpyDeviceToHost), "copy timing");
    for(int w = 0; w < 4; w++)
      fprintf(stderr, "[timing] multiplying wave %d: start -> ready %llu | wait for the first image, fragments, parameters %llu | waits at the chunk barriers %llu | "
                      "the chunks' k halves %llu | epilogue %llu | kernel %llu cycles\n", w, h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 7]);
    for(int w = 4; w < 8; w++)
      fprintf(stderr, "[timing] fetching wave %d: start -> two images requested %llu | wait for image 0 %llu | waits for an image %llu | waits at the chunk barriers %llu | "
                      "kernel %llu cycles\n", w, h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 7]);
  }
  return (double)ms / iters;
}


// nConv (2 or 4) convolutions 3x3 192 -> 192 on `batch` boards: chained = 0: one launch of the product shape per convolution, else launches of
// `chained` convolutions (conv_chain_kernel.h). timing: the chained launches carry cycle stamps (printed per wave of the middle board).
double benchConvChain(int batch, int nConv, int chained, int iters, int timing) {
  const char* dtEnv = getenv("KMX_BENCH_DTYPE");
  const int dtype = (dtEnv && std::string(dtEnv) == "fp16") ? DT_F16 : DT_BF16;
  const int S = 361, C = CHAIN_CHANNELS, X = 19, Y = 19;
  const size_t cells = (size_t)batch * S;
  uint32_t rng = 777;
  auto rnd = [&]() {
    rng = rng * 1664525u + 1013904223u;
    return ((rng >> 8) & 0xffff) / 65536.0f - 0.5f;
  };
  std::vector<FusedConv> fc;
  for(int k = 0; k < nConv; k++) {
    ConvDesc c;
    c.name = "bench"; c.ky = c.kx = 3; c.inC = c.outC = C;
    c.w.resize((size_t)9 * C * C);
    for(float& v : c.w) v = rnd() * 0.05f;
    BnDesc bn;
    bn.c = C; bn.act = KMX_ACT_MISH;
    bn.scale.assign(C, 1.0f);
    bn.bias.assign(C, 0.1f);
    fc.push_back(buildFusedConv(dtype, {{&c, &bn}}, nullptr));
  }
  std::vector<uint16_t> hx(cells * C);
  for(uint16_t& v : hx) v = dtype == DT_F16 ? floatToHalfBits(rnd()) : floatToBf16Bits(rnd());
  DevBuf x(hx.size() * 2, false), r(cells * C * 2), tmp(cells * C * 2), zero(ZERO_PAGE_ALLOC);
  x.upload(hx.data(), hx.size() * 2);
  std::vector<float> ones(cells, 1.0f);
  DevBuf mask(cells * sizeof(float), false);
  mask.upload(ones.data(), cells * sizeof(float));
  DevBuf dbg(8 * MAX_CHAIN * 8 * sizeof(unsigned long long));
  hipStream_t st = nullptr;
  auto launch = [&]() {
    if(chained == 0) {
      for(int k = 0; k < nConv; k++) {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.w = fc[k].w.get(); a.wFrag = fc[k].wFrag.get(); a.zeroPage = zero.get(); a.inC = C; a.nChunks = fc[k].nChunks; a.coutPad = fc[k].coutPad;
        a.N = batch; a.X = X; a.Y = Y;
        a.scale = fc[k].scale.as<float>(); a.bias = fc[k].bias.as<float>(); a.actKind = KMX_ACT_MISH; a.mask = mask.as<float>();
        a.actC = C; a.actBegin = 0; a.actEnd = C;
        if(k % 2 == 0) { a.in = x.get(); a.actOut = tmp.get(); }
        else {
          a.in = tmp.get(); a.actOut = x.get();
          a.resid = r.get(); a.residC = C; a.rawOut = r.get(); a.rawC = C; a.rawBegin = 0; a.rawEnd = C;
        }
        hipCheck(launchConv(dtype, 3, 23, a, st), "bench chain: convolution launch");
      }
      return;
    }
    for(int k0 = 0; k0 < nConv; k0 += chained) {
      ConvChainArgs ch;
      memset(&ch, 0, sizeof(ch));
      ch.in = x.get(); ch.zeroPage = zero.get(); ch.mask = mask.as<float>();
      ch.N = batch; ch.X = X; ch.Y = Y; ch.nConv = chained; ch.actKind = KMX_ACT_MISH;
      ch.dbg = timing ? dbg.as<unsigned long long>() : nullptr;
      for(int k = 0; k < chained; k++) {
        ChainConv& c = ch.conv[k];
        c.w = fc[k0 + k].w.get(); c.scale = fc[k0 + k].scale.as<float>(); c.bias = fc[k0 + k].bias.as<float>();
        c.resid = k % 2 == 1 ? r.get() : nullptr;
        c.rawOut = k % 2 == 1 ? r.get() : nullptr;
        c.actOut = k % 2 == 1 ? x.get() : tmp.get();
      }
      hipError_t e = timing ? (dtype == DT_F16 ? chaink::launchChainOne<TraitsF16, KMX_ACT_MISH, true>(ch, st) : chaink::launchChainOne<TraitsBF16, KMX_ACT_MISH, true>(ch, st))
                            : launchConvChain(dtype, ch, st);
      hipCheck(e, "bench chain: chain launch");
    }
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
  if(timing && chained != 0) {
    unsigned long long h[8 * MAX_CHAIN * 8];
    hipCheck(hipMemcpy(h, dbg.get(), sizeof(h), hipMemcpyDeviceToHost), "copy timing");
    for(int w = 0; w < 8; w++)
      for(int ci = 0; ci < chained; ci++) {
        const unsigned long long* p = h + (w * MAX_CHAIN + ci) * 8;
        fprintf(stderr, "[chain timing] wave %d conv %d: start %llu | loop %llu | wait+barrier %llu | epilogue %llu | closing wait+barrier %llu cycles\n", w, ci,
                p[0], p[1], p[2], p[3], p[4]);
      }
  }
  return (double)ms / iters;
}

// ---- two streams, half a launch apart --------------------------------------------------------------------------------
// Round 2 staggered the second half-batch stream by WHOLE launches (after k launches both streams again start their kernels
// at the same moment, so the phases of co-resident work-groups stay aligned). This measures what a stagger of a FRACTION of a
// launch buys: each of `nStreams` streams runs `launches` back-to-back launches of the same layer on its own `batch` boards;
// stream i first spins for i * delayUs microseconds. Returns the wall time (ms) from the common start to the last stream's end.
__global__ void spinKernel(unsigned long long ticks100MHz) {
  const unsigned long long t0 = wall_clock64();
  while(wall_clock64() - t0 < ticks100MHz) __builtin_amdgcn_s_sleep(8);
}

double benchConvStreams(int ks, int cfg, int cin, int cout, int batch, int nStreams, double delayUs, int launches, int epilogueMode) {
  const int dtype = DT_BF16, X = 19, Y = 19, S = X * Y;
  if(nStreams < 1 || nStreams > 4) throw Error(KMX_ERR_INVALID_ARG, "benchConvStreams: 1..4 streams");
  const size_t cells = (size_t)batch * S;
  const int inStride = roundUp(cin, 32), outStride = roundUp(cout, 32);
  ConvDesc c;
  c.name = "bench";
  c.ky = c.kx = ks;
  c.inC = cin;
  c.outC = cout;
  c.w.resize((size_t)ks * ks * cin * cout);
  uint32_t rng = 777;
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
  std::vector<float> ones(cells, 1.0f);
  DevBuf zero(ZERO_PAGE_ALLOC);
  struct PerStream {
    DevBuf in, resid, raw, act, mask;
    hipStream_t st = nullptr;
    hipEvent_t done = nullptr;
    ConvArgs a;
  };
  std::vector<std::unique_ptr<PerStream>> ps;
  for(int i = 0; i < nStreams; i++) {
    std::unique_ptr<PerStream> p(new PerStream());
    p->in = DevBuf(hin.size() * 2, false);
    p->in.upload(hin.data(), hin.size() * 2);
    p->resid = DevBuf(cells * outStride * 2);
    p->raw = DevBuf(cells * outStride * 2);
    p->act = DevBuf(cells * outStride * 2);
    p->mask = DevBuf(cells * sizeof(float), false);
    p->mask.upload(ones.data(), cells * sizeof(float));
    hipCheck(hipStreamCreateWithFlags(&p->st, hipStreamNonBlocking), "stream");
    hipCheck(hipEventCreate(&p->done), "event");
    ConvArgs& a = p->a;
    memset(&a, 0, sizeof(a));
    a.in = p->in.get(); a.w = fc.w.get(); a.wFrag = fc.wFrag.get(); a.zeroPage = zero.get(); a.inC = inStride; a.nChunks = fc.nChunks; a.coutPad = fc.coutPad;
    a.N = batch; a.X = X; a.Y = Y;
    if(epilogueMode == 1) {
      a.resid = p->resid.get(); a.residC = outStride;
      a.rawOut = p->raw.get(); a.rawC = outStride; a.rawBegin = 0; a.rawEnd = std::min(fc.coutPad, outStride);
    }
    a.actOut = p->act.get(); a.actC = outStride; a.actBegin = 0; a.actEnd = std::min(fc.coutPad, outStride);
    a.scale = fc.scale.as<float>(); a.bias = fc.bias.as<float>(); a.actKind = KMX_ACT_MISH; a.mask = p->mask.as<float>();
    ps.push_back(std::move(p));
  }
  hipStream_t st0 = nullptr;
  hipEvent_t e0, e1;
  hipCheck(hipEventCreate(&e0), "event");
  hipCheck(hipEventCreate(&e1), "event");
  auto round = [&](int n) {
    hipCheck(hipEventRecord(e0, st0), "record");
    for(auto& p : ps) hipCheck(hipStreamWaitEvent(p->st, e0, 0), "wait");
    for(int i = 0; i < nStreams; i++)
      if(i > 0 && delayUs > 0) hipLaunchKernelGGL(spinKernel, dim3(1), dim3(64), 0, ps[i]->st, (unsigned long long)(delayUs * i * 100.0));
    for(int l = 0; l < n; l++)
      for(auto& p : ps) hipCheck(launchConv(dtype, ks, cfg, p->a, p->st), "bench conv launch");
    for(auto& p : ps) {
      hipCheck(hipEventRecord(p->done, p->st), "record");
      hipCheck(hipStreamWaitEvent(st0, p->done, 0), "wait");
    }
    hipCheck(hipEventRecord(e1, st0), "record");
    hipCheck(hipStreamSynchronize(st0), "sync");
  };
  round(2);
  round(launches);
  float ms = 0;
  hipCheck(hipEventElapsedTime(&ms, e0, e1), "elapsed");
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for(auto& p : ps) {
    (void)hipEventDestroy(p->done);
    (void)hipStreamDestroy(p->st);
  }
  return (double)ms;
}

// ---- the seam kernel alone (192 -> 384 -> 192, mish, bf16) on `batch` boards of 19x19 ----------------------------------------
// timing != 0 runs the instrumented instantiation of the persistent kernel and prints work-group 0's per-wave phase sums.
double benchSeam(int batch, int iters, int timing) {
  const char* dtEnv = getenv("KMX_BENCH_DTYPE");  // fp16 | bf16 (default)
  const int dtype = dtEnv != nullptr && strcmp(dtEnv, "fp16") == 0 ? DT_F16 : DT_BF16;
  const int S = 361, C1 = 192, C2 = 384, C3 = 192;
  const size_t cells = (size_t)batch * S;
  uint32_t rng = 4711;
  auto rnd = [&]() {
    rng = rng * 1664525u + 1013904223u;
    return ((rng >> 8) & 0xffff) / 65536.0f - 0.5f;
  };
  auto conv1x1 = [&](int ic, int oc) {
    ConvDesc c;
    c.name = "seam"; c.ky = c.kx = 1; c.inC = ic; c.outC = oc;
    c.w.resize((size_t)ic * oc);
    for(float& v : c.w) v = rnd() * 0.1f;
    return c;
  };
  auto bnOf = [&](int c) {
    BnDesc b;
    b.c = c; b.act = KMX_ACT_MISH;
    b.scale.assign(c, 1.0f);
    b.bias.assign(c, 0.1f);
    return b;
  };
  const ConvDesc cd1 = conv1x1(C1, C2), cd2 = conv1x1(C2, C3);
  const BnDesc bn1 = bnOf(C2), bn2 = bnOf(C3);
  FusedConv f1 = buildFusedConv(dtype, {{&cd1, &bn1}}, nullptr), f2 = buildFusedConv(dtype, {{&cd2, &bn2}}, nullptr);
  std::vector<uint16_t> hx(cells * C1), hr(cells * C2);
  for(uint16_t& v : hx) v = floatToTBits(dtype, rnd());
  for(uint16_t& v : hr) v = floatToTBits(dtype, rnd());
  DevBuf x(hx.size() * 2, false), trunk(hr.size() * 2, false), midRaw(cells * C3 * 2), midAct(cells * C3 * 2), zero(ZERO_PAGE_ALLOC);
  x.upload(hx.data(), hx.size() * 2);
  trunk.upload(hr.data(), hr.size() * 2);
  std::vector<float> ones(cells, 1.0f);
  DevBuf mask(cells * sizeof(float), false);
  mask.upload(ones.data(), cells * sizeof(float));
  DevBuf dbg(8 * 9 * sizeof(unsigned long long));
  PwPairArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.in = x.get(); pa.inC = C1; pa.w1 = f1.w.get();
  pa.resid = trunk.get(); pa.rawOut = trunk.get(); pa.trunkC = C2;
  pa.scale1 = f1.scale.as<float>(); pa.bias1 = f1.bias.as<float>(); pa.actKind1 = KMX_ACT_MISH;
  pa.w2 = f2.w.get(); pa.rawOut2 = midRaw.get(); pa.actOut2 = midAct.get(); pa.midC = C3;
  pa.scale2 = f2.scale.as<float>(); pa.bias2 = f2.bias.as<float>(); pa.actKind2 = KMX_ACT_MISH;
  pa.mask = mask.as<float>(); pa.cells = (long long)cells; pa.zeroPage = zero.get();
  pa.dbg = timing ? dbg.as<unsigned long long>() : nullptr;
  pa.alone = 1;
  hipStream_t st = nullptr;
  auto launch = [&]() { hipCheck(launchPointwisePair(dtype, C1, C2, C3, pa, st), "bench seam launch"); };
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
  if(timing) {
    unsigned long long h[72];
    hipCheck(hipMemcpy(h, dbg.get(), sizeof(h), hipMemcpyDeviceToHost), "copy timing");
    const char* k = getenv("KMX_PW_KERNEL");
    if(k == nullptr || atoi(k) == 3) {
      for(int w = 0; w < 4; w++)  // pointwise3_kernel.h: four waves; a block = matrix micro-steps of one GEMM half beside the values of an epilogue half
        fprintf(stderr, "[seam timing, resident weights] wave %d: top (requests, wait X, BX) %llu | gemm1(0)+epi2'(1) %llu | gemm1(1)+epi1(0) %llu | BA0 %llu | "
                        "gemm2(0)+epi1(1) %llu | BA1 %llu | gemm2(1)+epi2(0) %llu | tail epi2(1) %llu | kernel %llu cycles\n",
                w, h[w * 9 + 0], h[w * 9 + 1], h[w * 9 + 2], h[w * 9 + 3], h[w * 9 + 4], h[w * 9 + 5], h[w * 9 + 6], h[w * 9 + 7], h[w * 9 + 8]);
    }
    else
    for(int w = 0; w < 8; w++)
      fprintf(stderr, "[seam timing] wave %d: top %llu | gemm1(0) %llu | P1 wait+barrier %llu | gemm1 next %llu | epilogue 1 %llu | P2/P3 wait+barrier %llu "
                      "| gemm2 steps %llu | epilogue 2 %llu | kernel %llu cycles\n",
              w, h[w * 9 + 0], h[w * 9 + 1], h[w * 9 + 2], h[w * 9 + 3], h[w * 9 + 4], h[w * 9 + 5], h[w * 9 + 6], h[w * 9 + 7], h[w * 9 + 8]);
  }
  return (double)ms / iters;
}

// ---- MFMA issue-rate microbenchmark: the practical ceiling the convolution's main loop is measured against ----
// mode bit 1: s_barrier after every 18 MFMAs (the convolution's step); bit 2: 12 ds_read_b128 per step feeding the MFMAs.
template <int MODE>
__global__ __launch_bounds__(512) void mfmaPeakKernel(int steps, int barEvery, unsigned long long* clocks, float* sink) {
  extern __shared__ __attribute__((aligned(256))) char smem[];
  typedef TraitsBF16 TR;
  typedef TR::V8 V8;
  const int lane = threadIdx.x & 63;
  f32x16 acc[9];
#pragma unroll
  for(int i = 0; i < 9; i++)
#pragma unroll
    for(int r = 0; r < 16; r++) acc[i][r] = 0.0f;
  V8 wf[2][3], af[2][3];
#pragma unroll
  for(int kk = 0; kk < 2; kk++)
#pragma unroll
    for(int j = 0; j < 3; j++)
#pragma unroll
      for(int i = 0; i < 8; i++) {
        wf[kk][j][i] = (TR::T)(0.001f * (float)(lane + j));
        af[kk][j][i] = (TR::T)(0.002f * (float)(lane + kk));
      }
  if(MODE & 2) {
    for(int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)smem)[i] = 0.001f * (float)(i & 255);
    __syncthreads();
  }
  int v0 = lane, v1 = lane * 7, v2 = lane ^ 5, v3 = lane + steps;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for(int s = 0; s < steps; s++) {
    if(MODE & 2) {
      const char* base = smem + ((s & 3) * 8192) + (lane & 31) * 64 + ((((lane >> 5)) ^ ((lane >> 2) & 3)) << 4);
#pragma unroll
      for(int kk = 0; kk < 2; kk++)
#pragma unroll
        for(int j = 0; j < 3; j++) {
          wf[kk][j] = *(const V8*)(base + j * 2048 + (kk * 32 ^ 0));
          af[kk][j] = *(const V8*)(base + 6144 + j * 2048 + (kk * 32 ^ 0) % 2048);
        }
    }
#pragma unroll
    for(int kk = 0; kk < 2; kk++)
#pragma unroll
      for(int ct = 0; ct < 3; ct++)
#pragma unroll
        for(int pt = 0; pt < 3; pt++) acc[ct * 3 + pt] = TR::mfma(wf[kk][ct], af[kk][pt], acc[ct * 3 + pt]);
    if(MODE & 4) {
      // the convolution's per-step address arithmetic: ~64 VALU operations competing for the vector issue slot
#pragma unroll
      for(int i = 0; i < 16; i++) {
        v0 = v0 * 3 + i; v1 = (v1 >> 1) ^ v0; v2 = v2 + v1; v3 = v3 ^ (v2 << 2);
      }
    }
    if((MODE & 1) && (s % barEvery) == 0) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  if((MODE & 4) && (v0 + v1 + v2 + v3) == 0x12345) sink[lane + 64] = 1.0f;
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float t = 0.0f;
#pragma unroll
  for(int i = 0; i < 9; i++)
#pragma unroll
    for(int r = 0; r < 16; r++) t += acc[i][r];
  if(t == 12345.678f) sink[lane] = t;
  if(blockIdx.x == 0 && threadIdx.x == 0) {
    clocks[0] = c1 - c0;
    clocks[1] = w1 - w0;
  }
}

// ---- what it costs a wave to ISSUE its operand traffic in the small-batch step shape (6 MFMAs per step, one wave per SIMD) ----
// VARIANT 0: MFMAs only; 1: + two LDS-DMA instructions (global_load_lds, 1 KiB each) per step; 2: + two plain global_load_dwordx4
// into registers, written to LDS with ds_write_b128 a step later (register staging). Sources are 2 KiB per wave of an L2-resident
// buffer; at most the requests of two steps are in flight. (kmx_bench_mfma, mode = 256 + VARIANT; tools/issue_cost.py.)
template <int VARIANT>
__global__ __launch_bounds__(256) void issueCostKernel(int steps, const char* src, unsigned long long* clocks, float* sink) {
  extern __shared__ __attribute__((aligned(256))) char smemIc[];
  typedef TraitsBF16 TR;
  typedef TR::V8 V8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smemIc;
  f32x16 acc[3];
#pragma unroll
  for(int i = 0; i < 3; i++)
#pragma unroll
    for(int r = 0; r < 16; r++) acc[i][r] = 0.0f;
  V8 wf, af[3];
#pragma unroll
  for(int i = 0; i < 8; i++) {
    wf[i] = (TR::T)(0.001f * (float)lane);
    af[0][i] = af[1][i] = af[2][i] = (TR::T)(0.002f * (float)(lane + i));
  }
  const char* mine = src + ((size_t)blockIdx.x % 64) * 8192 + wave * 2048 + lane * 16;
  u32x4 stage[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for(int s = 0; s < steps; s++) {
    const unsigned slot = ldsBase + (unsigned)(wave * 4 + (s & 1) * 2) * 1024u;
    if(VARIANT == 1) {
      dma16(mine, slot);
      dma16(mine + 1024, slot + 1024);
    }
    if(VARIANT == 2) {
      *(__attribute__((address_space(3))) u32x4*)(size_t)(slot + lane * 16) = stage[0];
      *(__attribute__((address_space(3))) u32x4*)(size_t)(slot + 1024 + lane * 16) = stage[1];
      stage[0] = *(const u32x4*)mine;
      stage[1] = *(const u32x4*)(mine + 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for(int kk = 0; kk < 2; kk++)
#pragma unroll
      for(int pt = 0; pt < 3; pt++) acc[pt] = TR::mfma(wf, af[pt], acc[pt]);
    __builtin_amdgcn_sched_barrier(0);
    if(VARIANT == 1) waitVm<2>();
  }
  waitVm<0>();
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float t = (float)(stage[0][0] + stage[1][3]);
#pragma unroll
  for(int i = 0; i < 3; i++)
#pragma unroll
    for(int r = 0; r < 16; r++) t += acc[i][r];
  if(t == 12345.678f) sink[lane] = t;
  if(blockIdx.x == 0 && threadIdx.x == 0) {
    clocks[0] = c1 - c0;
    clocks[1] = w1 - w0;
  }
}

// returns avg ms per launch; *tflops = achieved rate; *coreMhz = shader clock during the kernel (clock64 vs the 100 MHz wall clock)
double benchMfma(int wavesPerWg, int wgs, int mode, int steps, int iters, double* tflops, double* coreMhz) {
  DevBuf clk(16), sink(1024);
  hipStream_t st = nullptr;
  const int barEvery = (mode >> 4) > 0 ? (mode >> 4) : 1;
  DevBuf icSrc(64 * 8192 + 4096);
  auto launch = [&]() {
    dim3 g(wgs), b(wavesPerWg * 64);
    if(mode >= 256) {  // issue-cost microbenchmark: 4 waves per work-group; *tflops then returns shader cycles per step
      const char* srcp = (const char*)icSrc.get();
      if(mode == 256) hipLaunchKernelGGL(issueCostKernel<0>, g, dim3(256), 65536, st, steps, srcp, clk.as<unsigned long long>(), sink.as<float>());
      else if(mode == 257) hipLaunchKernelGGL(issueCostKernel<1>, g, dim3(256), 65536, st, steps, srcp, clk.as<unsigned long long>(), sink.as<float>());
      else hipLaunchKernelGGL(issueCostKernel<2>, g, dim3(256), 65536, st, steps, srcp, clk.as<unsigned long long>(), sink.as<float>());
      hipCheck(hipGetLastError(), "issue-cost bench launch");
      return;
    }
#define KMX_PEAK(M_) \
  case M_: hipLaunchKernelGGL(mfmaPeakKernel<M_>, g, b, 65536, st, steps, barEvery, clk.as<unsigned long long>(), sink.as<float>()); break;
    switch(mode & 7) {
      KMX_PEAK(0) KMX_PEAK(1) KMX_PEAK(2) KMX_PEAK(3) KMX_PEAK(4) KMX_PEAK(5) KMX_PEAK(6) KMX_PEAK(7)
    }
#undef KMX_PEAK
    hipCheck(hipGetLastError(), "mfma bench launch");
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
  unsigned long long h[2] = {0, 0};
  hipCheck(hipMemcpy(h, clk.get(), 16, hipMemcpyDeviceToHost), "copy clocks");
  const double avg = (double)ms / iters;
  if(tflops) *tflops = 18.0 * 32768.0 * steps * wavesPerWg * (double)wgs / (avg * 1e-3) / 1e12;
  if(coreMhz) *coreMhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
  if(mode >= 256 && tflops) *tflops = (double)h[0] / steps;  // shader cycles per step of wave 0 of work-group 0, last launch
  return avg;
}


// ---- what the matrix cores SUSTAIN, by operand data (kmx_bench_mfma_sustained; tools/mfma_power_probe.py; round 6) ----
// mfmaPeakKernel above multiplies a smooth ramp of small positive numbers and holds 2.38 GHz on the whole chip; the convolution's operands
// look like noise, and the chip clocks down under them whatever else the kernel does. This one runs the same two loops - the bare chain of
// 18 MFMAs on a 3 x 3 tile of accumulators, and the convolution's step shape (+ 12 ds_read_b128 + one s_barrier per step) - for SECONDS (the
// power controller needs tens of milliseconds to settle) on operands of a chosen kind: 0 zeros, 1 the smooth ramp, 2 uniform noise in
// [-1, 1), 3 the distributions of the bench's own operands (normal weights of a random-init 192-channel 3x3 layer x mish of a unit normal at 1/8). 8 waves per work-group, two per SIMD.
namespace {
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float operandValue(int kind, unsigned i, bool weight) {
  if(kind == 0) return 0.0f;
  if(kind == 1) return 0.001f * (float)(i & 255);
  const float u = (float)(int)(hash32(i) >> 8) * (1.0f / 8388608.0f);  // [0, 2)
  if(kind == 2) return u - 1.0f;
  // kind 3: what the bench's convolutions multiply - weights N(0, 2 / (9 x 192)) (modelgen.py's random init of a 192 -> 192 3x3 layer), the
  // image mish of a unit normal at 1/8 (the fp16 range transform): the values' DISTRIBUTION, not only their being non-constant
  const float g = sqrtf(-2.0f * __logf(0.5f * u + 1e-7f)) * __cosf(6.2831853f * ((float)(hash32(i ^ 0x9e3779b9u) >> 8) * (1.0f / 16777216.0f)));
  if(weight) return 0.034f * g;
  return 0.125f * g * tanhf(log1pf(__expf(g)));
}
// ORDER: in which order a k half's nine MFMAs go out - 0: weight fragment outer, image fragment inner (the convolution's order: the weight
// operand stays for three MFMAs, at every fourth both operands change); 1: the same as a snake (exactly one operand changes between any two
// consecutive MFMAs of a k half); 2: image fragment outer. Same accumulators, same K order per accumulator: the sums do not depend on it.
template <class TR, bool LDS, int ORDER>
__global__ __launch_bounds__(512) void mfmaSustainedKernel(int steps, int kind, unsigned long long* clocks, float* sink) {
  extern __shared__ __attribute__((aligned(256))) char smemSus[];
  typedef typename TR::V8 V8;
  typedef typename TR::T T;
  const int lane = threadIdx.x & 63;
  f32x16 acc[9];
#pragma unroll
  for(int i = 0; i < 9; i++)
#pragma unroll
    for(int r = 0; r < 16; r++) acc[i][r] = 0.0f;
  V8 wf[2][3], af[2][3];
#pragma unroll
  for(int kk = 0; kk < 2; kk++)
#pragma unroll
    for(int j = 0; j < 3; j++)
#pragma unroll
      for(int i = 0; i < 8; i++) {
        wf[kk][j][i] = TR::fromFloat(operandValue(kind, (unsigned)(threadIdx.x * 64 + kk * 24 + j * 8 + i), true));
        af[kk][j][i] = TR::fromFloat(operandValue(kind, (unsigned)(threadIdx.x * 64 + 4096 * 64 + kk * 24 + j * 8 + i), false));
      }
  if(LDS) {
    for(int i = threadIdx.x; i < 32768; i += blockDim.x)  // (a step reads weight fragments from the first 6 KB of its 8 KB quarter, image fragments behind them)
      ((T*)smemSus)[i] = TR::fromFloat(operandValue(kind, (unsigned)i * 2654435761u + blockIdx.x, (2 * i) % 8192 < 6144));
    __syncthreads();
  }
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for(int s = 0; s < steps; s++) {
    if(LDS) {
      const char* base = smemSus + ((s & 3) * 8192) + (lane & 31) * 64 + ((((lane >> 5)) ^ ((lane >> 2) & 3)) << 4);
#pragma unroll
      for(int kk = 0; kk < 2; kk++)
#pragma unroll
        for(int j = 0; j < 3; j++) {
          wf[kk][j] = *(const V8*)(base + j * 2048 + kk * 32);
          af[kk][j] = *(const V8*)(base + 6144 + j * 2048 + (kk * 32) % 2048);
        }
    }
#pragma unroll
    for(int kk = 0; kk < 2; kk++)
#pragma unroll
      for(int i = 0; i < 3; i++)
#pragma unroll
        for(int j = 0; j < 3; j++) {
          const int ct = ORDER == 2 ? j : i;
          const int pt = ORDER == 2 ? i : (ORDER == 1 && (i & 1)) ? 2 - j : j;
          acc[ct * 3 + pt] = TR::mfma(wf[kk][ct], af[kk][pt], acc[ct * 3 + pt]);
          __builtin_amdgcn_sched_barrier(0);  // the order written is the order issued
        }
    if(LDS) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float t = 0.0f;
#pragma unroll
  for(int i = 0; i < 9; i++)
#pragma unroll
    for(int r = 0; r < 16; r++) t += acc[i][r];
  if(t == 12345.678f) sink[lane] = t;
  if(blockIdx.x == 0 && threadIdx.x == 0) {
    clocks[0] = c1 - c0;
    clocks[1] = w1 - w0;
  }
}
}  // namespace
// shape: bit 0 the step shape (LDS reads + barrier) instead of the bare chain, bits 1-2 the ORDER above. Returns the seconds it ran; *tflops over the whole run, *coreMhz inside the last launch
double benchMfmaSustained(int wgs, int shape, int kind, int dtype, double seconds, double* tflops, double* coreMhz) {
  DevBuf clk(16), sink(1024);
  const int steps = 540;
  void (*kern)(int, int, unsigned long long*, float*) = nullptr;
#define KMX_SUS(TR_, L_, O_) if(dtype == TR_::DT && (shape & 1) == (L_ ? 1 : 0) && (shape >> 1) == O_) kern = mfmaSustainedKernel<TR_, L_, O_>;
  KMX_SUS(TraitsF16, false, 0) KMX_SUS(TraitsF16, true, 0) KMX_SUS(TraitsBF16, false, 0) KMX_SUS(TraitsBF16, true, 0)
  KMX_SUS(TraitsF16, false, 1) KMX_SUS(TraitsF16, true, 1) KMX_SUS(TraitsF16, false, 2) KMX_SUS(TraitsF16, true, 2)
#undef KMX_SUS
  if(kern == nullptr) throw Error(KMX_ERR_INVALID_ARG, "kmx_bench_mfma_sustained: no such loop (the issue orders 1 and 2 exist for fp16 only)");
  hipCheck(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536), "lds attribute");
  hipStream_t st;
  hipCheck(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "stream");
  auto launch = [&](int n) {
    for(int i = 0; i < n; i++) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 65536, st, steps, kind, clk.as<unsigned long long>(), sink.as<float>());
    hipCheck(hipGetLastError(), "sustained mfma launch");
  };
  hipEvent_t e0, e1;
  hipCheck(hipEventCreate(&e0), "event");
  hipCheck(hipEventCreate(&e1), "event");
  launch(10);
  hipCheck(hipEventRecord(e0, st), "record");
  launch(40);
  hipCheck(hipEventRecord(e1, st), "record");
  hipCheck(hipStreamSynchronize(st), "sync");
  float ms = 0;
  hipCheck(hipEventElapsedTime(&ms, e0, e1), "elapsed");
  int iters = (int)(seconds * 1e3 / ((double)ms / 40.0));
  if(iters < 1) iters = 1;
  if(iters > 200000) iters = 200000;
  hipCheck(hipEventRecord(e0, st), "record");
  launch(iters);
  hipCheck(hipEventRecord(e1, st), "record");
  hipCheck(hipStreamSynchronize(st), "sync");
  hipCheck(hipEventElapsedTime(&ms, e0, e1), "elapsed");
  unsigned long long h[2] = {0, 0};
  hipCheck(hipMemcpy(h, clk.get(), 16, hipMemcpyDeviceToHost), "copy clocks");
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(st);
  if(tflops) *tflops = 18.0 * 32768.0 * steps * 8.0 * (double)wgs * iters / ((double)ms * 1e-3) / 1e12;
  if(coreMhz) *coreMhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
  return (double)ms * 1e-3;
}

// ---- what a dependent launch costs before it does anything (kmx_bench_launch_floor; tools/launch_floor.py) ----
// A small pass is ~122 dependent launches of 11-17 us; the loops in them are a third of that. This measures the floor under a launch of
// the small-batch shapes' geometry: `launches` dependent launches of a kernel of 512 threads and `ldsBytes` of LDS on `wgs` work-groups,
// captured in ONE hipGraph (as the engine replays its schedule) and replayed. mode 0: the kernel ends at once; 1: every lane loads 16
// bytes that the launch before stored (one round trip through the memory the launches hand their tensors over in) and stores them again;
// 2: as 1, and the loaded value is first written to and read back from LDS behind a barrier (the work-group's LDS allocation is touched).
template <int MODE>
__global__ __launch_bounds__(512) void launchFloorKernel(const u32x4* in, u32x4* out) {
  extern __shared__ __attribute__((aligned(16))) char smemFloor[];
  if(MODE == 0) return;
  const size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
  u32x4 v = in[i];
  if(MODE == 2) {
    u32x4* l = (u32x4*)smemFloor;
    l[threadIdx.x] = v;
    __syncthreads();
    v = l[threadIdx.x ^ 1];
  }
  out[i] = v;
}
double benchLaunchFloor(int wgs, int ldsBytes, int mode, int launches, int iters) {
  DevBuf a((size_t)wgs * 512 * 16), b((size_t)wgs * 512 * 16);
  hipStream_t st;
  hipCheck(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "stream");
  auto kern = mode == 0 ? launchFloorKernel<0> : mode == 1 ? launchFloorKernel<1> : launchFloorKernel<2>;
  hipCheck(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes), "lds attribute");
  auto chain = [&] {
    for(int i = 0; i < launches; i++) {
      const u32x4* src = (const u32x4*)((i & 1) ? b.get() : a.get());
      u32x4* dst = (u32x4*)((i & 1) ? a.get() : b.get());
      hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), ldsBytes, st, src, dst);
    }
  };
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipCheck(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal), "capture");
  chain();
  hipCheck(hipStreamEndCapture(st, &graph), "end capture");
  hipCheck(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0), "instantiate");
  for(int i = 0; i < 3; i++) hipCheck(hipGraphLaunch(exec, st), "graph launch");
  hipCheck(hipStreamSynchronize(st), "sync");
  hipEvent_t e0, e1;
  hipCheck(hipEventCreate(&e0), "event");
  hipCheck(hipEventCreate(&e1), "event");
  hipCheck(hipEventRecord(e0, st), "record");
  for(int i = 0; i < iters; i++) hipCheck(hipGraphLaunch(exec, st), "graph launch");
  hipCheck(hipEventRecord(e1, st), "record");
  hipCheck(hipStreamSynchronize(st), "sync");
  float ms = 0;
  hipCheck(hipEventElapsedTime(&ms, e0, e1), "elapsed");
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipGraphExecDestroy(exec);
  (void)hipGraphDestroy(graph);
  (void)hipStreamDestroy(st);
  return (double)ms * 1e3 / ((double)iters * launches);  // us per launch
}

// ---- fault triage: the LDS squatter (kernels.h launchLdsSquatter; engine.cpp KMX_DEBUG_SQUAT) ----
namespace {
__global__ __launch_bounds__(64) void ldsSquatterKernel(int words, unsigned long long ticks, unsigned* corrupt) {
  extern __shared__ unsigned squat[];
  const unsigned tag = 0x5a5a0000u ^ blockIdx.x;
  for(int i = threadIdx.x; i < words; i += 64) squat[i] = tag + (unsigned)i;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();  // 100 MHz
  while(wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  unsigned bad = 0;
  for(int i = threadIdx.x; i < words; i += 64) bad += squat[i] != tag + (unsigned)i;
  if(bad != 0 && corrupt != nullptr) atomicAdd(corrupt, bad);
}
}  // namespace
hipError_t launchLdsSquatter(int blocks, int ldsBytes, int usec, unsigned* corrupt, hipStream_t stream) {
  if(blocks <= 0 || ldsBytes < 4 || ldsBytes > 64 * 1024 || usec < 0 || usec > 100000) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ldsSquatterKernel, dim3(blocks), dim3(64), (size_t)ldsBytes, stream, ldsBytes / 4, (unsigned long long)usec * 100ull, corrupt);
  return hipGetLastError();
}
}  // namespace kmx

// conv_chain_kernel.h — a CHAIN of 3x3 convolutions of one board in ONE launch, the activated image handed from one convolution to
// the next in LDS (round 4). Same arithmetic as consecutive launches of conv_kernel.h, bit for bit.
//
// Replaces, per launch, 2 or 4 consecutive ConvLayer::apply + BatchNormLayer::apply (+ residual) of the reference inside a
// nested-bottleneck block (eigenbackend.cpp:1103-1146 ResidualBlock::apply, called from :1266-1315): the inner residual blocks
//     A1: t = act(bn(conv(x)))     A2: r += conv(t); x' = act(bn'(r))      [B1, B2 the same on x', r]
// for the layer class that is 90 % of b18c384nbt's arithmetic: 192 -> 192 channels, 3x3.
//
// Why: at batch >= ~200 a work-group of conv_kernel.h owns a whole board and all 192 output
// channels, one work-group per CU and launch - so a layer's time is prologue + loop + epilogue, the matrix cores idle in the first and
// the last, and the next layer of the SAME board reads back through HBM what this one has just written. Here the work-group keeps the
// board: the epilogue of convolution i writes the activated 16-bit image straight into the LDS image of convolution i + 1 (the layout
// conv_kernel.h DMA-copies from HBM: [halo row][32 channels], slots XOR-swizzled), the weights of convolution i + 1 stream in while
// that epilogue runs, and only what a later launch needs goes to HBM: the raw residual stream and the last activated image.
//   LDS cannot hold a whole 192-channel image (6 chunks x 28 KB) beside the slab ring: chunks 0-2 (the channels of waves wn = 0) are
//   handed over in LDS, chunks 3-5 (waves wn = 1) go through a scratch tensor in HBM/L2 and are fetched by LDS-DMA into the slots that
//   chunks 0-2 leave behind while the loop works on chunks 1-3 - half the bytes of a hand-over through memory, none of its latency.
//
// Shape: the 8-wave x 192-channel shape of conv_kernel.h (cfg 23: wave (wm, wn) owns 96 cells x 96 channels, 3 x 3 MFMA tiles of
// 32 x 32; ring of 4 weight slabs, requests three steps ahead; waves 0-3 issue all LDS-DMA, spread over the step's MFMAs). Same
// MFMA instruction, operand roles, K order (chunk, tap, k half) and rounding points as conv_kernel.h: a chain's outputs are
// BIT-IDENTICAL to the separate launches (tests/test_engine_emulated.py on the CPU emulation, tests/test_gpu_layers.py on the MI355X).
#ifndef KMX_CONV_CHAIN_KERNEL_H_
#define KMX_CONV_CHAIN_KERNEL_H_

#include <atomic>

#include "conv_kernel.h"

namespace kmx {
namespace chaink {
using convk::dma16;
using convk::dma4;
using convk::waitVm;
using convk::ROWB;

constexpr int CH = 192;                  // channels in and out of every convolution of a chain
constexpr int NCHUNK = CH / KCHUNK;      // 6
constexpr int NT = 9, HALO = 1, MT = 3, WN = 3, WNW = 2;
constexpr int NWAVES = 8, NTHREADS = 512;
constexpr int D = 3, NSW = D + 1;        // requests three steps ahead, ring of four slabs
constexpr int NLOAD = 4;                 // waves 0-3 issue all LDS-DMA
constexpr int HPMAX = 21 * 21;
constexpr int NPA = (HPMAX * 4 + NLOAD * 64 - 1) / (NLOAD * 64);  // 7 requests per loader wave and board image
constexpr int ACT_BYTES = NPA * NLOAD * 64 * 16;                   // 28672: every request lands inside its slot
constexpr int NSLOT = 3;                 // image slots: chunk c lives in slot c % 3
constexpr int WPIECES = CH * 4;          // 16-byte pieces per slab
constexpr int NPW = WPIECES / (NLOAD * 64);                        // 3 requests per loader wave and slab
constexpr int W_BYTES = WPIECES * 16;    // 12288
constexpr int SLACK_BYTES = 1024;
constexpr int MASK_BYTES = NTHREADS * 4;  // one 4-byte request per lane: 384 cells + padding
constexpr int NTP = 256;                  // scale | bias of one convolution, each padded to 256 floats: one 4-byte request per lane
constexpr int PARAM_BYTES = 2 * NTP * 4;
constexpr int SLOT_OFFSET = 0;
constexpr int RING_OFFSET = NSLOT * ACT_BYTES;
constexpr int SLACK_OFFSET = RING_OFFSET + NSW * W_BYTES;
constexpr int MASK_OFFSET = SLACK_OFFSET + SLACK_BYTES;
constexpr int PARAM_OFFSET = MASK_OFFSET + MASK_BYTES;
constexpr int ldsBytes(int nConv) { return PARAM_OFFSET + nConv * PARAM_BYTES; }
static_assert(ldsBytes(MAX_CHAIN) <= 160 * 1024, "LDS budget exceeded");
static_assert(NPA + 1 <= NT && NPA <= NT + 1 - D, "the next image's requests precede the slab that the last tap waits for");
static_assert(2 * (WN * MT - WN - MT) >= 1 + NPW, "a step's requests ride behind MFMAs that carry no fragment read");

// TIMING (conv_bench.hip only): s_memtime stamps at the phase boundaries of every convolution, written per wave of the middle board to a.dbg
template <class TR, int KIND, bool TIMING = false>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void convChainKernel(const ConvChainArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  typedef typename TR::V4 V4;
  extern __shared__ __attribute__((aligned(256))) char smemChain[];
  const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smemChain;
  const unsigned bufW = ldsBase + RING_OFFSET;
  const unsigned slack = ldsBase + SLACK_OFFSET;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const int n = blockIdx.x;
  const int X = a.X, Y = a.Y, S = X * Y;
  const int W2 = X + 2 * HALO, HP = W2 * (Y + 2 * HALO);
  // GEMM column -> board cell and lane -> position inside a tile: as conv_kernel.h (16-lane groups of a ds_read_b128 read 16
  // consecutive image rows)
  const int mainCols = X >= 16 ? 16 * Y : 0;
  const int restW = X - 16;
  auto cellOf = [&](int m) -> int {
    if(m < mainCols) return (m >> 4) * X + (m & 15);
    if(X < 16) return m;
    const int k = m - mainCols;
    const int yy = k / restW;
    return yy * X + 16 + (k - yy * restW);
  };
  auto posOf = [](int l) -> int { return l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l; };
  const char* const zero = (const char*)a.zeroPage;
  const bool loader = wave < NLOAD;  // wave-uniform

  const unsigned khalf = lane >> 5;
  const unsigned c40 = khalf << 4;
  const bool waveActive = wm * (32 * MT) < S;
  // Per-lane addressing state (LDS-DMA sources, fragment rows) is RECOMPUTED at the top of every convolution of the chain from an
  // opaque copy of the lane index, and the epilogue derives its cell from scratch too: kept across the epilogue, these ~20 registers
  // sit on top of the 144 accumulators and the epilogue's own working set and the kernel spills (256 VGPRs + 152 bytes of scratch
  // per lane in the first version; -Rpass-analysis=kernel-resource-usage). A few hundred vector instructions per convolution
  // against ~100 k cycles of loop.
  unsigned srcOff[NPA];
  unsigned wOff0;
  unsigned wLane[2];
  unsigned aRow4[MT];
  auto laneSetup = [&](int ln) {
    // LDS-DMA sources of a board image, chunk 0: byte offset from the board's tensor, or bit 31 = zero page
#pragma unroll
    for(int j = 0; j < NPA; j++) {
      const int p = (j * NLOAD + (wave & 3)) * 64 + ln;
      const int hp = p >> 2;
      const int slot = (p & 3) ^ ((hp >> 2) & 3);
      unsigned off = 0x80000000u;
      if(hp < HP) {
        const int hy = hp / W2, hx = hp - hy * W2;
        const int y = hy - HALO, x = hx - HALO;
        if(y >= 0 && y < Y && x >= 0 && x < X) off = (unsigned)((y * X + x) * CH + slot * 8) * (unsigned)sizeof(T);
      }
      srcOff[j] = off;
    }
    wOff0 = (unsigned)((wave & 3) * 64 + ln) * 16u;  // request j of a slab: + j * NLOAD * 64 * 16
    const unsigned wXor = ((unsigned)ln >> 2) & 3;
#pragma unroll
    for(int kk = 0; kk < 2; kk++) wLane[kk] = bufW + (wn * (32 * WN) + (ln & 31)) * ROWB + (((kk * 2 + ((unsigned)ln >> 5)) ^ wXor) << 4);
#pragma unroll
    for(int pt = 0; pt < MT; pt++) {
      int j = wm * (32 * MT) + pt * 32 + posOf(ln & 31);
      j = cellOf(j < S ? j : S - 1);
      const int y = j / X, x = j - y * X;
      aRow4[pt] = (unsigned)((y + HALO) * W2 + (x + HALO)) << 2;
    }
  };

  auto ldsV8 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) V8*)addr; };
  auto ldsF4 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) f32x4*)addr; };
  auto ldsF1 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) float*)addr; };

  // ---- prologue: mask, every convolution's BN parameters, the first image and the first slabs ----
  {
    const int cellIdx = wave * 64 + lane;
    dma4(cellIdx < S ? (const void*)(a.mask + (size_t)n * S + cellIdx) : (const void*)zero, ldsBase + MASK_OFFSET + wave * 256);
  }
  for(int ci = 0; ci < a.nConv; ci++) {
    const int idx = wave * 64 + lane;  // 0..511: scale[0..255] | bias[0..255]
    const int arr = idx >> 8, c = idx & 255;
    const float* psrc = c < CH ? (arr == 0 ? a.conv[ci].scale : a.conv[ci].bias) + c : (const float*)zero;
    dma4(psrc, ldsBase + PARAM_OFFSET + ci * PARAM_BYTES + wave * 256);
  }

  // image source of the convolution being worked on (wave-uniform): conv 0 reads the chain's input, conv i > 0 the upper half of
  // the channels of conv i - 1's activated image from its scratch tensor
  const char* imgBoard = (const char*)a.in + (size_t)n * S * CH * sizeof(T);
  const char* wBase = (const char*)a.conv[0].w;
  // request j (0..NPA-1) of the image of `chunk` into its slot; dummy (zero page -> slack) when the chunk needs no request
  auto issueA = [&](int chunk, int j, bool real) {
    const unsigned off = srcOff[j];
    const char* src = (off & 0x80000000u) ? zero : imgBoard + off + (unsigned)chunk * (unsigned)(KCHUNK * sizeof(T));
    const unsigned dst = ldsBase + SLOT_OFFSET + (unsigned)(chunk % NSLOT) * ACT_BYTES + (unsigned)((j * NLOAD + (wave & 3)) * 64) * 16u;
    dma16(real ? src : zero, real ? dst : slack);
  };
  // request j (0..NPW-1) of the slab of `step`
  auto issueW1 = [&](int step, int j) {
    const bool live = step < NCHUNK * NT;
    const char* slab = wBase + (size_t)(live ? step : 0) * W_BYTES;
    dma16(slab + wOff0 + (unsigned)(j * NLOAD * 64 * 16), live ? bufW + (unsigned)(step % NSW) * W_BYTES + (unsigned)((j * NLOAD + (wave & 3)) * 64) * 16u : slack);
  };
  laneSetup(lane);
  if(loader) {
#pragma unroll
    for(int j = 0; j < NPA; j++) issueA(0, j, true);
#pragma unroll
    for(int s = 0; s < D; s++)
#pragma unroll
      for(int j = 0; j < NPW; j++) issueW1(s, j);
  }

  V8 wf[2][WN];
  V8 af[2][MT];
  unsigned aAddr[MT];
  f32x16 acc[WN][MT];
  const unsigned maskAddr = ldsBase + MASK_OFFSET;
  T* const trash = (T*)((char*)const_cast<void*>(a.zeroPage) + ZERO_PAGE_BYTES) + lane * 8;

  unsigned long long tPrev = TIMING ? __builtin_readcyclecounter() : 0;
  auto stamp = [&](int ci, int which) {  // phases of convolution ci: 0 prologue / restart, 1 loop, 2 wait + barrier after the loop, 3 epilogue, 4 closing wait + barrier
    if(!TIMING) return;
    const unsigned long long now = __builtin_readcyclecounter();
    if(a.dbg != nullptr && lane == 0 && (int)blockIdx.x == a.N / 2) a.dbg[(wave * MAX_CHAIN + ci) * 8 + which] = now - tPrev;
    tPrev = now;
  };
  for(int ci = 0; ci < a.nConv; ci++) {
    const bool first = ci == 0, last = ci + 1 == a.nConv;
    // which chunk's image is requested while the loop works on `chunk`: the next one (conv 0: everything comes from HBM) or the one
    // after it (conv i > 0: chunks 0-2 are in LDS already, chunks 3-5 follow into the slots they leave behind)
    const int dist = first ? 1 : 2;
    const int firstDma = first ? 1 : NSLOT;
    {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      laneSetup(ln);
    }

#pragma unroll
    for(int ct = 0; ct < WN; ct++)
#pragma unroll
      for(int pt = 0; pt < MT; pt++)
#pragma unroll
        for(int r = 0; r < 16; r++) acc[ct][pt][r] = 0.0f;

    // slab 0 of this convolution (and, conv 0, image 0) has landed; published by the barrier. For conv i > 0 the requests were issued
    // before the previous epilogue and that epilogue has waited for everything (vmcnt(0)) before its closing barrier.
    if(first) {
      if(loader) waitVm<(D - 1) * NPW>();
      else waitVm<0>();  // mask and parameters
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // fragments of step 0, k half 0
    {
#pragma unroll
      for(int ct = 0; ct < WN; ct++) wf[0][ct] = ldsV8(wLane[0] + ct * 32 * ROWB);
      unsigned sTap = (unsigned)(((0 - HALO) * W2 + (0 - HALO)) * 4) + ((ldsBase + SLOT_OFFSET) >> 4);
      asm volatile("" : "+s"(sTap));
#pragma unroll
      for(int pt = 0; pt < MT; pt++) {
        const unsigned q4 = aRow4[pt] + sTap;
        aAddr[pt] = (q4 << 4) | ((q4 ^ c40) & 0x30u);
        af[0][pt] = ldsV8(aAddr[pt]);
      }
    }

    stamp(ci, 0);
    int step = 0;
    for(int chunk = 0; chunk < NCHUNK; chunk++) {
      const unsigned curA = (unsigned)(chunk % NSLOT) * ACT_BYTES;
      const unsigned nextA = (unsigned)((chunk + 1) % NSLOT) * ACT_BYTES;
      const int dmaChunk = chunk + dist;
      const bool dmaReal = dmaChunk >= firstDma && dmaChunk < NCHUNK;
#pragma unroll
      for(int t = 0; t < NT; t++, step++) {
        // top of step s: slab s + 1 (requested in step s - 2, after that step's image request) has landed; younger: the requests of
        // step s - 1 (3 slab requests and, at taps 0..6, one image request)
        if(loader) {
          if((t + NT - 1) % NT < NPA) waitVm<NPW + 1>();
          else waitVm<NPW>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // first k half: the MFMAs of fragment set 0; behind them, one by one, the reads of set 1 (slab of this step, this tap) and
        // then this step's first requests
        const unsigned wb1 = wLane[1] + (unsigned)(step % NSW) * W_BYTES;
#pragma unroll
        for(int idx = 0; idx < WN * MT; idx++) {
          acc[idx / MT][idx % MT] = TR::mfma(wf[0][idx / MT], af[0][idx % MT], acc[idx / MT][idx % MT]);
          __builtin_amdgcn_sched_barrier(0);
          if(idx < WN) wf[1][idx] = ldsV8(wb1 + idx * 32 * ROWB);
          else if(idx < WN + MT) af[1][idx - WN] = ldsV8(aAddr[idx - WN] ^ 0x20u);
          else if(loader) {
            const int k = idx - (WN + MT);  // 0, 1, 2
            if(k == 0) {
              if(t < NPA) issueA(dmaChunk, t, dmaReal);
            }
            else issueW1(step + D, k - 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // second k half: fragment set 1; behind its MFMAs the reads of set 0 of the NEXT step (slab s + 1, next tap) and the rest of
        // the requests
        const unsigned wb0 = wLane[0] + (unsigned)((step + 1) % NSW) * W_BYTES;
        {
          const int tn = t + 1 < NT ? t + 1 : 0;
          unsigned sTap = (unsigned)(((tn / 3 - HALO) * W2 + (tn % 3 - HALO)) * 4) + ((ldsBase + SLOT_OFFSET + (t + 1 < NT ? curA : nextA)) >> 4);
          asm volatile("" : "+s"(sTap));
#pragma unroll
          for(int pt = 0; pt < MT; pt++) {
            const unsigned q4 = aRow4[pt] + sTap;
            aAddr[pt] = (q4 << 4) | ((q4 ^ c40) & 0x30u);
          }
        }
#pragma unroll
        for(int idx = 0; idx < WN * MT; idx++) {
          acc[idx / MT][idx % MT] = TR::mfma(wf[1][idx / MT], af[1][idx % MT], acc[idx / MT][idx % MT]);
          __builtin_amdgcn_sched_barrier(0);
          if(idx < WN) wf[0][idx] = ldsV8(wb0 + idx * 32 * ROWB);
          else if(idx < WN + MT) af[0][idx - WN] = ldsV8(aAddr[idx - WN]);
          else if(loader) {
            const int k = 3 + idx - (WN + MT);  // 3, 4, 5
            if(k - 1 < NPW) issueW1(step + D, k - 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // (the fragments read for "step 54" are never used: slab 54 % 4 and slot 0 hold old data, which is all they are)

    stamp(ci, 1);
    // ---- between two convolutions: every wave is out of the loop before anything overwrites a slot or a ring slab ----
    waitVm<0>();  // the trailing dummy requests
    if(!last) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // the next convolution's first slabs arrive while this epilogue runs
      wBase = (const char*)a.conv[ci + 1].w;
      imgBoard = (const char*)a.conv[ci].actOut + (size_t)n * S * CH * sizeof(T);
      if(loader) {
#pragma unroll
        for(int s = 0; s < D; s++)
#pragma unroll
          for(int j = 0; j < NPW; j++) issueW1(s, j);
      }
    }

    stamp(ci, 2);
    // ---- epilogue (conv_kernel.h's, plus the hand-over): straight from the accumulator layout ----
    const ChainConv& cv = a.conv[ci];
    const bool hasResid = cv.resid != nullptr;  // uniform
    const bool hasRaw = cv.rawOut != nullptr;   // uniform
    const unsigned scAddr = ldsBase + PARAM_OFFSET + ci * PARAM_BYTES, biAddr = scAddr + NTP * 4;
    T* const rawBoard = (T*)cv.rawOut + (size_t)n * S * CH;
    T* const actBoard = (T*)cv.actOut + (size_t)n * S * CH;
    const T* const residBoard = (const T*)cv.resid + (size_t)n * S * CH;
    auto epilogue = [&](auto residTag) {
      constexpr bool RESID = decltype(residTag)::value != 0;
      // the lane's column and cell, derived here (see laneSetup): nothing of the loop's addressing state is alive in this phase
      int lnE = lane;
      asm volatile("" : "+v"(lnE));
      const int myPos = posOf(lnE & 31);
      auto cellOfTileE = [&](int pt) -> int {
        const int rc = wm * (32 * MT) + pt * 32 + myPos;
        return cellOf(rc < S ? rc : S - 1);
      };
      u32x4 rq[2][2];
      auto loadResid = [&](int pt, int ct, u32x4 (&dst)[2]) {
        const T* const rrow = residBoard + (size_t)cellOfTileE(pt) * CH;
#pragma unroll
        for(int j = 0; j < 2; j++) dst[j] = *(const u32x4*)(rrow + wn * (32 * WN) + ct * 32 + 16 * j + 8 * khalf);
      };
      if(RESID) loadResid(0, 0, rq[0]);
#pragma unroll
      for(int pt = 0; pt < MT; pt++) {
        const int cellBase = wm * (32 * MT) + pt * 32;
        if(cellBase >= S) break;  // wave-uniform
        const bool live = cellBase + myPos < S;  // the same for both lanes of a pair
        const int cell = cellOfTileE(pt);
        const unsigned onBits = ldsF1(maskAddr + cell * 4) == 1.0f ? 0xffffffffu : 0u;
        T* const rawRow = rawBoard + (size_t)cell * CH;
        T* const actRow = actBoard + (size_t)cell * CH;
        // this lane's row of the LDS image: byte q * 64, logical 16-byte slot c at physical slot c ^ ((q >> 2) & 3)
        const int cy = cell / X, cx = cell - cy * X;
        const unsigned q = (unsigned)((cy + HALO) * W2 + (cx + HALO));
        const unsigned rowAddr = ldsBase + SLOT_OFFSET + q * ROWB;
        const unsigned rowXor = (q >> 2) & 3;
#pragma unroll
        for(int ct = 0; ct < WN; ct++) {
          const int chTile = wn * (32 * WN) + ct * 32;
          unsigned pOff = (unsigned)(chTile + 4 * khalf) * 4u;
          asm volatile("" : "+v"(pOff));
          u32x2 rp[4], op[4];
          u32x2 resP[4];
          if(RESID) {
            constexpr int NTILES = MT * WN;
            const int k = pt * WN + ct;
            if(k + 1 < NTILES) loadResid((k + 1) / WN, (k + 1) % WN, rq[(k + 1) & 1]);
            unpair(rq[k & 1], resP);
          }
#pragma unroll
          for(int g = 0; g < 4; g++) {
            const f32x4 sc = ldsF4(scAddr + pOff + 32 * g);
            const f32x4 bi = ldsF4(biAddr + pOff + 32 * g);
            f32x4 v;
#pragma unroll
            for(int i = 0; i < 4; i++) v[i] = acc[ct][pt][4 * g + i];
            if(RESID) {
              const V4 rr = __builtin_bit_cast(V4, resP[g]);
#pragma unroll
              for(int i = 0; i < 4; i++) v[i] += TR::toFloat(rr[i]);
            }
            V4 r, o;
#pragma unroll
            for(int i = 0; i < 4; i++) r[i] = TR::fromFloat(v[i]);
#pragma unroll
            for(int i = 0; i < 4; i += 2) {
              f32x2 x;
              x[0] = v[i] * sc[i] + bi[i];
              x[1] = v[i + 1] * sc[i + 1] + bi[i + 1];
              const f32x2 y = actK2<KIND>(x);
              o[i] = TR::fromFloat(y[0]);
              o[i + 1] = TR::fromFloat(y[1]);
            }
            rp[g] = __builtin_bit_cast(u32x2, r);
            op[g] = __builtin_bit_cast(u32x2, o);
            op[g][0] &= onBits;
            op[g][1] &= onBits;
          }
          if(RESID || hasRaw) {  // (a convolution with a residual always stores the raw stream)
            u32x4 rawQ[2];
            pairUp(rp, rawQ);
#pragma unroll
            for(int j = 0; j < 2; j++) {
              T* const dst = (live && hasRaw) ? rawRow + chTile + 16 * j + 8 * khalf : trash;
              *(u32x4*)dst = rawQ[j];
            }
          }
          u32x4 actQ[2];
          pairUp(op, actQ);
          // The activated image. Last convolution: to HBM, all of it. Otherwise the hand-over: the chunk of this channel tile is
          // wn * 3 + ct - waves wn = 0 hold chunks 0-2 and write them into LDS slots 0-2, the image layout of the next loop; waves
          // wn = 1 hold chunks 3-5 and store them to the scratch tensor, from where the next loop fetches them by LDS-DMA.
          if(last || wn == 1) {
#pragma unroll
            for(int j = 0; j < 2; j++) {
              T* const dst = live ? actRow + chTile + 16 * j + 8 * khalf : trash;
              *(u32x4*)dst = actQ[j];
            }
          }
          else if(live) {
#pragma unroll
            for(int j = 0; j < 2; j++) {
              const unsigned slot16 = (unsigned)(2 * j) + khalf;
              *(__attribute__((address_space(3))) u32x4*)(rowAddr + (unsigned)ct * ACT_BYTES + ((slot16 ^ rowXor) << 4)) = actQ[j];
            }
          }
        }
      }
    };
    if(waveActive) {
      if(hasResid) epilogue(ActKindTag<1>());
      else epilogue(ActKindTag<0>());
    }
    stamp(ci, 3);
    if(!last) {
      // the hand-over is complete when every wave's LDS writes are done and every wave's stores to the scratch tensor have been
      // acknowledged (the other waves' LDS-DMA reads them through the same L1); the next convolution's first slabs have landed too
      waitVm<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    stamp(ci, 4);
  }
}

template <class TR, int KIND, bool TIMING = false>
hipError_t launchChainOne(const ConvChainArgs& a, hipStream_t stream) {
  auto kern = convChainKernel<TR, KIND, TIMING>;
  constexpr int MAX_DEVICES = 64;
  static std::atomic<bool> attrSet[MAX_DEVICES];
  int dev = 0;
  hipError_t de = hipGetDevice(&dev);
  if(de != hipSuccess) return de;
  if(dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
  if(!attrSet[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes(MAX_CHAIN));
    if(e != hipSuccess) return e;
    attrSet[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(a.N), dim3(NTHREADS), ldsBytes(a.nConv), stream, a);
  return hipGetLastError();
}

}  // namespace chaink
}  // namespace kmx
#endif

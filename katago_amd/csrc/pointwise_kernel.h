// pointwise_kernel.h — two chained 1x1 convolutions in one launch, for the seam between two nested-bottleneck blocks:
//
//     trunk  += W1 * mid_act                         (block i:   finalConv, 1x1, C1 -> C2, + residual; eigenbackend.cpp:1308-1314)
//     t_act   = act(bn_{i+1}(trunk)) * mask          (block i+1: preBN + activation; eigenbackend.cpp:739-762)
//     mid     = W2 * t_act                           (block i+1: regularConv, 1x1, C2 -> C3)
//     mid_act = act(bn_inner(mid)) * mask            (first inner block's preBN)
//
// As two launches of the convolution kernel this seam moves (C1 + 2 C2 + 2 C2) + (C2 + 2 C3) 16-bit values per board cell
// through HBM and is memory-bound (measured 2.4 TB/s, 31 % of the step at 10 % of its FLOPs). A 1x1 convolution has no
// spatial coupling, so here the batch is ONE flat matrix of N*S cells: a work-group owns TM consecutive cells and all
// channels, the C2-channel activated image t_act is produced into LDS and consumed from there as the second GEMM's
// operand — it never exists in HBM. Traffic per cell: C1 + C2 (residual) + C2 (trunk) + 2 C3 values.
//
// Work-group = 8 waves, TM = 128 cells.
//   GEMM 1: D1[C2][TM], waves 2 (cells) x 4 (channels): a wave owns 64 cells x 32*WN1 channels (2 x WN1 MFMA tiles).
//           K = C1 in chunks of 32; the whole X tile [TM][C1] is fetched into LDS up front (LDS-DMA, swizzled on the source
//           side exactly like the convolution kernel's board image), W1 slabs [C2][32] ride a ring of 3, fetched two steps
//           ahead. Accumulators start from the residual stream.
//   epilogue 1: raw trunk values -> HBM (16-byte pieces: the lane pair (c, c + 32) of a tile column regroups its 8-byte
//           accumulator runs with v_permlane32_swap, device_common.h pairUp), activated values -> LDS in the image layout of GEMM 2.
//   GEMM 2: D2[C3][TM], waves 4 (cells) x 2 (channels): 32 cells x 32*WN2 channels per wave; K = C2, W2 slabs on the ring.
//   epilogue 2: mid raw and activated -> HBM.
// Same MFMA (v_mfma_f32_32x32x16), same operand roles (weights = A, cells = B), same K order (chunk, k-half) and the
// same rounding points as two launches of conv_kernel.h, so the result is BIT-IDENTICAL to the unfused schedule
// (tests/test_gpu_pointwise.py); the engine uses whichever is faster for the batch at hand.
#ifndef KMX_POINTWISE_KERNEL_H_
#define KMX_POINTWISE_KERNEL_H_

#include <atomic>

#include "device_common.h"

namespace kmx {
namespace pwk {

constexpr int ROWB = WROW_HALFS * 2;  // 64-byte LDS rows: four 16-byte slots, slot s of row r stored at s ^ ((r>>2)&3)
// Two shapes (the last template parameter NW of the kernel):
//   8 waves, 128 cells, ring of 3 weight slabs: LDS ~133 KB -> one work-group per CU. GEMM 1 waves 2 (cells) x 4 (channels),
//     GEMM 2 waves 4 x 2.
//   4 waves,  64 cells, ring of 2 weight slabs: LDS < 80 KB -> TWO work-groups per CU, 8 waves in all as before. Built to let
//     one group stream its tile in or its results out while the other multiplies (per work-group the seam spends ~30 k
//     cycles on HBM traffic and ~28 k on matrix + vector work, nothing overlapped). EXPERIMENT (KMX_PW_WAVES=4): measured
//     2 % slower than the 8-wave shape (the phases of co-resident groups did not overlap). GEMM 1 waves 2 x 2 (a wave
//     owns 192 channels), GEMM 2 waves 2 x 2.
template <int K1, int WN1, int WN2, int TM, int NW>
struct Geom {
  static constexpr int NWAVES = NW;
  static constexpr int NTHREADS = NWAVES * 64;
  static constexpr int RING = NW == 8 ? 3 : 2;  // weight slabs in LDS; slab c+RING-1 is requested in step c
  static constexpr int WM1 = 2;                 // wave rows (cells) of GEMM 1
  static constexpr int WNG1 = NWAVES / WM1;     // wave columns (channel groups) of GEMM 1
  static constexpr int WM2 = NW == 8 ? 4 : 2;
  static constexpr int WNG2 = NWAVES / WM2;
  static constexpr int C1 = 32 * K1;            // input channels of GEMM 1
  static constexpr int C2 = WNG1 * 32 * WN1;    // trunk channels
  static constexpr int K2 = C2 / 32;
  static constexpr int C3 = WNG2 * 32 * WN2;    // output channels of GEMM 2
  static constexpr int MT1 = TM / (32 * WM1);   // cell tiles per wave, GEMM 1
  static constexpr int MT2 = TM / (32 * WM2);   // cell tiles per wave, GEMM 2
  static_assert(TM % (32 * WM1) == 0 && TM % (32 * WM2) == 0, "a work-group's cells split into wave rows of whole 32-cell tiles");
  static constexpr int CHUNK_BYTES = TM * ROWB;          // one 32-channel chunk of an image tile
  static constexpr int X_BYTES = K1 * CHUNK_BYTES;
  static constexpr int W1_SLAB = C2 * ROWB;
  static constexpr int W2_SLAB = C3 * ROWB;
  static constexpr int A2_BYTES = K2 * CHUNK_BYTES;
  // LDS map. GEMM 1: [X | W1 ring]. After its loop: [A2 (over X and the head of the W1 ring) | W2 ring (behind A2)].
  static constexpr int W1_OFF = X_BYTES;
  static constexpr int PH1_END = W1_OFF + RING * W1_SLAB;
  static constexpr int W2_OFF = A2_BYTES;
  static constexpr int PH2_END = W2_OFF + RING * W2_SLAB;
  static constexpr int PIPE_END = PH1_END > PH2_END ? PH1_END : PH2_END;
  static constexpr int PARAM_OFF = PIPE_END;                              // scale1, bias1 [C2], scale2, bias2 [C3] as float
  static constexpr int MASK_OFF = PARAM_OFF + (2 * C2 + 2 * C3) * 4;      // TM floats
  static constexpr int SLACK_OFF = MASK_OFF + TM * 4;                     // 1 KiB: destination of padding DMA (never read)
  static constexpr int LDS_BYTES = SLACK_OFF + 1024;
  // DMA instructions (1 KiB each: 64 lanes x 16 bytes) per wave
  static constexpr int NPX = (K1 * TM * 4 + NTHREADS - 1) / NTHREADS;   // the X tile
  static constexpr int NPW1 = (C2 * 4 + NTHREADS - 1) / NTHREADS;       // one W1 slab
  static constexpr int NPW2 = (C3 * 4 + NTHREADS - 1) / NTHREADS;       // one W2 slab
  static_assert(TM * 4 == NTHREADS, "one DMA round of the work-group fills exactly one chunk of the X tile");
};

template <int N>
__device__ __forceinline__ void waitVm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void waitVmSel(int n) {  // n becomes a constant once the caller's branches are resolved
  switch(n) {
    case 1: waitVm<1>(); break;
    case 2: waitVm<2>(); break;
    case 3: waitVm<3>(); break;
    case 4: waitVm<4>(); break;
    case 5: waitVm<5>(); break;
    case 6: waitVm<6>(); break;
    case 7: waitVm<7>(); break;
    case 8: waitVm<8>(); break;
    case 9: waitVm<9>(); break;
    case 10: waitVm<10>(); break;
    case 11: waitVm<11>(); break;
    case 12: waitVm<12>(); break;
    case 13: waitVm<13>(); break;
    case 14: waitVm<14>(); break;
    case 15: waitVm<15>(); break;
    case 16: waitVm<16>(); break;
    case 17: waitVm<17>(); break;
    case 18: waitVm<18>(); break;
    case 19: waitVm<19>(); break;
    case 20: waitVm<20>(); break;
    case 21: waitVm<21>(); break;
    case 22: waitVm<22>(); break;
    case 23: waitVm<23>(); break;
    case 24: waitVm<24>(); break;
    case 25: waitVm<25>(); break;
    case 26: waitVm<26>(); break;
    case 27: waitVm<27>(); break;
    case 28: waitVm<28>(); break;
    case 29: waitVm<29>(); break;
    case 30: waitVm<30>(); break;
    case 31: waitVm<31>(); break;
    case 32: waitVm<32>(); break;
    case 33: waitVm<33>(); break;
    case 34: waitVm<34>(); break;
    case 35: waitVm<35>(); break;
    case 36: waitVm<36>(); break;
    case 37: waitVm<37>(); break;
    case 38: waitVm<38>(); break;
    case 39: waitVm<39>(); break;
    case 40: waitVm<40>(); break;
    default: waitVm<0>(); break;  // stricter than needed, never wrong
  }
}
// gfx950 barriers are "back-off" barriers: the compiler does not drain the LDS queue in front of them, so ds_writes that
// other waves read after the barrier are waited for explicitly
__device__ __forceinline__ void waitLds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void dma16(const void* gsrc, char* ldsWaveBase) {
  __builtin_amdgcn_global_load_lds(
    (const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)ldsWaveBase, 16, 0, 0);
}
template <class TR, int K1, int WN1, int WN2, int TM, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void pointwisePairKernel(const PwPairArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  typedef typename TR::V4 V4;
  typedef Geom<K1, WN1, WN2, TM, NW> G;
  constexpr int NWAVES = G::NWAVES, NTHREADS = G::NTHREADS, RING = G::RING;
  constexpr int K2 = G::K2, MT1 = G::MT1, MT2 = G::MT2, NPX = G::NPX, NPW1 = G::NPW1, NPW2 = G::NPW2;

  extern __shared__ __attribute__((aligned(256))) char smemPw[];
  char* const smem = smemPw;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned khalf = lane >> 5;
  // position of lane l (0..31) inside a 32-column tile, as in conv_kernel.h: the 16-lane groups a ds_read_b128 is served in
  // ({0-3,12-15,20-27}, {4-11,16-19,28-31}) then read 16 consecutive image rows, which the XOR swizzle spreads over all banks
  const int l31 = lane & 31;
  const int myPos = l31 < 4 ? l31 : l31 < 12 ? l31 + 12 : l31 < 16 ? l31 - 8 : l31 < 20 ? l31 + 8 : l31 < 28 ? l31 - 12 : l31;
  const long long cell0 = (long long)blockIdx.x * TM;
  const char* const zero = (const char*)a.zeroPage;
  char* const mySlack = smem + G::SLACK_OFF;
  float* const sc1S = (float*)(smem + G::PARAM_OFF);
  float* const bi1S = sc1S + G::C2;
  float* const sc2S = bi1S + G::C2;
  float* const bi2S = sc2S + G::C3;
  float* const maskS = (float*)(smem + G::MASK_OFF);

  // ---- requests: the X tile (chunk j = DMA round j), then W1 slabs 0 and 1 ----
  {
    const int p = wave * 64 + lane;           // piece of a chunk: row p/4, PHYSICAL slot p%4 = logical slot (p%4) ^ ((row>>2)&3)
    const int q = p >> 2;
    const int slot = (p & 3) ^ ((q >> 2) & 3);
    const bool live = cell0 + q < a.cells;
    const char* src = live ? (const char*)a.in + ((size_t)(cell0 + q) * a.inC + slot * 8) * sizeof(T) : zero;
#pragma unroll
    for(int j = 0; j < NPX; j++) dma16(live ? src + j * (KCHUNK * (int)sizeof(T)) : zero, smem + j * G::CHUNK_BYTES + wave * 1024);
  }
  auto issueW = [&](const void* w, int slabBytes, int npw, int ringOff, int step, int nSteps) {
    const bool liveStep = step < nSteps;
    const char* slab = (const char*)w + (size_t)(liveStep ? step : 0) * slabBytes;
    char* dst = smem + ringOff + (step % RING) * slabBytes;
    for(int j = 0; j < npw; j++) {
      const int pbase = (j * NWAVES + wave) * 64;
      const bool inRange = liveStep && pbase * 16 < slabBytes;
      dma16(slab + (size_t)(inRange ? pbase + lane : lane) * 16, inRange ? dst + pbase * 16 : mySlack);
    }
  };
#pragma unroll
  for(int c = 0; c < RING - 1; c++) issueW(a.w1, G::W1_SLAB, NPW1, G::W1_OFF, c, K1);

  // ---- per-channel parameters and the mask tile -> LDS (plain loads; published by the first barrier) ----
  for(int i = tid; i < G::C2; i += NTHREADS) {
    sc1S[i] = a.scale1[i];
    bi1S[i] = a.bias1[i];
  }
  for(int i = tid; i < G::C3; i += NTHREADS) {
    sc2S[i] = a.scale2[i];
    bi2S[i] = a.bias2[i];
  }
  if(tid < TM) maskS[tid] = cell0 + tid < a.cells ? a.mask[cell0 + tid] : 0.0f;
  waitLds();  // published by the first barrier of the loop below

  // ---- GEMM 1: accumulators start from the residual stream ----
  const int wm1 = wave / G::WNG1, wn1 = wave % G::WNG1;
  // The residual pieces are requested here and added in epilogue 1 (in fp32, before the rounding) - the same order of
  // operations as the convolution kernel's epilogue, which the unfused schedule runs: bit-identical results.
  f32x16 acc1[WN1][MT1];
  u32x4 rq[MT1][WN1][2];
#pragma unroll
  for(int pt = 0; pt < MT1; pt++) {
    const int cl = wm1 * (32 * MT1) + pt * 32 + myPos;
    const bool live = cell0 + cl < a.cells;
    // 16-byte pieces: the lane pair (c, c + 32) of a tile column loads channels [16 j + 8 h, +8) and exchanges halves
    // afterwards (device_common.h unpair)
    const T* const rrow = live ? (const T*)a.resid + (size_t)(cell0 + cl) * a.trunkC + wn1 * (32 * WN1) + 8 * khalf : (const T*)zero;
#pragma unroll
    for(int ct = 0; ct < WN1; ct++) {
#pragma unroll
      for(int j = 0; j < 2; j++) rq[pt][ct][j] = *(const u32x4*)(live ? rrow + ct * 32 + 16 * j : rrow);
#pragma unroll
      for(int r = 0; r < 16; r++) acc1[ct][pt][r] = 0.0f;
    }
  }

  const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto ldsV8 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) V8*)addr; };
  const unsigned wXor = (lane >> 2) & 3;  // (row>>2)&3 of a weight row: tile bases are multiples of 32 rows

  // step c: slab c has landed (two younger slabs' requests may be in flight) -> barrier -> request slab c+2 into the
  // ring slot whose last readers passed this barrier -> multiply
  {
    unsigned xRow[MT1];  // byte offset of this lane's image row inside a chunk, and its swizzle
    unsigned xXor[MT1];
#pragma unroll
    for(int pt = 0; pt < MT1; pt++) {
      const unsigned q = wm1 * (32 * MT1) + pt * 32 + myPos;
      xRow[pt] = q * ROWB;
      xXor[pt] = (q >> 2) & 3;
    }
    const unsigned wRow = (wn1 * (32 * WN1) + l31) * ROWB;
    for(int c = 0; c < K1; c++) {
      waitVm<(RING - 2) * NPW1>();  // everything but the younger slabs: X (oldest) and slab c
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issueW(a.w1, G::W1_SLAB, NPW1, G::W1_OFF, c + RING - 1, K1);
      const unsigned wBase = ldsBase + G::W1_OFF + (c % RING) * G::W1_SLAB + wRow;
      const unsigned xBase = ldsBase + c * G::CHUNK_BYTES;
#pragma unroll
      for(int kk = 0; kk < 2; kk++) {
        V8 wf[WN1], xf[MT1];
        const unsigned ls = kk * 2 + khalf;
#pragma unroll
        for(int ct = 0; ct < WN1; ct++) wf[ct] = ldsV8(wBase + ct * 32 * ROWB + ((ls ^ wXor) << 4));
#pragma unroll
        for(int pt = 0; pt < MT1; pt++) xf[pt] = ldsV8(xBase + xRow[pt] + ((ls ^ xXor[pt]) << 4));
#pragma unroll
        for(int ct = 0; ct < WN1; ct++)
#pragma unroll
          for(int pt = 0; pt < MT1; pt++) acc1[ct][pt] = TR::mfma(wf[ct], xf[pt], acc1[ct][pt]);
      }
    }
  }
  waitVm<0>();  // the trailing padding requests
  __builtin_amdgcn_s_barrier();  // every wave is done with X and the W1 ring: the LDS changes hands
  asm volatile("" ::: "memory");
#pragma unroll
  for(int c = 0; c < RING - 1; c++) issueW(a.w2, G::W2_SLAB, NPW2, G::W2_OFF, c, K2);

  // ---- epilogue 1: trunk raw -> HBM, activated trunk -> LDS image of GEMM 2 ----
  withActKind(a.actKind1, [&](auto kindTag) {
  constexpr int KIND = decltype(kindTag)::value;
#pragma unroll
  for(int pt = 0; pt < MT1; pt++) {
    const int cl = wm1 * (32 * MT1) + pt * 32 + myPos;
    const bool live = cell0 + cl < a.cells;
    // off-board cells of the activated image are zero whatever the arithmetic gave: result bits ANDed with all-ones / zeros
    const unsigned onBits = maskS[cl] == 1.0f ? 0xffffffffu : 0u;
    // stores are unconditional (a cell past the end writes to the trash area behind the zero page): their number is then a
    // constant, which the waits of GEMM 2's first steps count on
    T* const trash = (T*)((char*)const_cast<void*>(a.zeroPage) + ZERO_PAGE_BYTES) + lane * 8;
    T* const rawRow = live ? (T*)a.rawOut + (size_t)(cell0 + cl) * a.trunkC : nullptr;
    T* const actRow = live && a.actOut != nullptr ? (T*)a.actOut + (size_t)(cell0 + cl) * a.trunkC : nullptr;
    const bool hasActOut = a.actOut != nullptr;  // uniform
    const unsigned rowXor = ((unsigned)cl >> 2) & 3;
#pragma unroll
    for(int ct = 0; ct < WN1; ct++) {
      const int chunk = wn1 * WN1 + ct;  // 32-channel chunk of the trunk this tile covers
      u32x2 rp[4], op[4], resP[4];
      unpair(rq[pt][ct], resP);
#pragma unroll
      for(int g = 0; g < 4; g++) {
        const int c = chunk * 32 + 8 * g + 4 * khalf;
        const f32x4 sc = *(const f32x4*)(sc1S + c);
        const f32x4 bi = *(const f32x4*)(bi1S + c);
        V4 r, o;
        const V4 rr = __builtin_bit_cast(V4, resP[g]);
#pragma unroll
        for(int i = 0; i < 4; i += 2) {
          const float v0 = acc1[ct][pt][4 * g + i] + TR::toFloat(rr[i]), v1 = acc1[ct][pt][4 * g + i + 1] + TR::toFloat(rr[i + 1]);
          r[i] = TR::fromFloat(v0);
          r[i + 1] = TR::fromFloat(v1);
          f32x2 x;
          x[0] = v0 * sc[i] + bi[i];
          x[1] = v1 * sc[i + 1] + bi[i + 1];
          const f32x2 y = actK2<KIND>(x);
          o[i] = TR::fromFloat(y[0]);
          o[i + 1] = TR::fromFloat(y[1]);
        }
        rp[g] = __builtin_bit_cast(u32x2, r);
        op[g] = __builtin_bit_cast(u32x2, o);
        op[g][0] &= onBits;
        op[g][1] &= onBits;
      }
      // regroup into 16-byte runs (lane pair c, c + 32): this lane now holds channels chunk*32 + 16 j + 8 h + [0,8)
      u32x4 rawQ[2], oq[2];
      pairUp(rp, rawQ);
      pairUp(op, oq);
#pragma unroll
      for(int j = 0; j < 2; j++) {
        const int c = chunk * 32 + 16 * j + 8 * khalf;
        *(u32x4*)(live ? rawRow + c : trash) = rawQ[j];
        if(hasActOut) *(u32x4*)(live ? actRow + c : trash) = oq[j];
        // image layout: chunk `chunk`, row cl, logical 16-byte slot 2 j + h at physical slot ^ rowXor
        *(u32x4*)(smem + chunk * G::CHUNK_BYTES + cl * ROWB + (((2 * j + khalf) ^ rowXor) << 4)) = oq[j];
      }
    }
  }
  });
  waitLds();  // the image is read by other waves after the next barrier

  // ---- GEMM 2 ----
  const int wm2 = wave / G::WNG2, wn2 = wave % G::WNG2;
  f32x16 acc2[WN2][MT2];
#pragma unroll
  for(int ct = 0; ct < WN2; ct++)
#pragma unroll
    for(int pt = 0; pt < MT2; pt++)
#pragma unroll
      for(int r = 0; r < 16; r++) acc2[ct][pt][r] = 0.0f;
  {
    unsigned xRow[MT2], xXor[MT2];
#pragma unroll
    for(int pt = 0; pt < MT2; pt++) {
      const unsigned q = wm2 * (32 * MT2) + pt * 32 + myPos;
      xRow[pt] = q * ROWB;
      xXor[pt] = (q >> 2) & 3;
    }
    const unsigned wRow = (wn2 * (32 * WN2) + l31) * ROWB;
    // requests in flight, oldest first: slabs 0 .. RING-2, epilogue 1's stores, then slab c+RING-1 (requested at step c).
    // Waiting for slab c with "all but the youngest slabs" would, in the first RING-1 steps, also wait for the stores to be
    // acknowledged: there the stores (younger than the slab) are counted in.
    const int nStores1 = MT1 * WN1 * 2 * (a.actOut != nullptr ? 2 : 1);
    for(int c = 0; c < K2; c++) {
      if(c < RING - 1) waitVmSel((RING - 2) * NPW2 + nStores1);
      else waitVm<(RING - 2) * NPW2>();
      __builtin_amdgcn_s_barrier();  // c == 0: also publishes the activated image written above
      asm volatile("" ::: "memory");
      issueW(a.w2, G::W2_SLAB, NPW2, G::W2_OFF, c + RING - 1, K2);
      const unsigned wBase = ldsBase + G::W2_OFF + (c % RING) * G::W2_SLAB + wRow;
      const unsigned xBase = ldsBase + c * G::CHUNK_BYTES;
#pragma unroll
      for(int kk = 0; kk < 2; kk++) {
        V8 wf[WN2], xf[MT2];
        const unsigned ls = kk * 2 + khalf;
#pragma unroll
        for(int ct = 0; ct < WN2; ct++) wf[ct] = ldsV8(wBase + ct * 32 * ROWB + ((ls ^ wXor) << 4));
#pragma unroll
        for(int pt = 0; pt < MT2; pt++) xf[pt] = ldsV8(xBase + xRow[pt] + ((ls ^ xXor[pt]) << 4));
#pragma unroll
        for(int ct = 0; ct < WN2; ct++)
#pragma unroll
          for(int pt = 0; pt < MT2; pt++) acc2[ct][pt] = TR::mfma(wf[ct], xf[pt], acc2[ct][pt]);
      }
    }
  }
  waitVm<0>();

  // ---- epilogue 2: mid raw and activated -> HBM ----
  withActKind(a.actKind2, [&](auto kindTag) {
  constexpr int KIND = decltype(kindTag)::value;
#pragma unroll
  for(int pt = 0; pt < MT2; pt++) {
    const int cl = wm2 * (32 * MT2) + pt * 32 + myPos;
    const bool live = cell0 + cl < a.cells;  // the same for both lanes of a pair; no lane leaves before the exchange below
    const unsigned onBits = maskS[cl] == 1.0f ? 0xffffffffu : 0u;
    T* const rawRow = (T*)a.rawOut2 + (size_t)(cell0 + cl) * a.midC;
    T* const actRow = (T*)a.actOut2 + (size_t)(cell0 + cl) * a.midC;
#pragma unroll
    for(int ct = 0; ct < WN2; ct++) {
      u32x2 rp[4], op[4];
#pragma unroll
      for(int g = 0; g < 4; g++) {
        const int c = wn2 * (32 * WN2) + ct * 32 + 8 * g + 4 * khalf;
        const f32x4 sc = *(const f32x4*)(sc2S + c);
        const f32x4 bi = *(const f32x4*)(bi2S + c);
        V4 r, o;
#pragma unroll
        for(int i = 0; i < 4; i += 2) {
          const float v0 = acc2[ct][pt][4 * g + i], v1 = acc2[ct][pt][4 * g + i + 1];
          r[i] = TR::fromFloat(v0);
          r[i + 1] = TR::fromFloat(v1);
          f32x2 x;
          x[0] = v0 * sc[i] + bi[i];
          x[1] = v1 * sc[i + 1] + bi[i + 1];
          const f32x2 y = actK2<KIND>(x);
          o[i] = TR::fromFloat(y[0]);
          o[i + 1] = TR::fromFloat(y[1]);
        }
        rp[g] = __builtin_bit_cast(u32x2, r);
        op[g] = __builtin_bit_cast(u32x2, o);
        op[g][0] &= onBits;
        op[g][1] &= onBits;
      }
      u32x4 rq[2], oq[2];
      pairUp(rp, rq);
      pairUp(op, oq);
      if(live) {
#pragma unroll
        for(int j = 0; j < 2; j++) {
          const int c = wn2 * (32 * WN2) + ct * 32 + 16 * j + 8 * khalf;
          *(u32x4*)(rawRow + c) = rq[j];
          *(u32x4*)(actRow + c) = oq[j];
        }
      }
    }
  }
  });
}

template <class TR, int K1, int WN1, int WN2, int TM, int NW>
hipError_t launchPair(const PwPairArgs& a, hipStream_t stream) {
  typedef Geom<K1, WN1, WN2, TM, NW> G;
  static_assert(NW == 8 || G::LDS_BYTES <= 80 * 1024, "the 4-wave shape exists to fit two work-groups per CU");
  static_assert(G::LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
  static_assert(G::A2_BYTES <= G::PH1_END, "the activated image reuses the GEMM 1 operand area");
  auto kern = pointwisePairKernel<TR, K1, WN1, WN2, TM, NW>;
  constexpr int MAX_DEVICES = 64;  // the >64 KiB LDS opt-in is per function AND device (conv_kernel.h launchOne)
  static std::atomic<bool> attrSet[MAX_DEVICES];
  int dev = 0;
  hipError_t de = hipGetDevice(&dev);
  if(de != hipSuccess) return de;
  if(dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
  if(!attrSet[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if(e != hipSuccess) return e;
    attrSet[dev].store(true, std::memory_order_release);
  }
  if(a.cells <= 0) return hipErrorInvalidValue;
  const long long tiles = (a.cells + TM - 1) / TM;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(G::NTHREADS), G::LDS_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace pwk
}  // namespace kmx
#endif

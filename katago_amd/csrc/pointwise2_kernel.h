// pointwise2_kernel.h — the seam between two nested-bottleneck blocks (same arithmetic as pointwise_kernel.h, see there)
// as a PERSISTENT, software-pipelined kernel: a work-group walks tiles tile0, tile0 + grid, ... and, while it multiplies tile
// i, the inputs of tile i + 1 are already on their way.
//
// Why. pointwise_kernel.h runs  fetch X + residual | GEMM 1 | epilogue 1 | GEMM 2 | epilogue 2  strictly one after the other,
// one tile per work-group: ~30 k of its ~58 k cycles per tile are memory round trips with idle matrix cores, and every
// work-group of a round hits HBM at the same moment (measured 2.6-3.1 TB/s for a kernel that moves 248 MB).
// Here:
//   * the X tile of tile i + 1 (LDS-DMA) and its first W1 slabs are requested as soon as GEMM 1 of tile i has read the last of
//     tile i's X - they land during tile i's epilogues and second GEMM;
//   * the W2 slabs run on their own ring across the tile boundary, the W1 ring is refilled as soon as GEMM 1 is done;
//   * the 384-channel activated trunk image is never whole in LDS: epilogue 1 and GEMM 2 alternate in PARTS of 64 channels
//     (two 32-channel K chunks of GEMM 2), so that LDS has room for the X buffer to be prefetched into (the round-2 kernel
//     keeps 96 KB of activated image and cannot hold a second X);
//   * the residual pieces of part q + 2 are requested when part q has consumed its own (a ring of three in registers).
// Same MFMA, same operand roles, same K order (chunk, k half), same rounding points as pointwise_kernel.h and as two launches
// of conv_kernel.h: BIT-IDENTICAL results (tests/test_gpu_pointwise.py).
//
// Work-group = 8 waves = 4 (cell tiles of 32) x 2; TM = 128 cells.
//   GEMM 1: wave (ct, h) owns cells 32 ct + [0, 32) and the channel tiles 2 q + h, q = 0 .. NP-1 (every other 32-channel tile)
//           - so that every PART q = channel tiles {2 q, 2 q + 1} = K chunks {2 q, 2 q + 1} of GEMM 2 has one tile per wave.
//   GEMM 2: wave (ct, h) owns the same cells and channels 32 WN2 h + [0, 32 WN2).
// LDS: [X: K1 chunks][A2: 2 chunks][W1 ring: 2 slabs][W2 ring: 3 slabs][params][mask x 2][slack] = 154.5 KiB for 192->384->192.
//
// vmcnt bookkeeping. Every wave issues the same sequence of vector-memory operations per tile (requests that must not land go
// to a slack area / the zero page / a trash area), so "everything up to request Z has landed" is s_waitcnt vmcnt(N) with N =
// the number of operations issued after Z - a compile-time constant at every wait below; each states what it leaves in flight.
#ifndef KMX_POINTWISE2_KERNEL_H_
#define KMX_POINTWISE2_KERNEL_H_

#include <atomic>
#include <cstdlib>

#include "device_common.h"

namespace kmx {
namespace pw2 {

// the per-lane row pointers below pass through an opaque asm (see there), which hides their address space from the compiler:
// without this qualifier their accesses become FLAT instructions, which also count as LDS operations (lgkmcnt)
#define GLOBAL __attribute__((address_space(1)))
constexpr int ROWB = WROW_HALFS * 2;  // 64-byte LDS rows (four 16-byte slots, slot s of row r stored at s ^ ((r>>2)&3))
constexpr int TM = 128;               // cells per tile
constexpr int NWAVES = 8;
constexpr int NTHREADS = NWAVES * 64;

template <int K1, int K2, int WN2>
struct Geom {
  static constexpr int C1 = 32 * K1, C2 = 32 * K2, C3 = 64 * WN2;
  static constexpr int NP = K2 / 2;  // parts: 64 trunk channels = two K chunks of GEMM 2 each
  static_assert(K2 % 2 == 0 && K2 % 3 == 0, "parts of two chunks; the W2 ring of three slabs runs on across tiles");
  static constexpr int CHUNK_BYTES = TM * ROWB;
  static constexpr int X_OFF = 0, X_BYTES = K1 * CHUNK_BYTES;
  static constexpr int A2_OFF = X_OFF + X_BYTES, A2_BYTES = 2 * CHUNK_BYTES;
  static constexpr int W1_SLAB = C2 * ROWB;            // one K chunk of W1 in HBM: [C2 rows][64 B]
  static constexpr int W1_PART = K1 * 64 * ROWB;        // the 64 rows of one part for all K1 chunks, in LDS: [chunk][64 rows]
  static constexpr int W1_OFF = A2_OFF + A2_BYTES, W1_RING = 2;
  static constexpr int W2_SLAB = C3 * ROWB, W2_OFF = W1_OFF + W1_RING * W1_PART, W2_RING = 3;
  static constexpr int PARAM_OFF = W2_OFF + W2_RING * W2_SLAB;  // scale1, bias1 [C2], scale2, bias2 [C3]
  static constexpr int MASK_OFF = PARAM_OFF + (2 * C2 + 2 * C3) * 4;  // two tiles of TM floats
  static constexpr int SLACK_OFF = MASK_OFF + 2 * TM * 4;
  static constexpr int LDS_BYTES = SLACK_OFF + 1024;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
  static_assert((64 * (NP - 1) + 16 + 8) * 2 <= TRASH_BYTES && (32 * (WN2 - 1) + 16 + 8) * 2 <= TRASH_BYTES, "dead cells store at these offsets into the trash area");
  // 1 KiB requests (64 lanes x 16 bytes) per wave
  static constexpr int N_X = 1 + K1;                                 // the mask + one round of the work-group per chunk of X
  static constexpr int NPW1 = (4 * K1 + NWAVES - 1) / NWAVES;        // per W1 part (4 K1 KiB)
  static constexpr int NPW2 = (C3 * 4 + NTHREADS - 1) / NTHREADS;    // per W2 slab
  static constexpr int N_S2 = WN2 * 2 * 2;                           // stores of epilogue 2
  static constexpr int nR(int q) { return q + 1 < NP ? 2 : 0; }      // residual loads requested by part q (for part q + 1)
};

template <int N>
__device__ __forceinline__ void waitVm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt vmcnt(n) for an n that is a constant only after the part loop is unrolled (0 .. 63: the counter has six bits)
template <int LO, int HI>
__device__ __forceinline__ void waitVmRange(int n) {
  if constexpr(LO == HI) waitVm<LO>();
  else {
    constexpr int MID = (LO + HI) / 2;
    if(n <= MID) waitVmRange<LO, MID>(n);
    else waitVmRange<MID + 1, HI>(n);
  }
}
__device__ __forceinline__ void waitVmSel(int n) { waitVmRange<0, 63>(n); }
__device__ __forceinline__ void waitLds() {  // back-off barriers do not drain the LDS queue
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void wgBarrier() {
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned ldsWaveBase) {
  __builtin_amdgcn_global_load_lds(
    (const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)(size_t)ldsWaveBase, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* gsrc, unsigned ldsWaveBase) {
  __builtin_amdgcn_global_load_lds(
    (const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)(size_t)ldsWaveBase, 4, 0, 0);
}

// TIMING (conv_bench.hip only): s_memtime stamps between the phases, summed per wave over the tiles of work-group 0 into a.dbg:
// [0] tile top (wait + residual requests + barrier)  [1] GEMM 1 of part 0  [2] P1 wait + barrier + requests  [3] GEMM 1 of the next part
// [4] epilogue 1 (arithmetic, stores, image write)   [5] P2 / P3 wait + barrier + requests  [6] the GEMM-2 steps  [7] epilogue 2
template <class TR, int K1, int K2, int WN2, int KIND1, int KIND2, bool TIMING = false>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void pointwisePairPersistentKernel(const PwPairArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  typedef typename TR::V4 V4;
  typedef Geom<K1, K2, WN2> G;
  constexpr int NP = G::NP, NPW1 = G::NPW1, NPW2 = G::NPW2, NX = G::N_X;

  extern __shared__ __attribute__((aligned(256))) char smemPw2[];
  const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smemPw2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned khalf = lane >> 5;
  const int l31 = lane & 31;
  // position of lane l inside a 32-column tile (conv_kernel.h): the 16-lane groups of a ds_read_b128 read 16 consecutive rows
  const int myPos = l31 < 4 ? l31 : l31 < 12 ? l31 + 12 : l31 < 16 ? l31 - 8 : l31 < 20 ? l31 + 8 : l31 < 28 ? l31 - 12 : l31;
  const int cellTile = wave >> 1, half = wave & 1;
  const int cl = cellTile * 32 + myPos;  // this lane's cell (row of the tile) in both GEMMs
  const char* const zero = (const char*)a.zeroPage;
  T* const trash0 = (T*)((char*)const_cast<void*>(a.zeroPage) + ZERO_PAGE_BYTES);  // TRASH_BYTES of writable scratch
  const unsigned slack = ldsBase + G::SLACK_OFF;
  auto ldsV8 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) V8*)(size_t)addr; };
  auto ldsF4 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) f32x4*)(size_t)addr; };
  auto ldsF1 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) float*)(size_t)addr; };

  const long long numTiles = (a.cells + TM - 1) / TM;
  const long long stride = gridDim.x;

  // ---- request helpers: every call issues a fixed number of instructions ----
  const unsigned laneOff16 = (unsigned)lane * 16u;
  // N_X instructions: the mask tile (one 4-byte request per lane of waves 0, 1) and the X tile, chunk j = round j of the work-group
  auto issueX = [&](long long tile, int parity, bool live) {
    const long long cell0 = tile * TM;
    {
      const int cellIdx = wave * 64 + lane;
      const bool m = live && wave < 2 && cell0 + cellIdx < a.cells;
      dma4(m ? (const void*)(a.mask + cell0 + cellIdx) : (const void*)zero,
           live && wave < 2 ? ldsBase + G::MASK_OFF + (unsigned)parity * (TM * 4) + wave * 256 : slack);
    }
    const int p = wave * 64 + lane;  // piece of a chunk: row p/4, PHYSICAL slot p%4 = logical slot (p%4) ^ ((row>>2)&3)
    const int q = p >> 2;
    const int slot = (p & 3) ^ ((q >> 2) & 3);
    const bool rowLive = live && cell0 + q < a.cells;
    const char* src = rowLive ? (const char*)a.in + ((size_t)(cell0 + q) * a.inC + slot * 8) * sizeof(T) : zero;
#pragma unroll
    for(int j = 0; j < K1; j++)
      dma16(rowLive ? src + j * (KCHUNK * (int)sizeof(T)) : zero, live ? ldsBase + G::X_OFF + j * G::CHUNK_BYTES + wave * 1024 : slack);
  };
  // The W1 rows of part `part` (64 trunk channels) for ALL K1 chunks: K1 runs of 4 KiB out of the [chunk][C2 rows][64 B] weight
  // tensor, laid out in LDS as [chunk][64 rows]. 1 KiB request r of the 4 K1: chunk r / 4, rows 16 (r % 4) + [0, 16).
  auto issueW1Part = [&](int part, bool live) {
    const unsigned dst = ldsBase + G::W1_OFF + (unsigned)(part % G::W1_RING) * G::W1_PART;
    // (the base is made opaque at every call: otherwise all the request addresses of a tile - uniform and per-lane - are
    // computed once before the tile loop and kept in registers, ~100 of them, and the kernel spills)
    const char* w1 = (const char*)a.w1;
    asm volatile("" : "+s"(w1));
#pragma unroll
    for(int j = 0; j < NPW1; j++) {
      const int r = j * NWAVES + wave;
      const bool inRange = live && r < 4 * K1;
      const int rr = inRange ? r : 0;
      const char* src = w1 + (size_t)(rr >> 2) * G::W1_SLAB + (size_t)(part * 64 + (rr & 3) * 16) * ROWB;
      dma16(src + laneOff16, inRange ? dst + rr * 1024 : slack);
    }
  };
  auto issueW2 = [&](int slab, bool live) {
    const char* w2 = (const char*)a.w2;
    asm volatile("" : "+s"(w2));
    const char* src = w2 + (size_t)slab * G::W2_SLAB;
    const unsigned dst = ldsBase + G::W2_OFF + (unsigned)(slab % G::W2_RING) * G::W2_SLAB;
#pragma unroll
    for(int j = 0; j < NPW2; j++) {
      const int pbase = (j * NWAVES + wave) * 64;
      const bool inRange = live && pbase * 16 < G::W2_SLAB;
      dma16(src + (inRange ? pbase * 16 : 0) + laneOff16, inRange ? dst + pbase * 16 : slack);
    }
  };

  // ---- prologue: parameters by plain loads (nothing else is in flight yet), then the first tile's requests ----
  {
    float* const p = (float*)(smemPw2 + G::PARAM_OFF);
    for(int i = tid; i < G::C2; i += NTHREADS) {
      p[i] = a.scale1[i];
      p[G::C2 + i] = a.bias1[i];
    }
    for(int i = tid; i < G::C3; i += NTHREADS) {
      p[2 * G::C2 + i] = a.scale2[i];
      p[2 * G::C2 + G::C3 + i] = a.bias2[i];
    }
    waitLds();  // published by the first barrier
  }
  long long tile = blockIdx.x;
  issueX(tile, 0, true);
  issueW1Part(0, true);
  issueW1Part(1, true);
  issueW2(0, true);
  int parity = 0;  // which of the two mask buffers this tile uses (the tiles of one work-group alternate)

  // per-lane LDS addresses of the fragment reads and the image write, per k half kk (logical 16-byte slot 2 kk + khalf of the
  // lane's row, swizzled); opaque, so that every use is "this register + a constant the instruction carries"
  const unsigned xXor = ((unsigned)cl >> 2) & 3;
  const unsigned wXor = (lane >> 2) & 3;  // (row>>2)&3 of a weight row (tile bases are multiples of 32)
  unsigned xLane[2], w1Lane[2], w2Lane[2], a2wLane[2];
#pragma unroll
  for(int kk = 0; kk < 2; kk++) {
    const unsigned ls = kk * 2 + khalf;
    xLane[kk] = ldsBase + (unsigned)cl * ROWB + ((ls ^ xXor) << 4);                          // + X_OFF + chunk, or + A2_OFF + chunk
    w1Lane[kk] = ldsBase + G::W1_OFF + (unsigned)(half * 32 + l31) * ROWB + ((ls ^ wXor) << 4);         // + ring slot + chunk * 64 rows
    w2Lane[kk] = ldsBase + G::W2_OFF + (unsigned)(half * (32 * WN2) + l31) * ROWB + ((ls ^ wXor) << 4);  // + ring slot + tile * 32 rows
    a2wLane[kk] = ldsBase + G::A2_OFF + half * G::CHUNK_BYTES + (unsigned)cl * ROWB + (((2 * kk + khalf) ^ xXor) << 4);  // piece j = kk
    asm volatile("" : "+v"(xLane[kk]), "+v"(w1Lane[kk]), "+v"(w2Lane[kk]), "+v"(a2wLane[kk]));
  }
  // The two waves of a SIMD (wave w and w + 4: work-groups fill the SIMDs 0 -> 2 -> 1 -> 3 cyclically, twice) run the matrix
  // work and the vector work of a phase in OPPOSITE order: one multiplies while the other does epilogue arithmetic. In lock
  // step both epilogues (the bottleneck: ~1.3 k cycles of vector ALU per 32 x 32 tile) compete for the SIMD's vector issue
  // while its matrix pipe idles, then both wait on the matrix pipe.
  const bool matrixFirst = wave < 4;  // uniform
  // per-lane bases of the parameter reads of the two epilogues (opaque: see there)
  unsigned p1Lane = ldsBase + (unsigned)(32 * half + 4 * khalf) * 4u, p2Lane = ldsBase + (unsigned)(32 * WN2 * half + 4 * khalf) * 4u;
  asm volatile("" : "+v"(p1Lane), "+v"(p2Lane));

  // GEMM 1 of one part: this wave's 32 x 32 tile of D1 over all K1 chunks (X whole in LDS, the part's W1 rows whole in LDS)
  auto gemm1 = [&](int part, f32x16& acc) {
#pragma unroll
    for(int r = 0; r < 16; r++) acc[r] = 0.0f;
    // (fragments of two chunks at a time: left alone the scheduler hoists all 4 K1 reads of the part - 96 registers)
#pragma unroll
    for(int c = 0; c < K1; c++) {
#pragma unroll
      for(int kk = 0; kk < 2; kk++) {
        const V8 xf = ldsV8(xLane[kk] + (unsigned)(G::X_OFF + c * G::CHUNK_BYTES));
        const V8 wf = ldsV8(w1Lane[kk] + (unsigned)((part % G::W1_RING) * G::W1_PART + c * (64 * ROWB)));
        acc = TR::mfma(wf, xf, acc);
      }
      if(c % 2 == 1 && c + 1 < K1) __builtin_amdgcn_sched_barrier(0);
    }
  };

  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tPrev = TIMING ? __builtin_readcyclecounter() : 0;
  const unsigned long long tStart = tPrev;
  auto stamp = [&](int which) {
    if(!TIMING) return;
    const unsigned long long now = __builtin_readcyclecounter();
    seg[which] += now - tPrev;
    tPrev = now;
  };

  // first tile: its mask, X and the W1 rows of parts 0, 1 have landed (in flight at most: W2 slab 0); GEMM 1 of part 0
  f32x16 acc1[2];
  waitVm<NPW2>();
  wgBarrier();
  gemm1(0, acc1[0]);
  stamp(1);

  for(; tile < numTiles; tile += stride) {
    const long long cell0 = tile * TM;
    const bool hasNext = tile + stride < numTiles;  // uniform
    const bool live = cell0 + cl < a.cells;         // the same for both lanes of a pair
    const unsigned maskA = ldsBase + G::MASK_OFF + (unsigned)parity * (TM * 4);

    // residual pieces: 16 bytes per lane and request, the lane pair (c, c + 32) loads channels [16 j + 8 h, +8) of its tile;
    // requested one part ahead, two sets of registers (the compiler keeps their s_waitcnt)
    u32x4 rq[2][2];
    // (per-lane row pointers are opaque and the rest of an address a constant the instruction carries; dead cells read the zero
    // page and write the trash area - all at the same small offsets, which both areas are large enough for)
    const T* rrow = live ? (const T*)a.resid + (size_t)(cell0 + cl) * a.trunkC + 32 * half + 8 * khalf : (const T*)zero;
    asm volatile("" : "+v"(rrow));
    auto loadResid = [&](int q, u32x4 (&dst)[2]) {
#pragma unroll
      for(int j = 0; j < 2; j++) dst[j] = *(const GLOBAL u32x4*)(rrow + 64 * q + 16 * j);
    };
    loadResid(0, rq[0]);

    f32x16 acc2[WN2];
#pragma unroll
    for(int ct = 0; ct < WN2; ct++)
#pragma unroll
      for(int r = 0; r < 16; r++) acc2[ct][r] = 0.0f;
    const unsigned onBits = ldsF1(maskA + cl * 4) == 1.0f ? 0xffffffffu : 0u;  // off-board cells of activated images are zero
    T* rawRow = live ? (T*)a.rawOut + (size_t)(cell0 + cl) * a.trunkC + 32 * half + 8 * khalf : trash0;
    asm volatile("" : "+v"(rawRow));
    stamp(0);

#pragma unroll
    for(int q = 0; q < NP; q++) {
      // ---- P1. The W1 rows of part q + 1 (requested in P1 of part q - 1; parts 0, 1: before the tile) have landed; in flight at
      // most what part q - 1 issued after them: its W2 slab, residual loads, stores, its P2 slab.
      if(q >= 1 && q + 1 < NP) waitVmSel(NPW2 + G::nR(q - 1) + 2 + NPW2);
      wgBarrier();  // every wave has finished GEMM 1 of part q (its rows may be overwritten) and GEMM 2 of part q - 1 (so may the image)
      issueW1Part(q + 2 < NP ? q + 2 : q + 2 - NP, q + 2 < NP || hasNext);  // past the end: parts 0, 1 of the next tile
      issueW2(2 * q + 1, true);  // into the slot of slab 2 q - 2, read in GEMM 2 of part q - 1
      stamp(2);

      // ---- phase A: GEMM 1 of the NEXT part and epilogue 1 of this part (+ residual, raw trunk -> HBM, activated -> the LDS
      // image of GEMM 2), in opposite order on the two waves of a SIMD
      if(matrixFirst && q + 1 < NP) gemm1(q + 1, acc1[(q + 1) & 1]);
      stamp(3);
      {
        const int chTile = 2 * q + half;  // 32-channel tile of the trunk
        u32x2 rp[4], op[4], resP[4];
        unpair(rq[q & 1], resP);
        if(q + 1 < NP) loadResid(q + 1, rq[(q + 1) & 1]);
#pragma unroll
        for(int g = 0; g < 4; g++) {
          // (one per-lane base + a constant the instruction carries: left as one expression, every one of the 36 addresses of
          // a tile is computed before the tile loop and kept in a register of its own)
          const f32x4 sc = ldsF4(p1Lane + (unsigned)(G::PARAM_OFF + (64 * q + 8 * g) * 4));
          const f32x4 bi = ldsF4(p1Lane + (unsigned)(G::PARAM_OFF + G::C2 * 4 + (64 * q + 8 * g) * 4));
          V4 r, o;
          const V4 rr = __builtin_bit_cast(V4, resP[g]);
#pragma unroll
          for(int i = 0; i < 4; i += 2) {
            const float v0 = acc1[q & 1][4 * g + i] + TR::toFloat(rr[i]), v1 = acc1[q & 1][4 * g + i + 1] + TR::toFloat(rr[i + 1]);
            r[i] = TR::fromFloat(v0);
            r[i + 1] = TR::fromFloat(v1);
            f32x2 x;
            x[0] = v0 * sc[i] + bi[i];
            x[1] = v1 * sc[i + 1] + bi[i + 1];
            const f32x2 y = actK2<KIND1>(x);
            o[i] = TR::fromFloat(y[0]);
            o[i + 1] = TR::fromFloat(y[1]);
          }
          rp[g] = __builtin_bit_cast(u32x2, r);
          op[g] = __builtin_bit_cast(u32x2, o);
          op[g][0] &= onBits;
          op[g][1] &= onBits;
        }
        u32x4 rawQ[2], oq[2];
        pairUp(rp, rawQ);  // this lane now holds channels chTile*32 + 16 j + 8 h + [0, 8)
        pairUp(op, oq);
#pragma unroll
        for(int j = 0; j < 2; j++) *(GLOBAL u32x4*)(rawRow + 64 * q + 16 * j) = rawQ[j];
#pragma unroll
        for(int j = 0; j < 2; j++)  // image layout: chunk `half` of the part, row cl, logical 16-byte slot 2 j + h
          *(__attribute__((address_space(3))) u32x4*)(size_t)a2wLane[j] = oq[j];
      }
      stamp(4);
      if(!matrixFirst && q + 1 < NP) gemm1(q + 1, acc1[(q + 1) & 1]);
      stamp(3);
      waitLds();  // the image is read by other waves after the next barrier

      // ---- P2. W2 slabs 2 q (requested in P2 of part q - 1, or by the previous tile / the prologue) and 2 q + 1 (requested in P1
      // above) have landed; in flight at most this part's residual loads and stores.
      waitVmSel(G::nR(q) + 2);
      wgBarrier();  // also publishes the activated part
      // slab 2 q + 2 into the slot of slab 2 q - 1 (read in GEMM 2 of part q - 1); past the end: the next tile's slab 0
      issueW2(2 * q + 2 < K2 ? 2 * q + 2 : 0, 2 * q + 2 < K2 || hasNext);
      // the X buffer is free once GEMM 1 of the last part has been issued by every wave (phase A of part NP - 2): the next tile's X
      if(q == NP - 2) issueX(tile + stride, parity ^ 1, hasNext);
      stamp(5);
      // ---- GEMM 2 over the part's two K chunks
#pragma unroll
      for(int e = 0; e < 2; e++) {
        const int s = 2 * q + e;
#pragma unroll
        for(int kk = 0; kk < 2; kk++) {
          const V8 xf = ldsV8(xLane[kk] + (unsigned)(G::A2_OFF + e * G::CHUNK_BYTES));
          V8 wf[WN2];
#pragma unroll
          for(int ct = 0; ct < WN2; ct++) wf[ct] = ldsV8(w2Lane[kk] + (unsigned)((s % G::W2_RING) * G::W2_SLAB + ct * 32 * ROWB));
#pragma unroll
          for(int ct = 0; ct < WN2; ct++) acc2[ct] = TR::mfma(wf[ct], xf, acc2[ct]);
        }
      }
      stamp(6);
    }

    // ---- tile end: epilogue 2 (mid raw and activated -> HBM, N_S2 unconditional stores) beside GEMM 1 of the NEXT tile's part 0.
    // The next tile's mask, X and W1 rows of parts 0, 1 have landed; in flight at most what the last part issued after its W1 rows:
    // its W2 slab, its stores, the next tile's W2 slab 0.
    if(hasNext) {
      waitVm<NPW2 + 2 + NPW2>();
      wgBarrier();
      if(matrixFirst) gemm1(0, acc1[0]);
    }
    stamp(1);
    T* rawRow2 = live ? (T*)a.rawOut2 + (size_t)(cell0 + cl) * a.midC + 32 * WN2 * half + 8 * khalf : trash0;
    T* actRow2 = live ? (T*)a.actOut2 + (size_t)(cell0 + cl) * a.midC + 32 * WN2 * half + 8 * khalf : trash0;
    asm volatile("" : "+v"(rawRow2), "+v"(actRow2));
#pragma unroll
    for(int ct = 0; ct < WN2; ct++) {
      u32x2 rp[4], op[4];
#pragma unroll
      for(int g = 0; g < 4; g++) {
        const f32x4 sc = ldsF4(p2Lane + (unsigned)(G::PARAM_OFF + 2 * G::C2 * 4 + (32 * ct + 8 * g) * 4));
        const f32x4 bi = ldsF4(p2Lane + (unsigned)(G::PARAM_OFF + 2 * G::C2 * 4 + G::C3 * 4 + (32 * ct + 8 * g) * 4));
        V4 r, o;
#pragma unroll
        for(int i = 0; i < 4; i += 2) {
          const float v0 = acc2[ct][4 * g + i], v1 = acc2[ct][4 * g + i + 1];
          r[i] = TR::fromFloat(v0);
          r[i + 1] = TR::fromFloat(v1);
          f32x2 x;
          x[0] = v0 * sc[i] + bi[i];
          x[1] = v1 * sc[i + 1] + bi[i + 1];
          const f32x2 y = actK2<KIND2>(x);
          o[i] = TR::fromFloat(y[0]);
          o[i + 1] = TR::fromFloat(y[1]);
        }
        rp[g] = __builtin_bit_cast(u32x2, r);
        op[g] = __builtin_bit_cast(u32x2, o);
        op[g][0] &= onBits;
        op[g][1] &= onBits;
      }
      u32x4 rawQ[2], oq[2];
      pairUp(rp, rawQ);
      pairUp(op, oq);
#pragma unroll
      for(int j = 0; j < 2; j++) {
        *(GLOBAL u32x4*)(rawRow2 + ct * 32 + 16 * j) = rawQ[j];
        *(GLOBAL u32x4*)(actRow2 + ct * 32 + 16 * j) = oq[j];
      }
    }
    stamp(7);
    if(hasNext && !matrixFirst) gemm1(0, acc1[0]);
    stamp(1);
    parity ^= 1;
  }
  waitVm<0>();  // trailing requests into the slack area must land before the LDS is released
  if(TIMING && a.dbg != nullptr && lane == 0 && blockIdx.x == 0) {
    for(int i = 0; i < 8; i++) a.dbg[wave * 9 + i] = seg[i];
    a.dbg[wave * 9 + 8] = __builtin_readcyclecounter() - tStart;
  }
}

template <class TR, int K1, int K2, int WN2, int KIND1, int KIND2, bool TIMING = false>
hipError_t launchPersistent(const PwPairArgs& a, int maxGrid, hipStream_t stream) {
  typedef Geom<K1, K2, WN2> G;
  auto kern = pointwisePairPersistentKernel<TR, K1, K2, WN2, KIND1, KIND2, TIMING>;
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
  if(a.cells <= 0 || a.actOut != nullptr) return hipErrorInvalidValue;
  const long long tiles = (a.cells + TM - 1) / TM;
  // Balanced grid: with k = ceil(tiles / CUs) tiles per work-group, ceil(tiles / k) work-groups finish in the same k rounds as one
  // per CU would and leave the other CUs to whatever else is runnable - the other half-batch stream's kernel (361 tiles of a
  // 128-board half: 181 work-groups of two tiles instead of 256 of which 151 would walk a single tile). KMX_PW_BALANCE=0: one per CU.
  static const bool balance = [] {
    const char* e = getenv("KMX_PW_BALANCE");
    return e == nullptr || atoi(e) != 0;
  }();
  long long grid = tiles < maxGrid ? tiles : maxGrid;
  if(balance && tiles > maxGrid) {
    const long long perGroup = (tiles + maxGrid - 1) / maxGrid;
    grid = (tiles + perGroup - 1) / perGroup;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), G::LDS_BYTES, stream, a);
  return hipGetLastError();
}

#undef GLOBAL
}  // namespace pw2
}  // namespace kmx
#endif

// conv_small_kernel.h — the 3x3 convolution for SMALL batches: four waves multiply, four waves only fetch (round 4).
//
// While the chip is not full a layer takes as long as ONE of its work-groups does (they all run side by side on idle CUs), and a
// small pass is bound by the time INSIDE its kernels: a rocprofv3 trace of batch-8 passes of b18c384nbt shows 2.39 ms of kernel
// time in 2.62 ms of wall time, the 73 3x3 convolutions 21.4 us each (profiles/r04_steps/call1). What a step of such a
// work-group costs, per shape of conv_kernel.h (a work-group = one board x 32 output channels in all of them):
//   * 4 waves x 3 cell tiles (cfg 11): 6 MFMAs = 192 matrix-core cycles per wave and step, but 697 cycles per step - every wave also
//     issues two LDS-DMA requests per step at 100-160 cycles of issue each, a third of them padding (round 3 stamps);
//   * 12 waves x 1 cell tile (cfg 111, round 3): at most one request per wave and step, but every wave reads the SAME 2 KB weight
//     slab for its 2 MFMAs - 48 KB of LDS reads per step, 384 cycles at 128 B/cycle: ~600 cycles per step.
// This shape keeps the reuse of the first (a wave's weight fragments serve three cell tiles: 32 KB of LDS reads per step) and takes the
// requests off the multiplying waves altogether: waves 0-3 do what the four waves of cfg 11 do minus every request; waves 4-7 - one per
// SIMD, beside a multiplying wave - issue the slab and image requests of the step D steps ahead, wait for what the next step needs
// and meet the others at the step's barrier. Same decomposition (board x 32 channels: same DMA bytes, same LDS images), same MFMAs
// per output in the same K order (chunk, tap, k half) and the same epilogue arithmetic: BIT-IDENTICAL to every other shape
// (tests/test_kernels_latest_completion.py on the CPU emulation, tests/test_gpu_layers.py on the MI355X).
//
// Round 5, the default since then (REGW, below): the multiplying waves load their weight fragments straight from global memory into
// registers - from a copy of the weights in fragment order, ConvArgs::wFrag -, LDS holds only the board image, and there is ONE barrier
// per chunk; three instantiations: cfg 128 (a board x 32 channels), 127 (its cell tiles over three work-groups), 126 (a board x 64
// channels). The slab-ring form above stays instantiated behind KMX_CONV_TUNE=regw=0 as the measured baseline.
#ifndef KMX_CONV_SMALL_KERNEL_H_
#define KMX_CONV_SMALL_KERNEL_H_

#include <atomic>
#include <type_traits>

#include "conv_kernel.h"

namespace kmx {
namespace smallk {
using convk::dma16;
using convk::dma4;
using convk::waitVm;
using convk::ROWB;

constexpr int NT = 9, HALO = 1, MT = 3;
constexpr int NCOMPUTE = 4, NLOAD = 4, NWAVES = NCOMPUTE + NLOAD, NTHREADS = NWAVES * 64;
constexpr int NTILE = 32;                 // output channels per work-group
constexpr int HPMAX = 21 * 21;
constexpr int NPA = (HPMAX * 4 + NLOAD * 64 - 1) / (NLOAD * 64);  // 7 image requests per loader wave and chunk: one per tap 0..6
constexpr int ACT_BYTES = NPA * NLOAD * 64 * 16;                   // 28672
constexpr int W_BYTES = NTILE * ROWB;     // 2048: one request of waves 4 and 5 each
constexpr int SLACK_BYTES = 1024;
constexpr int MASK_BYTES = NTHREADS * 4;
constexpr int PARAM_BYTES = 256 * 4;      // scale | bias | per-board bias of the 32 channels: 96 floats, one 4-byte request per lane of waves 0-3
// How far ahead the fetching waves run. One work-group per CU (PACK = false): slabs SIX steps ahead on a ring of eight, the board image TWO
// chunks ahead in three buffers - at batch 1-8 every operand comes from HBM or the Infinity Cache (55 MB of weights cycle through 32 MB of L2
// per pass), ~0.6 us away, and three steps of ~0.2 us did not cover that: the multiplying waves waited at the barrier for the fetching
// waves' s_waitcnt. Two work-groups per CU (PACK = true, 128 registers): three steps and one chunk ahead, 68 KB of LDS each.
// DEPTH 0: three steps / one chunk ahead (all that fits twice per CU); 1: six steps / two chunks.
template <bool PACK, int DEPTH>
struct SG {
  static_assert(!PACK || DEPTH == 0, "two work-groups per CU leave 80 KB of LDS each");
  static_assert(DEPTH == 0 || DEPTH == 1, "");
  static constexpr int D = DEPTH == 0 ? 3 : 6;
  static constexpr int NSW = DEPTH == 0 ? 4 : 8;
  static constexpr int NSA = DEPTH == 1 ? 3 : 2;
  static constexpr int DIST = NSA - 1;  // the image of chunk c + DIST is requested while the loop works on chunk c
  static_assert((DIST + 1) * KCHUNK * 2 <= DEVBUF_TAIL_BYTES, "the image requests' run-ahead must stay inside the readable tail of a DevBuf");
  static constexpr int RING_OFFSET = NSA * ACT_BYTES;
  static constexpr int SLACK_OFFSET = RING_OFFSET + NSW * W_BYTES;
  static constexpr int MASK_OFFSET = SLACK_OFFSET + SLACK_BYTES;
  static constexpr int PARAM_OFFSET = MASK_OFFSET + MASK_BYTES;
  static constexpr int LDS_BYTES = PARAM_OFFSET + PARAM_BYTES;
  // image requests a slab-fetching wave issues at tap t (steady state): one at taps 0 .. NPA-1
  static constexpr int imgAt(int t) { return ((t % NT) + NT) % NT < NPA ? 1 : 0; }
  // top of step s (tap t): slab s + 1 - requested in step s + 1 - D, after that step's image request - has landed; younger: the requests
  // of steps s + 2 - D .. s - 1
  static constexpr int vmAt(int t) {
    int n = 0;
    for(int k = 1; k <= D - 2; k++) n += 1 + imgAt(t - k);
    return n;
  }
  // the D virtual steps -D .. -1 of the prologue issue what the steady state would: slab k in virtual step k - D, after an image request
  // where the steady state has one (pieces of image DIST, the last of them; its first pieces and all of images 0 .. DIST-1 go before)
  static constexpr int virtualImg() {
    int n = 0;
    for(int v = -D; v < 0; v++) n += imgAt(v);
    return n;
  }
};
// The slab waves never wait for an image by name. The image of chunk c + DIST is first read in the LAST step of chunk c + DIST - 1 (the
// fragments of a step are read behind the second half of the step before), i.e. after the barrier at the top of step 9 (c + DIST) - 1,
// where slab 9 (c + DIST) - requested in step 9 (c + DIST) - D - has landed: the image's last piece (tap NPA - 1 of chunk c) must not be
// requested later than that slab. (Round 4 ran a third depth - four steps / one chunk ahead - that broke this by one step: right under
// the emulator's latest-completion mode 1, in which the fetching waves run far ahead of the multiplying ones, 5 % faster on the MI355X and
// NOT bit-identical there; mode 2 of the emulator, which lands a copy at the barrier after its wait, now fails it. Deleted.)
template <bool PACK, int DEPTH>
constexpr bool imagesLandWithSlabs() { return NPA - 1 <= NT * SG<PACK, DEPTH>::DIST - SG<PACK, DEPTH>::D; }
static_assert(imagesLandWithSlabs<true, 0>() && imagesLandWithSlabs<false, 0>() && imagesLandWithSlabs<false, 1>(),
              "an image piece would be requested after the slab whose arrival stands for it");
static_assert(SG<false, 1>::LDS_BYTES <= 160 * 1024 && 2 * SG<true, 0>::LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
static_assert(SG<true, 0>::virtualImg() == 1 && SG<false, 1>::virtualImg() == 4, "prologue bookkeeping below");

// ---- REGW (round 5): the weights never pass through LDS ------------------------------------------------------------------------------
// In the shape above the four multiplying waves read, per k half, 4 x (1 KB of weight fragment + MTW KB of image fragments) from LDS -
// 16 KB = 128 cycles of the LDS port for 96 cycles of matrix work (MTW = 3), 8 KB = 64 cycles for 32 (MTW = 1) - the SAME weight fragment
// four times over, and nine barriers per chunk publish nothing but the next slab. Here every multiplying wave loads its weight fragments
// straight from global memory into registers (global_load_dwordx4 from ConvArgs::wFrag, the copy of the weights in MFMA-fragment order:
// a wave's load is 1 KB of consecutive bytes; the four waves' requests hit in the vector L1), a whole chunk ahead: a ring of 18 fragments
// = 72 registers, reloaded one k half after its use. What is left in LDS is the board image: three buffers (chunk c read, c + 1 published,
// c + 2 in flight), fetched by waves 4-7 as before. No slab ring, so nothing is published per step: ONE barrier per chunk instead of
// nine, the multiplying waves run a chunk's 18 k halves back to back with their image fragments read NSET - 1 k halves ahead. Same MFMAs
// per output in the same K order (chunk, tap, k half), same epilogue: bit-identical to the other shapes. What it bought and what the
// cycle stamps then showed a small launch's time to be made of: DESIGN.md 4.14, profiles/r05_steps/regw.
struct RWG {
  static constexpr int D = 0, NSW = 0;
  static constexpr int NSA = 3, DIST = 2;
  static_assert((DIST + 1) * KCHUNK * 2 <= DEVBUF_TAIL_BYTES, "the image requests' run-ahead must stay inside the readable tail of a DevBuf");
  static constexpr int RING_OFFSET = NSA * ACT_BYTES;
  static constexpr int SLACK_OFFSET = RING_OFFSET;
  static constexpr int MASK_OFFSET = SLACK_OFFSET + SLACK_BYTES;
  static constexpr int PARAM_OFFSET = MASK_OFFSET + MASK_BYTES;
  static constexpr int LDS_BYTES = PARAM_OFFSET + PARAM_BYTES;
  static constexpr int NHS = 2 * NT;  // k halves per chunk = weight fragments in the register ring
};
static_assert(RWG::LDS_BYTES <= 160 * 1024, "LDS budget exceeded");

// REGW: a weight fragment from global memory into registers (the lane's 16 bytes at uniform base + lane offset), and the wait before
// its use - hand-written, see the kernel. The wait takes the fragment as an in/out operand: no use can be scheduled ahead of it.
template <class V8>
__device__ __forceinline__ void gloadFrag(V8& dst, const char* base, unsigned off) {
#if defined(__AMDGCN__)
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(off), "s"(base));
#else
  dst = *(const V8*)(base + off);  // (the CPU emulation enters it in the wave's in-order queue: tests/test_engine_emulated.py)
#endif
}
template <int N, class V8>
__device__ __forceinline__ void waitFrag(V8& frag) {
#if defined(__AMDGCN__)
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(frag) : "n"(N));
#else
  (void)frag;  // (the CPU emulation: emu::waitVm(N))
#endif
}
// ... and the same for the image fragments out of LDS (ds_read_b128; a wave's LDS reads return in order): hipcc counts its own reads
// right (s_waitcnt lgkmcnt(5)) until the loop holds an inline-asm statement - with the loads above in the loop it drained the queue
// (lgkmcnt(0)) every other k half, i.e. waited out a whole LDS round trip there.
template <class V8>
__device__ __forceinline__ void ldsReadFrag(V8& dst, unsigned addr) {
#if defined(__AMDGCN__)
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
#else
  dst = *(const __attribute__((address_space(3))) V8*)addr;
#endif
}
template <int N, class V8>
__device__ __forceinline__ void waitLdsFrag(V8& frag) {
#if defined(__AMDGCN__)
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N));
#else
  (void)frag;
#endif
}
// ... for a count that is a constant only after the k-half loop is unrolled (the last chunk's reads count down, see the kernel)
template <class V8>
__device__ __forceinline__ void waitLdsFragSel(V8& frag, int n) {
  switch(n) {
    case 0: waitLdsFrag<0>(frag); break;
    case 1: waitLdsFrag<1>(frag); break;
    case 2: waitLdsFrag<2>(frag); break;
    case 3: waitLdsFrag<3>(frag); break;
    case 4: waitLdsFrag<4>(frag); break;
    case 5: waitLdsFrag<5>(frag); break;
    case 6: waitLdsFrag<6>(frag); break;
    case 7: waitLdsFrag<7>(frag); break;
    case 8: waitLdsFrag<8>(frag); break;
    case 9: waitLdsFrag<9>(frag); break;
    case 10: waitLdsFrag<10>(frag); break;
    case 11: waitLdsFrag<11>(frag); break;
    default: waitLdsFrag<0>(frag); break;  // stricter than needed, never wrong
  }
}
template <class V8>
__device__ __forceinline__ void waitFragSel(V8& frag, int n) {
  switch(n) {
    case 0: waitFrag<0>(frag); break;
    case 1: waitFrag<1>(frag); break;
    case 2: waitFrag<2>(frag); break;
    case 3: waitFrag<3>(frag); break;
    case 4: waitFrag<4>(frag); break;
    case 5: waitFrag<5>(frag); break;
    case 6: waitFrag<6>(frag); break;
    case 7: waitFrag<7>(frag); break;
    case 8: waitFrag<8>(frag); break;
    case 9: waitFrag<9>(frag); break;
    case 10: waitFrag<10>(frag); break;
    case 11: waitFrag<11>(frag); break;
    case 12: waitFrag<12>(frag); break;
    case 13: waitFrag<13>(frag); break;
    case 14: waitFrag<14>(frag); break;
    case 15: waitFrag<15>(frag); break;
    case 16: waitFrag<16>(frag); break;
    case 17: waitFrag<17>(frag); break;  // (never asked for by the kernel: the count its tests' deliberately wrong variant asks for)
    default: waitFrag<0>(frag); break;  // stricter than needed, never wrong
  }
}

// PACK (the second instantiation): register allocation capped at 128 per lane so that TWO work-groups share a CU (4 waves per SIMD, 2 x 68 KB
// of LDS) - for batches whose work-groups outnumber the CUs; the cap costs 48 bytes of scratch per lane outside the loop.
template <class TR, bool PACK, int DEPTH, int MTW, bool REGW, int WN, bool TIMING>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(PACK ? 4 : 2, PACK ? 4 : REGW ? 2 : 3))) void convSmallKernel(const ConvArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  typedef typename TR::V4 V4;
  extern __shared__ __attribute__((aligned(256))) char smemSmall[];
  const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smemSmall;
  static_assert(!REGW || (!PACK && DEPTH == 1), "the register-weights shape is one instantiation per (MTW, WN)");
  static_assert(WN == 1 || (REGW && WN == 2 && MTW == MT), "two channel tiles per wave exist for the unsplit register-weights shape only");
  // EARLY (the register-weights shapes): what a launch waits for first goes out first. In their first form the first image request left its
  // wave ~560 instructions into the kernel (seven pieces' source offsets, a division by the halo width each, were computed before the first
  // request), the first weight fragment ~250 (behind the cell bookkeeping of the epilogue), the first residual tile only after the loop.
  // Now an image piece is requested as soon as ITS offset is known (the division is a multiplication by a 16-bit reciprocal, exact below
  // 441 x 21), the fragments go out at the top of a multiplying wave, the first residual tile before the last chunk. Measured A/B on one
  // box (profiles/r05_steps/regw/call3_*): 12.32 -> 12.19 us per 3x3 launch at batch 8, 17.66 -> 17.38 at 32, 25.64 -> 25.42 at 64 - a
  // launch's fixed time is not its instruction count. The first form is deleted.
  constexpr bool EARLY = REGW;
  // TIMING (conv_bench.hip only, register-weights shapes): cycle stamps of one work-group's waves into ConvArgs::dbg, eight per wave -
  // multiplying waves: [0] start -> ready to wait, [1] the wait for the first image / fragments / parameters, [2] sum of the waits at the
  // chunk barriers, [3] sum of the chunks' 18 k halves, [4] epilogue, [7] whole kernel; fetching waves: [0] start -> the first two images
  // requested, [1] the wait for image 0, [2] sum of the waits for an image, [3] sum of the waits at the chunk barriers, [7] whole kernel
  static_assert(!TIMING || REGW, "");
  const unsigned long long tk0 = TIMING ? __builtin_readcyclecounter() : 0;
  unsigned long long tkA = 0, tkB = 0, tkC = 0, tkD = 0;
  auto stampOut = [&](unsigned long long t4) {
    if(TIMING && a.dbg != nullptr && (threadIdx.x & 63) == 0 && blockIdx.x == 0 && (int)blockIdx.y == a.N / 2 && blockIdx.z == 0) {
      unsigned long long* d = a.dbg + (threadIdx.x >> 6) * 8;
      d[0] = tkA; d[1] = tkB; d[2] = tkC; d[3] = tkD; d[4] = t4; d[7] = __builtin_readcyclecounter() - tk0;
    }
  };
  constexpr int NTILEW = NTILE * WN;  // output channels per work-group
  typedef std::conditional_t<REGW, RWG, SG<PACK, DEPTH>> G;
  constexpr int D = G::D, NSW = G::NSW, NSA = G::NSA, DIST = G::DIST;
  constexpr int MASK_OFFSET = G::MASK_OFFSET, PARAM_OFFSET = G::PARAM_OFFSET;
  const unsigned bufW = ldsBase + G::RING_OFFSET;
  const unsigned slack = ldsBase + G::SLACK_OFFSET;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= NCOMPUTE;  // wave-uniform
  const int lw = wave - NCOMPUTE;        // index among the loader waves
  const int wm = wave;                   // cell-tile group of a multiplying wave
  const int n = blockIdx.y;
  const int cout0 = blockIdx.x * NTILEW;
  const int X = a.X, Y = a.Y, S = X * Y;
  const int W2 = X + 2 * HALO, HP = W2 * (Y + 2 * HALO);
  const int inC = a.inC;
  const int nChunks = a.nChunks;
  const int nSteps = nChunks * NT;
  const char* const zero = (const char*)a.zeroPage;
  // GEMM column -> board cell, lane -> position inside a tile: as conv_kernel.h
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
  const int myPos = posOf(lane & 31);

  // ---- prologue, every wave: the board's mask; waves 0-3: the tile's parameters ----
  {
    const int cellIdx = wave * 64 + lane;
    dma4(cellIdx < S ? (const void*)(a.mask + (size_t)n * S + cellIdx) : (const void*)zero, ldsBase + MASK_OFFSET + wave * 256);
  }
  if(!loader) {
    const int idx = wave * 64 + lane;  // 0..255: scale [0,32) | bias [64,96) | per-board bias [128,160)
    const int arr = idx >> 6, c = idx & 63;
    const float* psrc = (const float*)zero;
    if(c < NTILEW) {
      if(arr == 0) psrc = a.scale + cout0 + c;
      else if(arr == 1) psrc = a.bias + cout0 + c;
      else if(arr == 2 && a.ncBias != nullptr) psrc = a.ncBias + (size_t)n * a.ncBiasStride + cout0 + c;
    }
    dma4(psrc, ldsBase + PARAM_OFFSET + wave * 256);
  }

  if(loader) {
    // =================================================== the fetching waves ===================================================
    const char* const inBoard = (const char*)a.in + (size_t)n * S * inC * sizeof(T);
    const char* const wBase = (const char*)a.w + (size_t)cout0 * ROWB;
    const size_t wSlabStride = (size_t)a.coutPad * ROWB;
    unsigned srcOff[NPA];  // byte offset from this board's tensor, or (bit 31) the zero page; + 64 bytes per chunk
    if constexpr(!EARLY) {
#pragma unroll
    for(int j = 0; j < NPA; j++) {
      const int p = (j * NLOAD + lw) * 64 + lane;
      const int hp = p >> 2;
      const int slot = (p & 3) ^ ((hp >> 2) & 3);
      unsigned off = 0x80000000u;
      if(hp < HP) {
        const int hy = hp / W2, hx = hp - hy * W2;
        const int y = hy - HALO, x = hx - HALO;
        if(y >= 0 && y < Y && x >= 0 && x < X) off = (unsigned)(((y * X + x) * inC + slot * 8) * (int)sizeof(T));
      }
      srcOff[j] = off;
    }
    }
    auto issueA = [&](int chunk, int j) {  // request j of the image of `chunk`; the pointer then moves on to the next chunk
      const bool live = chunk < nChunks;
      const unsigned off = srcOff[j];
      const char* src = (off & 0x80000000u) ? zero : inBoard + off;
      dma16(src, live ? ldsBase + (unsigned)(chunk % NSA) * ACT_BYTES + (unsigned)((j * NLOAD + lw) * 64) * 16u : slack);
      srcOff[j] = (off & 0x80000000u) ? off : off + KCHUNK * (unsigned)sizeof(T);
    };
    if constexpr(REGW) {
      // Images only, a whole image per chunk: image c + 2 is requested behind the barrier at the top of chunk c (every multiplying wave is
      // then done with image c - 1, whose buffer it takes) and waited for at the top of chunk c + 1, whose barrier publishes it - one chunk
      // before its first read (the multiplying waves read NSET - 1 k halves ahead, across the chunk boundary). Everything this wave has
      // in flight at a wait is one image: the counts are 0 and NPA.
      {
        const unsigned invW2 = (65536u + (unsigned)W2 - 1u) / (unsigned)W2;  // hp / W2 == hp * invW2 >> 16 for hp < 441, W2 <= 21
#pragma unroll
        for(int j = 0; j < NPA; j++) {
          const int p = (j * NLOAD + lw) * 64 + lane;
          const int hp = p >> 2;
          const int slot = (p & 3) ^ ((hp >> 2) & 3);
          unsigned off = 0x80000000u;
          if(hp < HP) {
            const int hy = (int)(((unsigned)hp * invW2) >> 16), hx = hp - hy * W2;
            const int y = hy - HALO, x = hx - HALO;
            if(y >= 0 && y < Y && x >= 0 && x < X) off = (unsigned)(((y * X + x) * inC + slot * 8) * (int)sizeof(T));
          }
          srcOff[j] = off;
          issueA(0, j);
        }
#pragma unroll
        for(int c = 1; c < DIST; c++)
#pragma unroll
          for(int j = 0; j < NPA; j++) issueA(c, j);
      }
      unsigned long long tq = 0;
      if(TIMING) { tq = __builtin_readcyclecounter(); tkA = tq - tk0; }
      waitVm<(DIST - 1) * NPA>();  // image 0 (and, older, this wave's piece of the mask) has landed; image 1 is in flight
      if(TIMING) tkB = __builtin_readcyclecounter() - tq;
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for(int chunk = 0; chunk < nChunks; chunk++) {
        if(TIMING) tq = __builtin_readcyclecounter();
        waitVm<0>();  // image chunk + 1
        if(TIMING) { const unsigned long long t = __builtin_readcyclecounter(); tkC += t - tq; tq = t; }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if(TIMING) tkD += __builtin_readcyclecounter() - tq;
#pragma unroll
        for(int j = 0; j < NPA; j++) issueA(chunk + DIST, j);
      }
      waitVm<0>();  // requests past the end went to the slack area: they must land before the LDS is released
      stampOut(0);
      return;
    }
    else {
    const unsigned wOff = (unsigned)(lw * 64 + lane) * 16u;  // waves 4 and 5: their KiB of a slab
    const bool slabWave = lw < 2;
    auto issueW = [&](int step) {  // one request (waves 4, 5)
      const bool live = step < nSteps;
      const char* slab = wBase + (size_t)(live ? step : 0) * wSlabStride;
      dma16(slab + wOff, live ? bufW + (unsigned)(step % (NSW > 0 ? NSW : 1)) * W_BYTES + (unsigned)lw * 1024u : slack);
    };
    // Fill the pipeline in the order the steady state would have: the D virtual steps -D .. -1 stand for the last D taps of a "chunk -1",
    // whose image requests would have been for image DIST - 1. So: images 0 .. DIST-2 whole, the first pieces of image DIST - 1, then per
    // virtual step the next piece of it where the steady state has an image request, followed by the step's slab (slab k in virtual step
    // k - D) - and the loop's compile-time wait counts hold from its first step on.
    constexpr int VIMG = G::virtualImg();  // pieces of image DIST - 1 that ride the virtual steps
#pragma unroll
    for(int c = 0; c + 1 < DIST; c++)
#pragma unroll
      for(int j = 0; j < NPA; j++) issueA(c, j);
#pragma unroll
    for(int j = 0; j < NPA - VIMG; j++) issueA(DIST - 1, j);
    {
      int piece = NPA - VIMG;
#pragma unroll
      for(int v = -D; v < 0; v++) {
        if(G::imgAt(v)) issueA(DIST - 1, piece++);
        if(slabWave) issueW(v + D);
      }
    }
    // slab 0 and image 0 (and the mask) have landed; in flight at most what was issued after slab 0 / after image 0
    // (one chunk ahead, D = 3: the last piece of image 0 rides virtual step -D, before slab 0)
    if(slabWave) waitVm<(D - 1) + VIMG - G::imgAt(-D)>();
    else waitVm<(DIST - 1) * NPA>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int step = 0;
    for(int chunk = 0; chunk < nChunks; chunk++) {
#pragma unroll
      for(int t = 0; t < NT; t++, step++) {
        // top of step s: slab s + 1 has landed - and with it every image request issued before it (G::vmAt). A wave that fetches no slabs
        // waits where the next chunk's image is first read, the last tap: for everything but the pieces of this chunk's own requests
        // (PACK: those ARE the next chunk's image)
        if(slabWave) convk::waitVmSel(G::vmAt(t));
        else if(t == NT - 1) waitVm<(DIST - 1) * NPA>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if(t < NPA) issueA(chunk + DIST, t);
        if(slabWave) issueW(step + D);
      }
    }
    waitVm<0>();  // requests past the end went to the slack area: they must land before the LDS is released
    return;
    }  // !REGW
  }

  // ===================================================== the multiplying waves =====================================================
  // REGW: the ring of weight fragments and its loads (described at the loop); EARLY: the first R - 1 k halves are requested here, before
  // the wave's cell bookkeeping
  constexpr int R = !REGW ? 1 : WN == 1 ? RWG::NHS : RWG::NHS / 2;
  V8 wf[R][WN];
  const char* const wTile = (const char*)a.wFrag + (size_t)cout0 * ROWB;
  const size_t wSlabStrideRw = (size_t)a.coutPad * ROWB;
  const unsigned wOffLane[2] = {(unsigned)lane * 16u, 1024u + (unsigned)lane * 16u};  // k half kk of a tile's 2 KB: 1 KB of 64 lanes x 16 bytes
  auto loadW = [&](int slot, const char* chunkBase, int hs) {  // the WN fragments of k half hs of the chunk at chunkBase
#pragma unroll
    for(int wn = 0; wn < WN; wn++) gloadFrag(wf[slot][wn], chunkBase + (size_t)(hs >> 1) * wSlabStrideRw + (size_t)(wn * NTILE * ROWB), wOffLane[hs & 1]);
  };
  if constexpr(REGW) {
#pragma unroll
    for(int hs = 0; hs + 1 < R; hs++) loadW(hs, wTile, hs);
  }
  const unsigned khalf = lane >> 5;
  const unsigned wXor = (lane >> 2) & 3;
  unsigned wLane[2];
#pragma unroll
  for(int kk = 0; kk < 2; kk++) wLane[kk] = bufW + (unsigned)(lane & 31) * ROWB + (((kk * 2 + khalf) ^ wXor) << 4);
  // MTW = 3: the wave's three cell tiles; MTW = 1 (grid z = 3): tile blockIdx.z of them - three work-groups share a board x 32 channels,
  // each fetches the whole image and slabs and does a third of the matrix work (see launchSmall)
  // MTW = 2 (register-weights only, cfg 125): the cell tiles over TWO work-groups - grid z = 0: tiles 0 and 1 of each wave, z = 1: tile 2
  // (its second "tile" lies past the wave's three: multiplied like any other, never stored) - for the batches between the three-way
  // split and one work-group per board x 32 channels
  static_assert(MTW != 2 || (REGW && WN == 1), "");
  const int pt0 = MTW == MT ? 0 : MTW == 2 ? 2 * (int)blockIdx.z : (int)blockIdx.z;
  unsigned aRow4[MTW];
  int cellOfTile[MTW];
#pragma unroll
  for(int pt = 0; pt < MTW; pt++) {
    int j = wm * (32 * MT) + (pt0 + pt) * 32 + myPos;
    j = cellOf(j < S ? j : S - 1);
    cellOfTile[pt] = j;
    const int y = j / X, x = j - y * X;
    aRow4[pt] = (unsigned)((y + HALO) * W2 + (x + HALO)) << 2;
  }
  const unsigned c40 = khalf << 4;
  const bool waveActive = wm * (32 * MT) + pt0 * 32 < S;
  auto ldsV8 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) V8*)addr; };
  auto ldsF4 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) f32x4*)addr; };
  auto ldsF1 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) float*)addr; };

  f32x16 acc[WN * MTW];  // [channel tile][cell tile]
#pragma unroll
  for(int pt = 0; pt < WN * MTW; pt++)
#pragma unroll
    for(int r = 0; r < 16; r++) acc[pt][r] = 0.0f;

  // the residual of (channel tile, cell tile) q = channel tile * MTW + cell tile, as the epilogue wants it (two 16-byte pieces per lane)
  auto loadResid = [&](int q, u32x4 (&dst)[2]) {
    const T* const rrow = (const T*)a.resid + ((size_t)n * S + cellOfTile[q % MTW]) * a.residC - a.rawBegin;
#pragma unroll
    for(int j = 0; j < 2; j++) {
      const int c = cout0 + (q / MTW) * 32 + 16 * j + 8 * khalf;
      const T* src = (c >= a.rawBegin && c < a.rawEnd) ? rrow + c : (const T*)zero;
      dst[j] = *(const u32x4*)src;
    }
  };
  u32x4 rqFirst[2];  // EARLY: tile 0's residual, requested before the last chunk

  if constexpr(REGW) {
    // ---- weights in registers: a chunk's 18 fragments a whole chunk ahead, one barrier per chunk ----
    constexpr int NHS = RWG::NHS;
    constexpr int NSET = MTW < MT ? 6 : 3;  // image-fragment register sets; the reads run NSET - 1 k halves ahead of their MFMAs
    static_assert(NHS % NSET == 0, "the set of a k half must be a compile-time index");
    static_assert(MTW * (NSET - 1) - 1 <= 11, "waitLdsFragSel knows counts up to 11");
    // the ring of weight fragments: R k halves of WN fragments each. One channel tile per wave: a whole chunk (18 x 4 registers); two: half
    // a chunk (9 x 8 registers, and a k half is twice as long)
    static_assert(NHS % R == 0 && (R - 2) * WN <= 17, "ring slots are compile-time indices; waitFragSel knows counts up to 17");
    // from the copy in fragment order (ConvArgs::wFrag): a wave's load is 1 KB of consecutive bytes, lane l at + 16 l. (Read from the
    // slab rows of ConvArgs::w - 32 rows of 64 bytes, the lane's slot in each - the same load touched 64 separate 16-byte pieces of 16 cache
    // lines: the shape whose cell tiles are split over three work-groups was SLOWER than its slab-ring twin that way, 13.5 against 12.4 us.)
    // The loads and their waits are hand-written: left to the compiler, a load whose use lies beyond the loop's back edge makes hipcc drain
    // the queue (s_waitcnt vmcnt(0)) at the top of every chunk - a memory round trip per chunk. A wave's loads return in order, WN
    // fragments are requested per k half, R - 1 k halves before their use: in the steady state the requests of R - 2 k halves are younger.
    const size_t wSlabStride = wSlabStrideRw;
    const char* wCur = wTile;  // this chunk's nine slabs
    // (k halves 0 .. R - 2 were requested at the top of the wave; k half R - 1 follows in the first k half: the ring's rule below)
    unsigned long long tq = 0;
    if(TIMING) { tq = __builtin_readcyclecounter(); tkA = tq - tk0; }
    waitVm<0>();  // this wave's mask and parameter requests (and, younger, the fragments - which the first MFMA needs anyway)
    __builtin_amdgcn_s_barrier();  // image 0, the mask and the parameters are published
    asm volatile("" ::: "memory");
    if(TIMING) tkB = __builtin_readcyclecounter() - tq;
    V8 af[NSET][MTW];
    // address of the image fragment of k half `hs` (0 .. NHS-1, or past the end: the next chunk's) of the chunk whose buffer is curA
    auto fragAddr = [&](int hs, unsigned curA, unsigned nextA, int pt) {
      const int h = hs < NHS ? hs : hs - NHS;
      const int t = h >> 1;
      const unsigned sTap = (unsigned)(((t / 3 - HALO) * W2 + (t % 3 - HALO)) * 4) + ((ldsBase + (hs < NHS ? curA : nextA)) >> 4);
      const unsigned q4 = aRow4[pt] + sTap;
      return ((q4 << 4) | ((q4 ^ c40) & 0x30u)) ^ ((h & 1) ? 0x20u : 0u);
    };
#pragma unroll
    for(int hs = 0; hs < NSET - 1; hs++)
#pragma unroll
      for(int pt = 0; pt < MTW; pt++) ldsReadFrag(af[hs][pt], fragAddr(hs, 0u, ACT_BYTES, pt));
    int chunk = 0;
    // one chunk: the barrier, then 18 k halves. LAST: the last chunk requests nothing beyond its own fragments (nothing stays in flight into
    // the epilogue, whose registers a late fragment would overwrite) and its waits count down with what is left in flight.
    // ROUND 6: that rule now holds for the IMAGE fragments too. Until then the last chunk went on reading NSET - 1 k halves ahead - the
    // fragments of a chunk that does not exist, never used and therefore never waited for. Their destination registers are dead to the
    // compiler: it gave them to the epilogue (the cfg 125 build: v[40:43], the row pointer of the residual prefetch), and when the LDS
    // data landed there AFTER the pointer had been computed the load went to an address made of image bits: the GPU exception
    // (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION) that killed production self-play in round 5's driver run - only under contention,
    // when LDS returns late enough: a pass alone on the chip never showed it, and the CPU emulation completes a read at once (DESIGN.md 0e;
    // tools/check_async_loads.py now walks the shipped code objects for exactly this and is a test).
    auto chunkBody = [&](auto lastTag) {
      constexpr bool LAST = decltype(lastTag)::value != 0;
      const unsigned curA = (unsigned)(chunk % NSA) * ACT_BYTES;
      const unsigned nextA = (unsigned)((chunk + 1) % NSA) * ACT_BYTES;
      const char* const wNext = wCur + NT * wSlabStride;
      if(TIMING) tq = __builtin_readcyclecounter();
      __builtin_amdgcn_s_barrier();  // publishes image chunk + 1; behind it the fetching waves overwrite image chunk - 1
      asm volatile("" ::: "memory");
      if(TIMING) { const unsigned long long t = __builtin_readcyclecounter(); tkC += t - tq; tq = t; }
#pragma unroll
      for(int hs = 0; hs < NHS; hs++) {
        // k half hs: its MFMAs; behind them, one by one, the image fragments of k half hs + NSET - 1 into the set the PREVIOUS k half
        // used, and the weight fragments of k half hs + R - 1 (this chunk's, or the next one's) into the ring slot the previous k half used
        const int setR = (hs + NSET - 1) % NSET;
        const int slot = hs % R;
        // k halves whose requests are younger than this one's: R - 2 in the steady state; in the last chunk only those that lie in the chunk
        const int younger = (!LAST || NHS - 1 - hs > R - 2) ? R - 2 : NHS - 1 - hs;
#pragma unroll
        for(int wn = 0; wn < WN; wn++) waitFragSel(wf[slot][wn], younger * WN);
        // LAST: no image fragment is read beyond the chunk's own
        const bool readAhead = !LAST || hs + NSET - 1 < NHS;
#pragma unroll
        for(int pt = 0; pt < MTW; pt++) {
          // an image fragment is read NSET - 1 k halves before its use, MTW reads per k half, in the order (k half, cell tile): at the use of
          // (hs, pt) the younger reads are those up to (hs + NSET - 1, pt - 1): MTW (NSET - 1) - 1 - in the last chunk at most what is left
          // of the chunk, (NHS - 1 - hs) MTW + (MTW - 1 - pt): its last use waits for everything
          constexpr int STEADY = MTW * (NSET - 1) - 1;
          const int left = (NHS - 1 - hs) * MTW + (MTW - 1 - pt);
          waitLdsFragSel(af[hs % NSET][pt], (!LAST || left > STEADY) ? STEADY : left);
#pragma unroll
          for(int wn = 0; wn < WN; wn++) {
            acc[wn * MTW + pt] = TR::mfma(wf[slot][wn], af[hs % NSET][pt], acc[wn * MTW + pt]);
            __builtin_amdgcn_sched_barrier(0);
            if(wn == 0) {
              if(pt == 0) {
                const int f = hs + R - 1;  // the k half whose fragments are requested now
                if(f < NHS) loadW(f % R, wCur, f);
                else if(!LAST) loadW(f % R, wNext, f - NHS);
              }
              if(readAhead) ldsReadFrag(af[setR][pt], fragAddr(hs + NSET - 1, curA, nextA, pt));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      if(TIMING) tkD += __builtin_readcyclecounter() - tq;
      wCur = wNext;
      chunk++;
    };
    while(chunk + 1 < nChunks) chunkBody(ActKindTag<0>());
    // (plain loads, older than everything the last chunk requests: the chunk's hand-written waits may only become stricter by them)
    if(EARLY && a.resid != nullptr) loadResid(0, rqFirst);
    chunkBody(ActKindTag<1>());
  }
  else {
  waitVm<0>();  // this wave's mask and parameter requests
  __builtin_amdgcn_s_barrier();  // slab 0 and image 0 are published
  asm volatile("" ::: "memory");
  V8 wf[2], af[2][MTW];
  unsigned aAddr[MTW];
  // fragments of step 0, k half 0
  {
    wf[0] = ldsV8(wLane[0]);
    unsigned sTap = (unsigned)(((0 - HALO) * W2 + (0 - HALO)) * 4) + (ldsBase >> 4);
    asm volatile("" : "+s"(sTap));
#pragma unroll
    for(int pt = 0; pt < MTW; pt++) {
      const unsigned q4 = aRow4[pt] + sTap;
      aAddr[pt] = (q4 << 4) | ((q4 ^ c40) & 0x30u);
      af[0][pt] = ldsV8(aAddr[pt]);
    }
  }
  int step = 0;
  for(int chunk = 0; chunk < nChunks; chunk++) {
    const unsigned curA = (unsigned)(chunk % NSA) * ACT_BYTES;
    const unsigned nextA = (unsigned)((chunk + 1) % NSA) * ACT_BYTES;
#pragma unroll
    for(int t = 0; t < NT; t++, step++) {
      __builtin_amdgcn_s_barrier();  // publishes slab step + 1 (and, at the last tap, the next chunk's image)
      asm volatile("" ::: "memory");
      // (Round 4 also measured the fragments read a WHOLE step ahead into a second register set - slower: 18.7 against 17.3 us per launch at
  // batch 1, as the same idea was for the twelve-wave shape; profiles/r04_steps/small_batch.)
  // first k half: the MFMAs of fragment set 0; behind them, one by one, the reads of set 1 (this step's slab, this tap)
      const unsigned wb1 = wLane[1] + (unsigned)(step % NSW) * W_BYTES;
#pragma unroll
      for(int pt = 0; pt < MTW; pt++) {
        acc[pt] = TR::mfma(wf[0], af[0][pt], acc[pt]);
        __builtin_amdgcn_sched_barrier(0);
        if(pt == 0) wf[1] = ldsV8(wb1);
        af[1][pt] = ldsV8(aAddr[pt] ^ 0x20u);
        __builtin_amdgcn_sched_barrier(0);
      }
      // second k half; behind its MFMAs the reads of set 0 of the NEXT step (slab step + 1, next tap)
      const unsigned wb0 = wLane[0] + (unsigned)((step + 1) % NSW) * W_BYTES;
      {
        const int tn = t + 1 < NT ? t + 1 : 0;
        unsigned sTap = (unsigned)(((tn / 3 - HALO) * W2 + (tn % 3 - HALO)) * 4) + ((ldsBase + (t + 1 < NT ? curA : nextA)) >> 4);
        asm volatile("" : "+s"(sTap));
#pragma unroll
        for(int pt = 0; pt < MTW; pt++) {
          const unsigned q4 = aRow4[pt] + sTap;
          aAddr[pt] = (q4 << 4) | ((q4 ^ c40) & 0x30u);
        }
      }
#pragma unroll
      for(int pt = 0; pt < MTW; pt++) {
        acc[pt] = TR::mfma(wf[1], af[1][pt], acc[pt]);
        __builtin_amdgcn_sched_barrier(0);
        if(pt == 0) wf[0] = ldsV8(wb0);
        af[0][pt] = ldsV8(aAddr[pt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  }  // !REGW
  // ---- epilogue: conv_kernel.h's, for one 32-channel tile per wave and cell tile ----
  if(!waveActive) {
    stampOut(0);
    return;
  }
  const unsigned long long tEpi = TIMING ? __builtin_readcyclecounter() : 0;
  const unsigned maskAddr = ldsBase + MASK_OFFSET;
  const unsigned scAddr = ldsBase + PARAM_OFFSET, biAddr = scAddr + 64 * 4, nbAddr = biAddr + 64 * 4;
  const bool hasResid = a.resid != nullptr;
  const bool hasNb = a.ncBias != nullptr;
  const bool anyRaw = a.rawEnd > a.rawBegin, anyAct = a.actEnd > a.actBegin;
  T* const rawBoard = (T*)a.rawOut + (size_t)n * S * a.rawC - a.rawBegin;
  T* const actBoard = (T*)a.actOut + (size_t)n * S * a.actC - a.actBegin;
  T* const trash = (T*)((char*)const_cast<void*>(a.zeroPage) + ZERO_PAGE_BYTES) + lane * 8;
  auto epilogue = [&](auto kindTag, auto residTag) {
    constexpr int KIND = decltype(kindTag)::value;
    constexpr bool RESID = decltype(residTag)::value != 0;
    u32x4 rq[2][2];
    // q = channel tile * MTW + cell tile (one channel tile per wave everywhere but in the 64-channel register-weights shape)
    if(RESID) {
      if(EARLY) {
        rq[0][0] = rqFirst[0];
        rq[0][1] = rqFirst[1];
      }
      else loadResid(0, rq[0]);
    }
#pragma unroll
    for(int q = 0; q < WN * MTW; q++) {
      const int pt = q % MTW, ct0 = cout0 + (q / MTW) * 32;
      const int cellBase = wm * (32 * MT) + (pt0 + pt) * 32;
      if(MTW == 2 && pt0 + pt >= MT) break;  // wave-uniform: the tile past the wave's three
      // wave-uniform. (Two channel tiles per wave: a cell tile off the board is walked with every piece going to the trash area - the
      // residual requests run one (channel tile, cell tile) ahead in a fixed order.)
      if(WN == 1 && cellBase >= S) break;
      const bool live = cellBase + myPos < S;
      const int cell = cellOfTile[pt];
      const unsigned onBits = ldsF1(maskAddr + cell * 4) == 1.0f ? 0xffffffffu : 0u;
      T* const rawRow = rawBoard + (size_t)cell * a.rawC;
      T* const actRow = actBoard + (size_t)cell * a.actC;
      unsigned pOff = (unsigned)(4 * khalf) * 4u + (unsigned)(q / MTW) * 128u;
      asm volatile("" : "+v"(pOff));
      u32x2 rp[4], op[4];
      u32x2 resP[4];
      if(RESID) {
        if(q + 1 < WN * MTW) loadResid(q + 1, rq[(q + 1) & 1]);
        unpair(rq[q & 1], resP);
      }
#pragma unroll
      for(int g = 0; g < 4; g++) {
        const f32x4 sc = ldsF4(scAddr + pOff + 32 * g);
        const f32x4 bi = ldsF4(biAddr + pOff + 32 * g);
        f32x4 v;
#pragma unroll
        for(int i = 0; i < 4; i++) v[i] = acc[q][4 * g + i];
        if(hasNb) v += ldsF4(nbAddr + pOff + 32 * g);
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
      if(RESID || anyRaw) {
        u32x4 rawQ[2];
        pairUp(rp, rawQ);
#pragma unroll
        for(int j = 0; j < 2; j++) {
          const int c = ct0 + 16 * j + 8 * khalf;
          T* const dst = (live && c >= a.rawBegin && c < a.rawEnd) ? rawRow + c : trash;
          *(u32x4*)dst = rawQ[j];
        }
      }
      if(RESID || anyAct) {
        u32x4 actQ[2];
        pairUp(op, actQ);
#pragma unroll
        for(int j = 0; j < 2; j++) {
          const int c = ct0 + 16 * j + 8 * khalf;
          T* const dst = (live && c >= a.actBegin && c < a.actEnd) ? actRow + c : trash;
          *(u32x4*)dst = actQ[j];
        }
      }
    }
  };
  withActKind(a.actKind, [&](auto kindTag) {
    if(hasResid) epilogue(kindTag, ActKindTag<1>());
    else epilogue(kindTag, ActKindTag<0>());
  });
  if(TIMING) stampOut(__builtin_readcyclecounter() - tEpi);
}

// MTW = 1: the cell tiles of a board x 32 channels over THREE work-groups (grid z) - for batches that leave most CUs idle (batch x
// channel tiles x 3 <= the CU count). A work-group's time at batch 1 is its 54 steps of six MFMAs and eight fragment reads per wave; with
// two MFMAs and four reads per step it is shorter, and three times as many CUs work. Every work-group still fetches the whole image
// and every slab (the fetching waves are unchanged): three times the L2 traffic, which is idle at these sizes. Outputs are computed by
// the same MFMAs in the same order: bit-identical.
template <class TR, bool PACK, int DEPTH, int MTW, bool REGW = false, int WN = 1, bool TIMING = false>
hipError_t launchSmall(const ConvArgs& a, hipStream_t stream) {
  if(a.coutPad % (NTILE * WN) != 0) return hipErrorInvalidValue;
  if(REGW && a.wFrag == nullptr) return hipErrorInvalidValue;  // the register-weights shapes read the copy in fragment order
  auto kern = convSmallKernel<TR, PACK, DEPTH, MTW, REGW, WN, TIMING>;
  constexpr int LDS_BYTES = std::conditional_t<REGW, RWG, SG<PACK, DEPTH>>::LDS_BYTES;
  constexpr int MAX_DEVICES = 64;  // the > 64 KiB LDS opt-in is per function AND device (conv_kernel.h launchOne)
  static std::atomic<bool> attrSet[MAX_DEVICES];
  int dev = 0;
  hipError_t de = hipGetDevice(&dev);
  if(de != hipSuccess) return de;
  if(dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
  if(!attrSet[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if(e != hipSuccess) return e;
    attrSet[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3(a.coutPad / (NTILE * WN), a.N, MTW == MT ? 1 : MTW == 2 ? 2 : MT), dim3(NTHREADS), LDS_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace smallk
}  // namespace kmx
#endif

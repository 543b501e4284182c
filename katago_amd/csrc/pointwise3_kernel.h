// pointwise3_kernel.h — the seam between two nested-bottleneck blocks (same arithmetic as pointwise_kernel.h, see there) with BOTH
// WEIGHT MATRICES RESIDENT ON THE CU: one wave per SIMD, 512 registers per lane, W1 and two thirds of W2 in the accumulator half of
// the register file, the last third of W2 in LDS - fetched ONCE per work-group, not once per tile.
//
// Why (round 5). pointwise2_kernel.h moves, per 128-cell tile, 343 KB of activations AND 294 KB of weights through
// the CU's memory pipe - W1 and W2 are consumed whole by every tile, LDS (160 KB) cannot hold them beside the tile, so they are
// re-fetched from L2 for every tile - at the ~12.5 B/cycle/CU that pipe sustains: 51 k cycles per tile of which the matrix cores need
// 14.6 k. A larger cell tile only halves the weight bytes per cell; keeping the weights ON the CU removes them:
//   * a work-group = 4 waves, ONE per SIMD (amdgpu_waves_per_eu(1, 1)): each wave owns 512 registers per lane, 256 of them AGPRs;
//   * MFMA A operands may be AGPRs (gfx90a and later): wave w keeps the W1 rows of ITS 96 trunk channels (3 tiles of 32 x K = 192:
//     36 fragments = 144 AGPRs) and the W2 rows of ITS mid tile w (32 channels x K = 384: 24 fragments = 96 AGPRs) - 240 AGPRs,
//     loaded once by plain global loads (which can target AGPRs directly) at the start of the persistent work-group;
//   * the two remaining mid tiles (4 and 5: 64 W2 rows, 48 KB) sit in LDS, also fetched once; tile 4 is multiplied by waves 0 / 1
//     (cell sub-tile 0 / 1), tile 5 by waves 2 / 3;
//   * hipcc does not put MFMA source operands of the BUILTIN into AGPRs under pressure (it spills, DESIGN.md 4.8: "hipcc will not put
//     this template's accumulators in AGPRs"), so the MFMA is an inline-asm statement whose weight operand carries the "a" constraint;
//     the wait states hipcc would pad around an MFMA are written out (mfmaSettle).
// A tile is 64 cells (2 sub-tiles of 32: with the weights resident the tile size no longer sets the weight traffic, and 64 cells let X
// be double-buffered beside the activated image). Per tile a wave issues 144 MFMAs (72 + 72) and ~1 800 vector instructions of
// epilogue arithmetic; the memory pipe moves 171 KB (X 24, residual 48, raw trunk 48, mid raw + act 48... per 64 cells) instead of 318.
//
// Flow per tile t of a work-group (tiles tile0, tile0 + grid, ...; `b` = t's X buffer):
//   top     request X(t + grid) -> buffer b^1 (LDS-DMA), the residual pieces of tile t (plain loads); wait for X(t); barrier B1
//   GEMM 1  D1[this wave's 96 channels][64 cells] over K = 192 from X[b]
//   epi 1   + residual, raw trunk -> HBM, act(bn(.)) mask -> the LDS image A2 (chunks 3w .. 3w+2, all 64 rows); barrier B2
//   GEMM 2  own mid tile w on both sub-tiles (weights in AGPRs) + the shared tile 4 + w/2 on sub-tile w%2 (weights in LDS), K = 384
//   epi 2   mid raw and act(bn'(.)) mask -> HBM
// Same MFMA, same operand roles (weights = A, cells = B), same K order (chunk, k half), same epilogue expressions and rounding points
// as pointwise_kernel.h / pointwise2_kernel.h / two launches of conv_kernel.h: BIT-IDENTICAL (tests/test_gpu_pointwise.py; emulated
// on the CPU in tests/test_engine_emulated.py).
//
// vmcnt bookkeeping as in pointwise2_kernel.h: every wave issues the same sequence of vector-memory operations per tile (requests
// that must not land go to a slack area / read the zero page / store to the trash area), so each wait is a compile-time count.
#ifndef KMX_POINTWISE3_KERNEL_H_
#define KMX_POINTWISE3_KERNEL_H_

#include <atomic>
#include <cstdlib>
#include <utility>

#include "device_common.h"

namespace kmx {
namespace pw3 {

#define GLOBAL __attribute__((address_space(1)))
constexpr int ROWB = WROW_HALFS * 2;  // 64-byte LDS rows (four 16-byte slots, slot s of row r stored at s ^ ((r>>2)&3))
constexpr int TM = 64;                // cells per tile: two sub-tiles of 32
constexpr int NSUB = TM / 32;
constexpr int NWAVES = 4;
constexpr int NTHREADS = NWAVES * 64;

template <int K1, int K2, int NT3>
struct Geom {
  static constexpr int C1 = 32 * K1, C2 = 32 * K2, C3 = 32 * NT3;
  static constexpr int T1 = K2 / NWAVES;   // trunk tiles (of 32 channels) per wave in GEMM 1
  static constexpr int NSH = NT3 - NWAVES; // mid tiles beyond one per wave: shared, their W2 rows in LDS
  static_assert(K2 % NWAVES == 0, "GEMM 1 deals whole 32-channel tiles to the four waves");
  static_assert(NSH == 2 && NSUB == 2, "two shared mid tiles, each multiplied by two waves on one cell sub-tile each");
  static_assert(T1 == 3, "mfmaSettle names three accumulators per phase");
  static constexpr int W_AGPRS = (T1 * K1 * 2 + K2 * 2) * 4;
  static_assert(W_AGPRS <= 256, "the resident weight fragments must fit the accumulator half of the register file");
  static constexpr int CHUNK_BYTES = TM * ROWB;
  static constexpr int X_OFF = 0, X_BUF = K1 * CHUNK_BYTES, X_BYTES = 2 * X_BUF;
  static constexpr int A2_OFF = X_OFF + X_BYTES, A2_BYTES = K2 * CHUNK_BYTES;
  static constexpr int WL_OFF = A2_OFF + A2_BYTES, WL_CHUNK = NSH * 32 * ROWB, WL_BYTES = K2 * WL_CHUNK;  // [chunk][64 rows of W2]
  static constexpr int PARAM_OFF = WL_OFF + WL_BYTES;                 // scale1, bias1 [C2], scale2, bias2 [C3]
  static constexpr int MASK_OFF = PARAM_OFF + (2 * C2 + 2 * C3) * 4;  // two tiles of TM floats
  static constexpr int SLACK_OFF = MASK_OFF + 2 * TM * 4;
  static constexpr int LDS_BYTES = SLACK_OFF + 1024;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
  static_assert(TM * 4 == NTHREADS, "one request round of the work-group fills exactly one chunk of the X tile");
  static_assert(WL_CHUNK == NWAVES * 1024, "one request round of the work-group fetches one chunk of the shared W2 rows");
  // vector-memory operations per wave and tile, in program order (the kernel's blocks):
  //   block 1  N_RS residual loads of sub-tile 0 | the previous tile's second epilogue of sub-tile 1: N_E2 stores, twice that on the waves
  //            whose shared mid tile sits on sub-tile 1, none in a work-group's first tile
  //   block 2  N_RS residual loads of sub-tile 1 | N_S1S raw-trunk stores of sub-tile 0      block 3  N_S1S raw-trunk stores of sub-tile 1
  //            | (the wait for the NEXT tile's X, then barrier BA1)
  //   block 4  N_X requests for the X of the tile after next | second epilogue of sub-tile 0: N_E2 stores, twice that on the waves whose
  //            shared mid tile sits on sub-tile 0
  static constexpr int N_X = 1 + K1;     // mask + one request per chunk of X
  static constexpr int N_RS = T1 * 2;    // residual loads of a sub-tile (16 bytes per lane each)
  static constexpr int N_S1S = T1 * 2;   // stores of epilogue 1 per sub-tile
  static constexpr int N_E2 = 2 * 2;     // stores of the second epilogue of one mid tile: (raw, act) x 2 pieces
  // At the wait for X(t + 1) (end of block 3 of tile t; requested in block 4 of tile t - 1, or by the prologue): the FEWEST operations
  // that can have been issued after its requests - a work-group's first tile: the residual loads and raw-trunk stores of blocks 1-3 and
  // nothing else. In the steady state the 3 N_E2 stores of the two second epilogues in between are waited for as well: the oldest
  // operations of most of a tile ago.
  static constexpr int VM_AFTER_X = N_RS + N_RS + N_S1S + N_S1S;
  static_assert(VM_AFTER_X <= 63, "s_waitcnt vmcnt has six bits");
};

template <int N>
__device__ __forceinline__ void waitVm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void waitLds() {
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

// v_mfma_f32_32x32x16 with the A operand (the weight fragment) in an AGPR quad. acc = W x (+ acc). hipcc treats the statement as one
// opaque instruction: it allocates the operands (the "a" constraint is what keeps 240 registers of weights in the accumulator file for
// the life of the work-group) and pads nothing - see mfmaSettle. On the CPU emulation (tests/fakehip) the builtin's stand-in runs.
template <class TR>
__device__ __forceinline__ void mfmaFirst(f32x16& acc, const typename TR::V8& w, const typename TR::V8& x) {
#if defined(__AMDGCN__)
  // (s_nop 1: the destination registers were last written by vector ALU instructions of the previous phase)
  if constexpr(TR::DT == DT_F16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(x));
  else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(x));
#else
  f32x16 z;
  for(int r = 0; r < 16; r++) z[r] = 0.0f;
  acc = TR::mfma(w, x, z);
#endif
}
template <class TR>
__device__ __forceinline__ void mfmaAcc(f32x16& acc, const typename TR::V8& w, const typename TR::V8& x) {
#if defined(__AMDGCN__)
  if constexpr(TR::DT == DT_F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
#else
  acc = TR::mfma(w, x, acc);
#endif
}
// the same with the weight fragment in VGPRs (the shared mid tile: its fragments come from LDS)
template <class TR>
__device__ __forceinline__ void mfmaFirstV(f32x16& acc, const typename TR::V8& w, const typename TR::V8& x) {
#if defined(__AMDGCN__)
  if constexpr(TR::DT == DT_F16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x));
  else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x));
#else
  mfmaFirst<TR>(acc, w, x);
#endif
}
template <class TR>
__device__ __forceinline__ void mfmaAccV(f32x16& acc, const typename TR::V8& w, const typename TR::V8& x) {
#if defined(__AMDGCN__)
  if constexpr(TR::DT == DT_F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
#else
  mfmaAcc<TR>(acc, w, x);
#endif
}
// An MFMA's result may be read by a vector ALU instruction only after the matrix pipe has written it back: 8 passes for 32x32x16
// = 12 wait states after the LAST MFMA into it (hipcc pads this for the builtin; nothing inside or after an asm statement). Sixteen
// states, once per GEMM phase; naming the accumulators keeps every reader below the statement.
__device__ __forceinline__ void mfmaSettle(f32x16& a0, f32x16& a1, f32x16& a2) {
#if defined(__AMDGCN__)
  asm volatile("s_nop 15" : "+v"(a0), "+v"(a1), "+v"(a2));
#else
  (void)a0; (void)a1; (void)a2;
#endif
}

// __builtin_amdgcn_sched_barrier mask: which instructions MAY cross - vector ALU (2), scalar ALU (4), transcendental (0x400); memory, LDS
// and inline-asm (the MFMAs) may not
constexpr int SCHED_ALU_ONLY = 0x2 | 0x4 | 0x400;
// compile-time loops: f(integral_constant<int, 0>) .. f(integral_constant<int, N - 1>) - every index a constant expression, so register
// arrays are indexed statically and `if constexpr` can pick a step's instructions
template <int... I, class F>
__device__ __forceinline__ void staticForImpl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void staticFor(F&& f) {
  staticForImpl(std::make_integer_sequence<int, N>(), f);
}
// NM matrix micro-steps spread evenly over NV pieces of vector work, in source order [steps of piece 0] piece 0 [steps of piece 1] ...:
// one wave per SIMD has nobody else to fill the matrix pipe while it does epilogue arithmetic, or the vector ALU while it multiplies -
// an MFMA runs for 32 cycles after its issue slot, and the independent vector instructions behind it issue meanwhile
template <int NM, int NV, class FM, class FV>
__device__ __forceinline__ void interleave(FM&& fm, FV&& fv) {
  staticFor<NV>([&](auto v) {
    constexpr int V = decltype(v)::value, M0 = NM * V / NV, M1 = NM * (V + 1) / NV;
    staticFor<M1 - M0>([&](auto m) { fm(std::integral_constant<int, M0 + decltype(m)::value>()); });
    fv(v);
    // pieces stay in source order (the scheduler would otherwise cluster the matrix steps) - but the two values of a PAIR may overlap:
    // a value is one dependent chain of ~12 vector instructions, two of them transcendental, and one wave per SIMD has only its own
    // instruction-level parallelism to cover their latencies. SCHED_ALU_ONLY lets vector / scalar ALU instructions cross, nothing else.
    if constexpr(V % 2 == 1) __builtin_amdgcn_sched_barrier(0);
    else __builtin_amdgcn_sched_barrier(SCHED_ALU_ONLY);
  });
}

// TIMING (conv_bench.hip only): s_memtime stamps between the blocks, summed per wave over the tiles of work-group 0 into a.dbg:
// [0] tile top (requests, wait for X, barrier BX)  [1] block 1  [2] block 2  [3] barrier BA0  [4] block 3  [5] barrier BA1  [6] block 4
// [7] the last tile's trailing epilogue  [8] total
template <class TR, int K1, int K2, int NT3, int KIND1, int KIND2, bool TIMING = false>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void pointwisePairResidentKernel(const PwPairArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  typedef typename TR::V4 V4;
  typedef Geom<K1, K2, NT3> G;
  constexpr int T1 = G::T1;

  extern __shared__ __attribute__((aligned(256))) char smemPw3[];
  const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smemPw3;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned khalf = lane >> 5;
  const int l31 = lane & 31;
  // position of lane l inside a 32-column tile (conv_kernel.h): the 16-lane groups of a ds_read_b128 read 16 consecutive rows
  const int myPos = l31 < 4 ? l31 : l31 < 12 ? l31 + 12 : l31 < 16 ? l31 - 8 : l31 < 20 ? l31 + 8 : l31 < 28 ? l31 - 12 : l31;
  const char* const zero = (const char*)a.zeroPage;
  T* const trash0 = (T*)((char*)const_cast<void*>(a.zeroPage) + ZERO_PAGE_BYTES);  // TRASH_BYTES of writable scratch
  const unsigned slack = ldsBase + G::SLACK_OFF;
  auto ldsV8 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) V8*)(size_t)addr; };
  auto ldsF1 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) float*)(size_t)addr; };

  const long long numTiles = (a.cells + TM - 1) / TM;
  const long long stride = gridDim.x;

  // ---- the resident weights: plain loads, once (rows slot-swizzled in HBM: logical slot s of row r at s ^ ((r>>2)&3)) ----
  const unsigned wXor = (lane >> 2) & 3;  // (r>>2)&3 of row 32 k + l31
  V8 w1f[T1][K1][2];  // this wave's trunk tiles T1 w + j: rows 32 (T1 w + j) + l31 of W1, per K chunk and k half
  V8 w2f[K2][2];      // mid tile w: rows 32 w + l31 of W2
#pragma unroll
  for(int j = 0; j < T1; j++)
#pragma unroll
    for(int c = 0; c < K1; c++)
#pragma unroll
      for(int kk = 0; kk < 2; kk++)
        w1f[j][c][kk] = *(const V8*)((const char*)a.w1 + ((size_t)c * G::C2 + 32 * (T1 * wave + j) + l31) * ROWB + (((2 * kk + khalf) ^ wXor) << 4));
#pragma unroll
  for(int c = 0; c < K2; c++)
#pragma unroll
    for(int kk = 0; kk < 2; kk++)
      w2f[c][kk] = *(const V8*)((const char*)a.w2 + ((size_t)c * G::C3 + 32 * wave + l31) * ROWB + (((2 * kk + khalf) ^ wXor) << 4));

  // ---- request helpers: every call issues a fixed number of instructions ----
  // N_X instructions: the mask tile (one 4-byte request per lane of wave 0) and the X tile, chunk j = round j of the work-group
  auto issueX = [&](long long tile, int parity, bool live) {
    const long long cell0 = tile * TM;
    {
      const bool m = live && wave == 0 && cell0 + lane < a.cells;
      dma4(m ? (const void*)(a.mask + cell0 + lane) : (const void*)zero, live && wave == 0 ? ldsBase + G::MASK_OFF + (unsigned)parity * (TM * 4) : slack);
    }
    const int p = wave * 64 + lane;  // piece of a chunk: row p/4, PHYSICAL slot p%4 = logical slot (p%4) ^ ((row>>2)&3)
    const int q = p >> 2;
    const int slot = (p & 3) ^ ((q >> 2) & 3);
    const bool rowLive = live && cell0 + q < a.cells;
    const char* src = rowLive ? (const char*)a.in + ((size_t)(cell0 + q) * a.inC + slot * 8) * sizeof(T) : zero;
#pragma unroll
    for(int j = 0; j < K1; j++)
      dma16(rowLive ? src + j * (KCHUNK * (int)sizeof(T)) : zero, live ? ldsBase + G::X_OFF + (unsigned)parity * G::X_BUF + j * G::CHUNK_BYTES + wave * 1024 : slack);
  };

  // ---- prologue: parameters by plain stores, the shared W2 rows and the first tile's X by LDS-DMA; everything landed before B1 ----
  {
    float* const p = (float*)(smemPw3 + G::PARAM_OFF);
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
  // rows 32 NWAVES .. C3-1 of W2 (the shared mid tiles), chunk by chunk: 4 KiB each, one request per wave
#pragma unroll
  for(int c = 0; c < K2; c++)
    dma16((const char*)a.w2 + ((size_t)c * G::C3 + 32 * NWAVES) * ROWB + wave * 1024 + lane * 16, ldsBase + G::WL_OFF + c * G::WL_CHUNK + wave * 1024);
  long long tile = blockIdx.x;
  issueX(tile, 0, true);
  issueX(tile + stride, 1, tile + stride < numTiles);
  waitVm<G::N_X>();  // all but the second tile's requests: weights, the shared W2 rows, the first tile's X and mask
  wgBarrier();       // ... for every wave (the parameters above are published too)
  int parity = 0;

  // per-lane LDS addresses (opaque, so that every use is "this register + a constant the instruction carries")
  unsigned xLane[NSUB][2], wlLane[2], a2wLane[NSUB][2];
  const unsigned shRow = (unsigned)(32 * (wave >> 1) + l31);  // this lane's row of the shared mid tile inside the LDS copy
#pragma unroll
  for(int kk = 0; kk < 2; kk++) {
    const unsigned ls = kk * 2 + khalf;
#pragma unroll
    for(int s = 0; s < NSUB; s++) {
      const unsigned cl = 32 * s + myPos;
      const unsigned xXor = (cl >> 2) & 3;
      xLane[s][kk] = ldsBase + cl * ROWB + ((ls ^ xXor) << 4);  // + X_OFF + buffer + chunk, or + A2_OFF + chunk
      a2wLane[s][kk] = ldsBase + G::A2_OFF + cl * ROWB + (((2 * kk + khalf) ^ xXor) << 4);  // piece j = kk of the lane pair; + chunk
      asm volatile("" : "+v"(xLane[s][kk]), "+v"(a2wLane[s][kk]));
    }
    wlLane[kk] = ldsBase + G::WL_OFF + shRow * ROWB + ((ls ^ wXor) << 4);  // + chunk
    asm volatile("" : "+v"(wlLane[kk]));
  }
  unsigned p1Lane = ldsBase + G::PARAM_OFF + (unsigned)(32 * T1 * wave + 4 * khalf) * 4u;  // scale1 of this wave's first trunk channel group
  unsigned p2Own = ldsBase + G::PARAM_OFF + 2 * G::C2 * 4 + (unsigned)(32 * wave + 4 * khalf) * 4u;
  unsigned p2Sh = ldsBase + G::PARAM_OFF + 2 * G::C2 * 4 + (unsigned)(32 * (NWAVES + (wave >> 1)) + 4 * khalf) * 4u;
  asm volatile("" : "+v"(p1Lane), "+v"(p2Own), "+v"(p2Sh));

  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tPrev = TIMING ? __builtin_readcyclecounter() : 0;
  const unsigned long long tStart = tPrev;
  auto stamp = [&](int which) {
    if(!TIMING) return;
    const unsigned long long now = __builtin_readcyclecounter();
    seg[which] += now - tPrev;
    tPrev = now;
  };

  // The whole tile loop is instantiated for SH = 0 and 1 (the sub-tile on which this wave multiplies its shared mid tile) under ONE
  // uniform branch: which MFMAs and which epilogue groups a block holds is then known at compile time.
  auto run = [&](auto shTag) {
    constexpr int SH = decltype(shTag)::value;
    constexpr int NE2_0 = SH == 0 ? 32 : 16, NE2_1 = SH == 1 ? 32 : 16;  // second-epilogue values per lane of sub-tile 0 / 1: own mid tile (+ the shared one)
    constexpr int NM1 = 2 * K1 * T1;                                    // MFMAs of GEMM 1 on one sub-tile
    constexpr int NM2_0 = SH == 0 ? 4 * K2 : 2 * K2, NM2_1 = SH == 1 ? 4 * K2 : 2 * K2;  // ... of GEMM 2 on sub-tile 0 / 1
    constexpr int NE1 = 16 * T1;                                        // first-epilogue values per lane of one sub-tile
    f32x16 acc1[NSUB][T1], acc2[NSUB], accS;
    u32x4 rq[NSUB][T1][2];
    bool live[NSUB] = {false, false};
    unsigned onBits[NSUB] = {0u, 0u};
    // sub-tile 1 of the PREVIOUS tile: its second epilogue runs beside the first GEMM block of this tile
    bool havePrev = false, prevLive1 = false;
    unsigned prevOn1 = 0u;
    long long prevCell0 = 0;
    V8 xfA, wlF;  // the fragments of the next k-step (activations; the shared tile's weights), read one step ahead
    u32x2 rp[4], op[4], resP[4];  // an epilogue tile in progress: its sixteen values fill them, the last regroups and stores

    // ---- matrix micro-steps: ONE MFMA each (a k-step I = 2 chunk + k half is T1 of them in GEMM 1, one or two in GEMM 2); the fragment
    // of the next k-step is read behind the first MFMA of this one (hipcc drains every LDS read before an asm that uses one) ----
    V8 xfN, wlN;
    auto g1Micro = [&](auto sTag, auto mTag, unsigned xb) {  // GEMM 1 on sub-tile S: this wave's T1 trunk tiles, weights from AGPRs
      constexpr int S = decltype(sTag)::value, M = decltype(mTag)::value, I = M / T1, J = M % T1, C = I >> 1, KK = I & 1;
      if constexpr(I == 0) mfmaFirst<TR>(acc1[S][J], w1f[J][C][KK], xfA);
      else mfmaAcc<TR>(acc1[S][J], w1f[J][C][KK], xfA);
      if constexpr(J == 0) {
        xfN = xfA;
        if constexpr(I + 1 < 2 * K1) xfN = ldsV8(xLane[S][(I + 1) & 1] + xb + (unsigned)(((I + 1) >> 1) * G::CHUNK_BYTES));
      }
      if constexpr(J == T1 - 1) xfA = xfN;
      __builtin_amdgcn_sched_barrier(SCHED_ALU_ONLY);  // the read stays HERE, a k-step ahead of its use (left alone, hipcc sinks it to just before that MFMA)
    };
    // GEMM 2 on sub-tile S: the own mid tile (AGPRs) and, on sub-tile SH, the shared one (LDS): NM2(S) micro-steps
    auto g2Micro = [&](auto sTag, auto mTag) {
      constexpr int S = decltype(sTag)::value, M = decltype(mTag)::value;
      constexpr bool BOTH = S == SH;
      constexpr int I = BOTH ? M / 2 : M, PART = BOTH ? M % 2 : 0, C = I >> 1, KK = I & 1;
      if constexpr(PART == 0) {
        if constexpr(I == 0) mfmaFirst<TR>(acc2[S], w2f[C][KK], xfA);
        else mfmaAcc<TR>(acc2[S], w2f[C][KK], xfA);
        xfN = xfA;
        wlN = wlF;
        if constexpr(I + 1 < 2 * K2) {
          xfN = ldsV8(xLane[S][(I + 1) & 1] + (unsigned)(G::A2_OFF + ((I + 1) >> 1) * G::CHUNK_BYTES));
          if constexpr(BOTH) wlN = ldsV8(wlLane[(I + 1) & 1] + (unsigned)(((I + 1) >> 1) * G::WL_CHUNK));
        }
        if constexpr(!BOTH) xfA = xfN;
      }
      else {
        if constexpr(I == 0) mfmaFirstV<TR>(accS, wlF, xfA);
        else mfmaAccV<TR>(accS, wlF, xfA);
        xfA = xfN;
        wlF = wlN;
      }
      __builtin_amdgcn_sched_barrier(SCHED_ALU_ONLY);
    };
    // ---- vector pieces: ONE value per lane each (group GR = four consecutive channels of the lane's cell, value I of it; values 2 H and
    // 2 H + 1 share dword H of the group's packed 16-bit results, and their parameters come as one 8-byte LDS read) ----
    auto ldsF2 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) f32x2*)(size_t)addr; };
    auto pack2 = [&](T lo, T hi) {
      V4 t;
      t[0] = lo; t[1] = hi; t[2] = lo; t[3] = hi;
      return __builtin_bit_cast(u32x2, t)[0];
    };
    f32x2 sc2, bi2;   // parameters of the value pair in progress
    f32x2 scN, biN;   // ... of the next pair: read a pair ahead (one wave per SIMD has nobody to cover an LDS round trip)
    T pendR, pendO;   // the pair's low value, waiting for its partner
    constexpr auto e1ParamOff = [](int q) { return (32 * (q >> 4) + 8 * ((q >> 2) & 3) + (q & 3)) * 4; };
    // epilogue 1, value Q = 16 j + 4 g + i of sub-tile S: + residual, raw trunk -> HBM, activated -> the LDS image of GEMM 2
    auto e1Val = [&](auto sTag, auto qTag, T* rawRow) {
      constexpr int S = decltype(sTag)::value, Q = decltype(qTag)::value, J = Q >> 4, GR = (Q >> 2) & 3, I = Q & 3, H = I >> 1;
      if constexpr((Q & 15) == 0) unpair(rq[S][J], resP);
      if constexpr((I & 1) == 0) {
        sc2 = scN;
        bi2 = biN;
        if constexpr(Q + 2 < 16 * T1) {
          scN = ldsF2(p1Lane + (unsigned)e1ParamOff(Q + 2));
          biN = ldsF2(p1Lane + (unsigned)(G::C2 * 4 + e1ParamOff(Q + 2)));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const V4 rr = __builtin_bit_cast(V4, resP[GR]);
      const float v = acc1[S][J][4 * GR + I] + TR::toFloat(rr[I]);
      const float x = v * sc2[I & 1] + bi2[I & 1];
      const T r = TR::fromFloat(v), o = TR::fromFloat(actK<KIND1>(x));
      if constexpr((I & 1) == 0) {
        pendR = r;
        pendO = o;
      }
      else {
        rp[GR][H] = pack2(pendR, r);
        op[GR][H] = pack2(pendO, o) & onBits[S];
      }
      if constexpr((Q & 15) == 15) {
        u32x4 rawQ[2], oq[2];
        pairUp(rp, rawQ);  // this lane now holds channels 32 (T1 w + J) + 16 i + 8 h + [0, 8)
        pairUp(op, oq);
#pragma unroll
        for(int i = 0; i < 2; i++) *(GLOBAL u32x4*)(rawRow + 32 * J + 16 * i) = rawQ[i];
#pragma unroll
        for(int i = 0; i < 2; i++)  // image layout: chunk T1 w + J, this lane's row, logical 16-byte slot 2 i + h
          *(__attribute__((address_space(3))) u32x4*)(size_t)(a2wLane[S][i] + (unsigned)((T1 * wave + J) * G::CHUNK_BYTES)) = oq[i];
      }
    };
    // epilogue 2, value V of sub-tile S: values 0..15 the own mid tile, 16..31 the shared one (sub-tile SH only): mid raw and activated -> HBM
    auto e2Step = [&](auto sTag, auto vTag, T* rawOwn, T* actOwn, T* rawSh, T* actSh, unsigned on) {
      constexpr int S = decltype(sTag)::value, V = decltype(vTag)::value, NV = S == 0 ? NE2_0 : NE2_1;
      constexpr int TILE = V >> 4, Q = V & 15, GR = Q >> 2, I = Q & 3, H = I >> 1;
      if constexpr((I & 1) == 0) {
        sc2 = scN;
        bi2 = biN;
        if constexpr(V + 2 < NV) {
          constexpr int Q2 = (V + 2) & 15;
          const unsigned pl = ((V + 2) >> 4) ? p2Sh : p2Own;
          scN = ldsF2(pl + (unsigned)((8 * (Q2 >> 2) + (Q2 & 3)) * 4));
          biN = ldsF2(pl + (unsigned)(G::C3 * 4 + (8 * (Q2 >> 2) + (Q2 & 3)) * 4));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      float v;
      if constexpr(TILE == 0) v = acc2[S][4 * GR + I];
      else v = accS[4 * GR + I];
      const float x = v * sc2[I & 1] + bi2[I & 1];
      const T r = TR::fromFloat(v), o = TR::fromFloat(actK<KIND2>(x));
      if constexpr((I & 1) == 0) {
        pendR = r;
        pendO = o;
      }
      else {
        rp[GR][H] = pack2(pendR, r);
        op[GR][H] = pack2(pendO, o) & on;
      }
      if constexpr(Q == 15) {
        T* const rawRow2 = TILE ? rawSh : rawOwn;
        T* const actRow2 = TILE ? actSh : actOwn;
        u32x4 rawQ[2], oq[2];
        pairUp(rp, rawQ);
        pairUp(op, oq);
#pragma unroll
        for(int i = 0; i < 2; i++) {
          *(GLOBAL u32x4*)(rawRow2 + 16 * i) = rawQ[i];
          *(GLOBAL u32x4*)(actRow2 + 16 * i) = oq[i];
        }
      }
    };
    auto e2Params = [&]() {  // the first pair's parameters, ahead of a second-epilogue block
      scN = ldsF2(p2Own);
      biN = ldsF2(p2Own + (unsigned)(G::C3 * 4));
    };
    auto e1Params = [&]() {
      scN = ldsF2(p1Lane);
      biN = ldsF2(p1Lane + (unsigned)(G::C2 * 4));
    };
    auto rows2 = [&](bool liveS, long long cellBase, int s, T*& rawOwn, T*& actOwn, T*& rawSh, T*& actSh) {
      const size_t row = (size_t)(cellBase + 32 * s + myPos) * a.midC + 8 * khalf;
      rawOwn = liveS ? (T*)a.rawOut2 + row + 32 * wave : trash0;
      actOwn = liveS ? (T*)a.actOut2 + row + 32 * wave : trash0;
      rawSh = liveS ? (T*)a.rawOut2 + row + 32 * (NWAVES + (wave >> 1)) : trash0;
      actSh = liveS ? (T*)a.actOut2 + row + 32 * (NWAVES + (wave >> 1)) : trash0;
      asm volatile("" : "+v"(rawOwn), "+v"(actOwn), "+v"(rawSh), "+v"(actSh));
    };
    auto loadResid = [&](auto sTag, long long cell0) {
      constexpr int S = decltype(sTag)::value;
      // 16 bytes per lane and request: the lane pair (c, c + 32) of sub-tile S loads channels 32 (T1 w + j) + 16 i + 8 h + [0, 8) of its
      // cell; dead cells read the zero page
      const T* rrow = live[S] ? (const T*)a.resid + (size_t)(cell0 + 32 * S + myPos) * a.trunkC + 32 * T1 * wave + 8 * khalf : (const T*)zero;
      asm volatile("" : "+v"(rrow));
#pragma unroll
      for(int j = 0; j < T1; j++)
#pragma unroll
        for(int i = 0; i < 2; i++) rq[S][j][i] = *(const GLOBAL u32x4*)(rrow + 32 * j + 16 * i);
    };
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};

    for(; tile < numTiles; tile += stride) {
      const long long cell0 = tile * TM;
      const unsigned maskA = ldsBase + G::MASK_OFF + (unsigned)parity * (TM * 4);
      const unsigned xb = (unsigned)(G::X_OFF + parity * G::X_BUF);

      // ---- block 1: GEMM 1 of sub-tile 0 | the previous tile's second epilogue of sub-tile 1 ----
      // (X(t) and its mask were published by the previous tile's barrier BA1 - the prologue's for the first tile)
#pragma unroll
      for(int s = 0; s < NSUB; s++) live[s] = cell0 + 32 * s + myPos < a.cells;  // the same for both lanes of a pair
      loadResid(S0, cell0);
#pragma unroll
      for(int s = 0; s < NSUB; s++) onBits[s] = ldsF1(maskA + (32 * s + myPos) * 4) == 1.0f ? 0xffffffffu : 0u;  // off-board cells of activated images are zero
      stamp(0);
      xfA = ldsV8(xLane[0][0] + xb);
      if(havePrev) {
        T *rawOwn, *actOwn, *rawSh, *actSh;
        rows2(prevLive1, prevCell0, 1, rawOwn, actOwn, rawSh, actSh);
        mfmaSettle(acc2[1], accS, accS);
        e2Params();
        interleave<NM1, NE2_1>([&](auto m) { g1Micro(S0, m, xb); }, [&](auto v) { e2Step(S1, v, rawOwn, actOwn, rawSh, actSh, prevOn1); });
      }
      else staticFor<NM1>([&](auto m) { g1Micro(S0, m, xb); });
      stamp(1);

      // ---- block 2: GEMM 1 of sub-tile 1 | epilogue 1 of sub-tile 0 ----
      loadResid(S1, cell0);
      {
        T* rawRow = live[0] ? (T*)a.rawOut + (size_t)(cell0 + myPos) * a.trunkC + 32 * T1 * wave + 8 * khalf : trash0;
        asm volatile("" : "+v"(rawRow));
        mfmaSettle(acc1[0][0], acc1[0][1], acc1[0][2]);
        xfA = ldsV8(xLane[1][0] + xb);
        e1Params();
        interleave<NM1, NE1>([&](auto m) { g1Micro(S1, m, xb); }, [&](auto v) { e1Val(S0, v, rawRow); });
      }
      stamp(2);
      waitLds();    // the image rows are read by the other waves after the barrier
      wgBarrier();  // BA0: the activated image of sub-tile 0 is whole; every wave is done with X[parity]
      stamp(3);

      // ---- block 3: GEMM 2 of sub-tile 0 | epilogue 1 of sub-tile 1 ----
      {
        T* rawRow = live[1] ? (T*)a.rawOut + (size_t)(cell0 + 32 + myPos) * a.trunkC + 32 * T1 * wave + 8 * khalf : trash0;
        asm volatile("" : "+v"(rawRow));
        mfmaSettle(acc1[1][0], acc1[1][1], acc1[1][2]);
        xfA = ldsV8(xLane[0][0] + (unsigned)G::A2_OFF);
        if constexpr(SH == 0) wlF = ldsV8(wlLane[0]);
        e1Params();
        interleave<NM2_0, NE1>([&](auto m) { g2Micro(S0, m); }, [&](auto v) { e1Val(S1, v, rawRow); });
      }
      stamp(4);
      waitLds();
      // the NEXT tile's X and mask (requested a tile ago, in block 4 of the previous tile or by the prologue) have landed: in flight at
      // most what was issued after them (VM_AFTER_X)
      waitVm<G::VM_AFTER_X>();
      wgBarrier();  // BA1: the activated image of sub-tile 1 is whole; the next tile's X is there for every wave
      stamp(5);

      // ---- block 4: GEMM 2 of sub-tile 1 | epilogue 2 of sub-tile 0 ----
      // the X of the tile after next into THIS tile's buffer (last read in block 2, before BA0) and mask slot (read at the top of the tile)
      issueX(tile + 2 * stride, parity, tile + 2 * stride < numTiles);
      {
        T *rawOwn, *actOwn, *rawSh, *actSh;
        rows2(live[0], cell0, 0, rawOwn, actOwn, rawSh, actSh);
        mfmaSettle(acc2[0], accS, accS);
        xfA = ldsV8(xLane[1][0] + (unsigned)G::A2_OFF);
        if constexpr(SH == 1) wlF = ldsV8(wlLane[0]);
        e2Params();
        interleave<NM2_1, NE2_0>([&](auto m) { g2Micro(S1, m); }, [&](auto v) { e2Step(S0, v, rawOwn, actOwn, rawSh, actSh, onBits[0]); });
      }
      stamp(6);
      havePrev = true;
      prevLive1 = live[1];
      prevOn1 = onBits[1];
      prevCell0 = cell0;
      parity ^= 1;
    }
    // the last tile's second epilogue of sub-tile 1
    if(havePrev) {
      T *rawOwn, *actOwn, *rawSh, *actSh;
      rows2(prevLive1, prevCell0, 1, rawOwn, actOwn, rawSh, actSh);
      mfmaSettle(acc2[1], accS, accS);
      e2Params();
      staticFor<NE2_1>([&](auto v) { e2Step(S1, v, rawOwn, actOwn, rawSh, actSh, prevOn1); });
    }
    stamp(7);
  };
  if(wave & 1) run(std::integral_constant<int, 1>());
  else run(std::integral_constant<int, 0>());
  waitVm<0>();  // trailing requests into the slack area must land before the LDS is released
  if(TIMING && a.dbg != nullptr && lane == 0 && blockIdx.x == 0) {
    for(int i = 0; i < 8; i++) a.dbg[wave * 9 + i] = seg[i];
    a.dbg[wave * 9 + 8] = __builtin_readcyclecounter() - tStart;
  }
}

template <class TR, int K1, int K2, int NT3, int KIND1, int KIND2, bool TIMING = false>
hipError_t launchResident(const PwPairArgs& a, int maxGrid, hipStream_t stream) {
  typedef Geom<K1, K2, NT3> G;
  auto kern = pointwisePairResidentKernel<TR, K1, K2, NT3, KIND1, KIND2, TIMING>;
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
  // balanced grid (pointwise2_kernel.h): ceil(tiles / k) work-groups of k = ceil(tiles / CUs) tiles each
  long long grid = tiles < maxGrid ? tiles : maxGrid;
  if(tiles > maxGrid) {
    const long long perGroup = (tiles + maxGrid - 1) / maxGrid;
    grid = (tiles + perGroup - 1) / perGroup;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), G::LDS_BYTES, stream, a);
  return hipGetLastError();
}

#undef GLOBAL
}  // namespace pw3
}  // namespace kmx
#endif

// conv_kernel.h — the hot kernel: im2col-free implicit-GEMM convolution on gfx950 matrix cores.
// (device template; instantiated by conv_mfma.hip for the product and by conv_bench.hip for ablations)
//
// Replaces, for one layer, the reference's ConvLayer::apply (eigenbackend.cpp:293-703; on CUDA a
// cudnnConvolutionForward / cublas GEMM, cudabackend.cpp:531-844) fused with the masked
// BatchNorm+activation that follows it (eigenbackend.cpp:739-762), the per-board bias add
// (:137-148) and the residual accumulate (:659-686).
//
// Decomposition: one work-group = one BOARD x NTILE output channels; 4 x WNW waves, wave (wm, wn) owns
// 96 board cells x 32*WN channels (3 x WN MFMA tiles of 32x32):
//   D[cout][cell] += sum_{tap, cin} W[tap][cout][cin] * X[cell + tap][cin]
//   - MFMA v_mfma_f32_32x32x16_{f16,bf16}; the A operand is the WEIGHT tile (rows = cout), the B
//     operand the ACTIVATION tile (cols = board cells), so each lane ends up holding 4 consecutive
//     output channels of one cell.
//   - K loop: input-channel chunks of 32 (outer) x filter taps (inner) = "steps". Per chunk the board's
//     activations INCLUDING a zero halo live in LDS as [cell][32ch] rows of 64 bytes; every tap reads
//     the same image at a row offset — no im2col, no per-tap global traffic.
//   - both LDS images are filled by global_load_lds (LDS-DMA, 16 B/lane), i.e. they are LINEAR copies of
//     what the DMA lanes fetch: the weights are pre-tiled AND pre-swizzled in HBM by the engine, the
//     board image is gathered through per-lane source pointers computed once per work-group and then only
//     advanced 64 bytes per chunk (halo lanes walk a zero page). No VGPR staging, no ds_write, no address
//     arithmetic on the vector ALU in the main loop (VALU instructions share the issue slot with the MFMAs:
//     64 of them per 18 MFMAs cost 24 % of the matrix rate, tools/mfma_peak.py).
//   - bank conflicts: (a) an XOR swizzle of the four 16-byte slots of a row with bits 2-3 of the row index (16
//     consecutive rows x one logical slot cover all 64 banks once) rather than padding — 20 % fewer DMA and LDS bytes
//     than 80-byte rows; (b) GEMM column -> board cell is permuted so that each fixed 16-lane group of a
//     ds_read_b128 reads 16 CONSECUTIVE image rows (no halo wrap inside a group): see cellOf/posOf below.
//     PMC, 3x3 192->192: SQ_LDS_BANK_CONFLICT 2.67M -> 0.55M cycles per launch.
//   - software pipeline of depth D: the weight slab of step s+D is requested in step s (ring of D+1 slabs) and
//     published by the barrier at the top of step s+D-1, one step before it is consumed, so that the first k-half
//     fragments of step s+1 can be read during step s. For 3x3/5x5 the next chunk's board image is requested PPS
//     DMA instructions per step into the second image buffer; for 1x1 (one step per chunk) whole images ride
//     the same ring. Every wave issues the SAME number of DMA instructions per step (padding with dummies into a
//     slack area), so that one compile-time s_waitcnt vmcnt(N) retires exactly "everything up to slab s+1"
//     while D-1 steps of requests stay in flight across the single s_barrier per step.
//   - LDS reads are software-pipelined too: each batch of fragment reads is issued right after the first MFMA of the
//     other fragment set (the compiler's own s_waitcnt before an MFMA drains ALL outstanding LDS reads).
//   - epilogue (round 2): straight from the accumulator layout - no LDS transpose, no work-group barrier. Per-channel
//     parameters come from LDS (4-byte LDS-DMA in the prologue), the residual is fetched one tile ahead and added in fp32,
//     lane pairs (c, c + 32) regroup their 8-byte runs into 16-byte pieces (v_permlane32_swap), stores are unconditional
//     (pieces that must not land go to a trash area) so that every s_waitcnt count is a compile-time constant. BN
//     scale/bias, activation, mask and the 16-bit rounding happen there; see the comments at the epilogue itself.
//   - 8-wave 3x3/5x5 shapes: waves 0-3 issue all the DMA (they are served first by the matrix core and have the slack).
//   Cycle account of a 3x3 192->192 work-group at batch 256 (profiles/r02_final/conv_timing.log): prologue 9.8 k, loop 82.8 k
//   (62.2 k of pure MFMA time), epilogue 19 k without / 41 k with residual. Variants that were built and measured slower
//   are listed in DESIGN.md 4.8 (profiles/r02_steps, r01_v6_experiment).
#ifndef KMX_CONV_KERNEL_H_
#define KMX_CONV_KERNEL_H_

#include <atomic>

#include "device_common.h"

namespace kmx {
namespace convk {

constexpr int ROWB = WROW_HALFS * 2;  // 64 bytes per LDS row = 4 slots of 16 bytes
constexpr int MT = 3;                 // board-cell tiles (of 32) per wave: 4 waves x 96 = 384 >= 361
constexpr int MAXLEN = 19;

// ablation switches (conv_bench.hip only; 0 in the product)
enum { ABL_NO_EPILOGUE = 1, ABL_NO_COMPUTE = 2, ABL_NO_DMA = 4, ABL_NO_LDS_READ = 8, ABL_NO_VMWAIT = 16, ABL_NO_W_DMA = 32, ABL_NO_A_DMA = 64,
       ABL_EPI_NOACT = 512, ABL_EPI_NOSTORE = 1024,
       ABL_TIMING = 2048 /* s_memtime stamps between the segments of a step, summed per wave into ConvArgs::dbg */,
       ABL_SPLIT = 262144 /* NOT an ablation: the cell tiles of a board over three work-groups, see the kernel */,
       ABL_BATCHED = 131072 /* the round-2 form of a step (A/B): the six fragment reads of a half-step as one batch behind its first MFMA, the step's LDS-DMA requests as one batch between the halves */, };

// Four waves along the cell dimension, each owning 3 tiles of 32 cells (4 x 96 = 384 >= 361). (Round 3's small-batch shape of twelve
// cell waves with one tile each lost to conv_small_kernel.h in round 4 and is gone)
template <int KS, int WN, int WNW, int D>
struct Geom {
  static constexpr int NWAVES = 4 * WNW;
  static constexpr int NTHREADS = NWAVES * 64;
  static constexpr int HALO = KS / 2;
  static constexpr int NT = KS * KS;
  static constexpr int HPMAX = (MAXLEN + 2 * HALO) * (MAXLEN + 2 * HALO);
  // 8-wave work-groups split the DMA work by ROLE: waves 0-3 (the older wave of each SIMD, which the matrix core serves
  // first and which otherwise idles at the barrier; tools/conv_timing.py) fetch the weight slabs, waves 4-7 the board image.
  // No dummy requests, per-role s_waitcnt constants, and a weight wave never waits behind an HBM-latency image piece.
  // 4-wave work-groups: every wave does both, padded with dummies to a constant per-step count.
  static constexpr bool ROLES = WNW == 2;
  static constexpr int NLW = !ROLES ? NWAVES : 4;  // waves that fetch weights
  static constexpr int NLA = !ROLES ? NWAVES : 4;  // waves that fetch the image
  static constexpr int NPA = (HPMAX * 4 + NLA * 64 - 1) / (NLA * 64);   // DMA instructions per image-loading wave per board image
  static constexpr int ACT_BYTES = (HPMAX * ROWB + 1023) / 1024 * 1024;  // instructions wholly past it go to the slack
  static constexpr int NTILE = 32 * WN * WNW;
  static constexpr int WPIECES = NTILE * 4;
  static constexpr int NPW = (WPIECES + NLW * 64 - 1) / (NLW * 64);  // DMA instructions per weight-loading wave per slab
  static constexpr int W_BYTES = (WPIECES * 16 + 1023) / 1024 * 1024;
  static_assert(D >= 2, "the slab of step s+1 is published at the top of step s: at least two steps of requests in flight");
  // image requests for chunks past the last one are still issued (into the slack): their sources run up to D chunks of 64 bytes past
  // the last cell's row, plus the lane's own 16-byte slot
  static_assert((D + 1) * KCHUNK * 2 <= DEVBUF_TAIL_BYTES, "the image requests' run-ahead must stay inside the readable tail of a DevBuf");
  // 3x3/5x5: the next chunk's image arrives PPS pieces per step during the first LS steps of the current chunk
  // (the fewest pieces per step for which it is complete D steps before the chunk ends); 1x1: whole images per step.
  static constexpr bool SPREAD = NT > 1;
  static constexpr int pickPPS() {
    for(int p = 1; p <= NPA; p++)
      if((NPA + p - 1) / p + (ROLES ? 1 : D) <= NT) return p;  // ROLES: complete before the last step of the chunk
    return NPA;
  }
  static constexpr int PPS = SPREAD ? pickPPS() : NPA;
  static constexpr int LS = (NPA + PPS - 1) / PPS;
  static constexpr int NSA = SPREAD ? 2 : D + 1;
  static constexpr int NSW = D + 1;
  static constexpr int SLACK_BYTES = 1024;  // destination of padding / past-the-end DMA instructions (never read; shared by all waves)
  static constexpr int NPM = (384 + NTHREADS - 1) / NTHREADS;          // 4-byte DMA instructions per wave for the mask
  static constexpr int MASK_BYTES = NPM * NTHREADS * 4;
  // per-channel parameters of the tile for the epilogue: scale | bias | per-board bias, each padded to whole 64-lane instructions
  static constexpr int NTP = (NTILE + 63) / 64 * 64;
  static constexpr int NPP = (3 * NTP / 64 + NWAVES - 1) / NWAVES;     // 4-byte DMA instructions per wave for them
  static constexpr int PARAM_BYTES = NPP * NWAVES * 256;
  static constexpr int PIPE_BYTES = NSA * ACT_BYTES + NSW * W_BYTES + SLACK_BYTES;
  static constexpr int MASK_OFFSET = PIPE_BYTES;
  static constexpr int PARAM_OFFSET = MASK_OFFSET + MASK_BYTES;
  static constexpr int LDS_BYTES = PARAM_OFFSET + PARAM_BYTES;
  // DMA instructions younger than the data of the current step when it is waited for
  // top of step s: everything up to slab s+1 has landed (requested in step s+1-D, followed there by its image piece)
  static constexpr int VMCNT = SPREAD ? PPS + (D - 2) * (NPW + PPS) : (D - 2) * (NPW + NPA);
  // before the loop: slab 0 and image 0
  static constexpr int VMCNT_PRO = SPREAD ? PPS + (D - 1) * (NPW + PPS) : (D - 1) * (NPW + NPA);
  // ROLES: a weight wave has only slabs in flight, an image wave only image pieces
  static constexpr int VMCNT_W = (D - 2) * NPW, VMCNT_PRO_W = (D - 1) * NPW;
  static constexpr int VMCNT_A1 = (D - 2) * NPA, VMCNT_PRO_A1 = (D - 1) * NPA;  // 1x1: whole images ride the ring
};

template <int N>
__device__ __forceinline__ void waitVm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt vmcnt(n) for an n that is a constant only after the tap loop is unrolled
__device__ __forceinline__ void waitVmSel(int n) {
  switch(n) {
    case 0: waitVm<0>(); break;
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
// LDS destinations are 32-bit LDS addresses (wave-uniform; lane l lands at +16 l), not generic pointers: a generic -> LDS
// pointer cast carries a null check, and hipcc (ROCm 7.2) mis-selects that compare when the pointer is a select of uniform values
__device__ __forceinline__ void dma16(const void* gsrc, unsigned ldsWaveBase) {
  __builtin_amdgcn_global_load_lds(
    (const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)(size_t)ldsWaveBase, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* gsrc, unsigned ldsWaveBase) {
  __builtin_amdgcn_global_load_lds(
    (const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)(size_t)ldsWaveBase, 4, 0, 0);
}

template <class TR, int KS, int WN, int WNW, int D, int ABL>
__global__ __launch_bounds__(256 * WNW) __attribute__((amdgpu_waves_per_eu(2, 2))) void convMfmaKernel(const ConvArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  typedef typename TR::V4 V4;
  typedef Geom<KS, WN, WNW, D> G;
  constexpr int HALO = G::HALO, NT = G::NT, NPA = G::NPA, NPW = G::NPW, NWAVES = G::NWAVES;
  constexpr bool SPREAD = G::SPREAD;
  // ABL_SPLIT (a product shape, not an ablation: conv_mfma.hip cfg 113): the wave's three cell tiles over THREE work-groups (grid z), one
  // each - every work-group fetches the whole image and every slab and does a third of the matrix work; for batches that leave CUs idle
  constexpr int MTL = (ABL & ABL_SPLIT) ? 1 : MT;  // cell tiles per wave in THIS work-group
  const int pt0 = (ABL & ABL_SPLIT) ? (int)blockIdx.z : 0;

  const unsigned long long tKernel0 = (ABL & ABL_TIMING) ? __builtin_readcyclecounter() : 0;  // work-group start
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;  // 32-bit LDS byte address of the allocation
  const unsigned bufA = ldsBase;
  const unsigned bufW = bufA + G::NSA * G::ACT_BYTES;
  const unsigned slack = bufW + G::NSW * G::W_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const int n = blockIdx.y;
  const int cout0 = blockIdx.x * G::NTILE;
  const int X = a.X, Y = a.Y, S = X * Y;
  const int W2 = X + 2 * HALO, HP = W2 * (Y + 2 * HALO);
  const int inC = a.inC;
  // GEMM column -> board cell. A ds_read_b128 is served in four fixed groups of 16 lanes ({0-3,12-15,20-27},
  // {4-11,16-19,28-31} and the same +32; MI355X_MICROARCH.md, LDS) and is conflict-free under the XOR swizzle when the 16
  // image rows of a group differ in (row mod 16). Consecutive cells do that inside a board row but not across the row
  // wrap (the halo skips two image rows; PMC: 30 % of the LDS cycles were conflicts with column = lane = cell). So:
  //   * columns are counted in "positions": position p of a 32-column tile sits in lane posLane(p), the first 16
  //     positions filling the first lane group, the next 16 the second;
  //   * positions 0..16Y-1 are the cells x < 16 of each board row (16 consecutive image rows per lane group), the
  //     remaining X-16 cells per row are collected at the end. Boards narrower than 16 keep position = cell.
  const int mainCols = X >= 16 ? 16 * Y : 0;
  const int restW = X - 16;
  auto cellOf = [&](int m) -> int {
    if(m < mainCols) return (m >> 4) * X + (m & 15);
    if(X < 16) return m;
    const int k = m - mainCols;
    const int yy = k / restW;
    return yy * X + 16 + (k - yy * restW);
  };
  // position of lane l (0..31) inside its tile: lanes 0-3,12-15,20-27 -> 0..15 ; 4-11,16-19,28-31 -> 16..31
  auto posOf = [](int l) -> int {
    return l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l;
  };
  const int myPos = posOf(lane & 31);

  const char* const inBoard = (const char*)a.in + (size_t)n * S * inC * sizeof(T);
  const char* const zero = (const char*)a.zeroPage;
  const unsigned mySlack = slack;

  // ---- per-lane DMA source POINTERS of the board image (halo and out-of-image lanes point into the zero page) ----
  // LDS piece p = row p/4, PHYSICAL slot p%4; it holds logical slot (p%4) ^ ((row>>2)&3) of that row. The pointers
  // advance by one 32-channel chunk (64 bytes) each time they are used, so the main loop spends no vector ALU
  // work on DMA addresses (vector ALU instructions compete with the MFMAs for the issue slot).
  constexpr bool ROLES = G::ROLES;
  // ONE: in the 8-wave 3x3 / 5x5 shapes waves 0-3 issue ALL the DMA (weight slabs and the next chunk's board image). They are
  // the older wave of each SIMD: the matrix core serves them first, and they used to idle ~27 % of the loop at the barrier
  // waiting for waves 4-7, which multiply at lower priority AND paid ~330 cycles per step for their image requests
  // (an LDS-DMA instruction costs its wave 100-200 cycles of issue). With every request on the waves that have the slack,
  // the younger waves only multiply. (tools/conv_timing.py)
  // (possible when the image pieces have all been requested before the slab that is waited for at the last tap: LS <= NT + 1 - D)
  constexpr bool ONE = ROLES && SPREAD && G::LS <= NT + 1 - D;
  // the step's fragment reads (and, in the ONE division, its DMA requests) spread over its MFMAs: needs a slot per read in each half
  constexpr bool SPREAD_STEP = WN * MTL >= WN + MTL && !(ABL & (ABL_BATCHED | ABL_TIMING));
  constexpr int SLOTS = WN * MTL - (WN + MTL);                           // MFMAs of a half-step that carry no fragment read
  constexpr bool SPREAD_DMA = SPREAD_STEP && ONE && 2 * SLOTS >= 1 + NPW;  // the image piece(s) of the tap + the slab's instructions
  const bool wLoader = !ROLES || wave < G::NLW;   // wave-uniform
  const bool aLoader = !ROLES || (ONE ? wave < 4 : wave >= G::NLW);
  const int lw = wave;                            // index among the weight-loading waves
  const int la = !ROLES ? wave : (wave & 3);  // index among the image-loading waves
  unsigned srcOff[NPA];  // byte offset from this board's tensor, or (bit 31 set) into the zero page; +64 per chunk
#pragma unroll
  for(int j = 0; j < NPA; j++) {
    int p = (j * G::NLA + la) * 64 + lane;
    int hp = p >> 2;
    int slot = (p & 3) ^ ((hp >> 2) & 3);
    unsigned off = 0x80000000u;
    if(hp < HP) {
      int hy = hp / W2;
      int hx = hp - hy * W2;
      int y = hy - HALO, x = hx - HALO;
      if(y >= 0 && y < Y && x >= 0 && x < X) off = (unsigned)(((y * X + x) * inC + slot * 8) * (int)sizeof(T));
    }
    srcOff[j] = off;
  }
  const char* const wBase = (const char*)a.w + (size_t)cout0 * ROWB;
  const size_t wSlabStride = (size_t)a.coutPad * ROWB;
  const int nChunks = a.nChunks;
  const int nSteps = nChunks * NT;
  unsigned wOff[NPW];  // per-lane byte offset inside a weight slab; the slab address itself is wave-uniform (SGPR base)
#pragma unroll
  for(int j = 0; j < NPW; j++) {
    const int p = (j * G::NLW + (lw % G::NLW)) * 64 + lane;
    wOff[j] = (unsigned)(p < G::WPIECES ? p : lane) * 16u;
  }

  // Every call issues exactly NPW instructions (into the slack area when `step` is past the end).
  auto issueW = [&](int step) {
    if(ABL & (ABL_NO_DMA | ABL_NO_W_DMA)) return;
    const bool live = step < nSteps;
    const char* slab = wBase + (size_t)(live ? step : 0) * wSlabStride;
    const unsigned dst = bufW + (step % G::NSW) * G::W_BYTES;
#pragma unroll
    for(int j = 0; j < NPW; j++) {
      const int pbase = (j * G::NLW + (lw % G::NLW)) * 64;
      const bool inRange = live && pbase < G::WPIECES;
      dma16(slab + wOff[j], inRange ? dst + pbase * 16 : mySlack);
    }
  };
  // One instruction of the board image of `chunk` (piece j; its pointer then moves on to the next chunk), or a
  // dummy into the slack area when j < 0.
  auto issueA = [&](int chunk, int j) {
    if(ABL & (ABL_NO_DMA | ABL_NO_A_DMA)) return;
    const int jj = j < 0 ? 0 : j;
    const int pbase = (jj * G::NLA + la) * 64;
    const bool live = j >= 0 && chunk < nChunks && pbase * 16 < G::ACT_BYTES;
    const unsigned off = srcOff[jj];
    const char* src = (off & 0x80000000u) ? zero + (off & 0x7fffffffu) : inBoard + off;
    dma16(src, live ? bufA + (chunk % G::NSA) * G::ACT_BYTES + pbase * 16 : mySlack);
    if(j >= 0) srcOff[jj] = off + KCHUNK * sizeof(T);
  };

  // ---- per-lane LDS read addressing (32-bit LDS byte addresses; kept to 3-4 vector ALU operations per fragment) ----
  const unsigned khalf = lane >> 5;  // which 8 of the 16 k-values of an MFMA this lane feeds
  // weights: row r = wn*32*WN + ct*32 + (lane&31); (r>>2)&3 does not depend on ct or wn (both multiples of 32 rows)
  const unsigned wXor = (lane >> 2) & 3;
  unsigned wLane[2];  // lane part of a weight fragment address, per k half
#pragma unroll
  for(int kk = 0; kk < 2; kk++)
    wLane[kk] = ldsBase + G::NSA * G::ACT_BYTES + (wn * (32 * WN) + (lane & 31)) * ROWB + (((kk * 2 + khalf) ^ wXor) << 4);
  // activations: 4 x (halo-image row) of this lane's cell in each of its MTL tiles; row q lives at byte q*64 and its
  // logical 16-byte slot c at physical slot c ^ ((q>>2)&3):  address = (4q << 4) | ((4q ^ (c<<4)) & 0x30)
  unsigned aRow4[MTL];
#pragma unroll
  for(int pt = 0; pt < MTL; pt++) {
    int j = wm * (32 * MT) + (pt0 + pt) * 32 + myPos;
    j = cellOf(j < S ? j : S - 1);  // columns beyond the board recompute the last one; never stored
    int y = j / X;
    int x = j - y * X;
    aRow4[pt] = (unsigned)((y + HALO) * W2 + (x + HALO)) << 2;
  }
  const unsigned c40 = khalf << 4;
  const bool waveActive = wm * (32 * MT) + pt0 * 32 < S;

  // Fragment readers. A step's 16 k-values per MFMA come as two halves kk = 0, 1 (register sets F0, F1).
  V8 wf[2][WN];
  V8 af[2][MTL];
  unsigned aAddr[MTL];  // kk=0 addresses of the tap last prepared; the kk=1 fragments sit 32 bytes away (slot ^ 2)
  auto ldsV8 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) V8*)addr; };
  auto readW = [&](int kk, int stepIdx) {
    if(ABL & (ABL_NO_LDS_READ | ABL_NO_COMPUTE)) return;
    const unsigned base = wLane[kk] + (unsigned)(stepIdx % G::NSW) * G::W_BYTES;
#pragma unroll
    for(int ct = 0; ct < WN; ct++) wf[kk][ct] = ldsV8(base + ct * 32 * ROWB);
  };
  // kk=0 fragments of tap t from the image buffer at byte offset imgOff (a multiple of 1024)
  auto readA0 = [&](unsigned imgOff, int t) {
    if(ABL & (ABL_NO_LDS_READ | ABL_NO_COMPUTE)) return;
    // 4 x (row offset of the tap) plus the buffer offset, both wave-uniform; opaque to the optimiser on purpose:
    // hoisted out of the chunk loop the NT*MTL addresses cost more registers than the two-wave-per-SIMD budget has
    unsigned sTap = (unsigned)(((t / KS - HALO) * W2 + (t % KS - HALO)) * 4) + ((ldsBase + imgOff) >> 4);
    asm volatile("" : "+s"(sTap));
#pragma unroll
    for(int pt = 0; pt < MTL; pt++) {
      const unsigned q4 = aRow4[pt] + sTap;
      aAddr[pt] = (q4 << 4) | ((q4 ^ c40) & 0x30u);
      af[0][pt] = ldsV8(aAddr[pt]);
    }
  };
  auto readA1 = [&]() {
    if(ABL & (ABL_NO_LDS_READ | ABL_NO_COMPUTE)) return;
#pragma unroll
    for(int pt = 0; pt < MTL; pt++) af[1][pt] = ldsV8(aAddr[pt] ^ 0x20u);
  };
  // MFMAs [first, last) of the WN*MTL that consume fragment set kk
  auto mfmaPart = [&](int kk, int first, int last, f32x16 (&acc)[WN][MTL]) {
    if(ABL & ABL_NO_COMPUTE) return;
#pragma unroll
    for(int ct = 0; ct < WN; ct++)
#pragma unroll
      for(int pt = 0; pt < MTL; pt++)
        if(ct * MTL + pt >= first && ct * MTL + pt < last) acc[ct][pt] = TR::mfma(wf[kk][ct], af[kk][pt], acc[ct][pt]);
  };
#pragma unroll
  for(int kk = 0; kk < 2; kk++) {
#pragma unroll
    for(int ct = 0; ct < WN; ct++)
#pragma unroll
      for(int i = 0; i < 8; i++) wf[kk][ct][i] = (T)(0.001f * (float)(lane + ct));
#pragma unroll
    for(int pt = 0; pt < MTL; pt++)
#pragma unroll
      for(int i = 0; i < 8; i++) af[kk][pt][i] = (T)(0.002f * (float)(lane + pt));
  }

  // ---- prologue, part 1: the oldest requests are this board's mask (4 bytes per lane) ----
#pragma unroll
  for(int j = 0; j < G::NPM; j++) {
    const int cellIdx = (j * NWAVES + wave) * 64 + lane;
    const float* msrc = cellIdx < S ? a.mask + (size_t)n * S + cellIdx : (const float*)zero;
    dma4(msrc, ldsBase + G::MASK_OFFSET + (j * NWAVES + wave) * 256);
  }
  // ... and the per-channel parameters of this tile for the epilogue (merged BN scale, bias, per-board bias)
#pragma unroll
  for(int j = 0; j < G::NPP; j++) {
    const int idx = (j * NWAVES + wave) * 64 + lane;
    const int arr = idx / G::NTP, c = idx % G::NTP;
    const float* psrc = (const float*)zero;
    if(c < G::NTILE) {
      if(arr == 0) psrc = a.scale + cout0 + c;
      else if(arr == 1) psrc = a.bias + cout0 + c;
      else if(arr == 2 && a.ncBias != nullptr) psrc = a.ncBias + (size_t)n * a.ncBiasStride + cout0 + c;
    }
    dma4(psrc, ldsBase + G::PARAM_OFFSET + (j * NWAVES + wave) * 256);
  }
  // ---- prologue, part 2: fill the pipeline with the same per-step instruction pattern the loop uses ----
  if(ONE) {
    if(wLoader) {
#pragma unroll
      for(int j = 0; j < NPA; j++) issueA(0, j);
#pragma unroll
      for(int s = 0; s < D; s++) issueW(s);
    }
  }
  else if(ROLES) {
    if(wLoader) {
#pragma unroll
      for(int s = 0; s < D; s++) issueW(s);
    }
    else if(SPREAD) {
#pragma unroll
      for(int j = 0; j < NPA; j++) issueA(0, j);
    }
    else {
#pragma unroll
      for(int s = 0; s < D; s++)
#pragma unroll
        for(int j = 0; j < NPA; j++) issueA(s, j);
    }
  }
  else if(SPREAD) {
#pragma unroll
    for(int j = 0; j < NPA; j++) issueA(0, j);
#pragma unroll
    for(int s = 0; s < D; s++) {
      issueW(s);
#pragma unroll
      for(int i = 0; i < G::PPS; i++) issueA(0, -1);  // dummies: keep the per-step DMA count constant
    }
  }
  else {
#pragma unroll
    for(int s = 0; s < D; s++) {
      issueW(s);
#pragma unroll
      for(int j = 0; j < NPA; j++) issueA(s, j);
    }
  }

  f32x16 acc[WN][MTL];
#pragma unroll
  for(int ct = 0; ct < WN; ct++)
#pragma unroll
    for(int pt = 0; pt < MTL; pt++)
#pragma unroll
      for(int r = 0; r < 16; r++) acc[ct][pt][r] = 0.0f;

  // ---- main loop ----
  // Invariant at the top of step s: slab s (and its board image) landed and was barrier-published one step EARLIER, and
  // the kk=0 fragments of step s are already in registers (or in flight). The step then runs
  //     wait+barrier (publishes slab s+1) | read F1(s) | MFMA F0(s) | DMA for step s+D | read F0(s+1) | MFMA F1(s)
  // so every LDS read has eight MFMAs (256 matrix-core cycles) to land in, and the matrix core only idles for the
  // barrier skew between waves.
  // image pieces requested at tap tt of a chunk (ONE)
  auto piecesAt = [&](int tt) -> int {
    const int left = NPA - tt * G::PPS;
    return left <= 0 ? 0 : left < G::PPS ? left : G::PPS;
  };
  auto issueStep = [&](int chunk, int t, int step) {
    if(ONE) {
      // image pieces FIRST, then the slab: everything older than the slab that a later step waits for has then landed too.
      // In the last chunk the pieces are dummies (the count per tap stays a compile-time constant).
      if(wLoader) {
#pragma unroll
        for(int i = 0; i < G::PPS; i++)
          if(t * G::PPS + i < NPA) issueA(chunk + 1, t * G::PPS + i);
        issueW(step + D);
      }
      return;
    }
    if(ROLES) {
      if(wLoader) issueW(step + D);
      else if(SPREAD) {
        // real pieces only; nothing after the image is complete (this wave waits with vmcnt(0) once per chunk)
        if(t * G::PPS < NPA && chunk + 1 < nChunks) {
#pragma unroll
          for(int i = 0; i < G::PPS; i++)
            if(t * G::PPS + i < NPA) issueA(chunk + 1, t * G::PPS + i);
        }
      }
      else {
#pragma unroll
        for(int j = 0; j < NPA; j++) issueA(chunk + D, j);
      }
      return;
    }
    issueW(step + D);
    if(SPREAD) {
#pragma unroll
      for(int i = 0; i < G::PPS; i++) issueA(chunk + 1, t * G::PPS + i < NPA ? t * G::PPS + i : -1);
    }
    else {
#pragma unroll
      for(int j = 0; j < NPA; j++) issueA(chunk + D, j);
    }
  };
  // (SPREAD_DMA) the k-th request of a step in the ONE division: k = 0 the image piece of this tap (if any), k = 1 .. NPW the
  // slab's instructions - the same requests in the same order as issueStep, one call each
  auto issueOneDma = [&](int chunk, int t, int step, int k) {
    if(!ONE || !wLoader) return;
    if(ABL & (ABL_NO_DMA)) return;
    if(k == 0) {
#pragma unroll
      for(int i = 0; i < G::PPS; i++)
        if(t * G::PPS + i < NPA) issueA(chunk + 1, t * G::PPS + i);
      return;
    }
    const int j = k - 1;
    if(j >= NPW) return;
    const int stepW = step + D;
    const bool live = stepW < nSteps;
    const char* slab = wBase + (size_t)(live ? stepW : 0) * wSlabStride;
    const unsigned dst = bufW + (stepW % G::NSW) * G::W_BYTES;
    const int pbase = (j * G::NLW + (lw % G::NLW)) * 64;
    const bool inRange = live && pbase < G::WPIECES;
    dma16(slab + wOff[j], inRange ? dst + pbase * 16 : mySlack);
  };
  // top-of-step wait of this wave's own requests: everything the barrier is about to publish
  auto waitStep = [&](int t) {
    if(ABL & (ABL_NO_DMA | ABL_NO_VMWAIT | ABL_NO_W_DMA | ABL_NO_A_DMA)) return;
    if(ONE) {
      // top of step s: slab s+1 (requested in step s+1-D, after that step's image pieces) has landed; younger: the requests of
      // steps s+2-D .. s-1
      if(wLoader) {
        int n = 0;
#pragma unroll
        for(int k = 1; k <= D - 2; k++) n += NPW + piecesAt((t - k + NT * D) % NT);
        waitVmSel(n);
      }
      return;
    }
    if(!ROLES) waitVm<G::VMCNT>();
    else if(wLoader) {
      waitVm<G::VMCNT_W>();
    }
    else if(!SPREAD) waitVm<G::VMCNT_A1>();
    else if(t == NT - 1) waitVm<0>();  // the next chunk's image, first read at the end of this step
  };
  if(ABL & (ABL_NO_DMA | ABL_NO_W_DMA | ABL_NO_A_DMA)) waitVm<0>();
  else if(!ROLES) waitVm<G::VMCNT_PRO>();
  else if(wLoader) waitVm<G::VMCNT_PRO_W>();
  else if(!SPREAD) waitVm<G::VMCNT_PRO_A1>();
  else waitVm<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  readW(0, 0);
  readA0(0, 0);
  // ABL_TIMING: cycles per segment of a step, summed over the steps (segments: wait+barrier | first MFMA + read F1 + MFMA F0
  // issue | DMA issue | first MFMA + read F0' + MFMA F1 issue); seg[4] = whole loop, seg[5..7] = prologue / loop end / kernel end
  unsigned long long seg[5] = {0, 0, 0, 0, 0};
  unsigned long long tPrev = 0;
  auto stamp = [&](int which) {
    if(!(ABL & ABL_TIMING)) return;
    const unsigned long long now = __builtin_readcyclecounter();
    seg[which] += now - tPrev;
    tPrev = now;
  };
  if(ABL & ABL_TIMING) tPrev = __builtin_readcyclecounter();
  const unsigned long long tLoop0 = tPrev;
  int step = 0;
  for(int chunk = 0; chunk < nChunks; chunk++) {
    const unsigned curA = (unsigned)(chunk % G::NSA) * G::ACT_BYTES;
    const unsigned nextA = (unsigned)((chunk + 1) % G::NSA) * G::ACT_BYTES;
#pragma unroll
    for(int t = 0; t < NT; t++, step++) {
      waitStep(t);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stamp(0);
      // The compiler's own s_waitcnt before an MFMA drains ALL outstanding LDS reads (lgkmcnt(0)), so each batch of
      // reads is issued right AFTER the first MFMA of the other fragment set: the wait it causes then only covers
      // reads that had eight MFMAs to land.
      if constexpr(SPREAD_STEP) {
        // Round 3: the WN + MTL fragment reads of a half-step ride behind its first MFMAs ONE BY ONE, and (8-wave 3x3 / 5x5 shapes) so
        // do the step's LDS-DMA requests behind the later ones. As batches - six ds_read_b128 behind the first MFMA, three or four
        // DMA instructions between the halves - they held the wave's issue stage while its matrix core ran dry: a DMA instruction
        // costs its wave 100-150 cycles of issue, and all eight waves hit the LDS with their batches right after the barrier.
        // 3x3 192->192, batch 256, same box: 64.3 -> 61.8 us (act only), 76.7 -> 73.1 us (residual); profiles/r03_steps/conv_spread_step.txt.
        // Same MFMAs in the same order: results are bit-identical to the batched form (ABL_BATCHED; the cycle stamps use it too).
        const unsigned wb1 = wLane[1] + (unsigned)(step % G::NSW) * G::W_BYTES;
#pragma unroll
        for(int idx = 0; idx < WN * MTL; idx++) {
          mfmaPart(0, idx, idx + 1, acc);
          __builtin_amdgcn_sched_barrier(0);
          if(idx < WN) wf[1][idx] = ldsV8(wb1 + idx * 32 * ROWB);
          else if(idx < WN + MTL) af[1][idx - WN] = ldsV8(aAddr[idx - WN] ^ 0x20u);
          if constexpr(SPREAD_DMA) {
            if(idx >= WN + MTL) issueOneDma(chunk, t, step, idx - (WN + MTL));  // requests 0 .. WN*MTL - WN - MTL - 1
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr(!SPREAD_DMA) issueStep(chunk, t, step);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned wb0 = wLane[0] + (unsigned)((step + 1) % G::NSW) * G::W_BYTES;
        {
          const int tn = t + 1 < NT ? t + 1 : 0;
          unsigned sTap = (unsigned)(((tn / KS - HALO) * W2 + (tn % KS - HALO)) * 4) + ((ldsBase + (t + 1 < NT ? curA : nextA)) >> 4);
          asm volatile("" : "+s"(sTap));
#pragma unroll
          for(int pt = 0; pt < MTL; pt++) {
            const unsigned q4 = aRow4[pt] + sTap;
            aAddr[pt] = (q4 << 4) | ((q4 ^ c40) & 0x30u);
          }
        }
#pragma unroll
        for(int idx = 0; idx < WN * MTL; idx++) {
          mfmaPart(1, idx, idx + 1, acc);
          __builtin_amdgcn_sched_barrier(0);
          if(idx < WN) wf[0][idx] = ldsV8(wb0 + idx * 32 * ROWB);
          else if(idx < WN + MTL) af[0][idx - WN] = ldsV8(aAddr[idx - WN]);
          if constexpr(SPREAD_DMA) {
            if(idx >= WN + MTL) issueOneDma(chunk, t, step, SLOTS + idx - (WN + MTL));  // the rest
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        continue;
      }
      mfmaPart(0, 0, 1, acc);
      __builtin_amdgcn_sched_barrier(0);
      readW(1, step);
      readA1();
      __builtin_amdgcn_sched_barrier(0);
      mfmaPart(0, 1, WN * MTL, acc);
      __builtin_amdgcn_sched_barrier(0);
      stamp(1);
      issueStep(chunk, t, step);  // (placing the image waves' requests at the start of the step instead measured 1-2 % slower)
      __builtin_amdgcn_sched_barrier(0);
      stamp(2);
      mfmaPart(1, 0, 1, acc);
      __builtin_amdgcn_sched_barrier(0);
      readW(0, step + 1);
      readA0(t + 1 < NT ? curA : nextA, t + 1 < NT ? t + 1 : 0);
      __builtin_amdgcn_sched_barrier(0);
      mfmaPart(1, 1, WN * MTL, acc);
      __builtin_amdgcn_sched_barrier(0);
      stamp(3);
    }
  }
  if(!(ABL & ABL_NO_DMA)) waitVm<0>();  // retire the trailing dummies before the LDS is reused / the wave exits
  const unsigned long long tLoop1 = (ABL & ABL_TIMING) ? __builtin_readcyclecounter() : 0;

  // ---- epilogue ----
  // Straight from the accumulator layout, no LDS round trip and no work-group barrier (nothing below writes LDS, so a wave
  // that leaves the loop early starts here while its SIMD partner still multiplies): lane (column, h) of a tile holds 4
  // consecutive channels per group g of ONE cell, so the cell index, its mask and its output rows are computed once per
  // tile; the per-channel parameters come from LDS (two broadcast reads per group); lane pairs (c, c + 32) regroup their
  // 8-byte runs into 16-byte pieces (device_common.h pairUp) and store them. The vector ALU is what bounds this phase
  // (~4 cycles per wave instruction, 144 outputs per lane): the row-walk it replaces spent a third of its instructions on
  // index arithmetic and staging.
  // (explicit LDS addresses, as for the fragment reads)
  auto ldsF4 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) f32x4*)addr; };
  auto ldsF1 = [&](unsigned addr) { return *(const __attribute__((address_space(3))) float*)addr; };
  const unsigned maskAddr = ldsBase + G::MASK_OFFSET;
  const unsigned scAddr = ldsBase + G::PARAM_OFFSET, biAddr = scAddr + G::NTP * 4, nbAddr = biAddr + G::NTP * 4;
  if(!waveActive) return;
  const bool hasResid = a.resid != nullptr;               // uniform
  const bool hasNb = a.ncBias != nullptr;                 // uniform; only the stem convolution has a per-board bias
  const bool anyRaw = a.rawEnd > a.rawBegin, anyAct = a.actEnd > a.actBegin;  // uniform
  const int actKindRt = (ABL & ABL_EPI_NOACT) ? KMX_ACT_IDENTITY : a.actKind;  // uniform for the launch
  T* const rawBoard = (T*)a.rawOut + (size_t)n * S * a.rawC - a.rawBegin;
  T* const actBoard = (T*)a.actOut + (size_t)n * S * a.actC - a.actBegin;
  // Stores are UNCONDITIONAL: a piece that must not be written (a column beyond the board's cells, a channel outside the
  // output's range) goes to a 1 KiB trash area behind the zero page instead of being predicated off. Besides saving the
  // exec-mask branches, this makes the number of stores per tile a compile-time constant - with stores inside branches the
  // compiler waits for the residual loads below with s_waitcnt vmcnt(0), i.e. for every store issued before them to be
  // acknowledged by memory, once per tile.
  T* const trash = (T*)((char*)const_cast<void*>(a.zeroPage) + ZERO_PAGE_BYTES) + lane * 8;
  int cellOfTile[MTL];
#pragma unroll
  for(int pt = 0; pt < MTL; pt++) {
    const int rc = wm * (32 * MT) + (pt0 + pt) * 32 + myPos;
    cellOfTile[pt] = cellOf(rc < S ? rc : S - 1);
  }
  float keep = 0.0f;  // ABL_NO_EPILOGUE: keeps the accumulators observable
  // The residual stream (trunk += conv(...), eigenbackend.cpp:659-686) is fetched HERE and added in fp32 before the rounding:
  // 16-byte pieces in the layout of the stores (lane pair (c, c + 32): channels 16 j + 8 h + [0, 8)), branch-free
  // (out-of-range pieces read the zero page), requested one (cell tile, channel tile) ahead of their use - two pieces in
  // flight besides the two being consumed; the accumulators leave no room for more (256 registers per lane at two waves per
  // SIMD). As the accumulators' initial value these loads sat in the prologue, on the critical path of every work-group at
  // the moment all of them pull their first operands from HBM. The body is instantiated with and without a residual so that
  // the launches without one carry no load bookkeeping at all.
  auto epilogue = [&](auto kindTag, auto residTag) {
  constexpr int KIND = decltype(kindTag)::value;
  constexpr bool RESID = decltype(residTag)::value != 0;
  u32x4 rq[2][2];
  auto loadResid = [&](int pt, int ct, u32x4 (&dst)[2]) {
    const T* const rrow = (const T*)a.resid + ((size_t)n * S + cellOfTile[pt]) * a.residC - a.rawBegin;
#pragma unroll
    for(int j = 0; j < 2; j++) {
      const int c = cout0 + wn * (32 * WN) + ct * 32 + 16 * j + 8 * khalf;
      const T* src = (c >= a.rawBegin && c < a.rawEnd) ? rrow + c : (const T*)zero;
      dst[j] = *(const u32x4*)src;
    }
  };
  if(RESID) loadResid(0, 0, rq[0]);
#pragma unroll
  for(int pt = 0; pt < MTL; pt++) {
    const int cellBase = wm * (32 * MT) + (pt0 + pt) * 32;
    if(cellBase >= S) break;  // wave-uniform
    if(ABL & ABL_NO_EPILOGUE) {
#pragma unroll
      for(int ct = 0; ct < WN; ct++)
#pragma unroll
        for(int r = 0; r < 16; r++) keep += acc[ct][pt][r];
      continue;
    }
    const int col = cellBase + myPos;
    const bool live = col < S;                 // the same for both lanes of a pair
    const int cell = cellOfTile[pt];
    // off-board cells of the activated image are ZERO whatever the arithmetic gave (the raw residual stream is never masked
    // off the board, and an overflowed fp16 value there must not reach the halo of the next 3x3 convolution as inf * 0 =
    // NaN): the result bits are ANDed with an all-ones / all-zeros word - no select, no branch per element
    const unsigned onBits = ldsF1(maskAddr + cell * 4) == 1.0f ? 0xffffffffu : 0u;
    T* const rawRow = rawBoard + (size_t)cell * a.rawC;
    T* const actRow = actBoard + (size_t)cell * a.actC;
#pragma unroll
    for(int ct = 0; ct < WN; ct++) {
      const int chTile = wn * (32 * WN) + ct * 32;  // first channel of this 32-channel tile inside the work-group's channels
      // the parameters are re-read from LDS for every tile (broadcast reads, cheap): kept across the cell tiles they would
      // occupy 96 registers next to the accumulators and spill. The empty asm hides the address from common-subexpression
      // elimination.
      unsigned pOff = (unsigned)(chTile + 4 * khalf) * 4u;
      asm volatile("" : "+v"(pOff));
      u32x2 rp[4], op[4];
      u32x2 resP[4];
      if(RESID) {
        constexpr int NTILES = MTL * WN;
        const int k = pt * WN + ct;
        // (a tile past the board still requests its pieces - of the last cell - so that the count in flight is a constant)
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
      // this lane's two 16-byte pieces of the tile: channels chTile + 16 j + 8 h + [0, 8)
      if(RESID || anyRaw) {
        u32x4 rawQ[2];
        pairUp(rp, rawQ);
#pragma unroll
        for(int j = 0; j < 2; j++) {
          const int c = cout0 + chTile + 16 * j + 8 * khalf;
          T* const dst = (live && c >= a.rawBegin && c < a.rawEnd) ? rawRow + c : trash;
          if(ABL & ABL_EPI_NOSTORE) asm volatile("" ::"v"(rawQ[j]));
          else *(u32x4*)dst = rawQ[j];
        }
      }
      if(RESID || anyAct) {
        u32x4 actQ[2];
        pairUp(op, actQ);
#pragma unroll
        for(int j = 0; j < 2; j++) {
          const int c = cout0 + chTile + 16 * j + 8 * khalf;
          T* const dst = (live && c >= a.actBegin && c < a.actEnd) ? actRow + c : trash;
          if(ABL & ABL_EPI_NOSTORE) asm volatile("" ::"v"(actQ[j]));
          else *(u32x4*)dst = actQ[j];
        }
      }
    }
  }
  };
  withActKind(actKindRt, [&](auto kindTag) {
    if(hasResid) epilogue(kindTag, ActKindTag<1>());
    else epilogue(kindTag, ActKindTag<0>());
  });
  if(ABL & ABL_NO_EPILOGUE) {
    if(keep == 12345.678f) ((float*)a.actOut)[lane] = keep;
  }
  if((ABL & ABL_TIMING) && a.dbg != nullptr && lane == 0 && blockIdx.x == 0 && (int)blockIdx.y == a.N / 2) {
    // one record of 8 counters per wave of the middle board's first work-group
    unsigned long long* rec = a.dbg + wave * 8;
    const unsigned long long tEnd = __builtin_readcyclecounter();
    for(int i = 0; i < 4; i++) rec[i] = seg[i];
    rec[4] = tLoop1 - tLoop0;
    rec[5] = tLoop0 - tKernel0;
    rec[6] = tEnd - tLoop1;
    rec[7] = tEnd - tKernel0;
  }
}

template <class TR, int KS, int WN, int WNW, int D, int ABL>
hipError_t launchOne(const ConvArgs& a, hipStream_t stream) {
  typedef Geom<KS, WN, WNW, D> G;
  constexpr int ldsBytes = G::LDS_BYTES;
  static_assert(ldsBytes <= 160 * 1024, "LDS budget exceeded");
  static_assert(!G::SPREAD || G::LS + (G::ROLES ? 1 : D) <= G::NT, "image pieces must land within their chunk");
  auto kern = convMfmaKernel<TR, KS, WN, WNW, D, ABL>;
  // The opt-in to more than 64 KiB of dynamic LDS is a property of the function ON ONE DEVICE, and one process may hold
  // handles on several GPUs (the reference runs one server thread per GPU in a single process): one flag per
  // instantiation and device. The call is idempotent, so a race between two handles' first launches is harmless.
  constexpr int MAX_DEVICES = 64;
  static std::atomic<bool> attrSet[MAX_DEVICES];
  int dev = 0;
  hipError_t de = hipGetDevice(&dev);
  if(de != hipSuccess) return de;
  if(dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
  if(!attrSet[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    if(e != hipSuccess) return e;
    attrSet[dev].store(true, std::memory_order_release);
  }
  if(a.coutPad % G::NTILE != 0) return hipErrorInvalidValue;
  dim3 grid(a.coutPad / G::NTILE, a.N, (ABL & ABL_SPLIT) ? MT : 1);
  hipLaunchKernelGGL(kern, grid, dim3(G::NTHREADS), ldsBytes, stream, a);
  return hipGetLastError();
}

}  // namespace convk
}  // namespace kmx
#endif

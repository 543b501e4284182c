// conv_kernel.h — the hot kernel: im2col-free implicit-GEMM convolution on gfx950 matrix cores.
// (device template; instantiated by conv_mfma.hip for the product and by conv_bench.hip for ablations)
//
// Replaces, for one layer, the reference's ConvLayer::apply (eigenbackend.cpp:293-703; on CUDA a
// cudnnConvolutionForward / cublas GEMM, cudabackend.cpp:531-844) fused with the masked
// BatchNorm+activation that follows it (eigenbackend.cpp:739-762), the per-board bias add
// (:137-148) and the residual accumulate (:659-686).
//
// Decomposition (one work-group = one BOARD x (64*WN) output channels, 8 waves = 4(M) x 2(N)):
//   D[cout][cell] += sum_{tap, cin} W[tap][cout][cin] * X[cell + tap][cin]
//   - MFMA v_mfma_f32_32x32x16_{f16,bf16}; the A operand is the WEIGHT tile (rows = cout), the B
//     operand the ACTIVATION tile (cols = board cells), so each lane ends up holding 4 consecutive
//     output channels of one cell (8-byte NHWC stores) instead of 16 cells of one channel.
//   - K loop: input-channel chunks of 32 (outer) x filter taps (inner) = "steps". Per chunk the board's
//     activations INCLUDING a zero halo live in LDS as [cell][32ch] rows of 80 bytes; every tap reads
//     the same image at a constant byte offset — no im2col, no per-tap global traffic.
//   - both LDS images are filled by global_load_lds (LDS-DMA, 16 B/lane): their layout is a plain
//     linear copy of the HBM layout (weights are pre-tiled by the engine; halo cells and the 16 pad
//     bytes of each row are sourced from a zero page), so no VGPR staging and no ds_write.
//   - 80-byte rows: 16 consecutive rows x 16 B cover all 64 banks once -> ds_read_b128 conflict-free.
//   - software pipeline of depth D: the weight slab of step s+D is requested at the top of step s (ring of
//     D+1 LDS slabs). For 3x3/5x5 the next chunk's board image is requested one DMA instruction per step
//     into the second image buffer; for 1x1 (one step per chunk) whole images ride the same ring.
//     Every wave issues the SAME number of DMA instructions per step (padding with dummies into a slack
//     area), so that one compile-time s_waitcnt vmcnt(N) retires exactly the data of the current step
//     while D-1 steps of requests stay in flight across the single s_barrier per step.
#ifndef KMX_CONV_KERNEL_H_
#define KMX_CONV_KERNEL_H_

#include "device_common.h"

namespace kmx {
namespace convk {

constexpr int ROWB = WROW_HALFS * 2;  // 80 bytes per LDS row
constexpr int MT = 3;                 // board-cell tiles (of 32) per wave: 4 waves x 96 = 384 >= 361
constexpr int NWAVES = 8;
constexpr int NTHREADS = NWAVES * 64;
constexpr int MAXLEN = 19;
constexpr int SLACK_BYTES = NWAVES * 1024;
constexpr int MASK_BYTES = NWAVES * 64 * 4;  // the board's mask (<= 361 floats) copied once per work-group

// ablation switches (conv_bench.hip only; 0 in the product)
enum { ABL_NO_EPILOGUE = 1, ABL_NO_COMPUTE = 2, ABL_NO_DMA = 4, ABL_NO_LDS_READ = 8, ABL_SETPRIO = 16, ABL_DIRECT_EPILOGUE = 32, ABL_ORDER2 = 64, ABL_SGB = 128, ABL_PINGPONG = 256, ABL_EPI_NOACT = 512, ABL_EPI_NOSTORE = 1024 };

template <int KS>
struct ConvGeom {
  static constexpr int HALO = KS / 2;
  static constexpr int NT = KS * KS;
  static constexpr int HPMAX = (MAXLEN + 2 * HALO) * (MAXLEN + 2 * HALO);
  static constexpr int NPA = (HPMAX * 5 + NTHREADS - 1) / NTHREADS;  // DMA instructions per wave per board image
  static constexpr int ACT_BYTES = (HPMAX * ROWB + 1023) / 1024 * 1024;  // DMA instructions wholly past it go to the slack
};
template <int WN>
struct WGeom {
  static constexpr int NTILE = 64 * WN;
  static constexpr int PIECES = NTILE * 5;
  static constexpr int NPW = (PIECES + NTHREADS - 1) / NTHREADS;
  static constexpr int W_BYTES = NPW * NWAVES * 1024;
};
template <int KS, int WN, int D>
struct Pipe {
  typedef ConvGeom<KS> G;
  typedef WGeom<WN> WG;
  static constexpr bool SPREAD = G::NT >= G::NPA + D;  // image pieces fit one per step within a chunk
  static constexpr int NSA = SPREAD ? 2 : D + 1;
  static constexpr int NSW = D + 1;
  static constexpr int PIPE_BYTES = NSA * G::ACT_BYTES + NSW * WG::W_BYTES + SLACK_BYTES;
  static constexpr int MASK_OFFSET = (PIPE_BYTES > NWAVES * 32 * (32 * WN + 4) * 4) ? PIPE_BYTES : NWAVES * 32 * (32 * WN + 4) * 4;
  static constexpr int STAGE_BYTES = NWAVES * 32 * (32 * WN + 4) * 4;  // epilogue transpose, reuses the same LDS
  static constexpr int LDS_BYTES = MASK_OFFSET + MASK_BYTES;  // mask tile sits past everything the epilogue reuses
  // DMA instructions younger than the data of the current step when it is waited for
  static constexpr int VMCNT = SPREAD ? 1 + (D - 1) * (WG::NPW + 1) : (D - 1) * (WG::NPW + G::NPA);
};

template <int N>
__device__ __forceinline__ void waitVm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void dma16(const void* gsrc, char* ldsWaveBase) {
  __builtin_amdgcn_global_load_lds(
    (const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)ldsWaveBase, 16, 0, 0);
}

template <class TR, int KS, int WN, int D, int ABL>
__global__ __launch_bounds__(NTHREADS) void convMfmaKernel(const ConvArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  typedef typename TR::V4 V4;
  typedef ConvGeom<KS> G;
  typedef WGeom<WN> WG;
  typedef Pipe<KS, WN, D> P;
  constexpr int HALO = G::HALO, NT = G::NT, NPA = G::NPA, NPW = WG::NPW;
  constexpr bool SPREAD = P::SPREAD;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const bufA = smem;
  char* const bufW = smem + P::NSA * G::ACT_BYTES;
  char* const slack = bufW + P::NSW * WG::W_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int n = blockIdx.y;
  const int cout0 = blockIdx.x * WG::NTILE;
  const int X = a.X, Y = a.Y, S = X * Y;
  const int W2 = X + 2 * HALO, HP = W2 * (Y + 2 * HALO);
  const int inC = a.inC;

  const T* const inBoard = (const T*)a.in + (size_t)n * S * inC;
  const char* const zero = (const char*)a.zeroPage;
  char* const mySlack = slack + wave * 1024;

  // ---- per-lane DMA source offsets of the board image (in T elements; -1 = zero page) ----
  int srcOff[NPA];
#pragma unroll
  for(int j = 0; j < NPA; j++) {
    int p = (j * NWAVES + wave) * 64 + lane;
    int hp = p / 5;
    int slot = p - hp * 5;
    int off = -1;
    if(hp < HP && slot < 4) {
      int hy = hp / W2;
      int hx = hp - hy * W2;
      int y = hy - HALO, x = hx - HALO;
      if(y >= 0 && y < Y && x >= 0 && x < X) off = (y * X + x) * inC + slot * 8;
    }
    srcOff[j] = off;
  }
  const char* const wBase = (const char*)a.w + (size_t)cout0 * ROWB;
  const size_t wSlabStride = (size_t)a.coutPad * ROWB;
  const int nChunks = a.nChunks;
  const int nSteps = nChunks * NT;

  // Every call issues exactly NPW instructions (a dummy slab when `step` is past the end).
  auto issueW = [&](int step) {
    if(ABL & ABL_NO_DMA) return;
    const bool live = step < nSteps;
    const char* slab = wBase + (size_t)step * wSlabStride;
    char* dst = bufW + (step % P::NSW) * WG::W_BYTES;
#pragma unroll
    for(int j = 0; j < NPW; j++) {
      int pbase = (j * NWAVES + wave) * 64;
      int p = pbase + lane;
      const char* src = (live && p < WG::PIECES) ? slab + (size_t)p * 16 : zero;
      dma16(src, live ? dst + pbase * 16 : mySlack);
    }
  };
  // One instruction of the board image of `chunk` (piece j), or a dummy when chunk is past the end / j >= NPA.
  auto issueA = [&](int chunk, int j, int off) {
    if(ABL & ABL_NO_DMA) return;
    int pbase = (j * NWAVES + wave) * 64;
    const bool live = chunk < nChunks && j < NPA && pbase * 16 < G::ACT_BYTES;
    const char* src = (live && off >= 0) ? (const char*)(inBoard + off + chunk * KCHUNK) : zero;
    dma16(src, live ? bufA + (chunk % P::NSA) * G::ACT_BYTES + pbase * 16 : mySlack);
  };

  // ---- per-lane LDS read offsets ----
  const int khalf = (lane >> 5) * 16;
  const int wOff = (wn * (32 * WN) + (lane & 31)) * ROWB + khalf;
  int aOff[MT];
#pragma unroll
  for(int pt = 0; pt < MT; pt++) {
    int j = wm * (32 * MT) + pt * 32 + (lane & 31);
    j = j < S ? j : S - 1;  // rows beyond the board recompute the last cell; never stored
    int y = j / X;
    int x = j - y * X;
    aOff[pt] = ((y + HALO) * W2 + (x + HALO)) * ROWB + khalf;
  }
  const bool waveActive = wm * (32 * MT) < S;

  // Accumulators start from the residual stream (trunk += conv(...), eigenbackend.cpp:659-686) instead of zero:
  // the only global LOADS of the kernel besides the DMA are issued here, underneath the pipeline fill, so that the
  // epilogue consists of stores alone. (vmcnt counts stores as well as loads: an epilogue that alternates residual
  // loads and stores pays one full memory round trip per row; measured 2-5x the K loop.)
  f32x16 acc[WN][MT];
#pragma unroll
  for(int ct = 0; ct < WN; ct++)
#pragma unroll
    for(int pt = 0; pt < MT; pt++)
#pragma unroll
      for(int r = 0; r < 16; r++) acc[ct][pt][r] = 0.0f;
  if(a.resid != nullptr && waveActive && !(ABL & ABL_DIRECT_EPILOGUE)) {
#pragma unroll
    for(int pt = 0; pt < MT; pt++) {
      const int cell = wm * (32 * MT) + pt * 32 + (lane & 31);
      if(cell < S) {
        const T* rrow = (const T*)a.resid + ((size_t)n * S + cell) * a.residC;
#pragma unroll
        for(int ct = 0; ct < WN; ct++)
#pragma unroll
          for(int g = 0; g < 4; g++) {
            const int c = cout0 + wn * (32 * WN) + ct * 32 + 8 * g + 4 * (lane >> 5);
            if(c >= a.rawBegin && c < a.rawEnd) {
              const V4 rr = *(const V4*)(rrow + (c - a.rawBegin));
#pragma unroll
              for(int i = 0; i < 4; i++) acc[ct][pt][4 * g + i] = TR::toFloat(rr[i]);
            }
          }
      }
    }
  }

  // ---- prologue: fill the pipeline with the same per-step instruction pattern the loop uses ----
  // first (oldest) request: this board's mask, 4 bytes per lane, read back from LDS by the epilogue
  {
    const int cellIdx = wave * 64 + lane;
    const float* msrc = cellIdx < S ? a.mask + (size_t)n * S + cellIdx : (const float*)zero;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)msrc,
                                     (__attribute__((address_space(3))) void*)(smem + P::MASK_OFFSET + wave * 256), 4, 0, 0);
  }
  if(SPREAD) {
#pragma unroll
    for(int j = 0; j < NPA; j++) issueA(0, j, srcOff[j]);
#pragma unroll
    for(int s = 0; s < D; s++) {
      issueW(s);
      issueA(nChunks, 0, -1);  // dummy: keeps the per-step DMA count constant
    }
  }
  else {
#pragma unroll
    for(int s = 0; s < D; s++) {
      issueW(s);
#pragma unroll
      for(int j = 0; j < NPA; j++) issueA(s, j, srcOff[j]);
    }
  }

  // ---- ping-pong schedule (ABL_PINGPONG) --------------------------------------------------------------------------
  // With one barrier per step all eight waves run [requests, fragment reads, 18 MFMAs] in phase, and the matrix pipe
  // idles while both waves of a SIMD sit in their load segment (measured: MFMA busy 29 %, 44 % of wave time in waits).
  // Ping-pong keeps ONE instruction stream but splits every step in two segments, [requests + fragment reads] and
  // [MFMAs], each behind its own barrier, and lets waves 4-7 (the SIMD partners of waves 0-3) enter the loop one
  // barrier late: whenever one wave of a SIMD is in a load segment its partner is in an MFMA segment. The waits
  // precede BOTH barriers because the late group must have its share of slab s landed by the barrier that releases the
  // early group's reads of slab s, which is the late group's pre-MFMA barrier; this needs D >= 2.
  constexpr bool PP = (ABL & ABL_PINGPONG) != 0;
  static_assert(!PP || D >= 2, "ping-pong needs a pipeline depth of at least 2");
  const int grp = wave >> 2;
  if(PP && grp == 1) __builtin_amdgcn_s_barrier();
  {
  int step = 0;
  for(int chunk = 0; chunk < nChunks; chunk++) {
    const char* const curA = bufA + (chunk % P::NSA) * G::ACT_BYTES;
#pragma unroll
    for(int t = 0; t < NT; t++, step++) {
      // (1) this step's slab (and image) has landed: own DMAs by counted vmcnt, everybody's by the barrier
      if(!(ABL & ABL_NO_DMA)) waitVm<P::VMCNT>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");

      // (2) requests for step + D   (3) MFMA over the 32 input channels of this (chunk, tap)
      auto requests = [&]() {
        issueW(step + D);
        if(SPREAD) {
          issueA(t < NPA ? chunk + 1 : nChunks, t < NPA ? t : 0, srcOff[t < NPA ? t : 0]);
        }
        else {
#pragma unroll
          for(int j = 0; j < NPA; j++) issueA(chunk + D, j, srcOff[j]);
        }
      };
      if(waveActive && !(ABL & ABL_NO_COMPUTE)) {
        if(!(ABL & ABL_ORDER2)) requests();
        const int dy = t / KS - HALO, dx = t % KS - HALO;
        const char* const aTap = curA + (dy * W2 + dx) * ROWB;
        const char* const wCur = bufW + (step % P::NSW) * WG::W_BYTES + wOff;
        // all fragments of the step are requested up front: LDS returns in order, so the second k-half's reads
        // complete underneath the first half's MFMAs (counted lgkmcnt) instead of stalling the matrix pipe twice
        V8 wf[2][WN];
        V8 af[2][MT];
#pragma unroll
        for(int kk = 0; kk < 2; kk++) {
          if(ABL & ABL_NO_LDS_READ) {
#pragma unroll
            for(int ct = 0; ct < WN; ct++)
#pragma unroll
              for(int i = 0; i < 8; i++) wf[kk][ct][i] = (T)(0.001f * (float)(lane + ct));
#pragma unroll
            for(int pt = 0; pt < MT; pt++)
#pragma unroll
              for(int i = 0; i < 8; i++) af[kk][pt][i] = (T)(0.002f * (float)(lane + pt));
          }
          else {
#pragma unroll
            for(int ct = 0; ct < WN; ct++) wf[kk][ct] = *(const V8*)(wCur + ct * 32 * ROWB + kk * 32);
#pragma unroll
            for(int pt = 0; pt < MT; pt++) af[kk][pt] = *(const V8*)(aTap + aOff[pt] + kk * 32);
          }
        }
        // ORDER2: the DMA requests come after the fragment reads in program order (they never touch the buffers
        // being read), so the scheduler may sink their address arithmetic and issue slots underneath the MFMAs
        if(ABL & ABL_ORDER2) requests();
        if(PP) {
          waitVm<P::VMCNT>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragment reads retire in the load segment, not the MFMA one
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        if(ABL & (ABL_SETPRIO | ABL_PINGPONG)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for(int kk = 0; kk < 2; kk++)
#pragma unroll
          for(int ct = 0; ct < WN; ct++)
#pragma unroll
            for(int pt = 0; pt < MT; pt++) acc[ct][pt] = TR::mfma(wf[kk][ct], af[kk][pt], acc[ct][pt]);
        if(ABL & (ABL_SETPRIO | ABL_PINGPONG)) __builtin_amdgcn_s_setprio(0);
        if(ABL & ABL_SGB) {
          // one MFMA, then a slice of the request code, repeated; the tail of the MFMAs follows
#pragma unroll
          for(int r = 0; r < 3; r++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);  // VALU
            __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);  // SALU
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // VMEM
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * WN * MT - 3, 0);
        }
      }
      else {
        requests();
        if(PP) {
          waitVm<P::VMCNT>();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      }
    }
  }
  }  // schedule
  if(PP && grp == 0) __builtin_amdgcn_s_barrier();
  if(!(ABL & ABL_NO_DMA)) waitVm<0>();  // retire the trailing dummies before the LDS is reused / the wave exits

  // ---- epilogue ----
  // After the MFMA chain a lane holds, per (ct,pt) tile, 4 consecutive channels of one cell — 8-byte pieces
  // scattered over 64 cache lines per instruction, which measured 2-5x the duration of the whole K loop. Instead each
  // wave transposes one 32-cell tile at a time through its private slice of the (now idle) LDS in fp32 and then
  // walks it row-wise: 16-byte loads/stores, 4*WN consecutive lanes per cell, i.e. whole contiguous runs of the
  // NHWC rows, for the residual read and for both outputs.
  const float* const maskBoard = (const float*)(smem + P::MASK_OFFSET);
  if(ABL & ABL_DIRECT_EPILOGUE) {
    if(!waveActive) return;
#pragma unroll
    for(int pt = 0; pt < MT; pt++) {
      const int cell = wm * (32 * MT) + pt * 32 + (lane & 31);
      if(cell >= S) continue;
      const size_t gcell = (size_t)n * S + cell;
      const float maskVal = maskBoard[cell];
#pragma unroll
      for(int ct = 0; ct < WN; ct++) {
#pragma unroll
        for(int g = 0; g < 4; g++) {
          const int c = cout0 + wn * (32 * WN) + ct * 32 + 8 * g + 4 * (lane >> 5);
          float v[4];
#pragma unroll
          for(int i = 0; i < 4; i++) v[i] = acc[ct][pt][4 * g + i];
          if(a.ncBias != nullptr) {
            const float4 b = *(const float4*)(a.ncBias + (size_t)n * a.ncBiasStride + c);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
          }
          if(c >= a.rawBegin && c < a.rawEnd) {
            if(a.resid != nullptr) {
              const V4 rr = *(const V4*)((const T*)a.resid + gcell * a.residC + (c - a.rawBegin));
#pragma unroll
              for(int i = 0; i < 4; i++) v[i] += TR::toFloat(rr[i]);
            }
            V4 o;
#pragma unroll
            for(int i = 0; i < 4; i++) o[i] = TR::fromFloat(v[i]);
            *(V4*)((T*)a.rawOut + gcell * a.rawC + (c - a.rawBegin)) = o;
          }
          if(c >= a.actBegin && c < a.actEnd) {
            const float4 sc = *(const float4*)(a.scale + c);
            const float4 bi = *(const float4*)(a.bias + c);
            V4 o;
            o[0] = TR::fromFloat(actApply(v[0] * sc.x + bi.x, a.actKind) * maskVal);
            o[1] = TR::fromFloat(actApply(v[1] * sc.y + bi.y, a.actKind) * maskVal);
            o[2] = TR::fromFloat(actApply(v[2] * sc.z + bi.z, a.actKind) * maskVal);
            o[3] = TR::fromFloat(actApply(v[3] * sc.w + bi.w, a.actKind) * maskVal);
            *(V4*)((T*)a.actOut + gcell * a.actC + (c - a.actBegin)) = o;
          }
        }
      }
    }
    return;
  }

  constexpr int ROWF = 32 * WN + 4;         // floats per staged row (+16 B: rows 16 B apart mod 128 -> conflict-free b128 writes)
  constexpr int STAGE_FLOATS = 32 * ROWF;   // one 32-cell x (32*WN)-channel tile per wave
  static_assert(NWAVES * STAGE_FLOATS * 4 <= P::LDS_BYTES, "epilogue staging does not fit the LDS");
  constexpr int PPC = 4 * WN;               // 8-channel pieces per cell
  constexpr int CPI = 64 / PPC;             // cells covered by one wave-wide access
  constexpr int NIT = (32 + CPI - 1) / CPI;
  const int pk = lane % PPC, pc = lane / PPC;
  const bool laneOn = pc < CPI;
  const int c8 = cout0 + wn * (32 * WN) + pk * 8;  // first of this lane's 8 output channels (fixed for all cells)
  const bool inRaw = c8 >= a.rawBegin && c8 < a.rawEnd;
  const bool inAct = c8 >= a.actBegin && c8 < a.actEnd;
  __builtin_amdgcn_s_barrier();             // every wave is done reading the operand images
  asm volatile("" ::: "memory");
  if(!waveActive) return;
  float* const stage = (float*)smem + wave * STAGE_FLOATS;
  float sc[8], bi[8], nb[8];
#pragma unroll
  for(int i = 0; i < 8; i++) {
    sc[i] = inAct ? a.scale[c8 + i] : 0.0f;
    bi[i] = inAct ? a.bias[c8 + i] : 0.0f;
    nb[i] = a.ncBias != nullptr ? a.ncBias[(size_t)n * a.ncBiasStride + c8 + i] : 0.0f;
  }
  {
    // Consume the parameter loads HERE, while no store is in flight: otherwise the compiler re-waits for them with
    // vmcnt(0) inside every row iteration below, and each of those waits then also drains the previous row's stores.
    float touch = 0.0f;
#pragma unroll
    for(int i = 0; i < 8; i++) touch += sc[i] + bi[i] + nb[i];
    asm volatile("" ::"v"(touch));
  }
#pragma unroll
  for(int pt = 0; pt < MT; pt++) {
    const int cellBase = wm * (32 * MT) + pt * 32;
    if(cellBase >= S) break;  // wave-uniform
    // (a) accumulators -> LDS, [cell][channel] fp32
#pragma unroll
    for(int ct = 0; ct < WN; ct++)
#pragma unroll
      for(int g = 0; g < 4; g++) {
        f32x4 v;
#pragma unroll
        for(int i = 0; i < 4; i++) v[i] = acc[ct][pt][4 * g + i];
        *(f32x4*)(stage + (lane & 31) * ROWF + ct * 32 + 8 * g + 4 * (lane >> 5)) = v;
      }
    if(ABL & ABL_NO_EPILOGUE) continue;
    // (b) row-wise walk: lane -> (cell pc of this group, 8 channels pk)
#pragma unroll
    for(int it = 0; it < NIT; it++) {
      const int cl = it * CPI + pc;
      const int cell = cellBase + cl;
      if(!laneOn || cl >= 32 || cell >= S) continue;
      const size_t gcell = (size_t)n * S + cell;
      const f32x4 lo = *(const f32x4*)(stage + cl * ROWF + pk * 8);
      const f32x4 hi = *(const f32x4*)(stage + cl * ROWF + pk * 8 + 4);
      float v[8];
#pragma unroll
      for(int i = 0; i < 4; i++) {
        v[i] = lo[i] + nb[i];
        v[4 + i] = hi[i] + nb[4 + i];
      }
      if(inRaw) {
        V8 o;
#pragma unroll
        for(int i = 0; i < 8; i++) o[i] = TR::fromFloat(v[i]);
        if(ABL & ABL_EPI_NOSTORE) asm volatile("" ::"v"(o)); else
        *(V8*)((T*)a.rawOut + gcell * a.rawC + (c8 - a.rawBegin)) = o;
      }
      if(inAct) {
        const float maskVal = maskBoard[cell];
        V8 o;
        // the activation kind is uniform for the launch: branch ONCE per row, not per element (a per-element switch
        // compiles to every activation being evaluated and selected - measured 12 us of a 20 us epilogue)
        const int kind = (ABL & ABL_EPI_NOACT) ? KMX_ACT_IDENTITY : a.actKind;
        if(kind == KMX_ACT_MISH) {
#pragma unroll
          for(int i = 0; i < 8; i++) o[i] = TR::fromFloat(actMish(v[i] * sc[i] + bi[i]) * maskVal);
        }
        else if(kind == KMX_ACT_RELU) {
#pragma unroll
          for(int i = 0; i < 8; i++) o[i] = TR::fromFloat(fmaxf(v[i] * sc[i] + bi[i], 0.0f) * maskVal);
        }
        else if(kind == KMX_ACT_SILU) {
#pragma unroll
          for(int i = 0; i < 8; i++) o[i] = TR::fromFloat(actSilu(v[i] * sc[i] + bi[i]) * maskVal);
        }
        else {
#pragma unroll
          for(int i = 0; i < 8; i++) o[i] = TR::fromFloat((v[i] * sc[i] + bi[i]) * maskVal);
        }
        if(ABL & ABL_EPI_NOSTORE) asm volatile("" ::"v"(o)); else
        *(V8*)((T*)a.actOut + gcell * a.actC + (c8 - a.actBegin)) = o;
      }
    }
  }
  if(ABL & ABL_NO_EPILOGUE) {
    // keep the staged values observable so the compiler cannot drop the accumulators
    if(stage[lane] == 12345.678f) ((float*)a.actOut)[lane] = stage[lane + 1];
  }
}

template <class TR, int KS, int WN, int D, int ABL>
hipError_t launchOne(const ConvArgs& a, hipStream_t stream) {
  typedef WGeom<WN> WG;
  typedef Pipe<KS, WN, D> P;
  constexpr int ldsBytes = P::LDS_BYTES;
  static_assert(ldsBytes <= 160 * 1024, "LDS budget exceeded");
  static_assert(!P::SPREAD || ConvGeom<KS>::NPA + D <= ConvGeom<KS>::NT, "image pieces must land within their chunk");
  auto kern = convMfmaKernel<TR, KS, WN, D, ABL>;
  static bool attrSet = false;  // per instantiation; idempotent
  if(!attrSet) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes);
    if(e != hipSuccess) return e;
    attrSet = true;
  }
  if(a.coutPad % WG::NTILE != 0) return hipErrorInvalidValue;
  dim3 grid(a.coutPad / WG::NTILE, a.N, 1);
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), ldsBytes, stream, a);
  return hipGetLastError();
}

}  // namespace convk
}  // namespace kmx
#endif

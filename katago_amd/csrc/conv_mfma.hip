// conv_mfma.hip — the hot kernel: im2col-free implicit-GEMM convolution on gfx950 matrix cores.
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
//   - K loop: input-channel chunks of 32 (outer) x filter taps (inner). Per chunk the board's
//     activations INCLUDING a zero halo live in LDS as [cell][32ch] rows of 80 bytes; every tap reads
//     the same image at a constant byte offset — no im2col, no per-tap global traffic.
//   - both LDS images are filled by global_load_lds (LDS-DMA, 16 B/lane): their layout is a plain
//     linear copy of the HBM layout (weights are pre-tiled by the engine; halo cells and the 16 pad
//     bytes of each row are sourced from a zero page), so no VGPR staging and no ds_write.
//   - 80-byte rows: 16 consecutive rows x 16 B cover all 64 banks once -> ds_read_b128 conflict-free.
//   - double-buffered: weight slab for step s+1 and one DMA instruction of the next chunk's board image
//     are issued at the top of step s; each wave waits its own DMA with a counted vmcnt and one
//     s_barrier per step publishes it.
#include "device_common.h"

namespace kmx {

namespace {

constexpr int ROWB = WROW_HALFS * 2;  // 80 bytes per LDS row
constexpr int MT = 3;                 // board-cell tiles (of 32) per wave: 4 waves x 96 = 384 >= 361
constexpr int NWAVES = 8;
constexpr int NTHREADS = NWAVES * 64;
constexpr int MAXLEN = 19;

template <int KS>
struct ConvGeom {
  static constexpr int HALO = KS / 2;
  static constexpr int NT = KS * KS;
  static constexpr int HPMAX = (MAXLEN + 2 * HALO) * (MAXLEN + 2 * HALO);
  static constexpr int NPA = (HPMAX * 5 + NTHREADS - 1) / NTHREADS;  // DMA instructions per wave per board image
  static constexpr int ACT_BYTES = NPA * NWAVES * 1024;              // incl. slack so every wave issues NPA
};
template <int WN>
struct WGeom {
  static constexpr int NTILE = 64 * WN;
  static constexpr int PIECES = NTILE * 5;
  static constexpr int NPW = (PIECES + NTHREADS - 1) / NTHREADS;
  static constexpr int W_BYTES = NPW * NWAVES * 1024;
};

template <int N>
__device__ __forceinline__ void waitVm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void dma16(const void* gsrc, char* ldsWaveBase) {
  __builtin_amdgcn_global_load_lds(
    (const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)ldsWaveBase, 16, 0, 0);
}

template <class TR, int KS, int WN>
__global__ __launch_bounds__(NTHREADS) void convMfmaKernel(const ConvArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  typedef typename TR::V4 V4;
  typedef ConvGeom<KS> G;
  typedef WGeom<WN> WG;
  constexpr int HALO = G::HALO, NT = G::NT, NPA = G::NPA, NPW = WG::NPW;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const bufA = smem;                      // 2 x ACT_BYTES
  char* const bufW = smem + 2 * G::ACT_BYTES;   // 2 x W_BYTES

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
  // weight slab of one (chunk, tap) for this work-group: NTILE rows of 80 bytes, contiguous in HBM
  const char* const wBase = (const char*)a.w + (size_t)cout0 * ROWB;
  const size_t wSlabStride = (size_t)a.coutPad * ROWB;

  auto issueW = [&](int step, int buf) {
    const char* slab = wBase + (size_t)step * wSlabStride;
#pragma unroll
    for(int j = 0; j < NPW; j++) {
      int pbase = (j * NWAVES + wave) * 64;
      int p = pbase + lane;
      const char* src = (p < WG::PIECES) ? slab + (size_t)p * 16 : zero;
      dma16(src, bufW + buf * WG::W_BYTES + pbase * 16);
    }
  };
  auto issueA = [&](int chunk, int buf, int j, int off) {
    int pbase = (j * NWAVES + wave) * 64;
    const char* src = (off >= 0) ? (const char*)(inBoard + off + chunk * KCHUNK) : zero;
    dma16(src, bufA + buf * G::ACT_BYTES + pbase * 16);
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

  f32x16 acc[WN][MT];
#pragma unroll
  for(int ct = 0; ct < WN; ct++)
#pragma unroll
    for(int pt = 0; pt < MT; pt++)
#pragma unroll
      for(int r = 0; r < 16; r++) acc[ct][pt][r] = 0.0f;

  const int nChunks = a.nChunks;
  const int nSteps = nChunks * NT;

  // ---- prologue: whole board image of chunk 0 and the first weight slab ----
#pragma unroll
  for(int j = 0; j < NPA; j++) issueA(0, 0, j, srcOff[j]);
  issueW(0, 0);

  int step = 0;
  for(int chunk = 0; chunk < nChunks; chunk++) {
    const char* const curA = bufA + (chunk & 1) * G::ACT_BYTES;
    const bool moreChunks = chunk + 1 < nChunks;
#pragma unroll
    for(int t = 0; t < NT; t++, step++) {
      // (1) the data of this step has landed: this wave's own DMAs, then everybody's.
      //     With NT >= NPA + 2 the board-image instruction issued in the previous step may stay in flight
      //     (it is only needed NT - NPA >= 2 steps later); otherwise drain everything.
      if(NT >= NPA + 2 && t >= 1 && t <= NPA && moreChunks)
        waitVm<1>();
      else
        waitVm<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");

      // (2) prefetch: weight slab of the next step, one instruction of the next chunk's board image
      if(step + 1 < nSteps) issueW(step + 1, (step + 1) & 1);
      if(moreChunks) {
        if(NT >= NPA + 2) {
          if(t < NPA) issueA(chunk + 1, (chunk + 1) & 1, t, srcOff[t < NPA ? t : 0]);
        }
        else if(t == 0) {
#pragma unroll
          for(int j = 0; j < NPA; j++) issueA(chunk + 1, (chunk + 1) & 1, j, srcOff[j]);
        }
      }

      // (3) MFMA over the 32 input channels of this (chunk, tap)
      if(waveActive) {
        const int dy = t / KS - HALO, dx = t % KS - HALO;
        const char* const aTap = curA + (dy * W2 + dx) * ROWB;
        const char* const wCur = bufW + (step & 1) * WG::W_BYTES + wOff;
#pragma unroll
        for(int kk = 0; kk < 2; kk++) {
          V8 wf[WN];
          V8 af[MT];
#pragma unroll
          for(int ct = 0; ct < WN; ct++) wf[ct] = *(const V8*)(wCur + ct * 32 * ROWB + kk * 32);
#pragma unroll
          for(int pt = 0; pt < MT; pt++) af[pt] = *(const V8*)(aTap + aOff[pt] + kk * 32);
#pragma unroll
          for(int ct = 0; ct < WN; ct++)
#pragma unroll
            for(int pt = 0; pt < MT; pt++) acc[ct][pt] = TR::mfma(wf[ct], af[pt], acc[ct][pt]);
        }
      }
    }
  }

  // ---- epilogue: lane holds, per (ct,pt) tile, channels c0 + 8g + 4*(lane>>5) + {0..3} of cell (lane&31) ----
  if(!waveActive) return;
  const float* const maskBoard = a.mask + (size_t)n * S;
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
        const bool inRaw = c >= a.rawBegin && c < a.rawEnd;
        const bool inAct = c >= a.actBegin && c < a.actEnd;
        if(inRaw) {
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
        if(inAct) {
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
}

template <class TR, int KS, int WN>
hipError_t launchOne(const ConvArgs& a, hipStream_t stream) {
  typedef ConvGeom<KS> G;
  typedef WGeom<WN> WG;
  constexpr int ldsBytes = 2 * G::ACT_BYTES + 2 * WG::W_BYTES;
  static_assert(ldsBytes <= 160 * 1024, "LDS budget exceeded");
  auto kern = convMfmaKernel<TR, KS, WN>;
  static bool attrSet = false;  // per instantiation; set once per process (idempotent, cheap)
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

template <class TR>
hipError_t launchT(int ks, int wn, const ConvArgs& a, hipStream_t stream) {
  if(ks == 3) {
    if(wn == 1) return launchOne<TR, 3, 1>(a, stream);
    if(wn == 2) return launchOne<TR, 3, 2>(a, stream);
    if(wn == 3) return launchOne<TR, 3, 3>(a, stream);
  }
  else if(ks == 1) {
    if(wn == 1) return launchOne<TR, 1, 1>(a, stream);
    if(wn == 2) return launchOne<TR, 1, 2>(a, stream);
    if(wn == 3) return launchOne<TR, 1, 3>(a, stream);
  }
  else if(ks == 5) {
    if(wn == 1) return launchOne<TR, 5, 1>(a, stream);
    if(wn == 2) return launchOne<TR, 5, 2>(a, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace

hipError_t launchConv(int dtype, int ks, int wn, const ConvArgs& a, hipStream_t stream) {
  if(a.X < 2 || a.Y < 2 || a.X > MAXLEN || a.Y > MAXLEN || a.N <= 0) return hipErrorInvalidValue;
  if(a.inC % 8 != 0 || a.coutPad % 64 != 0) return hipErrorInvalidValue;
  if(dtype == DT_F16) return launchT<TraitsF16>(ks, wn, a, stream);
  if(dtype == DT_BF16) return launchT<TraitsBF16>(ks, wn, a, stream);
  return hipErrorInvalidValue;
}

int chooseConvWN(int ks, int coutPad, int batch) {
  const int tiles = coutPad / 64;
  const int maxWN = ks == 5 ? 2 : 3;
  // the widest tile (fewest re-reads of the board image) that still gives every CU a work-group
  for(int wn = maxWN; wn > 1; wn--)
    if(tiles % wn == 0 && (tiles / wn) * batch >= 200) return wn;
  return 1;
}

}  // namespace kmx

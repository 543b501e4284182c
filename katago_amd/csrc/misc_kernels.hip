// misc_kernels.hip — the small, HBM-bound kernels around the MFMA convolution: input staging
// (symmetry + 16-bit conversion + mask), global pooling + bias + BN/act, the policy and value head
// tails, and the stand-alone ops the layer test hooks need. One work-group (256 threads) per board;
// all arithmetic in fp32. Reference citations are at the launch structs in kernels.h.
#include <cstring>

#include "device_common.h"

namespace kmx {

namespace {

constexpr int BT = 256;  // threads per board work-group

template <class TR>
__device__ __forceinline__ float ldT(const void* base, size_t idx) {
  return TR::toFloat(((const typename TR::T*)base)[idx]);
}
template <class TR>
__device__ __forceinline__ void stT(void* base, size_t idx, float v) {
  ((typename TR::T*)base)[idx] = TR::fromFloat(v);
}

__device__ __forceinline__ float blockSum(float v, float* red) {
  // red: BT floats of LDS
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for(int s = BT / 2; s > 0; s >>= 1) {
    if(tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  float r = red[0];
  __syncthreads();
  return r;
}

// ------------------------------------------------------------------------------------------------
template <class TR>
__global__ __launch_bounds__(BT) void inputExpandKernel(const InputArgs a) {
  __shared__ float red[BT];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int S = a.X * a.Y;
  const int sym = a.symmetry ? a.symmetry[n] : 0;
  typename TR::T* dst = (typename TR::T*)a.out + (size_t)n * S * KCHUNK;
  float localMask = 0.0f;
  typedef typename TR::V8 V8;
  // Per cell: ALL of the cell's loads first (22 bytes, one per bit plane, or 22 consecutive floats), then four vector stores of the
  // 32-channel row. (Until round 5 this was a per-channel loop of a load and a 2-byte store: the stores may alias the loads as far as the
  // compiler knows, so every channel waited out its own memory round trip - 22 of them in a row, 33 us per launch whatever the batch.)
  if(a.packed != nullptr) {
    // bit planes: plane c, cell p = bit (7 - p%8) of byte p/8 (packBits, dataio/trainingwrite.cpp:314-337)
    const int PB = (S + 7) / 8;
    const unsigned char* src = a.packed + (size_t)n * a.cin * PB;
    for(int p = tid; p < S; p += BT) {
      const int h = p / a.X, w = p - h * a.X;
      const int q = symDst(h, w, a.Y, a.X, sym, false);
      const int byte = p >> 3, shift = 7 - (p & 7);
      unsigned char b[KCHUNK];
#pragma unroll
      for(int c = 0; c < KCHUNK; c++) b[c] = c < a.cin ? src[(size_t)c * PB + byte] : (unsigned char)0;
      V8 row[KCHUNK / 8];
#pragma unroll
      for(int c = 0; c < KCHUNK; c++) row[c >> 3][c & 7] = TR::fromFloat((float)((b[c] >> shift) & 1));
      typename TR::T* dp = dst + (size_t)q * KCHUNK;
#pragma unroll
      for(int k = 0; k < KCHUNK / 8; k++) *(V8*)(dp + 8 * k) = row[k];
      const float m = (float)((b[0] >> shift) & 1);  // mask = input channel 0 (eigenbackend.cpp:2181)
      a.mask[(size_t)n * S + q] = m;
      localMask += m;
    }
  }
  else {
    const float* src = a.spatial + (size_t)n * S * a.cin;
    for(int p = tid; p < S; p += BT) {
      const int h = p / a.X, w = p - h * a.X;
      const int q = symDst(h, w, a.Y, a.X, sym, false);
      const float* sp = src + (size_t)p * a.cin;
      float v[KCHUNK];
#pragma unroll
      for(int c = 0; c < KCHUNK; c++) v[c] = c < a.cin ? sp[c] : 0.0f;
      V8 row[KCHUNK / 8];
#pragma unroll
      for(int c = 0; c < KCHUNK; c++) row[c >> 3][c & 7] = TR::fromFloat(v[c]);
      typename TR::T* dp = dst + (size_t)q * KCHUNK;
#pragma unroll
      for(int k = 0; k < KCHUNK / 8; k++) *(V8*)(dp + 8 * k) = row[k];
      const float m = v[0];  // mask = input channel 0 (eigenbackend.cpp:2181)
      a.mask[(size_t)n * S + q] = m;
      localMask += m;
    }
  }
  const float ms = blockSum(localMask, red);
  if(tid == 0) a.maskSum[n] = ms;
  // ncBias[n][c] = sum_g global[n][g] * W[g][c]
  const float* gl = a.global + (size_t)n * a.gin;
  if(a.meta == nullptr) {
    for(int c = tid; c < a.C; c += BT) {
      float s = 0.0f;
      for(int g = 0; g < a.gin; g++) s += gl[g] * a.wGlobal[(size_t)g * a.C + c];
      a.ncBias[(size_t)n * a.ncStride + c] = s;
    }
    return;
  }
  // sgf-metadata encoder: a 3-layer MLP per board whose output joins the global-feature bias
  HIP_DYNAMIC_SHARED(float, metaSm)  // in[metaIn], h1[metaC1], h2[metaC2]
  float* in = metaSm;
  float* h1 = in + a.metaIn;
  float* h2 = h1 + a.metaC1;
  for(int k = tid; k < a.metaIn; k += BT) in[k] = a.meta[(size_t)n * a.metaIn + k];
  __syncthreads();
  for(int j = tid; j < a.metaC1; j += BT) {
    float s = a.mB1[j];
    for(int k = 0; k < a.metaIn; k++) s += in[k] * a.mW1[(size_t)k * a.metaC1 + j];
    h1[j] = actApply(s, a.metaAct1);
  }
  __syncthreads();
  for(int i = tid; i < a.metaC2; i += BT) {
    float s = a.mB2[i];
    for(int j = 0; j < a.metaC1; j++) s += h1[j] * a.mW2[(size_t)j * a.metaC2 + i];
    h2[i] = actApply(s, a.metaAct2);
  }
  __syncthreads();
  for(int c = tid; c < a.C; c += BT) {
    float s = 0.0f;
    for(int g = 0; g < a.gin; g++) s += gl[g] * a.wGlobal[(size_t)g * a.C + c];
    float t = 0.0f;
    for(int i = 0; i < a.metaC2; i++) t += h2[i] * a.mW3[(size_t)i * a.C + c];
    a.ncBias[(size_t)n * a.ncStride + c] = s + t;
  }
}

// ------------------------------------------------------------------------------------------------
template <class TR>
__global__ __launch_bounds__(BT) void gpoolApplyKernel(const GPoolArgs a) {
  HIP_DYNAMIC_SHARED(float, sm)
  // layout: partSum[BT], partMax[BT], feat[3G], biasv[R]
  float* partSum = sm;
  float* partMax = sm + BT;
  float* feat = sm + 2 * BT;
  float* biasv = feat + 3 * a.G;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int S = a.S, G = a.G, R = a.R;
  const float* maskB = a.mask + (size_t)n * S;
  const size_t cell0 = (size_t)n * S;

  // phase 1: per-channel sum and max over the board. Threads = (cell group) x (channel).
  // groups = BT / Gc where Gc = channels handled per pass (<= BT).
  for(int c0 = 0; c0 < G; c0 += BT) {
    const int Gc = (G - c0) < BT ? (G - c0) : BT;
    int groups = BT / Gc;
    if(groups < 1) groups = 1;
    const int c = tid % Gc, grp = tid / Gc;
    float s = 0.0f, m = -1.0f;  // -1 init + (mask-1) trick: eigenbackend.cpp:156-166
    if(grp < groups) {
      for(int p = grp; p < S; p += groups) {
        const float x = ldT<TR>(a.g, (cell0 + p) * a.gStride + a.gOffset + c0 + c);
        s += x;
        m = fmaxf(m, x + (maskB[p] - 1.0f));
      }
    }
    partSum[tid] = s;
    partMax[tid] = m;
    __syncthreads();
    if(tid < Gc) {
      float ts = 0.0f, tm = -1.0f;
      for(int g = 0; g < groups; g++) {
        ts += partSum[g * Gc + tid];
        tm = fmaxf(tm, partMax[g * Gc + tid]);
      }
      const float div = a.maskSum[n];
      const float sqrtdiv = sqrtf(div);
      const float mean = ts / div;
      feat[c0 + tid] = mean;
      feat[G + c0 + tid] = mean * (sqrtdiv - 14.0f) * 0.1f;
      feat[2 * G + c0 + tid] = tm;
    }
    __syncthreads();
  }
  if(a.featOut != nullptr)
    for(int i = tid; i < 3 * G; i += BT) a.featOut[(size_t)n * 3 * G + i] = feat[i];
  // phase 2: bias[r] = sum_k feat[k] * W[k][r]
  for(int r = tid; r < R; r += BT) {
    float s = 0.0f;
    for(int k = 0; k < 3 * G; k++) s += feat[k] * a.w[(size_t)k * R + r];
    biasv[r] = s;
  }
  __syncthreads();
  // phase 3: r = act((r + bias)*scale + bnBias) * mask, in place
  const int total = S * R;
  for(int i = tid; i < total; i += BT) {
    const int p = i / R, r = i - p * R;
    const size_t idx = (cell0 + p) * a.rStride + a.rOffset + r;
    const float x = ldT<TR>(a.r, idx) + biasv[r];
    const float y = (maskB[p] == 1.0f) ? actApply(x * a.scale[r] + a.bias[r], a.actKind) : 0.0f;
    stT<TR>(a.r, idx, y);
  }
}

// Vectorised form of the kernel above (16-byte loads/stores, 512 threads per board) for channel counts that are
// multiples of 8 — every real net; the scalar kernel stays as the general fallback.
constexpr int GV_THREADS = 512;
template <class TR>
__global__ __launch_bounds__(GV_THREADS) void gpoolApplyVecKernel(const GPoolArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  HIP_DYNAMIC_SHARED(float, sm)
  const int n = blockIdx.x, tid = threadIdx.x;
  const int S = a.S, G = a.G, R = a.R;
  // Round 5: grid y = parts. Every part pools the board and multiplies (the same arithmetic in the same order, a few KB from L2), then
  // applies bias + BN + activation to ITS share of the cells: the last phase is 46 k activations per board on one CU - 8.5 us of vector
  // ALU - and at small batches most CUs have nothing to do. Values do not depend on the split.
  const int pBegin = (int)((long long)S * blockIdx.y / gridDim.y), pEnd = (int)((long long)S * (blockIdx.y + 1) / gridDim.y);
  const int NG = G / 8;                    // channel groups of 8
  const int cellGroups = GV_THREADS / NG;  // threads striding over the board per channel group
  float* partSum = sm;                     // [cellGroups][G]
  float* partMax = partSum + cellGroups * G;
  float* feat = partMax + cellGroups * G;  // [3G]
  float* biasv = feat + 3 * G;             // [R]
  const float* maskB = a.mask + (size_t)n * S;
  const size_t cell0 = (size_t)n * S;
  {
    const int cg = tid % NG, grp = tid / NG;
    float s[8], m[8];
#pragma unroll
    for(int i = 0; i < 8; i++) { s[i] = 0.0f; m[i] = -1.0f; }
    if(grp < cellGroups) {
      constexpr int UN1 = 6;  // a 19x19 board is 6 strides of 64 cell groups: all loads of a thread in flight at once
      for(int p0 = grp; p0 < S; p0 += cellGroups * UN1) {
        V8 xs[UN1];
        float mk[UN1];
#pragma unroll
        for(int u = 0; u < UN1; u++) {
          const int p = p0 + u * cellGroups;
          if(p < S) {
            xs[u] = *(const V8*)((const T*)a.g + (cell0 + p) * a.gStride + a.gOffset + cg * 8);
            mk[u] = maskB[p] - 1.0f;
          }
        }
#pragma unroll
        for(int u = 0; u < UN1; u++) {
          if(p0 + u * cellGroups < S) {
#pragma unroll
            for(int i = 0; i < 8; i++) {
              const float v = TR::toFloat(xs[u][i]);
              s[i] += v;
              m[i] = fmaxf(m[i], v + mk[u]);
            }
          }
        }
      }
#pragma unroll
      for(int i = 0; i < 8; i++) {
        partSum[grp * G + cg * 8 + i] = s[i];
        partMax[grp * G + cg * 8 + i] = m[i];
      }
    }
  }
  __syncthreads();
  if(tid < G) {
    float ts = 0.0f, tm = -1.0f;
    for(int g = 0; g < cellGroups; g++) {
      ts += partSum[g * G + tid];
      tm = fmaxf(tm, partMax[g * G + tid]);
    }
    const float div = a.maskSum[n];
    const float sqrtdiv = sqrtf(div);
    const float mean = ts / div;
    feat[tid] = mean;
    feat[G + tid] = mean * (sqrtdiv - 14.0f) * 0.1f;
    feat[2 * G + tid] = tm;
  }
  __syncthreads();
  if(a.featOut != nullptr && blockIdx.y == 0)
    for(int i = tid; i < 3 * G; i += GV_THREADS) a.featOut[(size_t)n * 3 * G + i] = feat[i];
  // [3G] x [3G][R] with every thread busy: the k range is split over KG thread groups (partials in LDS). One thread per
  // output looping over all 3G weights is a chain of ~200 L2 round trips and was a third of this kernel's 70 us.
  {
    const int KG = GV_THREADS / R > 0 ? (GV_THREADS / R < 8 ? GV_THREADS / R : 8) : 1;
    const int K = 3 * G;
    float* partial = partSum;  // [KG][R], the pooling partials are dead
    for(int idx = tid; idx < KG * R; idx += GV_THREADS) {
      const int kg = idx / R, r = idx - kg * R;
      const int k0 = kg * K / KG, k1 = (kg + 1) * K / KG;
      float acc = 0.0f;
#pragma unroll 16
      for(int k = k0; k < k1; k++) acc += feat[k] * a.w[(size_t)k * R + r];
      partial[idx] = acc;
    }
    __syncthreads();
    for(int r = tid; r < R; r += GV_THREADS) {
      float acc = 0.0f;
      for(int kg = 0; kg < KG; kg++) acc += partial[kg * R + r];
      biasv[r] = acc;
    }
  }
  __syncthreads();
  const int RV = R / 8;
  constexpr int UN = 12;  // loads of UN iterations in flight before the first is consumed: a 19x19 board of 128 channels is ONE burst per thread
  if(GV_THREADS % RV == 0) {
    // every thread keeps ONE group of 8 channels for all its cells: bias, BN scale and BN bias live in registers
    const int rv = tid % RV, pStep = GV_THREADS / RV;
    float bv[8], sc[8], bi[8];
#pragma unroll
    for(int k = 0; k < 8; k++) {
      bv[k] = biasv[rv * 8 + k];
      sc[k] = a.scale[rv * 8 + k];
      bi[k] = a.bias[rv * 8 + k];
    }
    T* const col = (T*)a.r + cell0 * a.rStride + a.rOffset + rv * 8;
    for(int p0 = pBegin + tid / RV; p0 < pEnd; p0 += pStep * UN) {
      V8 x[UN];
      float on[UN];
#pragma unroll
      for(int u = 0; u < UN; u++) {
        const int p = p0 + u * pStep;
        if(p < pEnd) {
          x[u] = *(const V8*)(col + (size_t)p * a.rStride);
          on[u] = maskB[p];
        }
      }
#pragma unroll
      for(int u = 0; u < UN; u++) {
        const int p = p0 + u * pStep;
        if(p < pEnd) {
          V8 y;
#pragma unroll
          for(int k = 0; k < 8; k++) {
            const float v = TR::toFloat(x[u][k]) + bv[k];
            y[k] = TR::fromFloat(on[u] == 1.0f ? actApply(v * sc[k] + bi[k], a.actKind) : 0.0f);
          }
          *(V8*)(col + (size_t)p * a.rStride) = y;
        }
      }
    }
    return;
  }
  const int total = (pEnd - pBegin) * RV;
  for(int i = tid; i < total; i += GV_THREADS) {
    const int p = pBegin + i / RV, rv = i % RV;
    T* ptr = (T*)a.r + (cell0 + p) * a.rStride + a.rOffset + rv * 8;
    V8 x = *(const V8*)ptr;
    const bool on = maskB[p] == 1.0f;
    V8 y;
#pragma unroll
    for(int k = 0; k < 8; k++) {
      const int r = rv * 8 + k;
      const float v = TR::toFloat(x[k]) + biasv[r];
      y[k] = TR::fromFloat(on ? actApply(v * a.scale[r] + a.bias[r], a.actKind) : 0.0f);
    }
    *(V8*)ptr = y;
  }
}

// ------------------------------------------------------------------------------------------------
template <class TR>
__global__ __launch_bounds__(BT) void policyFinalKernel(const PolicyArgs a) {
  HIP_DYNAMIC_SHARED(float, sm)
  float* hidden = sm;  // passHidden floats
  const int n = blockIdx.x, tid = threadIdx.x;
  const int S = a.X * a.Y, P = a.P, NP = a.NP;
  const int sym = a.symmetry ? a.symmetry[n] : 0;
  const float opt = a.optimism ? a.optimism[n] : 0.0f;
  const bool blend = (NP == 2 || NP == 4);  // channels 0,1 = policy, optimistic policy (eigenbackend.cpp:2553)
  float* out = a.out + (size_t)n * (S + 1);
  for(int p = tid; p < S; p += BT) {
    float l0 = 0.0f, l1 = 0.0f;
    const size_t base = ((size_t)n * S + p) * a.pStride + a.pOffset;
    for(int c = 0; c < P; c++) {
      const float x = ldT<TR>(a.p, base + c);
      l0 += x * a.w2[(size_t)c * NP];
      if(blend) l1 += x * a.w2[(size_t)c * NP + 1];
    }
    const float v = blend ? l0 + (l1 - l0) * opt : l0;
    const int h = p / a.X, w = p - h * a.X;
    out[symDst(h, w, a.Y, a.X, sym, true)] = v;
  }
  // pass logit
  const float* feat = a.feat + (size_t)n * a.G3;
  if(a.passHidden > 0) {
    for(int j = tid; j < a.passHidden; j += BT) {
      float s = 0.0f;
      for(int k = 0; k < a.G3; k++) s += feat[k] * a.wPass[(size_t)k * a.passHidden + j];
      hidden[j] = actApply(s + a.bPass[j], a.passAct);
    }
    __syncthreads();
    if(tid == 0) {
      float p0 = 0.0f, p1 = 0.0f;
      for(int j = 0; j < a.passHidden; j++) {
        p0 += hidden[j] * a.wPass2[(size_t)j * NP];
        if(blend) p1 += hidden[j] * a.wPass2[(size_t)j * NP + 1];
      }
      out[S] = blend ? p0 + (p1 - p0) * opt : p0;
    }
  }
  else if(tid == 0) {
    float p0 = 0.0f, p1 = 0.0f;
    for(int k = 0; k < a.G3; k++) {
      p0 += feat[k] * a.wPass[(size_t)k * NP];
      if(blend) p1 += feat[k] * a.wPass[(size_t)k * NP + 1];
    }
    out[S] = blend ? p0 + (p1 - p0) * opt : p0;
  }
}

// ------------------------------------------------------------------------------------------------
template <class TR>
__global__ __launch_bounds__(BT) void valueFinalKernel(const ValueArgs a) {
  HIP_DYNAMIC_SHARED(float, sm)
  // layout: part[BT], feat[3*V1], h[V2]
  float* part = sm;
  float* feat = sm + BT;
  float* hid = feat + 3 * a.V1;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int S = a.X * a.Y, V1 = a.V1, V2 = a.V2;
  const int sym = a.symmetry ? a.symmetry[n] : 0;
  const size_t cell0 = (size_t)n * S;
  // pooling (poolRowsValueHead)
  for(int c0 = 0; c0 < V1; c0 += BT) {
    const int Vc = (V1 - c0) < BT ? (V1 - c0) : BT;
    const int groups = BT / Vc;
    const int c = tid % Vc, grp = tid / Vc;
    float s = 0.0f;
    if(grp < groups)
      for(int p = grp; p < S; p += groups) s += ldT<TR>(a.v, (cell0 + p) * a.vStride + a.vOffset + c0 + c);
    part[tid] = s;
    __syncthreads();
    if(tid < Vc) {
      float ts = 0.0f;
      for(int g = 0; g < groups; g++) ts += part[g * Vc + tid];
      const float div = a.maskSum[n];
      const float sqrtdiv = sqrtf(div);
      const float mean = ts / div;
      feat[c0 + tid] = mean;
      feat[V1 + c0 + tid] = mean * (sqrtdiv - 14.0f) * 0.1f;
      feat[2 * V1 + c0 + tid] = mean * ((sqrtdiv - 14.0f) * (sqrtdiv - 14.0f) * 0.01f - 0.1f);
    }
    __syncthreads();
  }
  for(int j = tid; j < V2; j += BT) {
    float s = 0.0f;
    for(int k = 0; k < 3 * V1; k++) s += feat[k] * a.w2[(size_t)k * V2 + j];
    hid[j] = actApply(s + a.b2[j], a.v2Act);
  }
  __syncthreads();
  if(tid < 3) {
    float s = 0.0f;
    for(int j = 0; j < V2; j++) s += hid[j] * a.w3[(size_t)j * 3 + tid];
    a.value[(size_t)n * 3 + tid] = s + a.b3[tid];
  }
  else if(tid >= 32 && tid < 32 + 6) {
    const int k = tid - 32;
    float s = 0.0f;
    if(k < a.NSV) {
      for(int j = 0; j < V2; j++) s += hid[j] * a.wsv[(size_t)j * a.NSV + k];
      s += a.bsv[k];
    }
    a.score[(size_t)n * 6 + k] = s;
  }
  // ownership: 1x1 conv V1 -> 1, then inverse symmetry
  if(a.ownership != nullptr) {
    float* own = a.ownership + (size_t)n * S;
    for(int p = tid; p < S; p += BT) {
      float s = 0.0f;
      const size_t base = (cell0 + p) * a.vStride + a.vOffset;
      for(int c = 0; c < V1; c++) s += ldT<TR>(a.v, base + c) * a.wOwn[c];
      const int h = p / a.X, w = p - h * a.X;
      own[symDst(h, w, a.Y, a.X, sym, true)] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Vectorised tails (channel counts, strides and offsets multiples of 8 — every real net; the scalar kernels above stay
// as the general fallback). Same arithmetic, 16-byte loads, reductions spread over all threads: the scalar versions
// were chains of dependent 2-byte loads and took 73 us (policy) and 110 us (value) per batch of 256.
template <class TR>
__global__ __launch_bounds__(BT) void policyFinalVecKernel(const PolicyArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  HIP_DYNAMIC_SHARED(float, sm)
  const int n = blockIdx.x, tid = threadIdx.x;
  const int S = a.X * a.Y, P = a.P, NP = a.NP, H = a.passHidden;
  float* w2s = sm;               // [P][NP]
  float* hidden = w2s + P * NP;  // [H]
  float* partial = hidden + (H > 0 ? H : 1);  // [KG][H]
  float* wp2s = partial + 8 * (H > 0 ? H : 1);  // [H][NP]: the second pass matmul's weights (round 5: one thread walked them in global memory)
  const int sym = a.symmetry ? a.symmetry[n] : 0;
  const float opt = a.optimism ? a.optimism[n] : 0.0f;
  const bool blend = (NP == 2 || NP == 4);  // channels 0,1 = policy, optimistic policy (eigenbackend.cpp:2553)
  float* out = a.out + (size_t)n * (S + 1);
  for(int i = tid; i < P * NP; i += BT) w2s[i] = a.w2[i];
  if(H > 0)
    for(int i = tid; i < H * NP; i += BT) wp2s[i] = a.wPass2[i];
  __syncthreads();
  const int PV = P / 8;
  for(int p = tid; p < S; p += BT) {
    const T* row = (const T*)a.p + ((size_t)n * S + p) * a.pStride + a.pOffset;
    float l0 = 0.0f, l1 = 0.0f;
    for(int cv = 0; cv < PV; cv++) {
      const V8 x = *(const V8*)(row + cv * 8);
#pragma unroll
      for(int i = 0; i < 8; i++) {
        const float xv = TR::toFloat(x[i]);
        l0 += xv * w2s[(cv * 8 + i) * NP];
        if(blend) l1 += xv * w2s[(cv * 8 + i) * NP + 1];
      }
    }
    const float v = blend ? l0 + (l1 - l0) * opt : l0;
    const int h = p / a.X, w = p - h * a.X;
    out[symDst(h, w, a.Y, a.X, sym, true)] = v;
  }
  // pass logit
  const float* feat = a.feat + (size_t)n * a.G3;
  if(H > 0) {
    const int KG = BT / H > 0 ? (BT / H < 8 ? BT / H : 8) : 1;
    for(int idx = tid; idx < KG * H; idx += BT) {
      const int kg = idx / H, j = idx - kg * H;
      const int k0 = kg * a.G3 / KG, k1 = (kg + 1) * a.G3 / KG;
      float acc = 0.0f;
#pragma unroll 8
      for(int k = k0; k < k1; k++) acc += feat[k] * a.wPass[(size_t)k * H + j];
      partial[idx] = acc;
    }
    __syncthreads();
    for(int j = tid; j < H; j += BT) {
      float acc = 0.0f;
      for(int kg = 0; kg < KG; kg++) acc += partial[kg * H + j];
      hidden[j] = actApply(acc + a.bPass[j], a.passAct);
    }
    __syncthreads();
    if(tid == 0) {
      float p0 = 0.0f, p1 = 0.0f;
      for(int j = 0; j < H; j++) {
        p0 += hidden[j] * wp2s[j * NP];
        if(blend) p1 += hidden[j] * wp2s[j * NP + 1];
      }
      out[S] = blend ? p0 + (p1 - p0) * opt : p0;
    }
  }
  else if(tid == 0) {
    float p0 = 0.0f, p1 = 0.0f;
    for(int k = 0; k < a.G3; k++) {
      p0 += feat[k] * a.wPass[(size_t)k * NP];
      if(blend) p1 += feat[k] * a.wPass[(size_t)k * NP + 1];
    }
    out[S] = blend ? p0 + (p1 - p0) * opt : p0;
  }
}

template <class TR>
__global__ __launch_bounds__(BT) void valueFinalVecKernel(const ValueArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  HIP_DYNAMIC_SHARED(float, sm)
  const int n = blockIdx.x, tid = threadIdx.x;
  const int S = a.X * a.Y, V1 = a.V1, V2 = a.V2;
  const int NG = V1 / 8;          // channel groups of 8
  const int cellGroups = BT / NG;  // threads striding over the board per channel group
  const int KG = BT / V2 > 0 ? (BT / V2 < 8 ? BT / V2 : 8) : 1;
  // layout: feat[3*V1], hid[V2], wown[V1], w3s[3*V2], wsvs[6*V2], part[max(cellGroups*V1, KG*V2)]
  float* feat = sm;
  float* hid = feat + 3 * V1;
  float* wown = hid + V2;
  float* w3s = wown + V1;
  float* wsvs = w3s + 3 * V2;
  float* part = wsvs + 6 * V2;
  const int sym = a.symmetry ? a.symmetry[n] : 0;
  const size_t cell0 = (size_t)n * S;
  const T* vb = (const T*)a.v + cell0 * a.vStride + a.vOffset;
  // Round 5: every small weight vector the tail needs comes into LDS HERE, in one memory round trip that the pooling loads share - the
  // final 3- and NSV-wide products were 128 dependent global loads on nine threads, the longest single stretch of this kernel's 39 us
  for(int i = tid; i < V1; i += BT) wown[i] = a.wOwn[i];
  for(int i = tid; i < 3 * V2; i += BT) w3s[i] = a.w3[i];
  for(int i = tid; i < a.NSV * V2; i += BT) wsvs[i] = a.wsv[i];
  // pooling (poolRowsValueHead, eigenbackend.cpp:179-197): a thread's loads in batches of UNP, all of a batch in flight at once (the sums
  // are taken in the same order as one load at a time)
  {
    const int cg = tid % NG, grp = tid / NG;
    if(grp < cellGroups) {
      float s[8];
#pragma unroll
      for(int i = 0; i < 8; i++) s[i] = 0.0f;
      constexpr int UNP = 6;
      for(int p0 = grp; p0 < S; p0 += cellGroups * UNP) {
        V8 x[UNP];
#pragma unroll
        for(int u = 0; u < UNP; u++) {
          const int p = p0 + u * cellGroups;
          if(p < S) x[u] = *(const V8*)(vb + (size_t)p * a.vStride + cg * 8);
        }
#pragma unroll
        for(int u = 0; u < UNP; u++) {
          if(p0 + u * cellGroups < S) {
#pragma unroll
            for(int i = 0; i < 8; i++) s[i] += TR::toFloat(x[u][i]);
          }
        }
      }
#pragma unroll
      for(int i = 0; i < 8; i++) part[grp * V1 + cg * 8 + i] = s[i];
    }
  }
  __syncthreads();
  for(int c = tid; c < V1; c += BT) {
    float ts = 0.0f;
    for(int g = 0; g < cellGroups; g++) ts += part[g * V1 + c];
    const float div = a.maskSum[n];
    const float sqrtdiv = sqrtf(div);
    const float mean = ts / div;
    feat[c] = mean;
    feat[V1 + c] = mean * (sqrtdiv - 14.0f) * 0.1f;
    feat[2 * V1 + c] = mean * ((sqrtdiv - 14.0f) * (sqrtdiv - 14.0f) * 0.01f - 0.1f);
  }
  __syncthreads();
  // v2: [3V1] x [3V1][V2], k range split over KG thread groups
  for(int idx = tid; idx < KG * V2; idx += BT) {
    const int kg = idx / V2, j = idx - kg * V2;
    const int k0 = kg * 3 * V1 / KG, k1 = (kg + 1) * 3 * V1 / KG;
    float acc = 0.0f;
#pragma unroll 8
    for(int k = k0; k < k1; k++) acc += feat[k] * a.w2[(size_t)k * V2 + j];
    part[idx] = acc;
  }
  __syncthreads();
  for(int j = tid; j < V2; j += BT) {
    float acc = 0.0f;
    for(int kg = 0; kg < KG; kg++) acc += part[kg * V2 + j];
    hid[j] = actApply(acc + a.b2[j], a.v2Act);
  }
  __syncthreads();
  if(tid < 3) {
    float s = 0.0f;
    for(int j = 0; j < V2; j++) s += hid[j] * w3s[j * 3 + tid];
    a.value[(size_t)n * 3 + tid] = s + a.b3[tid];
  }
  else if(tid >= 32 && tid < 32 + 6) {
    const int k = tid - 32;
    float s = 0.0f;
    if(k < a.NSV) {
      for(int j = 0; j < V2; j++) s += hid[j] * wsvs[j * a.NSV + k];
      s += a.bsv[k];
    }
    a.score[(size_t)n * 6 + k] = s;
  }
  // ownership: 1x1 conv V1 -> 1, then inverse symmetry
  if(a.ownership != nullptr) {
    float* own = a.ownership + (size_t)n * S;
    for(int p = tid; p < S; p += BT) {
      const T* row = vb + (size_t)p * a.vStride;
      float s = 0.0f;
      for(int cv = 0; cv < NG; cv++) {
        const V8 x = *(const V8*)(row + cv * 8);
#pragma unroll
        for(int i = 0; i < 8; i++) s += TR::toFloat(x[i]) * wown[cv * 8 + i];
      }
      const int h = p / a.X, w = p - h * a.X;
      own[symDst(h, w, a.Y, a.X, sym, true)] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <class TR>
__global__ void bnActKernel(const BnActArgs a) {
  const size_t total = (size_t)a.N * a.S * a.C;
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t cell = i / a.C;
    const int c = (int)(i - cell * a.C);
    const float x = ldT<TR>(a.in, cell * a.stride + c);
    const float y = (a.mask[cell] == 1.0f) ? actApply(x * a.scale[c] + a.bias[c], a.actKind) : 0.0f;
    stT<TR>(a.out, cell * a.stride + c, y);
  }
}
template <class TR>
__global__ void floatToTKernel(const float* in, int inC, void* out, int outStride, size_t cells) {
  const size_t total = cells * outStride;
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t cell = i / outStride;
    const int c = (int)(i - cell * outStride);
    stT<TR>(out, i, c < inC ? in[cell * inC + c] : 0.0f);
  }
}
template <class TR>
__global__ void tToFloatKernel(const void* in, int inStride, int offset, float* out, int outC, size_t cells) {
  const size_t total = cells * outC;
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t cell = i / outC;
    const int c = (int)(i - cell * outC);
    out[i] = ldT<TR>(in, cell * inStride + offset + c);
  }
}

inline int gridFor(size_t total) {
  size_t b = (total + 255) / 256;
  if(b > 4096) b = 4096;
  if(b < 1) b = 1;
  return (int)b;
}

}  // namespace

#define KMX_DISPATCH(dtype, KERNEL, grid, block, lds, stream, ...)                               \
  do {                                                                                             \
    if((dtype) == DT_F16) hipLaunchKernelGGL(KERNEL<TraitsF16>, grid, block, lds, stream, __VA_ARGS__); \
    else if((dtype) == DT_BF16) hipLaunchKernelGGL(KERNEL<TraitsBF16>, grid, block, lds, stream, __VA_ARGS__); \
    else if((dtype) == DT_F32) hipLaunchKernelGGL(KERNEL<TraitsF32>, grid, block, lds, stream, __VA_ARGS__); \
    else return hipErrorInvalidValue;                                                              \
    return hipGetLastError();                                                                      \
  } while(0)

hipError_t launchInputExpand(int dtype, const InputArgs& a, hipStream_t stream) {
  if(a.cin > KCHUNK) return hipErrorInvalidValue;
  const size_t lds = a.meta ? sizeof(float) * ((size_t)a.metaIn + a.metaC1 + a.metaC2) : 0;
  KMX_DISPATCH(dtype, inputExpandKernel, dim3(a.N), dim3(BT), lds, stream, a);
}
hipError_t launchGPoolApply(int dtype, const GPoolArgs& a, hipStream_t stream) {
  const bool vec = a.G % 8 == 0 && a.R % 8 == 0 && a.gStride % 8 == 0 && a.rStride % 8 == 0 && a.gOffset % 8 == 0 &&
                   a.rOffset % 8 == 0 && a.G <= GV_THREADS && (GV_THREADS / (a.G / 8)) * a.G * 2 * sizeof(float) <= 48 * 1024;
  if(vec) {
    const int cellGroups = GV_THREADS / (a.G / 8);
    size_t lds = sizeof(float) * ((size_t)2 * cellGroups * a.G + 3 * a.G + a.R);
    // parts per board (the kernel's last phase): while the boards leave compute units idle
    const int parts = a.N >= 128 ? 1 : a.N >= 48 ? 2 : 4;
    KMX_DISPATCH(dtype, gpoolApplyVecKernel, dim3(a.N, parts), dim3(GV_THREADS), lds, stream, a);
  }
  size_t lds = sizeof(float) * (2 * BT + 3 * a.G + a.R);
  KMX_DISPATCH(dtype, gpoolApplyKernel, dim3(a.N), dim3(BT), lds, stream, a);
}
hipError_t launchPolicyFinal(int dtype, const PolicyArgs& a, hipStream_t stream) {
  if(a.P % 8 == 0 && a.pStride % 8 == 0 && a.pOffset % 8 == 0 && a.passHidden <= BT) {
    const int H = a.passHidden > 0 ? a.passHidden : 1;
    size_t lds = sizeof(float) * ((size_t)a.P * a.NP + H + (size_t)8 * H + (size_t)H * a.NP);
    KMX_DISPATCH(dtype, policyFinalVecKernel, dim3(a.N), dim3(BT), lds, stream, a);
  }
  size_t lds = sizeof(float) * (a.passHidden > 0 ? a.passHidden : 1);
  KMX_DISPATCH(dtype, policyFinalKernel, dim3(a.N), dim3(BT), lds, stream, a);
}
hipError_t launchValueFinal(int dtype, const ValueArgs& a, hipStream_t stream) {
  if(a.V1 % 8 == 0 && a.V1 / 8 <= BT && a.vStride % 8 == 0 && a.vOffset % 8 == 0 && a.V2 <= BT) {
    const size_t cellGroups = BT / (a.V1 / 8);
    const size_t partN = cellGroups * a.V1 > (size_t)8 * a.V2 ? cellGroups * a.V1 : (size_t)8 * a.V2;
    size_t lds = sizeof(float) * ((size_t)3 * a.V1 + a.V2 + a.V1 + 9 * (size_t)a.V2 + partN);
    KMX_DISPATCH(dtype, valueFinalVecKernel, dim3(a.N), dim3(BT), lds, stream, a);
  }
  size_t lds = sizeof(float) * (BT + 3 * a.V1 + a.V2);
  KMX_DISPATCH(dtype, valueFinalKernel, dim3(a.N), dim3(BT), lds, stream, a);
}
hipError_t launchBnAct(int dtype, const BnActArgs& a, hipStream_t stream) {
  KMX_DISPATCH(dtype, bnActKernel, dim3(gridFor((size_t)a.N * a.S * a.C)), dim3(256), 0, stream, a);
}
hipError_t launchFloatToT(int dtype, const float* in, int inC, void* out, int outStride, size_t cells, hipStream_t stream) {
  KMX_DISPATCH(dtype, floatToTKernel, dim3(gridFor(cells * outStride)), dim3(256), 0, stream, in, inC, out, outStride, cells);
}
hipError_t launchTToFloat(int dtype, const void* in, int inStride, int offset, float* out, int outC, size_t cells, hipStream_t stream) {
  KMX_DISPATCH(dtype, tToFloatKernel, dim3(gridFor(cells * outC)), dim3(256), 0, stream, in, inStride, offset, out, outC, cells);
}

// ---- host float -> 16-bit conversions (round to nearest even) ----
uint16_t floatToBf16Bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(r >> 16);
}
uint16_t floatToHalfBits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const uint32_t absu = u & 0x7fffffffu;
  if(absu > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);   // NaN
  if(absu >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // >= 65520 -> inf
  if(absu < 0x33000001u) return (uint16_t)sign;               // < 2^-25 -> 0
  int exp = (int)(absu >> 23) - 127;
  uint32_t mant = (absu & 0x7fffffu) | 0x800000u;
  if(exp < -14) {  // subnormal half
    int shift = -14 - exp;
    uint32_t m = mant >> (13 + shift);
    uint32_t rem = mant & ((1u << (13 + shift)) - 1u);
    uint32_t halfway = 1u << (12 + shift);
    if(rem > halfway || (rem == halfway && (m & 1u))) m++;
    return (uint16_t)(sign | m);
  }
  uint32_t h = ((uint32_t)(exp + 15) << 10) | ((mant & 0x7fffffu) >> 13);
  uint32_t rem = mant & 0x1fffu;
  if(rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
  return (uint16_t)(sign | h);
}

}  // namespace kmx

// transformer_kernels.hip — the layers of model-v17 transformer trunks that are not convolutions
// (SURVEY 8 rows a24 / f4). The Q/K/V, output and FFN projections run on the 1x1 convolution kernel; this file holds
//   rmsNormKernel      TransformerRMSNormLayer::apply (eigenbackend.cpp:885-915) and the RMSNorm trunk tip
//                      (RMSNormLayer::apply, eigenbackend.cpp:960-1031; per board through boardRmsKernel)
//   attentionKernel    RoPE (applyRoPE, eigenbackend.cpp:1417-1456; tables desc.cpp:1300-1363) + masked softmax
//                      attention with grouped-query heads (eigenbackend.cpp:1466-1560)
//   swiGluKernel       SiLU(linear1) * gate (eigenbackend.cpp:1674-1689)
//
// Verified on the MI355X (round 2): unit kernels against numpy restatements, whole nets against the reference PyTorch
// goldens and the oracle, and the reference's two trained transformer nets through its `testgpuerror` acceptance test
// (tests/test_gpu_transformer.py). Two attention kernels: attentionMfmaKernel (default, matrix cores) and attentionKernel
// (KMX_ATTENTION_VALU=1; one thread per query, the cross-check of the other).
#include "device_common.h"

#include <atomic>
#include <cstdlib>

namespace kmx {

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------------------
// RMSNorm over channels. 8 lanes per cell, 16-byte pieces; C and both strides are multiples of 8.
template <class TR>
__global__ __launch_bounds__(256) void rmsNormKernel(RmsNormArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  const int lane8 = threadIdx.x & 7;
  const size_t cells = (size_t)a.N * a.S;
  const size_t cell = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool valid = cell < cells;  // invalid lanes stay alive for the shuffles
  const T* in = (const T*)a.in + (valid ? cell : 0) * a.inStride;
  float ss = 0.0f;
  if(valid && a.boardRms == nullptr) {
    for(int c = lane8 * 8; c < a.C; c += 64) {
      const V8 v = *(const V8*)(in + c);
#pragma unroll
      for(int i = 0; i < 8; i++) {
        const float f = TR::toFloat(v[i]);
        ss += f * f;
      }
    }
  }
  ss += __shfl_xor(ss, 1);
  ss += __shfl_xor(ss, 2);
  ss += __shfl_xor(ss, 4);
  if(!valid) return;
  const bool onBoard = a.mask[cell] != 0.0f;
  const float r = a.boardRms != nullptr ? a.boardRms[cell / a.S] : rsqrtf(ss / (float)a.C + a.eps);
  T* out = (T*)a.out + cell * a.outStride;
  for(int c = lane8 * 8; c < a.outStride; c += 64) {
    V8 o;
    if(onBoard && c < a.C) {
      const V8 v = *(const V8*)(in + c);
#pragma unroll
      for(int i = 0; i < 8; i++) {
        float y = TR::toFloat(v[i]) * r * a.w[c + i];
        if(a.beta != nullptr) y += a.beta[c + i];
        o[i] = TR::fromFloat(actApply(y, a.actKind));
      }
    }
    else {
#pragma unroll
      for(int i = 0; i < 8; i++) o[i] = TR::fromFloat(0.0f);
    }
    *(V8*)(out + c) = o;
  }
}

// One RMS per board over on-board cells x channels: rms[n] = 1 / sqrt(sum / (count * C) + eps)
template <class TR>
__global__ __launch_bounds__(256) void boardRmsKernel(const void* inV, int inStride, int C, const float* mask, const float* maskSum,
                                                       int S, float eps, float* outRms) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  __shared__ float part[4];
  const int n = blockIdx.x;
  const T* in = (const T*)inV + (size_t)n * S * inStride;
  const int pieces = C / 8;
  float ss = 0.0f;
  for(int idx = threadIdx.x; idx < S * pieces; idx += 256) {
    const int cell = idx / pieces, c = (idx % pieces) * 8;
    if(mask[(size_t)n * S + cell] == 0.0f) continue;
    const V8 v = *(const V8*)(in + (size_t)cell * inStride + c);
#pragma unroll
    for(int i = 0; i < 8; i++) {
      const float f = TR::toFloat(v[i]);
      ss += f * f;
    }
  }
#pragma unroll
  for(int o = 1; o < 64; o <<= 1) ss += __shfl_xor(ss, o);
  if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  if(threadIdx.x == 0) {
    const float total = part[0] + part[1] + part[2] + part[3];
    outRms[n] = rsqrtf(total / (maskSum[n] * (float)C) + eps);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Attention of one (board, head) per work-group. K (rotated) and V of the head's KV head live in LDS as T[S][QDP] and
// T[S][VDP] (head dims zero-padded to QDP / VDP); every thread owns one query at a time and walks the keys with a
// running-max softmax, all lanes reading the same K/V row (LDS broadcast).
template <class TR, int QDP, int VDP>
__global__ __launch_bounds__(192) void attentionKernel(AttentionArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  HIP_DYNAMIC_SHARED(f32x4, smemAttn)  // dynamic LDS; the 16-byte element type carries the alignment the V8 reads need
  const int S = a.S, h = blockIdx.x, n = blockIdx.y;
  const int kvh = h / (a.H / a.KVH);
  T* const Ks = (T*)smemAttn;                            // [S][QDP]
  T* const Vs = Ks + (size_t)S * QDP;                    // [S][VDP]
  float* const Ms = (float*)(Vs + (size_t)S * VDP);      // [S]; (QDP + VDP) * 2 bytes per cell is a multiple of 16
  const T* const base = (const T*)a.qkv + (size_t)n * S * a.stride;
  const int numPairs = a.QD / 2;
  const size_t tableHead = (size_t)(a.ropeHeads > 1 ? kvh : 0) * numPairs * S;
  const float* const cosT = a.ropeCos != nullptr ? a.ropeCos + tableHead : nullptr;
  const float* const sinT = a.ropeSin != nullptr ? a.ropeSin + tableHead : nullptr;

  for(int idx = threadIdx.x; idx < S * (QDP / 2); idx += blockDim.x) {
    const int j = idx / (QDP / 2), p = idx % (QDP / 2);
    float k0 = 0.0f, k1 = 0.0f;
    const T* kp = base + (size_t)j * a.stride + a.kOff + kvh * a.QD;
    if(2 * p < a.QD) k0 = TR::toFloat(kp[2 * p]);
    if(2 * p + 1 < a.QD) k1 = TR::toFloat(kp[2 * p + 1]);
    if(cosT != nullptr && p < numPairs) {
      const float c = cosT[(size_t)p * S + j], s = sinT[(size_t)p * S + j];
      const float r0 = k0 * c - k1 * s, r1 = k0 * s + k1 * c;
      k0 = r0;
      k1 = r1;
    }
    Ks[(size_t)j * QDP + 2 * p] = TR::fromFloat(k0);
    Ks[(size_t)j * QDP + 2 * p + 1] = TR::fromFloat(k1);
  }
  for(int idx = threadIdx.x; idx < S * VDP; idx += blockDim.x) {
    const int j = idx / VDP, d = idx % VDP;
    Vs[idx] = d < a.VD ? base[(size_t)j * a.stride + a.vOff + kvh * a.VD + d] : TR::fromFloat(0.0f);
  }
  for(int j = threadIdx.x; j < S; j += blockDim.x) Ms[j] = a.mask[(size_t)n * S + j];
  __syncthreads();

  const float qscale = a.scale * LOG2E;  // softmax through exp2
  for(int q = threadIdx.x; q < S; q += blockDim.x) {
    T* const outp = (T*)a.out + ((size_t)n * S + q) * a.outStride + h * a.VD;
    if(Ms[q] == 0.0f) {  // masked queries contribute nothing (eigenbackend.cpp:1503-1508)
      for(int d = 0; d < a.VD; d++) outp[d] = TR::fromFloat(0.0f);
      continue;
    }
    float qv[QDP];
    const T* qp = base + (size_t)q * a.stride + h * a.QD;
#pragma unroll
    for(int p = 0; p < QDP / 2; p++) {
      float q0 = 0.0f, q1 = 0.0f;
      if(2 * p < a.QD) q0 = TR::toFloat(qp[2 * p]);
      if(2 * p + 1 < a.QD) q1 = TR::toFloat(qp[2 * p + 1]);
      if(cosT != nullptr && p < numPairs) {
        const float c = cosT[(size_t)p * S + q], s = sinT[(size_t)p * S + q];
        const float r0 = q0 * c - q1 * s, r1 = q0 * s + q1 * c;
        q0 = r0;
        q1 = r1;
      }
      qv[2 * p] = q0 * qscale;
      qv[2 * p + 1] = q1 * qscale;
    }
    float m = -INFINITY, l = 0.0f;
    float acc[VDP];
#pragma unroll
    for(int d = 0; d < VDP; d++) acc[d] = 0.0f;
    for(int j = 0; j < S; j++) {
      if(Ms[j] == 0.0f) continue;  // uniform over the work-group: masked keys (eigenbackend.cpp:1517-1519)
      float s = 0.0f;
#pragma unroll
      for(int d8 = 0; d8 < QDP / 8; d8++) {
        const V8 kk = *(const V8*)(Ks + (size_t)j * QDP + d8 * 8);
#pragma unroll
        for(int i = 0; i < 8; i++) s += qv[d8 * 8 + i] * TR::toFloat(kk[i]);
      }
      const float mNew = fmaxf(m, s);
      const float corr = __builtin_amdgcn_exp2f(m - mNew);  // first on-board key: exp2(-inf) = 0
      const float pj = __builtin_amdgcn_exp2f(s - mNew);
      l = l * corr + pj;
#pragma unroll
      for(int d8 = 0; d8 < VDP / 8; d8++) {
        const V8 vv = *(const V8*)(Vs + (size_t)j * VDP + d8 * 8);
#pragma unroll
        for(int i = 0; i < 8; i++) acc[d8 * 8 + i] = acc[d8 * 8 + i] * corr + pj * TR::toFloat(vv[i]);
      }
      m = mNew;
    }
    const float inv = 1.0f / l;
#pragma unroll
    for(int d = 0; d < VDP; d++)
      if(d < a.VD) outp[d] = TR::fromFloat(acc[d] * inv);
  }
}

// The same attention on the matrix cores. One work-group (4 waves) per (head, board); K (rotated) as T[SP][QDP] and V
// TRANSPOSED as T[VDP][SP] in LDS, SP = 384 padded cells. A wave owns query tiles of 32; per key tile of 32:
//   S^T[key][query] = K Q^T          QDP/16 MFMAs: A = K rows from LDS (one 16-byte read), B = the wave's Q fragments
//   running-max softmax per query    a query is a lane column; its 32 keys sit in the two lane halves: one
//                                    __shfl_xor(., 32) for the max, one for the sum
//   O^T[vd][query] += V^T P^T        2 MFMAs per 32-channel vd tile: B = P straight from the S^T accumulators — the
//                                    accumulator layout of a 32x32 tile is a valid B layout for two 16-deep MFMAs
//                                    under the key permutation  slot (half, i) -> key 4*half + (i & 3) + 8*(i >> 2)
//                                    (+16 for the second MFMA), which the A operand follows by reading V^T rows as
//                                    two 8-byte runs of 4 keys
// so the probabilities never leave registers. Scores are scaled in fp32 after the MFMA; P is rounded to T for the second
// product (as every flash-attention kernel does).
template <class TR, int QDP, int VDP>
__global__ __launch_bounds__(256) void attentionMfmaKernel(AttentionArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  typedef typename TR::V4 V4;
  constexpr int SP = 384, KT = SP / 32, NQF = QDP / 16, NVT = VDP / 32;
  HIP_DYNAMIC_SHARED(f32x4, smemAttn)
  const int S = a.S, h = blockIdx.x, n = blockIdx.y;
  const int kvh = h / (a.H / a.KVH);
  T* const Ks = (T*)smemAttn;                         // [SP][QDP]
  T* const Vt = Ks + (size_t)SP * QDP;                // [VDP][SP]
  float* const Ms = (float*)(Vt + (size_t)VDP * SP);  // [SP]; zero beyond the board buffer
  const T* const base = (const T*)a.qkv + (size_t)n * S * a.stride;
  const int numPairs = a.QD / 2;
  const size_t tableHead = (size_t)(a.ropeHeads > 1 ? kvh : 0) * numPairs * S;
  const float* const cosT = a.ropeCos != nullptr ? a.ropeCos + tableHead : nullptr;
  const float* const sinT = a.ropeSin != nullptr ? a.ropeSin + tableHead : nullptr;

  for(int idx = threadIdx.x; idx < SP * (QDP / 2); idx += blockDim.x) {
    const int j = idx / (QDP / 2), p = idx % (QDP / 2);
    float k0 = 0.0f, k1 = 0.0f;
    if(j < S) {
      const T* kp = base + (size_t)j * a.stride + a.kOff + kvh * a.QD;
      if(2 * p < a.QD) k0 = TR::toFloat(kp[2 * p]);
      if(2 * p + 1 < a.QD) k1 = TR::toFloat(kp[2 * p + 1]);
      if(cosT != nullptr && p < numPairs) {
        const float c = cosT[(size_t)p * S + j], sn = sinT[(size_t)p * S + j];
        const float r0 = k0 * c - k1 * sn, r1 = k0 * sn + k1 * c;
        k0 = r0;
        k1 = r1;
      }
    }
    Ks[(size_t)j * QDP + 2 * p] = TR::fromFloat(k0);
    Ks[(size_t)j * QDP + 2 * p + 1] = TR::fromFloat(k1);
  }
  for(int idx = threadIdx.x; idx < SP * VDP; idx += blockDim.x) {
    const int j = idx / VDP, d = idx % VDP;
    Vt[(size_t)d * SP + j] = (j < S && d < a.VD) ? base[(size_t)j * a.stride + a.vOff + kvh * a.VD + d] : TR::fromFloat(0.0f);
  }
  for(int j = threadIdx.x; j < SP; j += blockDim.x) Ms[j] = j < S ? a.mask[(size_t)n * S + j] : 0.0f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, hh = lane >> 5;
  const float qscale = a.scale * LOG2E;
  for(int qt = wave; qt < KT; qt += 4) {
    const int q = qt * 32 + col;
    const bool qIn = q < S;
    const bool qOn = qIn && Ms[q] != 0.0f;
    V8 qf[NQF];
#pragma unroll
    for(int kk = 0; kk < NQF; kk++)
#pragma unroll
      for(int i = 0; i < 8; i += 2) {
        const int d = 16 * kk + 8 * hh + i, p = d >> 1;
        float q0 = 0.0f, q1 = 0.0f;
        if(qIn) {
          const T* qp = base + (size_t)q * a.stride + h * a.QD;
          if(d < a.QD) q0 = TR::toFloat(qp[d]);
          if(d + 1 < a.QD) q1 = TR::toFloat(qp[d + 1]);
          if(cosT != nullptr && p < numPairs) {
            const float c = cosT[(size_t)p * S + q], sn = sinT[(size_t)p * S + q];
            const float r0 = q0 * c - q1 * sn, r1 = q0 * sn + q1 * c;
            q0 = r0;
            q1 = r1;
          }
        }
        qf[kk][i] = TR::fromFloat(q0);
        qf[kk][i + 1] = TR::fromFloat(q1);
      }
    float m = -INFINITY, l = 0.0f;
    f32x16 o[NVT];
#pragma unroll
    for(int vt = 0; vt < NVT; vt++)
#pragma unroll
      for(int v = 0; v < 16; v++) o[vt][v] = 0.0f;
    for(int kt = 0; kt < KT; kt++) {
      f32x16 s;
#pragma unroll
      for(int v = 0; v < 16; v++) s[v] = 0.0f;
#pragma unroll
      for(int kk = 0; kk < NQF; kk++) {
        const V8 kf = *(const V8*)(Ks + (size_t)(kt * 32 + col) * QDP + 16 * kk + 8 * hh);
        s = TR::mfma(kf, qf[kk], s);
      }
      // element v of this lane: key kt*32 + (v/4)*8 + hh*4 + v%4, query `col`
      float sv[16];
      float mx = -INFINITY;
#pragma unroll
      for(int v = 0; v < 16; v++) {
        const int key = kt * 32 + (v >> 2) * 8 + hh * 4 + (v & 3);
        sv[v] = Ms[key] != 0.0f ? s[v] * qscale : -INFINITY;
        mx = fmaxf(mx, sv[v]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mNew = fmaxf(m, mx);
      const float mSafe = mNew == -INFINITY ? 0.0f : mNew;  // nothing on the board so far: every exponent below is -inf -> 0
      const float corr = __builtin_amdgcn_exp2f(m - mSafe);
      float ps = 0.0f;
      V8 pf0, pf1;
#pragma unroll
      for(int v = 0; v < 8; v++) {
        const float p0 = __builtin_amdgcn_exp2f(sv[v] - mSafe), p1 = __builtin_amdgcn_exp2f(sv[v + 8] - mSafe);
        ps += p0 + p1;
        pf0[v] = TR::fromFloat(p0);
        pf1[v] = TR::fromFloat(p1);
      }
      ps += __shfl_xor(ps, 32);
      l = l * corr + ps;
      m = mNew;
#pragma unroll
      for(int vt = 0; vt < NVT; vt++) {
#pragma unroll
        for(int v = 0; v < 16; v++) o[vt][v] *= corr;
        const T* vrow = Vt + (size_t)(vt * 32 + col) * SP + kt * 32 + hh * 4;
        V8 af0, af1;
        const V4 a00 = *(const V4*)(vrow), a01 = *(const V4*)(vrow + 8), a10 = *(const V4*)(vrow + 16), a11 = *(const V4*)(vrow + 24);
#pragma unroll
        for(int i = 0; i < 4; i++) {
          af0[i] = a00[i];
          af0[4 + i] = a01[i];
          af1[i] = a10[i];
          af1[4 + i] = a11[i];
        }
        o[vt] = TR::mfma(af0, pf0, o[vt]);
        o[vt] = TR::mfma(af1, pf1, o[vt]);
      }
    }
    if(!qIn) continue;
    T* const outp = (T*)a.out + ((size_t)n * S + q) * a.outStride + h * a.VD;
    const float inv = qOn ? 1.0f / l : 0.0f;  // masked queries contribute nothing (eigenbackend.cpp:1503-1508)
#pragma unroll
    for(int vt = 0; vt < NVT; vt++)
#pragma unroll
      for(int g = 0; g < 4; g++) {
        const int vd0 = vt * 32 + g * 8 + hh * 4;
#pragma unroll
        for(int i = 0; i < 4; i++)
          if(vd0 + i < a.VD) outp[vd0 + i] = TR::fromFloat(qOn ? o[vt][4 * g + i] * inv : 0.0f);
      }
  }
}

// SwiGLU: h[c] = silu(a[c]) * g[c], a = channels [0, F), g = channels [gOff, gOff + F) of the fused projection
template <class TR>
__global__ __launch_bounds__(256) void swiGluKernel(SwiGluArgs a) {
  typedef typename TR::T T;
  typedef typename TR::V8 V8;
  const int pieces = a.outStride / 8;
  const size_t total = a.cells * pieces;
  for(size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t cell = idx / pieces;
    const int c = (int)(idx % pieces) * 8;
    V8 o;
    if(c < a.F) {
      const T* in = (const T*)a.in + cell * a.inStride;
      const V8 x = *(const V8*)(in + c);
      const V8 g = *(const V8*)(in + a.gOff + c);
#pragma unroll
      for(int i = 0; i < 8; i++) o[i] = TR::fromFloat(actSilu(TR::toFloat(x[i])) * TR::toFloat(g[i]));
    }
    else {
#pragma unroll
      for(int i = 0; i < 8; i++) o[i] = TR::fromFloat(0.0f);
    }
    *(V8*)((T*)a.out + cell * a.outStride + c) = o;
  }
}

int padDim(int d) { return d <= 8 ? 8 : d <= 16 ? 16 : d <= 32 ? 32 : d <= 64 ? 64 : -1; }

template <class TR, int QDP, int VDP>
hipError_t launchAttentionOne(const AttentionArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)a.S * (QDP + VDP) * 2 + (size_t)a.S * sizeof(float);
  auto kern = attentionKernel<TR, QDP, VDP>;
  if(lds > 64 * 1024) {  // opt in to the large LDS window, once per instantiation and device
    constexpr int MAX_DEVICES = 64;
    static std::atomic<bool> attrSet[MAX_DEVICES];
    int dev = 0;
    hipError_t de = hipGetDevice(&dev);
    if(de != hipSuccess) return de;
    if(dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    if(!attrSet[dev].load(std::memory_order_acquire)) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if(e != hipSuccess) return e;
      attrSet[dev].store(true, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.H, a.N), dim3(192), lds, stream, a);
  return hipGetLastError();
}
template <class TR, int QDP>
hipError_t launchAttentionV(int vdp, const AttentionArgs& a, hipStream_t stream) {
  switch(vdp) {
    case 8: return launchAttentionOne<TR, QDP, 8>(a, stream);
    case 16: return launchAttentionOne<TR, QDP, 16>(a, stream);
    case 32: return launchAttentionOne<TR, QDP, 32>(a, stream);
    case 64: return launchAttentionOne<TR, QDP, 64>(a, stream);
    default: return hipErrorInvalidValue;
  }
}
template <class TR>
hipError_t launchAttentionQ(int qdp, int vdp, const AttentionArgs& a, hipStream_t stream) {
  switch(qdp) {
    case 8: return launchAttentionV<TR, 8>(vdp, a, stream);
    case 16: return launchAttentionV<TR, 16>(vdp, a, stream);
    case 32: return launchAttentionV<TR, 32>(vdp, a, stream);
    case 64: return launchAttentionV<TR, 64>(vdp, a, stream);
    default: return hipErrorInvalidValue;
  }
}

template <class TR, int QDP, int VDP>
hipError_t launchAttentionMfmaOne(const AttentionArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)384 * QDP * 2 + (size_t)VDP * 384 * 2 + (size_t)384 * sizeof(float);
  auto kern = attentionMfmaKernel<TR, QDP, VDP>;
  if(lds > 64 * 1024) {
    constexpr int MAX_DEVICES = 64;
    static std::atomic<bool> attrSet[MAX_DEVICES];
    int dev = 0;
    hipError_t de = hipGetDevice(&dev);
    if(de != hipSuccess) return de;
    if(dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    if(!attrSet[dev].load(std::memory_order_acquire)) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if(e != hipSuccess) return e;
      attrSet[dev].store(true, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.H, a.N), dim3(256), lds, stream, a);
  return hipGetLastError();
}
template <class TR>
hipError_t launchAttentionMfmaQ(int qdp, int vdp, const AttentionArgs& a, hipStream_t stream) {
  if(qdp == 16 && vdp == 32) return launchAttentionMfmaOne<TR, 16, 32>(a, stream);
  if(qdp == 32 && vdp == 32) return launchAttentionMfmaOne<TR, 32, 32>(a, stream);
  if(qdp == 64 && vdp == 32) return launchAttentionMfmaOne<TR, 64, 32>(a, stream);
  if(qdp == 16 && vdp == 64) return launchAttentionMfmaOne<TR, 16, 64>(a, stream);
  if(qdp == 32 && vdp == 64) return launchAttentionMfmaOne<TR, 32, 64>(a, stream);
  if(qdp == 64 && vdp == 64) return launchAttentionMfmaOne<TR, 64, 64>(a, stream);
  return hipErrorInvalidValue;
}

}  // namespace

bool attentionDimsSupported(int qHeadDim, int vHeadDim) { return padDim(qHeadDim) > 0 && padDim(vHeadDim) > 0; }

hipError_t launchRmsNorm(int dtype, const RmsNormArgs& a, hipStream_t stream) {
  if(a.C % 8 != 0 || a.inStride % 8 != 0 || a.outStride % 8 != 0 || a.C > a.inStride || a.C > a.outStride) return hipErrorInvalidValue;
  const size_t cells = (size_t)a.N * a.S;
  const dim3 grid((unsigned)((cells + 31) / 32));
  if(dtype == DT_F16) hipLaunchKernelGGL(rmsNormKernel<TraitsF16>, grid, dim3(256), 0, stream, a);
  else if(dtype == DT_BF16) hipLaunchKernelGGL(rmsNormKernel<TraitsBF16>, grid, dim3(256), 0, stream, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launchBoardRms(int dtype, const void* in, int inStride, int C, const float* mask, const float* maskSum, int N, int S,
                          float eps, float* outRms, hipStream_t stream) {
  if(C % 8 != 0 || inStride % 8 != 0 || C > inStride) return hipErrorInvalidValue;
  if(dtype == DT_F16)
    hipLaunchKernelGGL(boardRmsKernel<TraitsF16>, dim3(N), dim3(256), 0, stream, in, inStride, C, mask, maskSum, S, eps, outRms);
  else if(dtype == DT_BF16)
    hipLaunchKernelGGL(boardRmsKernel<TraitsBF16>, dim3(N), dim3(256), 0, stream, in, inStride, C, mask, maskSum, S, eps, outRms);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launchAttention(int dtype, const AttentionArgs& a, hipStream_t stream) {
  const int qdp = padDim(a.QD), vdp = padDim(a.VD);
  if(qdp < 0 || vdp < 0 || a.H < 1 || a.KVH < 1 || a.H % a.KVH != 0) return hipErrorInvalidValue;
  if(a.ropeCos != nullptr && a.QD % 2 != 0) return hipErrorInvalidValue;
  // matrix-core version unless KMX_ATTENTION_VALU=1 (the plain kernel is kept as the cross-check of the other)
  const char* valu = getenv("KMX_ATTENTION_VALU");
  if(a.S <= 384 && !(valu != nullptr && valu[0] == '1')) {
    const int q16 = qdp < 16 ? 16 : qdp, v32 = vdp < 32 ? 32 : vdp;
    if(dtype == DT_F16) return launchAttentionMfmaQ<TraitsF16>(q16, v32, a, stream);
    if(dtype == DT_BF16) return launchAttentionMfmaQ<TraitsBF16>(q16, v32, a, stream);
    return hipErrorInvalidValue;
  }
  if(dtype == DT_F16) return launchAttentionQ<TraitsF16>(qdp, vdp, a, stream);
  if(dtype == DT_BF16) return launchAttentionQ<TraitsBF16>(qdp, vdp, a, stream);
  return hipErrorInvalidValue;
}

hipError_t launchSwiGlu(int dtype, const SwiGluArgs& a, hipStream_t stream) {
  if(a.F % 8 != 0 || a.gOff % 8 != 0 || a.inStride % 8 != 0 || a.outStride % 8 != 0 || a.F > a.outStride) return hipErrorInvalidValue;
  size_t blocks = (a.cells * (a.outStride / 8) + 255) / 256;
  if(blocks > 8192) blocks = 8192;
  if(blocks < 1) blocks = 1;
  if(dtype == DT_F16) hipLaunchKernelGGL(swiGluKernel<TraitsF16>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  else if(dtype == DT_BF16) hipLaunchKernelGGL(swiGluKernel<TraitsBF16>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace kmx

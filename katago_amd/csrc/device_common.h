// device_common.h — element traits (fp16 / bf16), activation functions and symmetry index maps
// shared by the gfx950 kernels.
#ifndef KMX_DEVICE_COMMON_H_
#define KMX_DEVICE_COMMON_H_

#include <hip/hip_runtime.h>

#include "../../include/katamx.h"
#include "kernels.h"

namespace kmx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));

struct TraitsF16 {
  typedef _Float16 T;
  typedef h16x8 V8;
  typedef h16x4 V4;
  static constexpr int DT = DT_F16;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float toFloat(T x) { return (float)x; }
  static __device__ __forceinline__ T fromFloat(float x) { return (T)x; }
};
struct TraitsBF16 {
  typedef __bf16 T;
  typedef b16x8 V8;
  typedef b16x4 V4;
  static constexpr int DT = DT_BF16;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float toFloat(T x) { return (float)x; }
  static __device__ __forceinline__ T fromFloat(float x) { return (T)x; }
};

// Activations in fp32. Mish follows the reference's form x*tanh(softplus(x)) with softplus linearised
// above 20 (eigenbackend.cpp:754): tanh(log1p(e)) = (e^2+2e)/(e^2+2e+2), e = exp(min(x,20)); for x > 20
// the tanh argument exceeds 20 and tanh saturates to exactly 1 in fp32, as does this rational form.
__device__ __forceinline__ float actMish(float x) {
  // e = exp(min(x,20)) through the hardware base-2 exponential; n = e^2 + 2e; mish = x * n / (n + 2)
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.4426950408889634f);
  const float n = e * (e + 2.0f);
  return x * n * __builtin_amdgcn_rcpf(n + 2.0f);  // v_rcp_f32 (1 ulp): outputs are rounded to 16 bits anyway
}
__device__ __forceinline__ float actSilu(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.4426950408889634f));
}
__device__ __forceinline__ float actApply(float x, int kind) {
  if(kind == KMX_ACT_RELU) return fmaxf(x, 0.0f);
  if(kind == KMX_ACT_MISH) return actMish(x);
  if(kind == KMX_ACT_SILU) return actSilu(x);
  return x;
}

// Index map of copyWithSymmetry (nninputs.cpp:529-577) for one channel-last image: the cell (h,w) of the
// source lands on cell index symDst(...) of the destination. reverse=false for inputs, true for outputs.
__device__ __forceinline__ int symDst(int h, int w, int hSize, int wSize, int symmetry, bool reverse) {
  bool transpose = (symmetry & 4) != 0 && hSize == wSize;
  bool flipX = (symmetry & 2) != 0;
  bool flipY = (symmetry & 1) != 0;
  if(transpose && !reverse) {
    bool t = flipX;
    flipX = flipY;
    flipY = t;
  }
  int hStrideNew = wSize, wStrideNew = 1, base = 0;
  if(flipY) {
    base += (hSize - 1) * hStrideNew;
    hStrideNew = -hStrideNew;
  }
  if(flipX) {
    base += (wSize - 1) * wStrideNew;
    wStrideNew = -wStrideNew;
  }
  if(transpose) {
    int t = hStrideNew;
    hStrideNew = wStrideNew;
    wStrideNew = t;
  }
  return base + h * hStrideNew + w * wStrideNew;
}

}  // namespace kmx
#endif

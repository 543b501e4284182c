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

// fp32 storage (DT_F32, kernels.h): the small kernels' element type in the verification mode; no matrix-core form
typedef float f32x8 __attribute__((ext_vector_type(8)));
struct TraitsF32 {
  typedef float T;
  typedef f32x8 V8;
  typedef float V4 __attribute__((ext_vector_type(4)));
  static constexpr int DT = DT_F32;
  static __device__ __forceinline__ float toFloat(T x) { return x; }
  static __device__ __forceinline__ T fromFloat(float x) { return x; }
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16-byte global accesses for values that live in the accumulator layout of a 32x32 MFMA tile.
// Lane (column c, half h = lane >> 5) of a tile holds rows 8g + 4h + i (g, i = 0..3): for one g that is 4 consecutive 16-bit
// channels = 8 bytes, and straight loads/stores of it are 8-byte accesses scattered 16 bytes apart - half the bytes per
// memory instruction of what the vector memory path moves best (MI355X_MICROARCH.md: an epilogue of dwordx2 stores is
// store-ISSUE-bound, dwordx4 halves it). Lanes c and c + 32 hold the two halves of every 16-byte run, so one
// v_permlane32_swap per dword regroups them: after pairUp() lane h = 0 holds channels [0,8) and [16,24) of the tile's 32,
// lane h = 1 holds [8,16) and [24,32), each as one 16-byte piece. unpair() is the same exchange backwards (loads).
// Every lane of the wave must execute these (lane exchange): call them outside divergent control flow.
__device__ __forceinline__ void swapUpperLower(unsigned& x, unsigned& y) {  // lanes 32-63 of x <-> lanes 0-31 of y
  const u32x2 r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0];
  y = r[1];
}
// p[g] = the lane's 4 channels of group g (two dwords). Returns q[j] = 16 bytes: channels 16 j + 8 h + [0,8).
__device__ __forceinline__ void pairUp(const u32x2 (&p)[4], u32x4 (&q)[2]) {
#pragma unroll
  for(int j = 0; j < 2; j++) {
    unsigned x0 = p[2 * j][0], x1 = p[2 * j][1], y0 = p[2 * j + 1][0], y1 = p[2 * j + 1][1];
    swapUpperLower(x0, y0);
    swapUpperLower(x1, y1);
    q[j][0] = x0; q[j][1] = x1; q[j][2] = y0; q[j][3] = y1;
  }
}
__device__ __forceinline__ void unpair(const u32x4 (&q)[2], u32x2 (&p)[4]) {
#pragma unroll
  for(int j = 0; j < 2; j++) {
    unsigned x0 = q[j][0], x1 = q[j][1], y0 = q[j][2], y1 = q[j][3];
    swapUpperLower(x0, y0);
    swapUpperLower(x1, y1);
    p[2 * j][0] = x0; p[2 * j][1] = x1; p[2 * j + 1][0] = y0; p[2 * j + 1][1] = y1;
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// min(x, c) as the single v_min_f32 it is on the hardware: fminf() makes the compiler canonicalise x first (a v_max_f32 x, x
// per element, for signalling NaNs), and the epilogues evaluate this once per output value
__device__ __forceinline__ float minPlain(float x, float c) {
#if defined(__AMDGCN__)
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c));
  return r;
#else
  return fminf(x, c);
#endif
}

// Activations in fp32. Mish = x tanh(softplus(x)) (the reference linearises softplus above 20, eigenbackend.cpp:754: there
// tanh is 1 to the last bit). With u = 1 + e^x: tanh(log u) = (u^2 - 1) / (u^2 + 1), so
//     mish(x) = x - 2x / (u^2 + 1)
// - 5 plain vector instructions and 2 transcendentals (v_exp_f32, v_rcp_f32) per value, no clamp: for large x, e^x and the
// denominator overflow to +inf, the reciprocal is 0 and the result is x exactly; for very negative x it is x - x = 0 where
// mish itself is below 1e-30. Absolute error ~1e-7 |x|: far inside the 16-bit rounding of every output these feed.
// Round 2 evaluated PAIRS with packed-fp32 instructions and one shared reciprocal (v_pk_*_f32, 3 transcendentals per pair).
// The round-3 ISA of the seam kernel showed what that costs: 80 v_mov_b64 per 16 values to marshal operands into aligned
// register pairs, on top of packed instructions that issue no faster than two plain ones (MI355X_MICROARCH.md: packed fp32 is
// "an anti-lever" beside MFMAs). Plain scalar code it is, in every kernel (the epilogues stay bit-identical to one another).
__device__ __forceinline__ float actMish(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
  const float u = e + 1.0f;
  return __builtin_fmaf(x * -2.0f, __builtin_amdgcn_rcpf(__builtin_fmaf(u, u, 1.0f)), x);
}
// mish of a tensor that carries 1/8 of its values: f(8x)/8 = x tanh(softplus(8x)) (desc.cpp:421-445; model_desc.cpp scaledBy8)
__device__ __forceinline__ float actMishScale8(float x) {
  const float e = __builtin_amdgcn_exp2f(x * (8.0f * 1.4426950408889634f));
  const float u = e + 1.0f;
  return __builtin_fmaf(x * -2.0f, __builtin_amdgcn_rcpf(__builtin_fmaf(u, u, 1.0f)), x);
}
__device__ __forceinline__ f32x2 actMish2(f32x2 x) {
  f32x2 y;
  y[0] = actMish(x[0]);
  y[1] = actMish(x[1]);
  return y;
}
__device__ __forceinline__ float actSilu(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.4426950408889634f));
}
__device__ __forceinline__ float actApply(float x, int kind) {
  if(kind == KMX_ACT_RELU) return fmaxf(x, 0.0f);
  if(kind == KMX_ACT_MISH) return actMish(x);
  if(kind == KMX_ACT_SILU) return actSilu(x);
  if(kind == KMX_ACT_MISH_SCALE8) return actMishScale8(x);
  return x;
}

// The activation kind is uniform for a launch: kernels branch ONCE (withActKind) into a body instantiated for the kind, instead of
// once per element or per row - and only that body's code is fetched.
template <int KIND>
__device__ __forceinline__ float actK(float x) {
  return KIND == KMX_ACT_MISH ? actMish(x) : KIND == KMX_ACT_RELU ? fmaxf(x, 0.0f) : KIND == KMX_ACT_SILU ? actSilu(x)
         : KIND == KMX_ACT_MISH_SCALE8 ? actMishScale8(x) : x;
}
template <int K>
struct ActKindTag {
  static constexpr int value = K;
};
// the same on a pair of values (the mish pair shares a reciprocal)
template <int KIND>
__device__ __forceinline__ f32x2 actK2(f32x2 x) {
  if(KIND == KMX_ACT_MISH) return actMish2(x);
  f32x2 y;
  y[0] = actK<KIND>(x[0]);
  y[1] = actK<KIND>(x[1]);
  return y;
}
template <class F>
__device__ __forceinline__ void withActKind(int kind, F&& f) {
  if(kind == KMX_ACT_MISH) f(ActKindTag<KMX_ACT_MISH>());
  else if(kind == KMX_ACT_RELU) f(ActKindTag<KMX_ACT_RELU>());
  else if(kind == KMX_ACT_SILU) f(ActKindTag<KMX_ACT_SILU>());
  else if(kind == KMX_ACT_MISH_SCALE8) f(ActKindTag<KMX_ACT_MISH_SCALE8>());
  else f(ActKindTag<KMX_ACT_IDENTITY>());
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

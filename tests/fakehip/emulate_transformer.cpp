// TEST INFRASTRUCTURE: katago_amd/csrc/transformer_kernels.hip compiled for the host and executed on the CPU (see
// emul/hip/hip_runtime.h), behind three C entry points with the signatures of the library's unit hooks
// kmx_test_rmsnorm / kmx_test_attention / kmx_test_swiglu. tests/test_transformer_kernels_emulated.py compares them with
// numpy restatements of the reference formulas: the kernels' index arithmetic, masking, RoPE, running-max softmax and
// 16-bit layouts are exercised before they ever see a GPU. What this cannot show: anything hardware-specific (LDS
// alignment and banking, the precision of v_exp_f32, occupancy). Nothing in the product links or loads this file.
#include <hip/hip_runtime.h>  // the stand-in (-I tests/fakehip/emul)

#include "../../katago_amd/csrc/transformer_kernels.hip"

#include "emul/emu_runtime.inc"

namespace {
using namespace kmx;
int roundUp32(int c) { return (c + 31) / 32 * 32; }
template <class TR>
std::vector<typename TR::T> toT(const float* in, size_t cells, int C, int stride) {
  std::vector<typename TR::T> v(cells * stride + 256);  // tail like DEVBUF_TAIL
  for(size_t i = 0; i < v.size(); i++) v[i] = TR::fromFloat(0.0f);
  for(size_t c = 0; c < cells; c++)
    for(int k = 0; k < C; k++) v[c * stride + k] = TR::fromFloat(in[c * C + k]);
  return v;
}
template <class TR>
void fromT(const std::vector<typename TR::T>& v, size_t cells, int C, int stride, float* out) {
  for(size_t c = 0; c < cells; c++)
    for(int k = 0; k < C; k++) out[c * C + k] = TR::toFloat(v[c * stride + k]);
}

template <class TR>
int rmsnormT(int N, int S, int C, float eps, const float* w, const float* beta, int act, int perBoard, const float* in, const float* mask,
             float* out) {
  const size_t cells = (size_t)N * S;
  const int stride = roundUp32(C);
  auto x = toT<TR>(in, cells, C, stride);
  std::vector<typename TR::T> y(cells * stride + 256);
  for(auto& e : y) e = TR::fromFloat(123.0f);  // the kernel must overwrite everything it owns, padding included
  std::vector<float> rms(N, 0.0f), ms(N, 0.0f);
  for(int b = 0; b < N; b++)
    for(int p = 0; p < S; p++) ms[b] += mask[(size_t)b * S + p];
  RmsNormArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x.data(); a.inStride = stride; a.out = y.data(); a.outStride = stride; a.C = C; a.eps = eps;
  a.w = w; a.beta = beta; a.actKind = act; a.mask = mask; a.N = N; a.S = S;
  if(perBoard) {
    if(launchBoardRms(TR::DT, x.data(), stride, C, mask, ms.data(), N, S, eps, rms.data(), nullptr) != hipSuccess) return -1;
    a.boardRms = rms.data();
  }
  if(launchRmsNorm(TR::DT, a, nullptr) != hipSuccess) return -2;
  for(size_t c = 0; c < cells; c++)
    for(int k = C; k < stride; k++)
      if(TR::toFloat(y[c * stride + k]) != 0.0f) return -3;  // channel padding must be zeroed
  fromT<TR>(y, cells, C, stride, out);
  return 0;
}

template <class TR>
int attentionT(int N, int S, int H, int KVH, int QD, int VD, const float* cosT, const float* sinT, int ropeHeads, const float* q, const float* k,
               const float* v, const float* mask, float* out) {
  const size_t cells = (size_t)N * S;
  const int kOff = (H * QD + 7) / 8 * 8, vOff = kOff + (KVH * QD + 7) / 8 * 8, ctot = vOff + (KVH * VD + 7) / 8 * 8;
  std::vector<float> cat(cells * ctot, 0.0f);
  for(size_t c = 0; c < cells; c++) {
    std::copy(q + c * H * QD, q + (c + 1) * H * QD, cat.begin() + c * ctot);
    std::copy(k + c * KVH * QD, k + (c + 1) * KVH * QD, cat.begin() + c * ctot + kOff);
    std::copy(v + c * KVH * VD, v + (c + 1) * KVH * VD, cat.begin() + c * ctot + vOff);
  }
  const int stride = roundUp32(ctot), outStride = roundUp32(H * VD);
  auto x = toT<TR>(cat.data(), cells, ctot, stride);
  std::vector<typename TR::T> y(cells * outStride + 256);
  for(auto& e : y) e = TR::fromFloat(0.0f);  // the engine's buffer is zero-initialised; the kernel writes [0, H*VD) only
  AttentionArgs a;
  memset(&a, 0, sizeof(a));
  a.qkv = x.data(); a.stride = stride; a.kOff = kOff; a.vOff = vOff;
  a.H = H; a.KVH = KVH; a.QD = QD; a.VD = VD;
  a.ropeCos = cosT; a.ropeSin = sinT; a.ropeHeads = ropeHeads > 1 ? KVH : 1;
  a.mask = mask; a.out = y.data(); a.outStride = outStride; a.scale = 1.0f / sqrtf((float)QD); a.N = N; a.S = S;
  if(launchAttention(TR::DT, a, nullptr) != hipSuccess) return -1;
  fromT<TR>(y, cells, H * VD, outStride, out);
  return 0;
}

template <class TR>
int swigluT(int N, int S, int F, const float* a1, const float* g, float* out) {
  const size_t cells = (size_t)N * S;
  std::vector<float> cat(cells * 2 * F);
  for(size_t c = 0; c < cells; c++) {
    std::copy(a1 + c * F, a1 + (c + 1) * F, cat.begin() + c * 2 * F);
    std::copy(g + c * F, g + (c + 1) * F, cat.begin() + c * 2 * F + F);
  }
  const int stride = roundUp32(2 * F), outStride = roundUp32(F);
  auto x = toT<TR>(cat.data(), cells, 2 * F, stride);
  std::vector<typename TR::T> y(cells * outStride + 256);
  for(auto& e : y) e = TR::fromFloat(123.0f);
  SwiGluArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x.data(); a.inStride = stride; a.gOff = F; a.F = F; a.out = y.data(); a.outStride = outStride; a.cells = cells;
  if(launchSwiGlu(TR::DT, a, nullptr) != hipSuccess) return -1;
  for(size_t c = 0; c < cells; c++)
    for(int k = F; k < outStride; k++)
      if(TR::toFloat(y[c * outStride + k]) != 0.0f) return -3;
  fromT<TR>(y, cells, F, outStride, out);
  return 0;
}
}  // namespace

extern "C" {
// dtype: 0 = fp16, 1 = bf16 (kmx::DT_F16 / DT_BF16)
int emu_rmsnorm(int dtype, int N, int S, int C, float eps, const float* w, const float* beta, int act, int perBoard, const float* in,
                const float* mask, float* out) {
  return dtype == 0 ? rmsnormT<TraitsF16>(N, S, C, eps, w, beta, act, perBoard, in, mask, out)
                    : rmsnormT<TraitsBF16>(N, S, C, eps, w, beta, act, perBoard, in, mask, out);
}
int emu_attention(int dtype, int N, int S, int H, int KVH, int QD, int VD, const float* cosT, const float* sinT, int ropeHeads, const float* q,
                  const float* k, const float* v, const float* mask, float* out) {
  return dtype == 0 ? attentionT<TraitsF16>(N, S, H, KVH, QD, VD, cosT, sinT, ropeHeads, q, k, v, mask, out)
                    : attentionT<TraitsBF16>(N, S, H, KVH, QD, VD, cosT, sinT, ropeHeads, q, k, v, mask, out);
}
int emu_swiglu(int dtype, int N, int S, int F, const float* a, const float* g, float* out) {
  return dtype == 0 ? swigluT<TraitsF16>(N, S, F, a, g, out) : swigluT<TraitsBF16>(N, S, F, a, g, out);
}
}

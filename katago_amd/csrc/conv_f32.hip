// conv_f32.hip — the convolution contract of kernels.h (ConvArgs) in fp32 storage and arithmetic: the verification mode behind
// KMX_PREC_FP32 (useFP16Mode = False, cpp/neuralnet/nninterface.h:50-63).
//
// Replaces, like conv_kernel.h, ConvLayer::apply + the masked BatchNorm / activation that follows + the per-board bias add + the
// residual accumulate (eigenbackend.cpp:293-763) - but as the plainest kernel that honours the contract: one thread per (board cell,
// output channel), the K loop in the order (chunk of 32 input channels, tap, channel) over the SAME weight tensor layout the engine
// builds for the 16-bit kernels (T[chunk][tap][coutPad][32], the four 8-value slots of a row at slot ^ ((cout>>2)&3)), fp32 values,
// fp32 accumulation by fmaf. Correct, not fast: this mode exists so that the reference's testgpuerror can build its fp32 evaluator on
// the device (command/gputest.cpp:122-133) and so that the fixed-seed searches have a mode without 16-bit rounding - nothing is tuned.
#include "device_common.h"

namespace kmx {

namespace {
constexpr int F32_THREADS = 256;

__global__ __launch_bounds__(F32_THREADS) void convF32Kernel(const ConvArgs a, int ks) {
  const int S = a.X * a.Y, nt = ks * ks, halo = ks / 2;
  const int n = blockIdx.y;
  const int idx = blockIdx.x * F32_THREADS + threadIdx.x;  // cell * coutPad + c: neighbouring threads share a cell's input row
  if(idx >= S * a.coutPad) return;
  const int cell = idx / a.coutPad, c = idx - cell * a.coutPad;
  const int y = cell / a.X, x = cell - y * a.X;
  const size_t gcell = (size_t)n * S + cell;
  const float* in = (const float*)a.in;
  const float* w = (const float*)a.w;
  float acc = 0.0f;
  if(a.resid != nullptr && c >= a.rawBegin && c < a.rawEnd) acc = ((const float*)a.resid)[gcell * a.residC + (c - a.rawBegin)];
  const int swz = (c >> 2) & 3;
  for(int chunk = 0; chunk < a.nChunks; chunk++)
    for(int t = 0; t < nt; t++) {
      const int yy = y + t / ks - halo, xx = x + t % ks - halo;
      if(yy < 0 || yy >= a.Y || xx < 0 || xx >= a.X) continue;
      const float* irow = in + ((size_t)n * S + yy * a.X + xx) * a.inC + chunk * KCHUNK;
      const float* wr = w + (((size_t)chunk * nt + t) * a.coutPad + c) * WROW_HALFS;
      float s = 0.0f;
#pragma unroll
      for(int k = 0; k < KCHUNK; k++) s = fmaf(irow[k], wr[(((k >> 3) ^ swz) << 3) + (k & 7)], s);
      acc += s;
    }
  const float v = acc + (a.ncBias != nullptr ? a.ncBias[(size_t)n * a.ncBiasStride + c] : 0.0f);
  if(c >= a.rawBegin && c < a.rawEnd) ((float*)a.rawOut)[gcell * a.rawC + (c - a.rawBegin)] = v;
  if(c >= a.actBegin && c < a.actEnd)
    ((float*)a.actOut)[gcell * a.actC + (c - a.actBegin)] = a.mask[gcell] == 1.0f ? actApply(v * a.scale[c] + a.bias[c], a.actKind) : 0.0f;
}
}  // namespace

hipError_t launchConvF32(int ks, const ConvArgs& a, hipStream_t stream) {
  if(a.X < 2 || a.Y < 2 || a.X > 19 || a.Y > 19 || a.N <= 0) return hipErrorInvalidValue;
  if(a.inC % 8 != 0 || a.coutPad % 32 != 0 || a.inC < a.nChunks * KCHUNK) return hipErrorInvalidValue;
  if(ks != 1 && ks != 3 && ks != 5) return hipErrorInvalidValue;
  const int total = a.X * a.Y * a.coutPad;
  hipLaunchKernelGGL(convF32Kernel, dim3((total + F32_THREADS - 1) / F32_THREADS, a.N), dim3(F32_THREADS), 0, stream, a, ks);
  return hipGetLastError();
}

}  // namespace kmx

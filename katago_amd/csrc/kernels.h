// kernels.h — launch interface of the gfx950 kernels (conv_mfma.hip, misc_kernels.hip).
//
// Data layout in HBM:
//   activations : T[N][S][Cs]   "NHWC", S = nnY*nnX, Cs = channel stride, a multiple of 32;
//                 channels beyond the real count are zero; cells off the board are zero.
//   conv weights: T[chunk][tap][coutPad][32]  chunk = 32 input channels, tap = ky*KS+kx; within a row the four
//                 8-value slots are stored at slot ^ ((cout>>2)&3): the LDS image the kernel wants (conflict-free
//                 ds_read_b128) is then a plain linear copy for global_load_lds.
//   T is fp16 or bf16 (KMX_PREC_FP16 / KMX_PREC_BF16); all accumulation and epilogue math is fp32.
#ifndef KMX_KERNELS_H_
#define KMX_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmx {

// DT_F32 (round 5): fp32 storage and arithmetic - the verification mode behind KMX_PREC_FP32 (useFP16Mode = False,
// cpp/neuralnet/nninterface.h:50-63; what the reference's testgpuerror builds beside the 16-bit evaluator, command/gputest.cpp:122-133).
// Same schedule, same buffers at four bytes per value, the small kernels instantiated for float; the convolutions run a plain
// one-thread-per-output kernel (conv_f32.hip) - correct, not fast. Fused seams, chained convolutions and transformer blocks are 16-bit only.
enum { DT_F16 = 0, DT_BF16 = 1, DT_F32 = 2 };
inline int dtSize(int dtype) { return dtype == DT_F32 ? 4 : 2; }

constexpr int KCHUNK = 32;      // input channels per K chunk
constexpr int ZERO_PAGE_BYTES = 16384;  // >= (inC/32 + 4) * 64
constexpr int TRASH_BYTES = 1024;       // writable scratch BEHIND the zero page: where the convolution's stores that must not land go (16 bytes per lane)
constexpr int ZERO_PAGE_ALLOC = ZERO_PAGE_BYTES + TRASH_BYTES;
// readable bytes behind EVERY device allocation (engine.cpp DevBuf): the image requests of the convolution kernels keep advancing their
// per-lane source pointers 64 bytes per chunk past the last chunk - up to RING DEPTH chunks beyond the last cell's row, destination
// redirected to a slack area, the read itself still issued. Each kernel family static_asserts its run-ahead against this.
constexpr int DEVBUF_TAIL_BYTES = 512;
constexpr int WROW_HALFS = 32;  // halfs per weight/activation LDS row (64 bytes = four 16-byte slots, XOR-swizzled)

// One fused convolution: out = epilogue( conv(in, w) ).
// Epilogue per output channel c in [0, coutPad) and board cell p:
//   v = acc(p,c) + ncBias[n][c] (if ncBias) + resid[n][p][c - rawBegin] (if resid and c in raw range)
//   raw range [rawBegin, rawEnd):  rawOut[n][p][c - rawBegin] = v
//   act range [actBegin, actEnd):  actOut[n][p][c - actBegin] = act(v*scale[c] + bias[c]) * mask[n][p]
// (reference ops fused here: BatchNormLayer::apply eigenbackend.cpp:739-762, addNCBiasInplace :137-148,
//  the residual accumulate of ConvLayer::apply :659-686)
struct ConvArgs {
  const void* in;
  const void* w;
  // 3x3, 16-bit only (else null): the SAME weights in MFMA-fragment order for the shapes that load them straight into registers
  // (conv_small_kernel.h REGW) - T[chunk][tap][coutPad/32][k half of the chunk (2)][lane (64)][8]: lane l holds output channel
  // 32 tile + (l & 31), input channels 16 khalf + 8 (l >> 5) + [0, 8) of the chunk - a wave's fragment load is 1 KB of consecutive bytes
  const void* wFrag;
  const void* zeroPage;  // ZERO_PAGE_BYTES of zeros (halo lanes walk it 64 bytes per input-channel chunk), followed by TRASH_BYTES of writable scratch
  int inC;               // channel stride of `in`
  int nChunks;           // ceil(real Cin / 32)
  int coutPad;           // multiple of 32
  int N, X, Y;
  const float* ncBias;
  int ncBiasStride;
  const void* resid;
  int residC;
  void* rawOut;
  int rawC, rawBegin, rawEnd;
  void* actOut;
  int actC, actBegin, actEnd;
  const float* scale;
  const float* bias;
  int actKind;
  const float* mask;  // [N][S]
  unsigned long long* dbg;  // instrumentation only (ABL_TIMING variants of conv_bench.hip): per-segment cycle sums
};

// KS in {1,3,5}; cfg = 10*WNW + WN (WNW waves along channels: 1 = 4-wave, 2 = 8-wave work-group; WN = 32-channel
// tiles per wave). Returns hipSuccess or an error (unsupported combination -> hipErrorInvalidValue).
hipError_t launchConv(int dtype, int ks, int cfg, const ConvArgs& a, hipStream_t stream);
hipError_t launchConvF32(int ks, const ConvArgs& a, hipStream_t stream);  // conv_f32.hip: the DT_F32 form of the contract above (cfg is ignored)
int chooseConvCfg(int ks, int coutPad, int batch);
bool convCfgInstantiated(int ks, int cfg);  // is there a kernel for this (kernel size, shape)?
const char* convTuneError();  // null, or why the debug override KMX_CONV_TUNE could not be parsed (engine construction then fails)

// A chain of 2..MAX_CHAIN 3x3 convolutions 192 -> 192 of one board as ONE launch (conv_chain_kernel.h): the inner residual blocks of a
// nested-bottleneck block. Convolution i reads the activated image of convolution i - 1 (conv 0: `in`); every tensor has channel
// stride 192. Epilogue of convolution i, per channel c and cell p:  v = acc + resid[p][c] (if resid);  rawOut[p][c] = v (if rawOut);
// act = actKind((v * scale[c] + bias[c])) * mask[p]: the LAST convolution stores all of it to actOut; an earlier one hands channels
// [0, 96) to its successor inside the CU and stores only channels [96, 192) to actOut - a scratch tensor nobody else may rely on.
constexpr int MAX_CHAIN = 4;
constexpr int CHAIN_CHANNELS = 192;
struct ChainConv {
  const void* w;       // T[6][9][192][32], the FusedConv layout
  const float* scale;  // [192] merged BN of the activation that follows this convolution
  const float* bias;
  const void* resid;   // T[N][S][192] or null (may alias rawOut)
  void* rawOut;        // T[N][S][192] or null; a convolution with a residual must have one
  void* actOut;        // T[N][S][192], never null
};
struct ConvChainArgs {
  const void* in;        // T[N][S][192]
  const void* zeroPage;  // as ConvArgs::zeroPage
  const float* mask;     // [N][S]
  int N, X, Y;
  int nConv;
  int actKind;           // one activation kind for the whole chain
  ChainConv conv[MAX_CHAIN];
  unsigned long long* dbg;  // instrumentation only (conv_bench.hip benchConvChain): per-wave cycle sums of the phases of each convolution
};
hipError_t launchConvChain(int dtype, const ConvChainArgs& a, hipStream_t stream);
bool convChainSupported(int actKind);  // is there a kernel for this activation?

// The seam between two nested-bottleneck blocks as ONE launch (pointwise_kernel.h): block i's closing 1x1 convolution
// (+ residual), block i+1's preBN + activation, block i+1's opening 1x1 convolution and the first inner block's
// preBN + activation. The activated trunk image only exists in LDS. Cells are the flat N*S index.
struct PwPairArgs {
  // ALIASING CONTRACT: `in` may be the same buffer as actOut2 / rawOut2 only if inC == midC (a work-group then writes only rows
  // it has already fetched: 128 consecutive cells, all channels); the engine fuses a seam under that condition only.
  const void* in;        // T [cells][inC]: activated mid image of block i
  int inC;
  const void* w1;        // T [C1/32][C2][32], rows slot-swizzled (the FusedConv layout)
  const void* resid;     // T [cells][trunkC]: trunk raw (may alias rawOut)
  void* rawOut;          // T [cells][trunkC]
  int trunkC;            // channel stride of the trunk tensors (>= C2)
  void* actOut;          // T [cells][trunkC] or null: the activated trunk image, only written when something else reads it
  const float* scale1;   // [C2] merged BN of block i+1's preBN
  const float* bias1;
  int actKind1;
  const void* w2;        // T [C2/32][C3][32]
  void* rawOut2;         // T [cells][midC]
  void* actOut2;         // T [cells][midC]
  int midC;              // channel stride of the mid tensors (>= C3)
  const float* scale2;   // [C3]
  const float* bias2;
  int actKind2;
  const float* mask;     // [cells]
  long long cells;       // N * S
  const void* zeroPage;  // >= 64 readable zero bytes
  unsigned long long* dbg;  // instrumentation only (conv_bench.hip benchSeam): per-wave cycle sums of the persistent kernel's phases
  // 1: the launch has the chip to itself. The persistent, software-pipelined kernel (one work-group per CU for the whole launch) is
  // taken when the launch is alone (39.6 k against 37.9 k evals/s at batch 256 on one stream, 28.6 k against 28.1 k at batch 128) or
  // has at least two tiles per CU whatever runs beside it (two streams of 256 boards: 40.5 k against 39.1 k; the batcher's 256-row
  // batches, two in flight: 39.6 k against 37.4 k rows/s). A launch of fewer tiles beside other streams' kernels - the two 128-board
  // halves of a split batch: 361 tiles each - takes the one-tile-per-work-group kernel: short work-groups free their CU tile by tile
  // and let the other stream's kernels in (41.5 k against 40.5-40.9 k on one box, equal within noise on another;
  // profiles/r03_steps/seam_two_streams*.txt).
  int alone;
};
hipError_t launchPointwisePair(int dtype, int c1, int c2, int c3, const PwPairArgs& a, hipStream_t stream);
bool pointwisePairSupported(int c1, int c2, int c3);  // is there a kernel for these channel counts?

// Input staging: fp32 NHWC rows (not symmetrised) -> T[N][S][32] symmetrised + mask + maskSum +
// ncBias[n][C] = W_global^T * global[n]   (copyInputsWithSymmetry nninputs.cpp:529-597, Model::apply
// eigenbackend.cpp:2181-2182, initialMatMul eigenbackend.cpp:1928-1930)
struct InputArgs {
  const float* spatial;  // [N][S][cin], or null when `packed` is given
  const unsigned char* packed;  // [N][cin][ceil(S/8)] bit planes, MSB first (binaryInputNCHWPacked layout), or null
  const float* global;   // [N][gin]
  const int* symmetry;   // [N] device
  int cin, gin;
  int N, X, Y;
  void* out;             // T[N][S][32]
  float* mask;           // [N][S]
  float* maskSum;        // [N]
  const float* wGlobal;  // [gin][C]
  float* ncBias;         // [N][ncStride]
  int C, ncStride;
  // sgf-metadata encoder (null meta = none): ncBias += W3^T act2(W2^T act1(W1^T meta + b1) + b2)   (eigenbackend.cpp:1848-1860)
  const float* meta;     // [N][metaIn] device, or null
  int metaIn, metaC1, metaC2, metaAct1, metaAct2;
  const float* mW1;      // [metaIn][metaC1]
  const float* mB1;
  const float* mW2;      // [metaC1][metaC2]
  const float* mB2;
  const float* mW3;      // [metaC2][C]
};
hipError_t launchInputExpand(int dtype, const InputArgs& a, hipStream_t stream);

// Global pooling + bias + BN/act for a gpool residual block and for the policy head:
//   feat = poolRowsGPool(g)  (eigenbackend.cpp:152-177);  b = W^T feat (MatMulLayer);
//   r[n][p][c] = act((r + b[c])*scale[c] + bias[c]) * mask      (addNCBiasInplace + BatchNormLayer)
struct GPoolArgs {
  const void* g;  // T, activated gpool channels
  int gStride, gOffset, G;
  void* r;        // T, in place
  int rStride, rOffset, R;
  const float* w;  // [3G][R]
  const float* scale;
  const float* bias;  // [R]
  int actKind;
  const float* mask;
  const float* maskSum;
  float* featOut;  // [N][3G] or null
  int N, S;
};
hipError_t launchGPoolApply(int dtype, const GPoolArgs& a, hipStream_t stream);

// Policy head tail: p2Conv (1x1), pass logits, optimism blend, inverse symmetry
// (PolicyHead::apply eigenbackend.cpp:2019-2035; getOutput :2553-2567)
struct PolicyArgs {
  const void* p;  // T p1 after BN/act
  int pStride, pOffset, P;
  const float* w2;       // [P][NP]
  int NP;                // policy channels 1,2,4
  const float* feat;     // [N][3G]
  int G3;
  const float* wPass;    // [3G][passHidden or NP]
  const float* bPass;    // [passHidden] or null
  const float* wPass2;   // [passHidden][NP] or null
  int passHidden;        // 0 => single matmul
  int passAct;
  const int* symmetry;
  const float* optimism;  // [N] device
  float* out;             // [N][S+1]
  int N, X, Y;
};
hipError_t launchPolicyFinal(int dtype, const PolicyArgs& a, hipStream_t stream);

// Value head tail: value pooling, v2/v3/sv3 MLP, ownership conv + inverse symmetry
// (ValueHead::apply eigenbackend.cpp:2100-2113; poolRowsValueHead :179-197)
struct ValueArgs {
  const void* v;  // T v1 after BN/act
  int vStride, vOffset, V1;
  const float* w2;  // [3V1][V2]
  const float* b2;
  int V2, v2Act;
  const float* w3;  // [V2][3]
  const float* b3;
  const float* wsv;  // [V2][NSV]
  const float* bsv;
  int NSV;
  const float* wOwn;  // [V1]
  const float* maskSum;
  const int* symmetry;
  float* value;      // [N][3]
  float* score;      // [N][6]
  float* ownership;  // [N][S]
  int N, X, Y;
};
hipError_t launchValueFinal(int dtype, const ValueArgs& a, hipStream_t stream);

// ---- model-v17 transformer trunks (transformer_kernels.hip; experimental, see the header of that file) ----
// RMSNorm over channels: out = act(in * r * w + beta) on on-board cells, 0 elsewhere and in channels [C, outStride);
// r = 1/sqrt(mean_c(in^2) + eps) per cell, or boardRms[n] (launchBoardRms) for the per-board trunk tip.
struct RmsNormArgs {
  const void* in;   // T [N][S][inStride]
  int inStride;
  void* out;        // T [N][S][outStride]
  int outStride;
  int C;
  float eps;
  const float* w;     // [C]
  const float* beta;  // [C] or null
  int actKind;
  const float* boardRms;  // [N] or null
  const float* mask;      // [N][S]
  int N, S;
};
hipError_t launchRmsNorm(int dtype, const RmsNormArgs& a, hipStream_t stream);
hipError_t launchBoardRms(int dtype, const void* in, int inStride, int C, const float* mask, const float* maskSum, int N, int S,
                          float eps, float* outRms, hipStream_t stream);
// Masked softmax attention with grouped-query heads and 2D RoPE, one work-group per (head, board).
struct AttentionArgs {
  const void* qkv;  // T [N][S][stride]: q at [0, H*QD), k at [kOff, kOff + KVH*QD), v at [vOff, vOff + KVH*VD)
  int stride, kOff, vOff;
  int H, KVH, QD, VD;
  const float* ropeCos;  // [ropeHeads][QD/2][S], or null without RoPE
  const float* ropeSin;
  int ropeHeads;         // 1 = one table for all heads (fixed theta), KVH = per KV head (learnable frequencies)
  const float* mask;     // [N][S]
  void* out;             // T [N][S][outStride], head h at channels [h*VD, (h+1)*VD)
  int outStride;
  float scale;           // 1/sqrt(QD)
  int N, S;
};
hipError_t launchAttention(int dtype, const AttentionArgs& a, hipStream_t stream);
bool attentionDimsSupported(int qHeadDim, int vHeadDim);
// SwiGLU: out[c] = silu(in[c]) * in[gOff + c] for c < F; channels [F, outStride) = 0
struct SwiGluArgs {
  const void* in;
  int inStride, gOff, F;
  void* out;
  int outStride;
  size_t cells;
};
hipError_t launchSwiGlu(int dtype, const SwiGluArgs& a, hipStream_t stream);

// Stand-alone masked BN + activation (only the layer test hooks need it un-fused).
struct BnActArgs {
  const void* in;
  void* out;
  int stride, C;
  const float* scale;
  const float* bias;
  int actKind;
  const float* mask;
  int N, S;
};
hipError_t launchBnAct(int dtype, const BnActArgs& a, hipStream_t stream);

// fp32 <-> T conversion of dense activation tensors (test hooks / debugging)
hipError_t launchFloatToT(int dtype, const float* in, int inC, void* out, int outStride, size_t cells, hipStream_t stream);
hipError_t launchTToFloat(int dtype, const void* in, int inStride, int offset, float* out, int outC, size_t cells, hipStream_t stream);
// Fault triage (KMX_DEBUG_SQUAT, engine.cpp; tests): `blocks` one-wave work-groups that hold `ldsBytes` of LDS each for `usec` microseconds,
// fill them with a pattern and count, into *corrupt, the words that no longer hold it at the end. Launched beside a kernel under test it
// makes that kernel's work-groups start at a NONZERO LDS base on the compute units they share - where a kernel runs whenever another
// stream's kernels are on the chip - and notices writes that land outside the kernel's own allocation.
hipError_t launchLdsSquatter(int blocks, int ldsBytes, int usec, unsigned* corrupt, hipStream_t stream);

// host helpers
uint16_t floatToHalfBits(float f);
uint16_t floatToBf16Bits(float f);
inline uint16_t floatToTBits(int dtype, float f) { return dtype == DT_F16 ? floatToHalfBits(f) : floatToBf16Bits(f); }

}  // namespace kmx
#endif

// model_desc.h — host-side description of a KataGo convolutional net, parsed from the reference's
// model file format (.bin / .txt, optionally gzipped).
//
// Format reference: cpp/neuralnet/desc.cpp:40-90 (float blocks), :110-155 (conv), :208-289 (batchnorm),
// :382-403 (activation), :451-479 (matmul), :518-535 (matbias), :566-593 / :652-702 / :783-818 (blocks),
// :1444-1562 (block stack), :1669-1768 (trunk), :2051-2155 (policy head), :2242-2340 (value head),
// :2441-2615 (model header). Weights are kept in MODEL-FILE order; the engine re-tiles them for the GPU.
#ifndef KMX_MODEL_DESC_H_
#define KMX_MODEL_DESC_H_

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace kmx {

struct ModelError : std::runtime_error {
  int code;
  ModelError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

struct ConvDesc {
  std::string name;
  int ky = 0, kx = 0, inC = 0, outC = 0;
  std::vector<float> w;  // [ky][kx][ic][oc]  (file order)
  float at(int y, int x, int ic, int oc) const { return w[(((size_t)y * kx + x) * inC + ic) * outC + oc]; }
};
struct BnDesc {  // batch norm merged to scale/bias, plus the activation that follows it
  std::string name;
  int c = 0;
  int act = 0;  // KMX_ACT_*
  std::vector<float> scale, bias;
};
struct MatMulDesc {
  std::string name;
  int inC = 0, outC = 0;
  std::vector<float> w;  // [ic][oc]
};
struct MatBiasDesc {
  std::string name;
  int c = 0;
  std::vector<float> w;
};

// RMSNorm inside transformer blocks (TransformerRMSNormDesc, desc.cpp:1125-1144): y = x / rms(x over channels) * w
struct TRmsDesc {
  std::string name;
  int c = 0;
  float eps = 0.0f;
  std::vector<float> w;
};

enum class BlockKind { Ordinary, GPool, Nested, Attention, FFN };

struct BlockDesc {
  BlockKind kind = BlockKind::Ordinary;
  std::string name;
  // Ordinary: preBN, regularConv, midBN, finalConv
  // GPool:    preBN, regularConv, gpoolConv, gpoolBN, gpoolToBiasMul, midBN, finalConv
  // Nested:   preBN, regularConv (= preConv), inner[], midBN (= postBN), finalConv (= postConv)
  BnDesc preBN, midBN, gpoolBN;
  ConvDesc regularConv, finalConv, gpoolConv;
  MatMulDesc gpoolToBiasMul;
  std::vector<BlockDesc> inner;
  // Attention (TransformerAttentionDesc, desc.cpp:1173-1258): preLN, qProj/kProj/vProj/outProj, 2D RoPE
  // FFN       (TransformerFFNDesc, desc.cpp:1371-1405):       preLN, linear1, linearGate (SwiGLU), linear2
  TRmsDesc preLN;
  int numHeads = 0, numKVHeads = 0, qHeadDim = 0, vHeadDim = 0;
  bool useRope = false, learnableRope = false;
  float ropeTheta = 0.0f;
  std::vector<float> ropeFreqs;  // learnable: [numKVHeads][qHeadDim/2][2] = (omega_x, omega_y)
  MatMulDesc qProj, kProj, vProj, outProj;
  int ffnChannels = 0;
  MatMulDesc linear1, linearGate, linear2;
  bool isTransformer() const { return kind == BlockKind::Attention || kind == BlockKind::FFN; }
};

struct ModelDesc {
  std::string name;
  int version = 0;
  int numInputChannels = 0, numInputGlobalChannels = 0;
  int numPolicyChannels = 0, numValueChannels = 0, numScoreValueChannels = 0, numOwnershipChannels = 0;
  float postProcess[7] = {20.0f, 20.0f, 20.0f, 20.0f, 40.0f, 0.25f, 30.0f};  // desc.h ModelPostProcessParams defaults
  int trunkC = 0, midC = 0, regularC = 0, gpoolC = 0, numBlocks = 0;

  ConvDesc initialConv;
  MatMulDesc initialMatMul;
  // sgf-metadata encoder (desc.cpp:1571-1625): meta[192] -> mul1+bias1+act1 -> mul2+bias2+act2 -> mul3 -> [trunkC],
  // added to the trunk like the global-feature bias (eigenbackend.cpp:1929-1932). metaEncoderVersion 0 = none.
  int metaEncoderVersion = 0, numInputMetaChannels = 0;
  MatMulDesc metaMul1, metaMul2, metaMul3;
  MatBiasDesc metaBias1, metaBias2;
  int metaAct1 = 0, metaAct2 = 0;
  std::vector<BlockDesc> blocks;
  BnDesc trunkTipBN;
  // trunkNormKind != 0: the tip is an RMSNorm (RMSNormLayerDesc, desc.cpp:1069-1095) with gamma/beta and an activation;
  // rmsSpatial: one RMS per board over on-board cells x channels instead of one per cell
  int trunkNormKind = 0, trunkTipAct = 0;
  bool rmsSpatial = false;
  float rmsEps = 0.0f;
  std::vector<float> rmsGamma, rmsBeta;
  bool hasTransformerBlocks = false;

  ConvDesc p1Conv, g1Conv, p2Conv;
  BnDesc g1BN, p1BN;
  MatMulDesc gpoolToBiasMul, gpoolToPassMul, gpoolToPassMul2;
  MatBiasDesc gpoolToPassBias;
  int passAct = 0;
  bool hasPassMLP = false;

  ConvDesc v1Conv, vOwnershipConv;
  BnDesc v1BN;
  MatMulDesc v2Mul, v3Mul, sv3Mul;
  MatBiasDesc v2Bias, v3Bias, sv3Bias;
  int v2Act = 0;

  int64_t numParameters = 0;
  double macPerPosition = 0.0;  // direct-convolution MACs per board point
  std::string sha256;           // hex digest of the uncompressed file contents

  // The fp16 range transform of the reference (desc.cpp:2718-2736; see model_desc.cpp): a copy of this net whose tensors all
  // carry 1/8 of their values and whose outputs are unchanged. scale8Applies(): standard batch norm trunk, no transformer
  // blocks, activations among identity / relu / mish.
  bool scale8Applied = false;
  bool scale8Applies() const;
  std::unique_ptr<ModelDesc> scaledBy8() const;

  // Throws ModelError. expectedSha256 may be empty (no check).
  static std::unique_ptr<ModelDesc> loadFromFile(const std::string& path, const std::string& expectedSha256);
};

std::string sha256Hex(const unsigned char* data, size_t len);

}  // namespace kmx
#endif

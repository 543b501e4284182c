// model_desc.cpp — parser for KataGo model files. See model_desc.h for the format citations.
#include "model_desc.h"

#include <cstdlib>

#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/katamx.h"

namespace kmx {

// ------------------------------------------------------------------------------------------------
// SHA-256 (FIPS 180-4). The reference verifies the digest of the file bytes as stored on disk
// (cpp/core/fileutils.cpp:117-141).
namespace {
inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
const uint32_t K256[64] = {
  0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
  0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
  0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
  0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
  0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
  0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
  0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
void sha256Block(uint32_t h[8], const unsigned char* p) {
  uint32_t w[64];
  for(int i = 0; i < 16; i++)
    w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
  for(int i = 16; i < 64; i++) {
    uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for(int i = 0; i < 64; i++) {
    uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
    uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
}  // namespace

std::string sha256Hex(const unsigned char* data, size_t len) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t full = len / 64;
  for(size_t i = 0; i < full; i++) sha256Block(h, data + 64 * i);
  unsigned char tail[128];
  size_t rem = len - 64 * full;
  memset(tail, 0, sizeof(tail));
  memcpy(tail, data + 64 * full, rem);
  tail[rem] = 0x80;
  size_t tlen = rem + 1 + 8 <= 64 ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for(int i = 0; i < 8; i++) tail[tlen - 1 - i] = (unsigned char)(bits >> (8 * i));
  sha256Block(h, tail);
  if(tlen == 128) sha256Block(h, tail + 64);
  char out[65];
  for(int i = 0; i < 8; i++) snprintf(out + 8 * i, 9, "%08x", h[i]);
  return std::string(out, 64);
}

// ------------------------------------------------------------------------------------------------
namespace {

[[noreturn]] void bad(const std::string& msg) { throw ModelError(KMX_ERR_MODEL, msg); }

bool endsWithLower(const std::string& s, const char* suf) {
  size_t n = strlen(suf);
  if(s.size() < n) return false;
  for(size_t i = 0; i < n; i++)
    if(tolower((unsigned char)s[s.size() - n + i]) != suf[i]) return false;
  return true;
}

std::vector<unsigned char> readWholeFile(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if(!f) throw ModelError(KMX_ERR_IO, "could not open model file " + path);
  std::vector<unsigned char> data;
  unsigned char chunk[1 << 16];
  size_t n;
  while((n = fread(chunk, 1, sizeof(chunk), f)) > 0) {
    data.insert(data.end(), chunk, chunk + n);
    if(n < sizeof(chunk)) break;  // end of file or error: do not read again
  }
  bool err = ferror(f) != 0;
  fclose(f);
  if(err) throw ModelError(KMX_ERR_IO, "error while reading model file " + path);
  return data;
}

std::vector<unsigned char> gunzip(const std::vector<unsigned char>& in, const std::string& path) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if(inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) throw ModelError(KMX_ERR_INTERNAL, "zlib inflateInit2 failed");
  std::vector<unsigned char> out;
  out.resize(in.size() * 2 + (1 << 20));
  zs.next_in = const_cast<unsigned char*>(in.data());
  zs.avail_in = (uInt)in.size();
  size_t produced = 0;
  for(;;) {
    if(produced == out.size()) out.resize(out.size() * 2);
    size_t room = out.size() - produced;
    if(room > (1u << 30)) room = 1u << 30;
    zs.next_out = out.data() + produced;
    zs.avail_out = (uInt)room;
    int rc = inflate(&zs, Z_NO_FLUSH);
    produced += room - zs.avail_out;
    if(rc == Z_STREAM_END) break;
    if(rc != Z_OK && rc != Z_BUF_ERROR) {
      inflateEnd(&zs);
      throw ModelError(KMX_ERR_IO, "could not decompress " + path + " (not a gzip file?)");
    }
    if(rc == Z_BUF_ERROR && zs.avail_in == 0) {
      inflateEnd(&zs);
      throw ModelError(KMX_ERR_IO, "truncated gzip stream in " + path);
    }
  }
  inflateEnd(&zs);
  out.resize(produced);
  return out;
}

// Token / float-block reader over the uncompressed bytes.
class Reader {
 public:
  Reader(const std::vector<unsigned char>& d, bool binaryFloats) : d_(d), pos_(0), binary_(binaryFloats) {}

  std::string token(const char* what) {
    while(pos_ < d_.size() && isspace(d_[pos_])) pos_++;
    if(pos_ >= d_.size()) bad(std::string("unexpected end of model file while reading ") + what);
    size_t b = pos_;
    while(pos_ < d_.size() && !isspace(d_[pos_])) pos_++;
    return std::string((const char*)&d_[b], pos_ - b);
  }
  int integer(const char* what) {
    std::string t = token(what);
    char* e;
    long v = strtol(t.c_str(), &e, 10);
    if(e == t.c_str() || *e != 0) bad(std::string(what) + ": expected an integer but found '" + t + "'");
    return (int)v;
  }
  float real(const char* what) {
    std::string t = token(what);
    char* e;
    float v = strtof(t.c_str(), &e);
    if(e == t.c_str() || *e != 0) bad(std::string(what) + ": expected a number but found '" + t + "'");
    return v;
  }
  // desc.cpp:40-90
  std::vector<float> floats(size_t n, const std::string& name) {
    std::vector<float> out(n);
    if(!binary_) {
      for(size_t i = 0; i < n; i++) out[i] = real(name.c_str());
    }
    else {
      int skipped = 0;
      while(pos_ < d_.size() && d_[pos_] != '@') {
        pos_++;
        if(++skipped > 100)
          bad(name + ": could not read float weights. Invalid model - perhaps you are trying to load a .txt.gz model as a .bin.gz model?");
      }
      if(pos_ + 5 > d_.size() || memcmp(&d_[pos_], "@BIN@", 5) != 0)
        bad(name + ": did not find expected header for binary float block");
      pos_ += 5;
      if(pos_ + 4 * n > d_.size()) bad(name + ": did not find the expected number of floats in binary float block");
      memcpy(out.data(), &d_[pos_], 4 * n);  // little-endian host assumed (x86-64)
      pos_ += 4 * n;
    }
    for(size_t i = 0; i < n; i++)
      if(!std::isfinite(out[i])) bad(name + ": Nan or infinite neural net weight or parameter");
    return out;
  }
  void expectZeros(int count, const char* what) {
    for(int i = 0; i < count; i++)
      if(integer(what) != 0) bad(std::string("unknown/unsupported ") + what);
  }

 private:
  const std::vector<unsigned char>& d_;
  size_t pos_;
  bool binary_;
};

ConvDesc parseConv(Reader& r) {
  ConvDesc c;
  c.name = r.token("conv name");
  c.ky = r.integer("convYSize");
  c.kx = r.integer("convXSize");
  c.inC = r.integer("inChannels");
  c.outC = r.integer("outChannels");
  int dy = r.integer("dilationY"), dx = r.integer("dilationX");
  if(c.ky <= 0 || c.kx <= 0) bad(c.name + ": convolution filter sizes must be positive");
  if(c.inC <= 0 || c.outC <= 0) bad(c.name + ": number of in and out channels must be positive");
  if(c.ky % 2 != 1 || c.kx % 2 != 1) bad(c.name + ": convolution filter sizes must be odd, found even sizes");
  if(dy != 1 || dx != 1) bad(c.name + ": dilated convolutions are not supported by this backend");
  c.w = r.floats((size_t)c.ky * c.kx * c.inC * c.outC, c.name);
  return c;
}
BnDesc parseBn(Reader& r) {
  BnDesc b;
  b.name = r.token("bn name");
  b.c = r.integer("numChannels");
  float eps = r.real("epsilon");
  int hasScale = r.integer("hasScale"), hasBias = r.integer("hasBias");
  if(b.c < 1) bad(b.name + ": numChannels < 1");
  if(!(eps > 0) || !std::isfinite(eps)) bad(b.name + ": epsilon is not positive and finite");
  std::vector<float> mean = r.floats(b.c, b.name), var = r.floats(b.c, b.name), scale, bias;
  if(hasScale) scale = r.floats(b.c, b.name);
  if(hasBias) bias = r.floats(b.c, b.name);
  b.scale.resize(b.c);
  b.bias.resize(b.c);
  for(int i = 0; i < b.c; i++) {  // computeMerged, desc.cpp:272-279
    b.scale[i] = (hasScale ? scale[i] : 1.0f) / std::sqrt(var[i] + eps);
    b.bias[i] = (hasBias ? bias[i] : 0.0f) - b.scale[i] * mean[i];
  }
  return b;
}
int parseAct(Reader& r, int version) {
  (void)r.token("activation name");
  if(version < 11) return KMX_ACT_RELU;
  std::string k = r.token("activation kind");
  if(k == "ACTIVATION_IDENTITY") return KMX_ACT_IDENTITY;
  if(k == "ACTIVATION_RELU") return KMX_ACT_RELU;
  if(k == "ACTIVATION_MISH") return KMX_ACT_MISH;
  if(k == "ACTIVATION_SILU") return KMX_ACT_SILU;
  bad("unknown activation " + k);
}
MatMulDesc parseMatMul(Reader& r) {
  MatMulDesc m;
  m.name = r.token("matmul name");
  m.inC = r.integer("inChannels");
  m.outC = r.integer("outChannels");
  if(m.inC <= 0 || m.outC <= 0) bad(m.name + ": number of in and out channels must be positive");
  m.w = r.floats((size_t)m.inC * m.outC, m.name);
  return m;
}
MatBiasDesc parseMatBias(Reader& r) {
  MatBiasDesc m;
  m.name = r.token("matbias name");
  m.c = r.integer("numChannels");
  if(m.c <= 0) bad(m.name + ": number of channels must be positive");
  m.w = r.floats(m.c, m.name);
  return m;
}

TRmsDesc parseTRms(Reader& r) {
  TRmsDesc t;
  t.name = r.token("rmsnorm name");
  t.c = r.integer("numChannels");
  t.eps = r.real("epsilon");
  if(t.c < 1) bad(t.name + ": number of channels must be positive");
  if(!(t.eps > 0.0f) || t.eps > 1.0f) bad(t.name + ": epsilon out of range");
  t.w = r.floats(t.c, t.name);
  return t;
}

std::vector<BlockDesc> parseStack(Reader& r, int version, int numBlocks, int trunkC, const std::string& owner);

BlockDesc parseBlock(Reader& r, int version, int trunkC, const std::string& owner) {
  BlockDesc b;
  std::string kind = r.token("block kind");
  if(kind == "ordinary_block") {
    b.kind = BlockKind::Ordinary;
    b.name = r.token("block name");
    b.preBN = parseBn(r);
    b.preBN.act = parseAct(r, version);
    b.regularConv = parseConv(r);
    b.midBN = parseBn(r);
    b.midBN.act = parseAct(r, version);
    b.finalConv = parseConv(r);
    if(b.preBN.c != b.regularConv.inC || b.midBN.c != b.regularConv.outC || b.midBN.c != b.finalConv.inC)
      bad(b.name + ": residual block channel counts are inconsistent");
  }
  else if(kind == "gpool_block") {
    b.kind = BlockKind::GPool;
    b.name = r.token("block name");
    b.preBN = parseBn(r);
    b.preBN.act = parseAct(r, version);
    b.regularConv = parseConv(r);
    b.gpoolConv = parseConv(r);
    b.gpoolBN = parseBn(r);
    b.gpoolBN.act = parseAct(r, version);
    b.gpoolToBiasMul = parseMatMul(r);
    b.midBN = parseBn(r);
    b.midBN.act = parseAct(r, version);
    b.finalConv = parseConv(r);
    if(b.preBN.c != b.regularConv.inC || b.preBN.c != b.gpoolConv.inC || b.gpoolBN.c != b.gpoolConv.outC ||
       b.gpoolBN.c * 3 != b.gpoolToBiasMul.inC || b.midBN.c != b.regularConv.outC ||
       b.midBN.c != b.gpoolToBiasMul.outC || b.midBN.c != b.finalConv.inC)
      bad(b.name + ": gpool block channel counts are inconsistent");
  }
  else if(kind == "nested_bottleneck_block") {
    b.kind = BlockKind::Nested;
    b.name = r.token("block name");
    int n = r.integer("nested numBlocks");
    if(n < 1) bad(b.name + ": nested bottleneck res block num blocks must be positive");
    b.preBN = parseBn(r);
    b.preBN.act = parseAct(r, version);
    b.regularConv = parseConv(r);
    b.inner = parseStack(r, version, n, b.regularConv.outC, b.name);
    b.midBN = parseBn(r);
    b.midBN.act = parseAct(r, version);
    b.finalConv = parseConv(r);
    if(b.preBN.c != b.regularConv.inC || b.midBN.c != b.regularConv.outC || b.midBN.c != b.finalConv.inC)
      bad(b.name + ": nested block channel counts are inconsistent");
  }
  else if(kind == "transformer_attention_block") {
    b.kind = BlockKind::Attention;
    b.name = r.token("block name");
    b.numHeads = r.integer("numHeads");
    b.numKVHeads = r.integer("numKVHeads");
    b.qHeadDim = r.integer("qHeadDim");
    b.vHeadDim = r.integer("vHeadDim");
    b.useRope = r.integer("useRope") != 0;
    b.learnableRope = r.integer("learnableRope") != 0;
    if(b.numHeads < 1 || b.numKVHeads < 1 || b.qHeadDim < 1 || b.vHeadDim < 1) bad(b.name + ": transformer attention dimensions must be positive");
    if(b.numHeads % b.numKVHeads != 0) bad(b.name + ": numHeads must be divisible by numKVHeads");
    if(b.useRope && b.qHeadDim % 2 != 0) bad(b.name + ": qHeadDim must be even for RoPE");
    b.preLN = parseTRms(r);
    b.qProj = parseMatMul(r);
    b.kProj = parseMatMul(r);
    b.vProj = parseMatMul(r);
    b.outProj = parseMatMul(r);
    if(b.preLN.c != trunkC || b.qProj.inC != trunkC || b.kProj.inC != trunkC || b.vProj.inC != trunkC ||
       b.qProj.outC != b.numHeads * b.qHeadDim || b.kProj.outC != b.numKVHeads * b.qHeadDim ||
       b.vProj.outC != b.numKVHeads * b.vHeadDim || b.outProj.inC != b.numHeads * b.vHeadDim || b.outProj.outC != trunkC)
      bad(b.name + ": transformer attention channel counts are inconsistent");
    if(b.useRope) {
      (void)r.token("rope parameter name");
      if(b.learnableRope) {
        const int kvh = r.integer("rope numKVHeads"), np = r.integer("rope numPairs"), two = r.integer("rope freq dims");
        if(kvh != b.numKVHeads || np != b.qHeadDim / 2 || two != 2) bad(b.name + ": learnable rope frequency table has the wrong shape");
        b.ropeFreqs = r.floats((size_t)kvh * np * 2, b.name);
      }
      else {
        b.ropeTheta = r.real("rope theta");
        if(!(b.ropeTheta > 0.0f) || !std::isfinite(b.ropeTheta)) bad(b.name + ": rope theta must be positive and finite");  // desc.cpp:1248
      }
    }
    return b;
  }
  else if(kind == "transformer_ffn_block") {
    b.kind = BlockKind::FFN;
    b.name = r.token("block name");
    const int nc = r.integer("numChannels");
    b.ffnChannels = r.integer("ffnChannels");
    const bool useSwiGLU = r.integer("useSwiGLU") != 0;
    if(nc < 1 || b.ffnChannels < 1) bad(b.name + ": transformer ffn channels must be positive");
    b.preLN = parseTRms(r);
    b.linear1 = parseMatMul(r);
    if(useSwiGLU) b.linearGate = parseMatMul(r);
    b.linear2 = parseMatMul(r);
    if(nc != trunkC || b.preLN.c != trunkC || b.linear1.inC != nc || b.linear1.outC != b.ffnChannels ||
       (useSwiGLU && (b.linearGate.inC != nc || b.linearGate.outC != b.ffnChannels)) || b.linear2.inC != b.ffnChannels ||
       b.linear2.outC != nc)
      bad(b.name + ": transformer ffn channel counts are inconsistent");
    if(!useSwiGLU)  // the reference's own backends refuse it too (eigenbackend.cpp:1631-1633, cudabackend.cpp:1996)
      throw ModelError(KMX_ERR_UNSUPPORTED, b.name + ": non-SwiGLU transformer FFN is not supported");
    return b;
  }
  else
    bad(owner + ": found unknown block kind: " + kind);
  if(b.preBN.c != trunkC || b.finalConv.outC != trunkC) bad(owner + ": " + b.name + " does not match the channel count of its residual stream");
  return b;
}
std::vector<BlockDesc> parseStack(Reader& r, int version, int numBlocks, int trunkC, const std::string& owner) {
  std::vector<BlockDesc> v;
  v.reserve(numBlocks);
  for(int i = 0; i < numBlocks; i++) v.push_back(parseBlock(r, version, trunkC, owner));
  return v;
}

bool anyTransformer(const std::vector<BlockDesc>& blocks) {
  for(const BlockDesc& b : blocks)
    if(b.isTransformer() || anyTransformer(b.inner)) return true;
  return false;
}

void countBlock(const BlockDesc& b, double& mac, int64_t& params) {
  if(b.isTransformer()) {
    // projections only; the QK^T / PV products cost area*numHeads*(qHeadDim+vHeadDim) MACs per point more (board dependent)
    for(const MatMulDesc* m : {&b.qProj, &b.kProj, &b.vProj, &b.outProj, &b.linear1, &b.linearGate, &b.linear2}) {
      mac += (double)m->inC * m->outC;
      params += (int64_t)m->inC * m->outC;
    }
    params += b.preLN.c + (int64_t)b.ropeFreqs.size();
    return;
  }
  auto conv = [&](const ConvDesc& c) {
    mac += (double)c.ky * c.kx * c.inC * c.outC;
    params += (int64_t)c.ky * c.kx * c.inC * c.outC;
  };
  conv(b.regularConv);
  conv(b.finalConv);
  params += 2 * b.preBN.c + 2 * b.midBN.c;
  if(b.kind == BlockKind::GPool) {
    conv(b.gpoolConv);
    params += 2 * b.gpoolBN.c + (int64_t)b.gpoolToBiasMul.inC * b.gpoolToBiasMul.outC;
  }
  for(const BlockDesc& i : b.inner) countBlock(i, mac, params);
}

}  // namespace

std::unique_ptr<ModelDesc> ModelDesc::loadFromFile(const std::string& path, const std::string& expectedSha256) {
  bool isGz = endsWithLower(path, ".gz");
  bool binaryFloats;
  if(endsWithLower(path, ".txt") || endsWithLower(path, ".txt.gz")) binaryFloats = false;
  else if(endsWithLower(path, ".bin") || endsWithLower(path, ".bin.gz") || isGz) binaryFloats = true;
  else
    bad("Model file should end with .txt, .bin, .txt.gz, .bin.gz, or possibly just .gz: " + path);

  std::vector<unsigned char> raw = readWholeFile(path);
  std::string digest = sha256Hex(raw.data(), raw.size());
  if(!expectedSha256.empty()) {
    std::string want = expectedSha256;
    for(char& ch : want) ch = (char)tolower((unsigned char)ch);
    if(want != digest)
      throw ModelError(KMX_ERR_MODEL, "File " + path + " sha256 was " + digest + " which does not match the expected sha256 " + expectedSha256);
  }
  std::vector<unsigned char> data = isGz ? gunzip(raw, path) : std::move(raw);

  std::unique_ptr<ModelDesc> mp(new ModelDesc());
  ModelDesc& m = *mp;
  m.sha256 = digest;
  try {
    Reader r(data, binaryFloats);
    m.name = r.token("model name");
    m.version = r.integer("model version");
    if(m.version < 0) bad("This neural net has an invalid version, you probably specified the wrong file.");
    if(m.version < 8)
      throw ModelError(KMX_ERR_UNSUPPORTED, "model version " + std::to_string(m.version) + " uses pre-v7 input features, which the katamx backend does not implement");
    if(m.version > 17) bad("This neural net requires a newer version: model version " + std::to_string(m.version));
    m.numInputChannels = r.integer("numInputChannels");
    m.numInputGlobalChannels = r.integer("numInputGlobalChannels");
    if(m.version >= 13)
      for(int i = 0; i < 7; i++) {
        m.postProcess[i] = r.real("post-process multiplier");
        if(!(m.postProcess[i] > 0)) bad(m.name + ": post-process multipliers must be positive");
      }
    if(m.version >= 15) {
      m.metaEncoderVersion = r.integer("metaEncoderVersion");
      // a flag the host reads from its own ModelDesc (nneval.cpp:306); any other value is a file from the future (desc.cpp:2538-2548)
      const int preferPassAlive = r.integer("preferPassAliveUnderSuicideRules");
      if(preferPassAlive != 0 && preferPassAlive != 1) bad(m.name + ": model preferPassAliveUnderSuicideRules unexpected value");
      r.expectZeros(6, "model option");
      if(m.metaEncoderVersion < 0) bad(m.name + ": model metaEncoderVersion unexpected value");
      if(m.metaEncoderVersion > 1)  // modelversion.cpp:83-89: only version 1 (192 input features) exists
        throw ModelError(KMX_ERR_UNSUPPORTED, m.name + ": metaEncoderVersion " + std::to_string(m.metaEncoderVersion) + " is not supported");
    }
    // trunk
    std::string trunkName = r.token("trunk name");
    m.numBlocks = r.integer("numBlocks");
    m.trunkC = r.integer("trunkNumChannels");
    m.midC = r.integer("midNumChannels");
    m.regularC = r.integer("regularNumChannels");
    (void)r.integer("dilatedNumChannels");
    m.gpoolC = r.integer("gpoolNumChannels");
    if(m.version >= 15) {
      m.trunkNormKind = r.integer("trunkNormKind");
      r.expectZeros(5, "trunk option");
      if(m.trunkNormKind != 0 && m.trunkNormKind != 1) bad(trunkName + ": unknown trunkNormKind");  // STANDARD / RMSNORM only, desc.cpp:1698
    }
    if(m.numBlocks < 1) bad(trunkName + ": trunk num blocks must be positive");
    if(m.trunkC <= 0 || m.midC <= 0 || m.regularC <= 0 || m.gpoolC <= 0) bad(trunkName + ": all numbers of channels must be positive");
    m.initialConv = parseConv(r);
    m.initialMatMul = parseMatMul(r);
    if(m.initialConv.outC != m.trunkC || m.initialMatMul.outC != m.trunkC) bad(trunkName + ": initial layers do not produce trunkNumChannels");
    if(m.metaEncoderVersion > 0) {  // SGFMetadataEncoderDesc, desc.cpp:1571-1625
      const std::string ename = r.token("sgf metadata encoder name");
      m.numInputMetaChannels = r.integer("numInputMetaChannels");
      if(m.numInputMetaChannels != 192) bad(ename + ": number of in channels did not match expected (192)");
      m.metaMul1 = parseMatMul(r);
      m.metaBias1 = parseMatBias(r);
      m.metaAct1 = parseAct(r, m.version);
      m.metaMul2 = parseMatMul(r);
      m.metaBias2 = parseMatBias(r);
      m.metaAct2 = parseAct(r, m.version);
      m.metaMul3 = parseMatMul(r);
      if(m.metaMul1.inC != m.numInputMetaChannels || m.metaMul1.outC != m.metaBias1.c || m.metaMul2.inC != m.metaMul1.outC ||
         m.metaMul2.outC != m.metaBias2.c || m.metaMul3.inC != m.metaMul2.outC || m.metaMul3.outC != m.trunkC)
        bad(ename + ": sgf metadata encoder channel counts are inconsistent");
    }
    m.blocks = parseStack(r, m.version, m.numBlocks, m.trunkC, trunkName);
    m.hasTransformerBlocks = anyTransformer(m.blocks);
    if(m.trunkNormKind == 0) {
      m.trunkTipBN = parseBn(r);
      m.trunkTipBN.act = parseAct(r, m.version);
      if(m.trunkTipBN.c != m.trunkC) bad(trunkName + ": trunkTipBN.numChannels != trunkNumChannels");
    }
    else {  // RMSNormLayerDesc, desc.cpp:1069-1095
      const std::string nname = r.token("trunk tip rmsnorm name");
      const int c = r.integer("numChannels");
      m.rmsEps = r.real("epsilon");
      m.rmsSpatial = r.integer("spatial") != 0;
      const int cgroup = r.integer("cgroupSize");
      if(c != m.trunkC) bad(nname + ": numChannels != trunkNumChannels");
      if(!(m.rmsEps > 0.0f) || m.rmsEps > 1.0f) bad(nname + ": epsilon out of range");
      if(cgroup != 0) throw ModelError(KMX_ERR_UNSUPPORTED, nname + ": grouped spatial RMSNorm is not supported");
      m.rmsGamma = r.floats(c, nname);
      m.rmsBeta = r.floats(c, nname);
      m.trunkTipAct = parseAct(r, m.version);
    }

    // policy head
    std::string pname = r.token("policy head name");
    if(m.version >= 17) {
      m.numPolicyChannels = r.integer("policyOutChannels");
      if(m.numPolicyChannels != 2 && m.numPolicyChannels != 4) bad(pname + ": invalid policyOutChannels");
      r.expectZeros(3, "policy option");
    }
    else if(m.version == 16) m.numPolicyChannels = 4;
    else if(m.version >= 12) m.numPolicyChannels = 2;
    else m.numPolicyChannels = 1;
    m.p1Conv = parseConv(r);
    m.g1Conv = parseConv(r);
    m.g1BN = parseBn(r);
    m.g1BN.act = parseAct(r, m.version);
    m.gpoolToBiasMul = parseMatMul(r);
    m.p1BN = parseBn(r);
    m.p1BN.act = parseAct(r, m.version);
    m.p2Conv = parseConv(r);
    m.gpoolToPassMul = parseMatMul(r);
    if(m.version >= 15) {
      m.hasPassMLP = true;
      m.gpoolToPassBias = parseMatBias(r);
      m.passAct = parseAct(r, m.version);
      m.gpoolToPassMul2 = parseMatMul(r);
    }
    if(m.p1Conv.outC != m.p1BN.c || m.g1Conv.outC != m.g1BN.c || m.gpoolToBiasMul.inC != 3 * m.g1BN.c ||
       m.gpoolToBiasMul.outC != m.p1BN.c || m.p2Conv.inC != m.p1BN.c || m.p2Conv.outC != m.numPolicyChannels ||
       m.gpoolToPassMul.inC != 3 * m.g1BN.c)
      bad(pname + ": policy head channel counts are inconsistent");
    if(m.hasPassMLP) {
      if(m.gpoolToPassMul.outC != m.gpoolToPassBias.c || m.gpoolToPassMul.outC != m.gpoolToPassMul2.inC ||
         m.gpoolToPassMul2.outC != m.numPolicyChannels)
        bad(pname + ": pass head channel counts are inconsistent");
    }
    else if(m.gpoolToPassMul.outC != m.numPolicyChannels)
      bad(pname + ": gpoolToPassMul.outChannels != policy channels");
    if(m.p1Conv.ky != 1 || m.p1Conv.kx != 1 || m.g1Conv.ky != 1 || m.g1Conv.kx != 1 || m.p2Conv.ky != 1 || m.p2Conv.kx != 1)
      throw ModelError(KMX_ERR_UNSUPPORTED, pname + ": policy head convolutions must be 1x1");

    // value head
    std::string vname = r.token("value head name");
    if(m.version >= 17) r.expectZeros(3, "value option");
    m.v1Conv = parseConv(r);
    m.v1BN = parseBn(r);
    m.v1BN.act = parseAct(r, m.version);
    m.v2Mul = parseMatMul(r);
    m.v2Bias = parseMatBias(r);
    m.v2Act = parseAct(r, m.version);
    m.v3Mul = parseMatMul(r);
    m.v3Bias = parseMatBias(r);
    m.sv3Mul = parseMatMul(r);
    m.sv3Bias = parseMatBias(r);
    m.vOwnershipConv = parseConv(r);
    m.numValueChannels = m.v3Mul.outC;
    m.numScoreValueChannels = m.sv3Mul.outC;
    m.numOwnershipChannels = m.vOwnershipConv.outC;
    if(m.v1Conv.outC != m.v1BN.c || m.v2Mul.inC != 3 * m.v1BN.c || m.v2Mul.outC != m.v2Bias.c || m.v2Mul.outC != m.v3Mul.inC ||
       m.v3Mul.outC != 3 || m.v3Bias.c != 3 || m.sv3Mul.inC != m.v2Mul.outC || m.sv3Mul.outC != m.sv3Bias.c ||
       m.sv3Mul.outC != (m.version >= 9 ? 6 : 4) || m.vOwnershipConv.inC != m.v1Conv.outC || m.vOwnershipConv.outC != 1)
      bad(vname + ": value head channel counts are inconsistent");
    if(m.v1Conv.ky != 1 || m.v1Conv.kx != 1 || m.vOwnershipConv.ky != 1 || m.vOwnershipConv.kx != 1)
      throw ModelError(KMX_ERR_UNSUPPORTED, vname + ": value head convolutions must be 1x1");

    if(m.numInputChannels != m.initialConv.inC) bad(m.name + ": numInputChannels != trunk.initialConv.inChannels");
    if(m.numInputGlobalChannels != m.initialMatMul.inC) bad(m.name + ": numInputGlobalChannels != trunk.initialMatMul.inChannels");
    if(m.numInputChannels != 22 || m.numInputGlobalChannels != 19) bad(m.name + ": expected 22 spatial and 19 global input features (inputs v7)");
    if(m.trunkC != m.p1Conv.inC || m.trunkC != m.g1Conv.inC || m.trunkC != m.v1Conv.inC) bad(m.name + ": head input channels != trunk channels");
  }
  catch(const ModelError& e) {
    throw ModelError(e.code, "Error loading or parsing model file " + path + ": " + e.what());
  }

  double mac = 0;
  int64_t params = 0;
  auto conv = [&](const ConvDesc& c) {
    mac += (double)c.ky * c.kx * c.inC * c.outC;
    params += (int64_t)c.ky * c.kx * c.inC * c.outC;
  };
  conv(m.initialConv);
  params += (int64_t)m.initialMatMul.inC * m.initialMatMul.outC;
  for(const BlockDesc& b : m.blocks) countBlock(b, mac, params);
  conv(m.p1Conv); conv(m.g1Conv); conv(m.p2Conv); conv(m.v1Conv); conv(m.vOwnershipConv);
  m.macPerPosition = mac;
  m.numParameters = params;
  return mp;
}

// ---- the fp16 range transform (SURVEY 8 row a23 ii; reference: ModelDesc::applyScale8ToReduceActivations, desc.cpp:2718-2736,
// and the per-layer pieces :355-366 batch norm, :421-445 activation, :552-556 bias, :1974-1986 trunk, :2217-2224 policy head,
// :2389-2396 value head). Every tensor of the net carries 1/8 of its value: the stem's outputs are scaled (initial convolution,
// global-feature matmul, metadata encoder's last matmul), every additive constant follows (merged BN biases, head biases),
// convolutions and matmuls are linear and need nothing, and an activation f becomes g(x) = f(8x)/8 - identity and relu are
// their own g, mish becomes x * tanh(softplus(8x)) (KMX_ACT_MISH_SCALE8). Global pooling is linear in the values (its off-board
// filler of -1 for the max stays below every activated value). The reference undoes the factor in NNEvaluator's
// post-processing (outputScaleMultiplier = 8, nneval.cpp:962,1123-1131,1245); here the LAST linear layer of each output -
// p2Conv, the pass head's final matmul, v3, sv3, the ownership convolution, all of which the engine evaluates in fp32 - carries
// the factor 8 (exact in binary floating point), so the C ABI keeps returning plain logits whatever the precision.
static bool actScales(int act) { return act == KMX_ACT_IDENTITY || act == KMX_ACT_RELU || act == KMX_ACT_MISH; }
static int actScaled8(int act) { return act == KMX_ACT_MISH ? KMX_ACT_MISH_SCALE8 : act; }
static bool blocksScale(const std::vector<BlockDesc>& blocks) {
  for(const BlockDesc& b : blocks) {
    if(b.isTransformer()) return false;
    if(!actScales(b.preBN.act) || !actScales(b.midBN.act)) return false;
    if(b.kind == BlockKind::GPool && !actScales(b.gpoolBN.act)) return false;
    if(b.kind == BlockKind::Nested && !blocksScale(b.inner)) return false;
  }
  return true;
}
bool ModelDesc::scale8Applies() const {
  if(trunkNormKind != 0 || hasTransformerBlocks) return false;  // an RMSNorm would undo the factor (desc.cpp:2721-2729)
  if(!blocksScale(blocks)) return false;
  return actScales(trunkTipBN.act) && actScales(g1BN.act) && actScales(p1BN.act) && actScales(v1BN.act) && actScales(passAct) && actScales(v2Act);
}
static void scaleBn(BnDesc& bn) {
  for(float& b : bn.bias) b *= 0.125f;
  bn.act = actScaled8(bn.act);
}
static void scaleBlocks(std::vector<BlockDesc>& blocks) {
  for(BlockDesc& b : blocks) {
    scaleBn(b.preBN);
    scaleBn(b.midBN);
    if(b.kind == BlockKind::GPool) scaleBn(b.gpoolBN);
    if(b.kind == BlockKind::Nested) scaleBlocks(b.inner);
  }
}
std::unique_ptr<ModelDesc> ModelDesc::scaledBy8() const {
  if(!scale8Applies()) throw ModelError(KMX_ERR_UNSUPPORTED, name + ": the 1/8 activation scaling does not apply to this architecture");
  std::unique_ptr<ModelDesc> mp(new ModelDesc(*this));
  ModelDesc& m = *mp;
  for(float& w : m.initialConv.w) w *= 0.125f;
  for(float& w : m.initialMatMul.w) w *= 0.125f;
  if(m.metaEncoderVersion > 0)
    for(float& w : m.metaMul3.w) w *= 0.125f;
  scaleBlocks(m.blocks);
  scaleBn(m.trunkTipBN);
  scaleBn(m.g1BN);
  scaleBn(m.p1BN);
  scaleBn(m.v1BN);
  for(float& w : m.gpoolToPassBias.w) w *= 0.125f;
  m.passAct = actScaled8(m.passAct);
  for(float& w : m.v2Bias.w) w *= 0.125f;
  m.v2Act = actScaled8(m.v2Act);
  // outputs back to their own scale: v3 / sv3 biases stay as they are in the file (0.125 b * 8)
  for(float& w : m.p2Conv.w) w *= 8.0f;
  for(float& w : (m.hasPassMLP ? m.gpoolToPassMul2 : m.gpoolToPassMul).w) w *= 8.0f;
  for(float& w : m.v3Mul.w) w *= 8.0f;
  for(float& w : m.sv3Mul.w) w *= 8.0f;
  for(float& w : m.vOwnershipConv.w) w *= 8.0f;
  m.scale8Applied = true;
  return mp;
}

}  // namespace kmx

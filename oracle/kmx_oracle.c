/*
 * kmx_oracle.c — CPU ORACLE (fp32, plain C + OpenMP).  TEST INFRASTRUCTURE ONLY — see kmx_oracle.h.
 *
 * Each function cites the reference file:line whose arithmetic it restates. The convolution is a
 * direct cross-correlation (the reference's Winograd, eigenbackend.cpp:293-703, computes the same
 * sums with a different rounding order, ~1e-6 relative).
 */
#define _GNU_SOURCE
#include "kmx_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
static __thread char g_err[1024];
const char* okmx_last_error(void) { return g_err; }
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

/* ---------------------------------- model description -------------------------------------- */
typedef struct {
  char name[160];
  int ky, kx, ic, oc;
  float* w; /* kept in model-FILE order [ky][kx][ic][oc] (desc.cpp:130) */
} OConv;
typedef struct {
  char name[160];
  int c;
  int act;      /* activation that follows this BN */
  float* scale; /* merged: scale/sqrt(var+eps)          (desc.cpp:272-279) */
  float* bias;  /* merged: bias - mergedScale*mean */
} OBn;
typedef struct {
  char name[160];
  int ic, oc;
  float* w; /* [ic][oc] (desc.cpp:461-476) */
} OMatMul;
typedef struct {
  char name[160];
  int c;
  float* w;
} OMatBias;

enum { BLK_ORDINARY = 0, BLK_GPOOL = 2, BLK_NESTED = 3, BLK_ATTENTION = 4, BLK_FFN = 5 }; /* desc.h:374-379 numbering */

typedef struct { /* TransformerRMSNormDesc (desc.h:259-276): weight only */
  int c;
  float eps;
  float* w;
} OTRms;

typedef struct OBlock {
  int kind;
  char name[160];
  /* ordinary: preBN regularConv midBN finalConv
   * gpool:    preBN regularConv gpoolConv gpoolBN gpoolToBiasMul midBN finalConv
   * nested:   preBN(preConv=regularConv) blocks postBN(=midBN) postConv(=finalConv) */
  OBn preBN, midBN, gpoolBN;
  OConv regularConv, finalConv, gpoolConv;
  OMatMul gpoolToBiasMul;
  int numBlocks;
  struct OBlock* blocks;
  /* transformer_attention_block (TransformerAttentionDesc, desc.h:278-321): preLN qProj kProj vProj outProj [rope]
   * transformer_ffn_block (TransformerFFNDesc, desc.h:323-346):             preLN linear1 [linearGate] linear2 */
  OTRms preLN;
  int numHeads, numKVHeads, qHeadDim, vHeadDim, useRope, learnableRope;
  float ropeTheta;
  float* ropeFreqs; /* learnable: [numKVHeads][qHeadDim/2][2] */
  OMatMul qProj, kProj, vProj, outProj;
  int ffnChannels, useSwiGLU;
  OMatMul linear1, linearGate, linear2;
} OBlock;

struct okmx_model {
  kmx_model_info info;
  int version;
  int numBlocks, C;
  OConv initialConv;
  OMatMul initialMatMul;
  /* SGFMetadataEncoderDesc (desc.cpp:1571-1625); metaEncoderVersion 0 = none */
  int metaEncoderVersion;
  OMatMul metaMul1, metaMul2, metaMul3;
  OMatBias metaBias1, metaBias2;
  int metaAct1, metaAct2;
  OBlock* blocks;
  OBn trunkTipBN;
  /* trunkNormKind != 0: RMSNormLayerDesc tip (desc.h:238-257) + activation instead of the batch norm */
  int trunkNormKind, trunkTipAct, rmsSpatial;
  float rmsEps;
  float* rmsGamma;
  float* rmsBeta;
  /* policy head (desc.cpp:2084-2104) */
  OConv p1Conv, g1Conv, p2Conv;
  OBn g1BN, p1BN;
  OMatMul gpoolToBiasMul, gpoolToPassMul, gpoolToPassMul2;
  OMatBias gpoolToPassBias;
  int passAct;
  /* value head (desc.cpp:2261-2273) */
  OConv v1Conv, vOwnershipConv;
  OBn v1BN;
  OMatMul v2Mul, v3Mul, sv3Mul;
  OMatBias v2Bias, v3Bias, sv3Bias;
  int v2Act;
  int64_t numParams;
  double macPerPos;
};

/* ---------------------------------- parser ------------------------------------------------- */
typedef struct {
  const unsigned char* buf;
  size_t len, pos;
  int binary;
  int err;
  char errmsg[512];
} Rd;

static void rd_fail(Rd* r, const char* fmt, ...) {
  if(r->err) return;
  r->err = 1;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(r->errmsg, sizeof(r->errmsg), fmt, ap);
  va_end(ap);
}
static int rd_token(Rd* r, char* out, size_t outlen) {
  if(r->err) { out[0] = 0; return 0; }
  while(r->pos < r->len && isspace(r->buf[r->pos])) r->pos++;
  if(r->pos >= r->len) { rd_fail(r, "unexpected end of model file"); out[0] = 0; return 0; }
  size_t n = 0;
  while(r->pos < r->len && !isspace(r->buf[r->pos])) {
    if(n + 1 < outlen) out[n++] = (char)r->buf[r->pos];
    r->pos++;
  }
  out[n] = 0;
  return 1;
}
static int rd_int(Rd* r, const char* what) {
  char t[256];
  if(!rd_token(r, t, sizeof(t))) return 0;
  char* e;
  long v = strtol(t, &e, 10);
  if(e == t || *e != 0) rd_fail(r, "%s: expected integer, got '%s'", what, t);
  return (int)v;
}
static float rd_float(Rd* r, const char* what) {
  char t[256];
  if(!rd_token(r, t, sizeof(t))) return 0;
  char* e;
  float v = strtof(t, &e);
  if(e == t) rd_fail(r, "%s: expected float, got '%s'", what, t);
  return v;
}
/* readFloats (desc.cpp:40-90): text tokens, or "@BIN@" + little-endian fp32 block */
static float* rd_floats(Rd* r, size_t n, const char* name) {
  float* out = (float*)calloc(n ? n : 1, sizeof(float)); /* zeros, not garbage, if parsing fails half-way */
  if(r->err) return out;
  if(!r->binary) {
    for(size_t i = 0; i < n; i++) out[i] = rd_float(r, name);
  } else {
    int before = 0;
    while(r->pos < r->len && r->buf[r->pos] != '@') {
      r->pos++;
      if(++before > 100) { rd_fail(r, "%s: could not find binary float block", name); return out; }
    }
    if(r->pos + 5 > r->len || memcmp(r->buf + r->pos, "@BIN@", 5) != 0) {
      rd_fail(r, "%s: did not find expected header for binary float block", name);
      return out;
    }
    r->pos += 5;
    if(r->pos + 4 * n > r->len) { rd_fail(r, "%s: truncated binary float block", name); return out; }
    memcpy(out, r->buf + r->pos, 4 * n);
    r->pos += 4 * n;
  }
  for(size_t i = 0; i < n; i++)
    if(!isfinite(out[i])) { rd_fail(r, "%s: Nan or infinite neural net weight or parameter", name); break; }
  return out;
}

static void parse_conv(Rd* r, OConv* c) { /* ConvLayerDesc, desc.cpp:110-155 */
  rd_token(r, c->name, sizeof(c->name));
  c->ky = rd_int(r, "convYSize");
  c->kx = rd_int(r, "convXSize");
  c->ic = rd_int(r, "inChannels");
  c->oc = rd_int(r, "outChannels");
  int dy = rd_int(r, "dilationY"), dx = rd_int(r, "dilationX");
  if(r->err) return;
  if(c->ky <= 0 || c->kx <= 0 || c->ic <= 0 || c->oc <= 0 || (c->ky % 2) != 1 || (c->kx % 2) != 1) {
    rd_fail(r, "%s: bad conv sizes", c->name);
    return;
  }
  if(dy != 1 || dx != 1) { rd_fail(r, "%s: dilated convolutions are not supported", c->name); return; }
  c->w = rd_floats(r, (size_t)c->ky * c->kx * c->ic * c->oc, c->name);
}
static void parse_bn(Rd* r, OBn* b) { /* BatchNormLayerDesc, desc.cpp:208-289 */
  rd_token(r, b->name, sizeof(b->name));
  b->c = rd_int(r, "numChannels");
  float eps = rd_float(r, "epsilon");
  int hasScale = rd_int(r, "hasScale"), hasBias = rd_int(r, "hasBias");
  if(r->err) return;
  if(b->c < 1 || !(eps > 0)) { rd_fail(r, "%s: bad batchnorm header", b->name); return; }
  float* mean = rd_floats(r, b->c, b->name);
  float* var = rd_floats(r, b->c, b->name);
  float* scale = hasScale ? rd_floats(r, b->c, b->name) : NULL;
  float* bias = hasBias ? rd_floats(r, b->c, b->name) : NULL;
  b->scale = (float*)malloc(sizeof(float) * b->c);
  b->bias = (float*)malloc(sizeof(float) * b->c);
  if(!r->err)
    for(int i = 0; i < b->c; i++) {
      float s = scale ? scale[i] : 1.0f, bb = bias ? bias[i] : 0.0f;
      b->scale[i] = s / sqrtf(var[i] + eps);
      b->bias[i] = bb - b->scale[i] * mean[i];
    }
  free(mean); free(var); free(scale); free(bias);
  b->act = KMX_ACT_IDENTITY;
}
static int parse_act(Rd* r, int version) { /* ActivationLayerDesc, desc.cpp:382-403 */
  char t[160];
  rd_token(r, t, sizeof(t)); /* name */
  if(version >= 11) {
    rd_token(r, t, sizeof(t));
    if(!strcmp(t, "ACTIVATION_IDENTITY")) return KMX_ACT_IDENTITY;
    if(!strcmp(t, "ACTIVATION_RELU")) return KMX_ACT_RELU;
    if(!strcmp(t, "ACTIVATION_MISH")) return KMX_ACT_MISH;
    if(!strcmp(t, "ACTIVATION_SILU")) return KMX_ACT_SILU;
    rd_fail(r, "unknown activation %s", t);
  }
  return KMX_ACT_RELU;
}
static void parse_matmul(Rd* r, OMatMul* m) { /* MatMulLayerDesc, desc.cpp:451-479 */
  rd_token(r, m->name, sizeof(m->name));
  m->ic = rd_int(r, "inChannels");
  m->oc = rd_int(r, "outChannels");
  if(r->err) return;
  if(m->ic <= 0 || m->oc <= 0) { rd_fail(r, "%s: bad matmul sizes", m->name); return; }
  m->w = rd_floats(r, (size_t)m->ic * m->oc, m->name);
}
static void parse_matbias(Rd* r, OMatBias* m) { /* MatBiasLayerDesc, desc.cpp:518-535 */
  rd_token(r, m->name, sizeof(m->name));
  m->c = rd_int(r, "numChannels");
  if(r->err) return;
  if(m->c <= 0) { rd_fail(r, "%s: bad matbias size", m->name); return; }
  m->w = rd_floats(r, m->c, m->name);
}

static void parse_block_stack(Rd* r, int version, int numBlocks, int trunkC, OBlock* blocks);

static void parse_trms(Rd* r, OTRms* t) { /* TransformerRMSNormDesc, desc.cpp:1125-1144 */
  char name[160];
  rd_token(r, name, sizeof(name));
  t->c = rd_int(r, "numChannels");
  t->eps = rd_float(r, "epsilon");
  if(r->err) return;
  if(t->c < 1 || !(t->eps > 0) || t->eps > 1.0f) { rd_fail(r, "%s: bad transformer rmsnorm header", name); return; }
  t->w = rd_floats(r, t->c, name);
}

static void parse_block(Rd* r, int version, OBlock* b, int trunkC) { /* desc.cpp:1444-1562 */
  char kind[160];
  rd_token(r, kind, sizeof(kind));
  if(r->err) return;
  memset(b, 0, sizeof(*b));
  if(!strcmp(kind, "ordinary_block")) { /* ResidualBlockDesc, desc.cpp:566-593 */
    b->kind = BLK_ORDINARY;
    rd_token(r, b->name, sizeof(b->name));
    parse_bn(r, &b->preBN);
    b->preBN.act = parse_act(r, version);
    parse_conv(r, &b->regularConv);
    parse_bn(r, &b->midBN);
    b->midBN.act = parse_act(r, version);
    parse_conv(r, &b->finalConv);
    if(r->err) return;
    if(b->preBN.c != b->regularConv.ic || b->midBN.c != b->regularConv.oc || b->midBN.c != b->finalConv.ic ||
       b->preBN.c != trunkC || b->finalConv.oc != trunkC)
      rd_fail(r, "%s: residual block channel mismatch", b->name);
  } else if(!strcmp(kind, "gpool_block")) { /* GlobalPoolingResidualBlockDesc, desc.cpp:652-702 */
    b->kind = BLK_GPOOL;
    rd_token(r, b->name, sizeof(b->name));
    parse_bn(r, &b->preBN);
    b->preBN.act = parse_act(r, version);
    parse_conv(r, &b->regularConv);
    parse_conv(r, &b->gpoolConv);
    parse_bn(r, &b->gpoolBN);
    b->gpoolBN.act = parse_act(r, version);
    parse_matmul(r, &b->gpoolToBiasMul);
    parse_bn(r, &b->midBN);
    b->midBN.act = parse_act(r, version);
    parse_conv(r, &b->finalConv);
    if(r->err) return;
    if(b->preBN.c != b->regularConv.ic || b->preBN.c != b->gpoolConv.ic || b->gpoolBN.c != b->gpoolConv.oc ||
       b->gpoolBN.c * 3 != b->gpoolToBiasMul.ic || b->midBN.c != b->regularConv.oc ||
       b->midBN.c != b->gpoolToBiasMul.oc || b->midBN.c != b->finalConv.ic || b->preBN.c != trunkC ||
       b->finalConv.oc != trunkC)
      rd_fail(r, "%s: gpool block channel mismatch", b->name);
  } else if(!strcmp(kind, "nested_bottleneck_block")) { /* NestedBottleneckResidualBlockDesc, desc.cpp:783-818 */
    b->kind = BLK_NESTED;
    rd_token(r, b->name, sizeof(b->name));
    b->numBlocks = rd_int(r, "numBlocks");
    if(r->err) return;
    if(b->numBlocks < 1) { rd_fail(r, "%s: nested block numBlocks must be positive", b->name); return; }
    parse_bn(r, &b->preBN);
    b->preBN.act = parse_act(r, version);
    parse_conv(r, &b->regularConv); /* preConv */
    if(r->err) return;
    b->blocks = (OBlock*)calloc(b->numBlocks, sizeof(OBlock));
    parse_block_stack(r, version, b->numBlocks, b->regularConv.oc, b->blocks);
    parse_bn(r, &b->midBN); /* postBN */
    b->midBN.act = parse_act(r, version);
    parse_conv(r, &b->finalConv); /* postConv */
    if(r->err) return;
    if(b->preBN.c != b->regularConv.ic || b->midBN.c != b->regularConv.oc || b->midBN.c != b->finalConv.ic ||
       b->preBN.c != trunkC || b->finalConv.oc != trunkC)
      rd_fail(r, "%s: nested block channel mismatch", b->name);
  } else if(!strcmp(kind, "transformer_attention_block")) { /* TransformerAttentionDesc, desc.cpp:1173-1258 */
    b->kind = BLK_ATTENTION;
    rd_token(r, b->name, sizeof(b->name));
    b->numHeads = rd_int(r, "numHeads");
    b->numKVHeads = rd_int(r, "numKVHeads");
    b->qHeadDim = rd_int(r, "qHeadDim");
    b->vHeadDim = rd_int(r, "vHeadDim");
    b->useRope = rd_int(r, "useRope") != 0;
    b->learnableRope = rd_int(r, "learnableRope") != 0;
    if(r->err) return;
    if(b->numHeads < 1 || b->numKVHeads < 1 || b->numHeads % b->numKVHeads != 0 || b->qHeadDim < 1 || b->vHeadDim < 1 ||
       (b->useRope && b->qHeadDim % 2 != 0)) { rd_fail(r, "%s: bad attention header", b->name); return; }
    parse_trms(r, &b->preLN);
    parse_matmul(r, &b->qProj);
    parse_matmul(r, &b->kProj);
    parse_matmul(r, &b->vProj);
    parse_matmul(r, &b->outProj);
    if(r->err) return;
    if(b->qProj.oc != b->numHeads * b->qHeadDim || b->kProj.oc != b->numKVHeads * b->qHeadDim ||
       b->vProj.oc != b->numKVHeads * b->vHeadDim || b->outProj.ic != b->numHeads * b->vHeadDim || b->qProj.ic != trunkC ||
       b->outProj.oc != trunkC || b->preLN.c != trunkC) { rd_fail(r, "%s: attention channel mismatch", b->name); return; }
    if(b->useRope) {
      char t[160];
      rd_token(r, t, sizeof(t)); /* rope freqs / theta name */
      if(b->learnableRope) {
        int kvh = rd_int(r, "ropeNumKVHeads"), np = rd_int(r, "ropeNumPairs"), d2 = rd_int(r, "ropeDim2");
        if(r->err) return;
        if(kvh != b->numKVHeads || np != b->qHeadDim / 2 || d2 != 2) { rd_fail(r, "%s: bad learnable rope header", b->name); return; }
        b->ropeFreqs = rd_floats(r, (size_t)kvh * np * 2, b->name);
      } else {
        b->ropeTheta = rd_float(r, "ropeTheta");
        if(!r->err && !(b->ropeTheta > 0.0f)) rd_fail(r, "%s: rope theta must be positive", b->name);
      }
    }
  } else if(!strcmp(kind, "transformer_ffn_block")) { /* TransformerFFNDesc, desc.cpp:1371-1405 */
    b->kind = BLK_FFN;
    rd_token(r, b->name, sizeof(b->name));
    int nc = rd_int(r, "numChannels");
    b->ffnChannels = rd_int(r, "ffnChannels");
    b->useSwiGLU = rd_int(r, "useSwiGLU") != 0;
    if(r->err) return;
    parse_trms(r, &b->preLN);
    parse_matmul(r, &b->linear1);
    if(b->useSwiGLU) parse_matmul(r, &b->linearGate);
    parse_matmul(r, &b->linear2);
    if(r->err) return;
    if(nc != trunkC || b->preLN.c != trunkC || b->linear1.ic != nc || b->linear1.oc != b->ffnChannels ||
       (b->useSwiGLU && (b->linearGate.ic != nc || b->linearGate.oc != b->ffnChannels)) || b->linear2.ic != b->ffnChannels ||
       b->linear2.oc != nc) { rd_fail(r, "%s: ffn channel mismatch", b->name); return; }
    if(!b->useSwiGLU) rd_fail(r, "%s: non-SwiGLU transformer FFN is not supported (as in the reference's Eigen backend, eigenbackend.cpp:1631-1633)", b->name);
  } else {
    rd_fail(r, "found unknown block kind: %s", kind);
  }
}
static void parse_block_stack(Rd* r, int version, int numBlocks, int trunkC, OBlock* blocks) {
  for(int i = 0; i < numBlocks && !r->err; i++) parse_block(r, version, &blocks[i], trunkC);
}

static void free_conv(OConv* c) { free(c->w); }
static void free_bn(OBn* b) { free(b->scale); free(b->bias); }
static void free_block(OBlock* b) {
  free_bn(&b->preBN); free_bn(&b->midBN); free_bn(&b->gpoolBN);
  free_conv(&b->regularConv); free_conv(&b->finalConv); free_conv(&b->gpoolConv);
  free(b->gpoolToBiasMul.w);
  for(int i = 0; i < b->numBlocks; i++) free_block(&b->blocks[i]);
  free(b->blocks);
  free(b->preLN.w); free(b->ropeFreqs);
  free(b->qProj.w); free(b->kProj.w); free(b->vProj.w); free(b->outProj.w);
  free(b->linear1.w); free(b->linearGate.w); free(b->linear2.w);
}
void okmx_model_free(okmx_model* m) {
  if(!m) return;
  free_conv(&m->initialConv);
  free(m->initialMatMul.w);
  free(m->metaMul1.w); free(m->metaMul2.w); free(m->metaMul3.w); free(m->metaBias1.w); free(m->metaBias2.w);
  for(int i = 0; i < m->numBlocks && m->blocks; i++) free_block(&m->blocks[i]);
  free(m->blocks);
  free_bn(&m->trunkTipBN);
  free(m->rmsGamma); free(m->rmsBeta);
  free_conv(&m->p1Conv); free_conv(&m->g1Conv); free_conv(&m->p2Conv);
  free_bn(&m->g1BN); free_bn(&m->p1BN);
  free(m->gpoolToBiasMul.w); free(m->gpoolToPassMul.w); free(m->gpoolToPassMul2.w); free(m->gpoolToPassBias.w);
  free_conv(&m->v1Conv); free_conv(&m->vOwnershipConv);
  free_bn(&m->v1BN);
  free(m->v2Mul.w); free(m->v3Mul.w); free(m->sv3Mul.w);
  free(m->v2Bias.w); free(m->v3Bias.w); free(m->sv3Bias.w);
  free(m);
}

static double conv_mac(const OConv* c) { return (double)c->ky * c->kx * c->ic * c->oc; }
static int64_t conv_params(const OConv* c) { return (int64_t)c->ky * c->kx * c->ic * c->oc; }
static void block_counts(const OBlock* b, double* mac, int64_t* params) {
  if(b->kind == BLK_ATTENTION || b->kind == BLK_FFN) {
    /* projections only: the QK^T / PV products cost area*numHeads*(qHeadDim+vHeadDim) MACs per point on top, which
       depends on the board and is left out of this per-point figure */
    const OMatMul* mm[7] = {&b->qProj, &b->kProj, &b->vProj, &b->outProj, &b->linear1, &b->linearGate, &b->linear2};
    for(int i = 0; i < 7; i++)
      if(mm[i]->w) { *mac += (double)mm[i]->ic * mm[i]->oc; *params += (int64_t)mm[i]->ic * mm[i]->oc; }
    *params += b->preLN.c;
    if(b->kind == BLK_ATTENTION && b->learnableRope) *params += (int64_t)b->numKVHeads * (b->qHeadDim / 2) * 2;
    return;
  }
  *mac += conv_mac(&b->regularConv) + conv_mac(&b->finalConv);
  *params += conv_params(&b->regularConv) + conv_params(&b->finalConv) + 2 * b->preBN.c + 2 * b->midBN.c;
  if(b->kind == BLK_GPOOL) {
    *mac += conv_mac(&b->gpoolConv);
    *params += conv_params(&b->gpoolConv) + 2 * b->gpoolBN.c + (int64_t)b->gpoolToBiasMul.ic * b->gpoolToBiasMul.oc;
  }
  for(int i = 0; i < b->numBlocks; i++) block_counts(&b->blocks[i], mac, params);
}

static int load_file(const char* path, unsigned char** out, size_t* outlen) {
  gzFile f = gzopen(path, "rb"); /* transparently reads plain files too */
  if(!f) return 0;
  size_t cap = 1 << 24, len = 0;
  unsigned char* buf = (unsigned char*)malloc(cap);
  for(;;) {
    if(len == cap) { cap *= 2; buf = (unsigned char*)realloc(buf, cap); }
    int n = gzread(f, buf + len, (unsigned)((cap - len) > (1u << 30) ? (1u << 30) : (cap - len)));
    if(n < 0) { free(buf); gzclose(f); return 0; }
    if(n == 0) break;
    len += (size_t)n;
  }
  gzclose(f);
  *out = buf;
  *outlen = len;
  return 1;
}
static int ends_with(const char* s, const char* suf) {
  size_t a = strlen(s), b = strlen(suf);
  if(a < b) return 0;
  for(size_t i = 0; i < b; i++)
    if(tolower((unsigned char)s[a - b + i]) != suf[i]) return 0;
  return 1;
}

/* ModelDesc::ModelDesc + TrunkDesc + PolicyHeadDesc + ValueHeadDesc
 * (desc.cpp:2441-2615, 1669-1768, 2051-2155, 2242-2340) */
static int parse_model(Rd* r, okmx_model* m) {
  kmx_model_info* info = &m->info;
  char tok[256];
  rd_token(r, tok, sizeof(tok));
  snprintf(info->name, sizeof(info->name), "%.*s", (int)sizeof(info->name) - 1, tok);
  m->version = info->model_version = rd_int(r, "modelVersion");
  if(r->err) return 0;
  if(m->version < 8 || m->version > 17) {
    rd_fail(r, "model version %d is not supported (need 8..17, inputs v7)", m->version);
    return 0;
  }
  info->num_input_channels = rd_int(r, "numInputChannels");
  info->num_input_global_channels = rd_int(r, "numInputGlobalChannels");
  info->td_score_multiplier = 20.0f; /* defaults: desc.h ModelPostProcessParams */
  info->score_mean_multiplier = 20.0f;
  info->score_stdev_multiplier = 20.0f;
  info->lead_multiplier = 20.0f;
  info->variance_time_multiplier = 40.0f;
  info->shortterm_value_error_multiplier = 0.25f;
  info->shortterm_score_error_multiplier = 30.0f;
  info->output_scale_multiplier = 1.0f;
  if(m->version >= 13) {
    info->td_score_multiplier = rd_float(r, "tdScoreMultiplier");
    info->score_mean_multiplier = rd_float(r, "scoreMeanMultiplier");
    info->score_stdev_multiplier = rd_float(r, "scoreStdevMultiplier");
    info->lead_multiplier = rd_float(r, "leadMultiplier");
    info->variance_time_multiplier = rd_float(r, "varianceTimeMultiplier");
    info->shortterm_value_error_multiplier = rd_float(r, "shorttermValueErrorMultiplier");
    info->shortterm_score_error_multiplier = rd_float(r, "shorttermScoreErrorMultiplier");
  }
  int metaEncoderVersion = 0;
  if(m->version >= 15) {
    metaEncoderVersion = rd_int(r, "metaEncoderVersion");
    const int preferPassAlive = rd_int(r, "preferPassAliveUnderSuicideRules"); /* desc.cpp:2538-2548: a flag, 0 or 1 */
    if(!r->err && preferPassAlive != 0 && preferPassAlive != 1) rd_fail(r, "model preferPassAliveUnderSuicideRules unexpected value");
    for(int i = 0; i < 6; i++)
      if(rd_int(r, "unused model option") != 0) rd_fail(r, "unknown/unsupported model option");
  }
  if(r->err) return 0;
  if(metaEncoderVersion < 0 || metaEncoderVersion > 1) { rd_fail(r, "unsupported metaEncoderVersion"); return 0; } /* modelversion.cpp:83-89 */
  m->metaEncoderVersion = info->meta_encoder_version = metaEncoderVersion;

  /* trunk */
  rd_token(r, tok, sizeof(tok));
  m->numBlocks = info->num_blocks = rd_int(r, "numBlocks");
  m->C = info->trunk_num_channels = rd_int(r, "trunkNumChannels");
  info->mid_num_channels = rd_int(r, "midNumChannels");
  (void)rd_int(r, "regularNumChannels");
  (void)rd_int(r, "dilatedNumChannels");
  (void)rd_int(r, "gpoolNumChannels");
  if(m->version >= 15) {
    int trunkNormKind = rd_int(r, "trunkNormKind");
    for(int i = 0; i < 5; i++)
      if(rd_int(r, "unused trunk option") != 0) rd_fail(r, "unknown/unsupported trunk option");
    if(!r->err && (trunkNormKind < 0 || trunkNormKind > 3)) rd_fail(r, "unknown trunkNormKind");
    m->trunkNormKind = trunkNormKind;
  }
  if(r->err) return 0;
  if(m->numBlocks < 1 || m->C < 1) { rd_fail(r, "bad trunk header"); return 0; }
  parse_conv(r, &m->initialConv);
  parse_matmul(r, &m->initialMatMul);
  if(r->err) return 0;
  if(m->metaEncoderVersion > 0) { /* SGFMetadataEncoderDesc, desc.cpp:1571-1625 */
    rd_token(r, tok, sizeof(tok));
    info->num_input_meta_channels = rd_int(r, "numInputMetaChannels");
    parse_matmul(r, &m->metaMul1);
    parse_matbias(r, &m->metaBias1);
    m->metaAct1 = parse_act(r, m->version);
    parse_matmul(r, &m->metaMul2);
    parse_matbias(r, &m->metaBias2);
    m->metaAct2 = parse_act(r, m->version);
    parse_matmul(r, &m->metaMul3);
    if(r->err) return 0;
    if(info->num_input_meta_channels != 192 || m->metaMul1.ic != 192 || m->metaMul1.oc != m->metaBias1.c ||
       m->metaMul2.ic != m->metaMul1.oc || m->metaMul2.oc != m->metaBias2.c || m->metaMul3.ic != m->metaMul2.oc ||
       m->metaMul3.oc != m->C) { rd_fail(r, "sgf metadata encoder channel counts are inconsistent"); return 0; }
  }
  m->blocks = (OBlock*)calloc(m->numBlocks, sizeof(OBlock));
  parse_block_stack(r, m->version, m->numBlocks, m->C, m->blocks);
  if(m->trunkNormKind == 0) {
    parse_bn(r, &m->trunkTipBN);
    m->trunkTipBN.act = parse_act(r, m->version);
  } else { /* RMSNormLayerDesc, desc.cpp:1069-1095 */
    rd_token(r, tok, sizeof(tok));
    int c = rd_int(r, "numChannels");
    m->rmsEps = rd_float(r, "epsilon");
    m->rmsSpatial = rd_int(r, "spatial") != 0;
    int cgroup = rd_int(r, "cgroupSize");
    if(r->err) return 0;
    if(c != m->C || !(m->rmsEps > 0) || m->rmsEps > 1.0f || cgroup != 0) { rd_fail(r, "bad or unsupported trunk tip rmsnorm"); return 0; }
    m->rmsGamma = rd_floats(r, c, "trunk tip rmsnorm gamma");
    m->rmsBeta = rd_floats(r, c, "trunk tip rmsnorm beta");
    m->trunkTipAct = parse_act(r, m->version);
  }
  if(r->err) return 0;

  /* policy head */
  rd_token(r, tok, sizeof(tok));
  int policyOut;
  if(m->version >= 17) {
    policyOut = rd_int(r, "policyOutChannels");
    for(int i = 0; i < 3; i++)
      if(rd_int(r, "unused policy option") != 0) rd_fail(r, "unknown/unsupported policy option");
  } else if(m->version == 16) policyOut = 4;
  else if(m->version >= 12) policyOut = 2;
  else policyOut = 1;
  info->num_policy_channels = policyOut;
  parse_conv(r, &m->p1Conv);
  parse_conv(r, &m->g1Conv);
  parse_bn(r, &m->g1BN);
  m->g1BN.act = parse_act(r, m->version);
  parse_matmul(r, &m->gpoolToBiasMul);
  parse_bn(r, &m->p1BN);
  m->p1BN.act = parse_act(r, m->version);
  parse_conv(r, &m->p2Conv);
  parse_matmul(r, &m->gpoolToPassMul);
  if(m->version >= 15) {
    parse_matbias(r, &m->gpoolToPassBias);
    m->passAct = parse_act(r, m->version);
    parse_matmul(r, &m->gpoolToPassMul2);
  }
  if(r->err) return 0;
  if(m->p2Conv.oc != policyOut) { rd_fail(r, "p2Conv.outChannels != policyOutChannels"); return 0; }

  /* value head */
  rd_token(r, tok, sizeof(tok));
  if(m->version >= 17)
    for(int i = 0; i < 3; i++)
      if(rd_int(r, "unused value option") != 0) rd_fail(r, "unknown/unsupported value option");
  parse_conv(r, &m->v1Conv);
  parse_bn(r, &m->v1BN);
  m->v1BN.act = parse_act(r, m->version);
  parse_matmul(r, &m->v2Mul);
  parse_matbias(r, &m->v2Bias);
  m->v2Act = parse_act(r, m->version);
  parse_matmul(r, &m->v3Mul);
  parse_matbias(r, &m->v3Bias);
  parse_matmul(r, &m->sv3Mul);
  parse_matbias(r, &m->sv3Bias);
  parse_conv(r, &m->vOwnershipConv);
  if(r->err) return 0;
  info->num_value_channels = m->v3Mul.oc;
  info->num_score_value_channels = m->sv3Mul.oc;
  info->num_ownership_channels = m->vOwnershipConv.oc;
  if(info->num_value_channels != 3 || info->num_ownership_channels != 1 ||
     info->num_score_value_channels != (m->version >= 9 ? 6 : 4)) {
    rd_fail(r, "unexpected value head output sizes");
    return 0;
  }
  if(info->num_input_channels != m->initialConv.ic || info->num_input_global_channels != m->initialMatMul.ic ||
     m->C != m->p1Conv.ic || m->C != m->g1Conv.ic || m->C != m->v1Conv.ic || info->num_input_channels != 22 ||
     info->num_input_global_channels != 19) {
    rd_fail(r, "model input/trunk channel mismatch");
    return 0;
  }
  /* parameter and MAC counts (per board point; dense layers ignored) */
  double mac = conv_mac(&m->initialConv);
  int64_t params = conv_params(&m->initialConv) + (int64_t)m->initialMatMul.ic * m->initialMatMul.oc;
  for(int i = 0; i < m->numBlocks; i++) block_counts(&m->blocks[i], &mac, &params);
  mac += conv_mac(&m->p1Conv) + conv_mac(&m->g1Conv) + conv_mac(&m->p2Conv) + conv_mac(&m->v1Conv) +
         conv_mac(&m->vOwnershipConv);
  params += conv_params(&m->p1Conv) + conv_params(&m->g1Conv) + conv_params(&m->p2Conv) + conv_params(&m->v1Conv) +
            conv_params(&m->vOwnershipConv);
  m->macPerPos = mac;
  m->numParams = params;
  info->num_parameters = params;
  info->flops_per_position = 2.0 * mac;
  return 1;
}

int okmx_model_load(const char* path, const char* expected_sha256, okmx_model** out) {
  (void)expected_sha256; /* the oracle does not verify hashes */
  if(!path || !out) return fail(KMX_ERR_INVALID_ARG, "okmx_model_load: null argument");
  *out = NULL;
  int binary;
  if(ends_with(path, ".txt") || ends_with(path, ".txt.gz")) binary = 0;
  else if(ends_with(path, ".bin") || ends_with(path, ".bin.gz") || ends_with(path, ".gz")) binary = 1;
  else return fail(KMX_ERR_MODEL, "Model file should end with .txt, .bin, .txt.gz, .bin.gz: %s", path);
  unsigned char* buf;
  size_t len;
  if(!load_file(path, &buf, &len)) return fail(KMX_ERR_IO, "could not read model file %s", path);
  okmx_model* m = (okmx_model*)calloc(1, sizeof(okmx_model));
  Rd r;
  memset(&r, 0, sizeof(r));
  r.buf = buf; r.len = len; r.binary = binary;
  int ok = parse_model(&r, m);
  free(buf);
  if(!ok || r.err) {
    okmx_model_free(m);
    return fail(KMX_ERR_MODEL, "Error loading or parsing model file %s: %s", path, r.errmsg);
  }
  *out = m;
  return KMX_OK;
}
int okmx_model_info_get(const okmx_model* m, kmx_model_info* out) {
  if(!m || !out) return fail(KMX_ERR_INVALID_ARG, "okmx_model_info_get: null argument");
  *out = m->info;
  return KMX_OK;
}

/* ---------------------------------- layer math --------------------------------------------- */
static inline float act_apply(float x, int act) {
  switch(act) {
    case KMX_ACT_IDENTITY: return x;
    case KMX_ACT_RELU: return x > 0.0f ? x : 0.0f;
    case KMX_ACT_MISH: /* x*tanh(softplus(x)), softplus linearised above 20 (eigenbackend.cpp:754) */
      return x * tanhf(log1pf(expf(x < 20.0f ? x : 20.0f)) + ((x > 20.0f ? x : 20.0f) - 20.0f));
    case KMX_ACT_SILU: return x / (expf(-x) + 1.0f);
    default: return NAN;
  }
}

/* ConvLayer::apply (eigenbackend.cpp:293-703): cross-correlation, zero padding, stride 1, no bias.
 * in [n][Y][X][ic], out [n][Y][X][oc]; accumulate adds into out (eigenbackend.cpp:659-686). */
static void conv_apply(const OConv* c, int n, int X, int Y, const float* in, float* out, int accumulate) {
  const int ic = c->ic, oc = c->oc, ky = c->ky, kx = c->kx, py = ky / 2, px = kx / 2;
#pragma omp parallel for collapse(2) schedule(static)
  for(int b = 0; b < n; b++) {
    for(int y = 0; y < Y; y++) {
      float* acc = (float*)malloc(sizeof(float) * oc);
      for(int x = 0; x < X; x++) {
        for(int o = 0; o < oc; o++) acc[o] = 0.0f;
        for(int dy = 0; dy < ky; dy++) {
          int yy = y + dy - py;
          if(yy < 0 || yy >= Y) continue;
          for(int dx = 0; dx < kx; dx++) {
            int xx = x + dx - px;
            if(xx < 0 || xx >= X) continue;
            const float* ip = in + (((size_t)b * Y + yy) * X + xx) * ic;
            const float* wp = c->w + ((size_t)dy * kx + dx) * ic * oc;
            for(int i = 0; i < ic; i++) {
              float a = ip[i];
              if(a == 0.0f) continue;
              const float* wr = wp + (size_t)i * oc;
#pragma omp simd
              for(int o = 0; o < oc; o++) acc[o] += a * wr[o];
            }
          }
        }
        float* op = out + (((size_t)b * Y + y) * X + x) * oc;
        if(accumulate)
          for(int o = 0; o < oc; o++) op[o] += acc[o];
        else
          for(int o = 0; o < oc; o++) op[o] = acc[o];
      }
      free(acc);
    }
  }
}

/* BatchNormLayer::apply (eigenbackend.cpp:739-762): mask==1 ? act(x*scale+bias) : 0 */
static void bnact_apply(const OBn* bn, int n, int S, const float* in, float* out, const float* mask) {
  const int C = bn->c;
#pragma omp parallel for schedule(static)
  for(int p = 0; p < n * S; p++) {
    const float* ip = in + (size_t)p * C;
    float* op = out + (size_t)p * C;
    if(mask[p] == 1.0f)
      for(int c = 0; c < C; c++) op[c] = act_apply(ip[c] * bn->scale[c] + bn->bias[c], bn->act);
    else
      for(int c = 0; c < C; c++) op[c] = 0.0f;
  }
}
/* MatMulLayer::apply (eigenbackend.cpp:836-839): out[n][oc] = sum_ic in[n][ic]*W[ic][oc] */
static void matmul_apply(const OMatMul* m, int n, const float* in, float* out) {
  for(int b = 0; b < n; b++)
    for(int o = 0; o < m->oc; o++) {
      float s = 0.0f;
      for(int i = 0; i < m->ic; i++) s += in[(size_t)b * m->ic + i] * m->w[(size_t)i * m->oc + o];
      out[(size_t)b * m->oc + o] = s;
    }
}
static void matbias_apply(const OMatBias* m, int n, float* x) { /* eigenbackend.cpp:856-862 */
  for(int b = 0; b < n; b++)
    for(int c = 0; c < m->c; c++) x[(size_t)b * m->c + c] += m->w[c];
}
/* addNCBiasInplace (eigenbackend.cpp:137-148): every cell, including off-board */
static void add_nc_bias(int n, int S, int C, float* x, const float* bias) {
#pragma omp parallel for schedule(static)
  for(int p = 0; p < n * S; p++) {
    const float* bp = bias + (size_t)(p / S) * C;
    float* xp = x + (size_t)p * C;
    for(int c = 0; c < C; c++) xp[c] += bp[c];
  }
}
/* poolRowsGPool (eigenbackend.cpp:152-177) -> out [n][3C]: mean, mean*(sqrt(ms)-14)*0.1, max(x+(mask-1)) */
static void pool_gpool(int n, int S, int C, const float* in, float* out, const float* mask, const float* maskSum) {
  for(int b = 0; b < n; b++)
    for(int c = 0; c < C; c++) {
      float s = 0.0f, m = -1.0f;
      for(int p = 0; p < S; p++) {
        float x = in[((size_t)b * S + p) * C + c];
        s += x;
        float v = x + (mask[(size_t)b * S + p] - 1.0f);
        if(v > m) m = v;
      }
      float div = maskSum[b], sqrtdiv = sqrtf(div), mean = s / div;
      out[(size_t)b * 3 * C + c] = mean;
      out[(size_t)b * 3 * C + C + c] = mean * (sqrtdiv - 14.0f) * 0.1f;
      out[(size_t)b * 3 * C + 2 * C + c] = m;
    }
}
/* poolRowsValueHead (eigenbackend.cpp:179-197) */
static void pool_value(int n, int S, int C, const float* in, float* out, const float* maskSum) {
  for(int b = 0; b < n; b++)
    for(int c = 0; c < C; c++) {
      float s = 0.0f;
      for(int p = 0; p < S; p++) s += in[((size_t)b * S + p) * C + c];
      float div = maskSum[b], sqrtdiv = sqrtf(div), mean = s / div;
      out[(size_t)b * 3 * C + c] = mean;
      out[(size_t)b * 3 * C + C + c] = mean * (sqrtdiv - 14.0f) * 0.1f;
      out[(size_t)b * 3 * C + 2 * C + c] = mean * ((sqrtdiv - 14.0f) * (sqrtdiv - 14.0f) * 0.01f - 0.1f);
    }
}

static float* falloc(size_t n) { return (float*)malloc(sizeof(float) * (n ? n : 1)); }


/* TransformerRMSNormLayer::apply (eigenbackend.cpp:885-915): per cell over channels, weight only, masked cells -> 0 */
static void trms_apply(const OTRms* t, int n, int S, const float* in, float* out, const float* mask) {
  const int C = t->c;
#pragma omp parallel for schedule(static)
  for(int p = 0; p < n * S; p++) {
    const float* ip = in + (size_t)p * C;
    float* op = out + (size_t)p * C;
    if(mask[p] == 0.0f) { for(int c = 0; c < C; c++) op[c] = 0.0f; continue; }
    float sumSq = 0.0f;
    for(int c = 0; c < C; c++) sumSq += ip[c] * ip[c];
    const float rms = 1.0f / sqrtf(sumSq / (float)C + t->eps);
    for(int c = 0; c < C; c++) op[c] = ip[c] * rms * t->w[c];
  }
}
/* matmul over all cells: out[p][oc] = sum_ic in[p][ic] W[ic][oc]  (MatMulLayer on the (C, seqLen*N) view) */
static void matmul_cells(const OMatMul* m, size_t cells, const float* in, float* out) {
#pragma omp parallel for schedule(static)
  for(long p = 0; p < (long)cells; p++) {
    const float* ip = in + (size_t)p * m->ic;
    float* op = out + (size_t)p * m->oc;
    for(int o = 0; o < m->oc; o++) op[o] = 0.0f;
    for(int i = 0; i < m->ic; i++) {
      const float v = ip[i];
      const float* wr = m->w + (size_t)i * m->oc;
      for(int o = 0; o < m->oc; o++) op[o] += v * wr[o];
    }
  }
}
/* TransformerAttentionDesc::computeRopeCosSin (desc.cpp:1300-1363) with paddedNNXYLen = X*Y */
static void rope_tables(const OBlock* b, int X, int Y, float* cosT, float* sinT) {
  const int S = X * Y, numPairs = b->qHeadDim / 2;
  if(b->learnableRope) {
    for(int h = 0; h < b->numKVHeads; h++)
      for(int p = 0; p < numPairs; p++) {
        const float fx = b->ropeFreqs[(h * numPairs + p) * 2 + 0], fy = b->ropeFreqs[(h * numPairs + p) * 2 + 1];
        for(int y = 0; y < Y; y++)
          for(int x = 0; x < X; x++) {
            const float angle = (float)x * fx + (float)y * fy;
            cosT[(h * numPairs + p) * S + y * X + x] = cosf(angle);
            sinT[(h * numPairs + p) * S + y * X + x] = sinf(angle);
          }
      }
  } else {
    const int perDim = numPairs / 2, dimHalf = b->qHeadDim / 2;
    for(int p = 0; p < numPairs; p++)
      for(int y = 0; y < Y; y++)
        for(int x = 0; x < X; x++) {
          float angle;
          if(p < perDim) angle = (float)y * (1.0f / powf(b->ropeTheta, (float)(2 * p) / (float)dimHalf));
          else angle = (float)x * (1.0f / powf(b->ropeTheta, (float)(2 * (p - perDim)) / (float)dimHalf));
          cosT[p * S + y * X + x] = cosf(angle);
          sinT[p * S + y * X + x] = sinf(angle);
        }
  }
}
/* applyRoPE (eigenbackend.cpp:1417-1456): rotate channel pairs (2p, 2p+1) of every head */
static void rope_apply(const OBlock* b, int n, int S, float* data, int numBufHeads, const float* cosT, const float* sinT) {
  const int numPairs = b->qHeadDim / 2, D = numBufHeads * b->qHeadDim;
#pragma omp parallel for schedule(static)
  for(int cell = 0; cell < n * S; cell++) {
    const int xy = cell % S;
    float* row = data + (size_t)cell * D;
    for(int h = 0; h < numBufHeads; h++)
      for(int p = 0; p < numPairs; p++) {
        const int t = b->learnableRope ? ((h * b->numKVHeads / numBufHeads) * numPairs + p) * S + xy : p * S + xy;
        const float c = cosT[t], sn = sinT[t];
        float* v = row + h * b->qHeadDim + 2 * p;
        const float x0 = v[0], x1 = v[1];
        v[0] = x0 * c - x1 * sn;
        v[1] = x0 * sn + x1 * c;
      }
  }
}
/* TransformerAttentionBlock::apply (eigenbackend.cpp:1376-1600) */
static void attention_apply(const OBlock* b, int n, int X, int Y, float* trunk, const float* mask) {
  const int S = X * Y, C = b->qProj.ic, H = b->numHeads, KVH = b->numKVHeads, QD = b->qHeadDim, VD = b->vHeadDim;
  const size_t NS = (size_t)n * S;
  float* ln = falloc(NS * C);
  float* q = falloc(NS * H * QD);
  float* k = falloc(NS * KVH * QD);
  float* v = falloc(NS * KVH * VD);
  float* att = falloc(NS * H * VD);
  trms_apply(&b->preLN, n, S, trunk, ln, mask);
  matmul_cells(&b->qProj, NS, ln, q);
  matmul_cells(&b->kProj, NS, ln, k);
  matmul_cells(&b->vProj, NS, ln, v);
  if(b->useRope) {
    const int numPairs = QD / 2;
    const size_t tsz = (size_t)(b->learnableRope ? KVH : 1) * numPairs * S;
    float* cosT = falloc(tsz);
    float* sinT = falloc(tsz);
    rope_tables(b, X, Y, cosT, sinT);
    rope_apply(b, n, S, q, H, cosT, sinT);
    rope_apply(b, n, S, k, KVH, cosT, sinT);
    free(cosT); free(sinT);
  }
  const float scale = 1.0f / sqrtf((float)QD);
  const int group = H / KVH;
#pragma omp parallel for schedule(dynamic) collapse(2)
  for(int bi = 0; bi < n; bi++)
    for(int h = 0; h < H; h++) {
      const int kvh = h / group;
      const float* maskN = mask + (size_t)bi * S;
      float* scores = (float*)malloc(sizeof(float) * S);
      for(int qi = 0; qi < S; qi++) {
        float* out = att + ((size_t)bi * S + qi) * H * VD + h * VD;
        for(int d = 0; d < VD; d++) out[d] = 0.0f;
        if(maskN[qi] == 0.0f) continue; /* masked queries contribute nothing (eigenbackend.cpp:1503-1508) */
        const float* qv = q + ((size_t)bi * S + qi) * H * QD + h * QD;
        float maxVal = -1e30f;
        for(int ki = 0; ki < S; ki++) {
          if(maskN[ki] == 0.0f) continue;
          const float* kv = k + ((size_t)bi * S + ki) * KVH * QD + kvh * QD;
          float s = 0.0f;
          for(int d = 0; d < QD; d++) s += qv[d] * kv[d];
          s *= scale;
          scores[ki] = s;
          if(s > maxVal) maxVal = s;
        }
        float sumExp = 0.0f;
        for(int ki = 0; ki < S; ki++) {
          if(maskN[ki] == 0.0f) { scores[ki] = 0.0f; continue; }
          scores[ki] = expf(scores[ki] - maxVal);
          sumExp += scores[ki];
        }
        const float inv = 1.0f / sumExp;
        for(int ki = 0; ki < S; ki++) {
          const float w = scores[ki] * inv;
          if(w == 0.0f) continue;
          const float* vv = v + ((size_t)bi * S + ki) * KVH * VD + kvh * VD;
          for(int d = 0; d < VD; d++) out[d] += w * vv[d];
        }
      }
      free(scores);
    }
  matmul_cells(&b->outProj, NS, att, ln); /* ln reused as the block output */
#pragma omp parallel for schedule(static)
  for(long p = 0; p < (long)NS; p++)
    for(int c = 0; c < C; c++) trunk[(size_t)p * C + c] += ln[(size_t)p * C + c] * mask[p];
  free(ln); free(q); free(k); free(v); free(att);
}
/* TransformerFFNBlock::apply (eigenbackend.cpp:1645-1718), SwiGLU */
static void ffn_apply(const OBlock* b, int n, int S, float* trunk, const float* mask) {
  const int C = b->linear1.ic, F = b->ffnChannels;
  const size_t NS = (size_t)n * S;
  float* ln = falloc(NS * C);
  float* a = falloc(NS * F);
  float* g = falloc(NS * F);
  trms_apply(&b->preLN, n, S, trunk, ln, mask);
  matmul_cells(&b->linear1, NS, ln, a);
  matmul_cells(&b->linearGate, NS, ln, g);
  for(size_t i = 0; i < NS * F; i++) a[i] = a[i] / (1.0f + expf(-a[i])) * g[i];
  matmul_cells(&b->linear2, NS, a, ln);
#pragma omp parallel for schedule(static)
  for(long p = 0; p < (long)NS; p++)
    for(int c = 0; c < C; c++) trunk[(size_t)p * C + c] += ln[(size_t)p * C + c] * mask[p];
  free(ln); free(a); free(g);
}
/* RMSNormLayer::apply for the trunk tip (eigenbackend.cpp:960-1031): gamma, beta, activation; per cell or (spatial) per board */
static void rms_tip_apply(const okmx_model* m, int n, int S, const float* in, float* out, const float* mask) {
  const int C = m->C;
  for(int bi = 0; bi < n; bi++) {
    float boardRms = 0.0f;
    if(m->rmsSpatial) {
      float sumSq = 0.0f;
      int count = 0;
      for(int p = 0; p < S; p++) {
        if(mask[(size_t)bi * S + p] == 0.0f) continue;
        const float* ip = in + ((size_t)bi * S + p) * C;
        for(int c = 0; c < C; c++) sumSq += ip[c] * ip[c];
        count++;
      }
      boardRms = 1.0f / sqrtf(sumSq / ((float)count * (float)C) + m->rmsEps);
    }
    for(int p = 0; p < S; p++) {
      const float* ip = in + ((size_t)bi * S + p) * C;
      float* op = out + ((size_t)bi * S + p) * C;
      if(mask[(size_t)bi * S + p] == 0.0f) { for(int c = 0; c < C; c++) op[c] = 0.0f; continue; }
      float rms = boardRms;
      if(!m->rmsSpatial) {
        float sumSq = 0.0f;
        for(int c = 0; c < C; c++) sumSq += ip[c] * ip[c];
        rms = 1.0f / sqrtf(sumSq / (float)C + m->rmsEps);
      }
      for(int c = 0; c < C; c++) op[c] = act_apply(ip[c] * rms * m->rmsGamma[c] + m->rmsBeta[c], m->trunkTipAct);
    }
  }
}

static void blockstack_apply(const OBlock* blocks, int numBlocks, int n, int X, int Y, float* trunk,
                             const float* mask, const float* maskSum);

/* ResidualBlock / GlobalPoolingResidualBlock / NestedBottleneckResidualBlock ::apply
 * (eigenbackend.cpp:1103-1146, 1150-1230, 1266-1315). trunk is updated in place. */
static void block_apply(const OBlock* b, int n, int X, int Y, float* trunk, const float* mask, const float* maskSum) {
  const int S = X * Y;
  const size_t NS = (size_t)n * S;
  if(b->kind == BLK_ORDINARY) {
    float* t = falloc(NS * b->preBN.c);
    float* mid = falloc(NS * b->regularConv.oc);
    bnact_apply(&b->preBN, n, S, trunk, t, mask);
    conv_apply(&b->regularConv, n, X, Y, t, mid, 0);
    bnact_apply(&b->midBN, n, S, mid, mid, mask);
    conv_apply(&b->finalConv, n, X, Y, mid, trunk, 1);
    free(t); free(mid);
  } else if(b->kind == BLK_GPOOL) {
    const int G = b->gpoolConv.oc, R = b->regularConv.oc;
    float* t = falloc(NS * b->preBN.c);
    float* r = falloc(NS * R);
    float* g = falloc(NS * G);
    float* gp = falloc((size_t)n * 3 * G);
    float* gb = falloc((size_t)n * R);
    bnact_apply(&b->preBN, n, S, trunk, t, mask);
    conv_apply(&b->regularConv, n, X, Y, t, r, 0);
    conv_apply(&b->gpoolConv, n, X, Y, t, g, 0);
    bnact_apply(&b->gpoolBN, n, S, g, g, mask);
    pool_gpool(n, S, G, g, gp, mask, maskSum);
    matmul_apply(&b->gpoolToBiasMul, n, gp, gb);
    add_nc_bias(n, S, R, r, gb);
    bnact_apply(&b->midBN, n, S, r, r, mask);
    conv_apply(&b->finalConv, n, X, Y, r, trunk, 1);
    free(t); free(r); free(g); free(gp); free(gb);
  } else if(b->kind == BLK_ATTENTION) {
    attention_apply(b, n, X, Y, trunk, mask);
  } else if(b->kind == BLK_FFN) {
    ffn_apply(b, n, S, trunk, mask);
  } else { /* nested bottleneck */
    const int M = b->regularConv.oc;
    float* t = falloc(NS * b->preBN.c);
    float* mid = falloc(NS * M);
    bnact_apply(&b->preBN, n, S, trunk, t, mask);
    conv_apply(&b->regularConv, n, X, Y, t, mid, 0);
    blockstack_apply(b->blocks, b->numBlocks, n, X, Y, mid, mask, maskSum);
    bnact_apply(&b->midBN, n, S, mid, mid, mask);
    conv_apply(&b->finalConv, n, X, Y, mid, trunk, 1);
    free(t); free(mid);
  }
}
static void blockstack_apply(const OBlock* blocks, int numBlocks, int n, int X, int Y, float* trunk,
                             const float* mask, const float* maskSum) {
  for(int i = 0; i < numBlocks; i++) block_apply(&blocks[i], n, X, Y, trunk, mask, maskSum);
}

/* copyWithSymmetry (nninputs.cpp:529-577), NHWC branch, nSize = 1 */
void okmx_copy_with_symmetry(const float* src, float* dst, int hSize, int wSize, int cSize, int symmetry, int reverse) {
  int transpose = (symmetry & 0x4) != 0 && hSize == wSize;
  int flipX = (symmetry & 0x2) != 0;
  int flipY = (symmetry & 0x1) != 0;
  if(transpose && !reverse) { int t = flipX; flipX = flipY; flipY = t; }
  int hStride = wSize * cSize, wStride = cSize;
  int hBaseNew = 0, hStrideNew = hStride, wBaseNew = 0, wStrideNew = wStride;
  if(flipY) { hBaseNew = (hSize - 1) * hStrideNew; hStrideNew = -hStrideNew; }
  if(flipX) { wBaseNew = (wSize - 1) * wStrideNew; wStrideNew = -wStrideNew; }
  if(transpose) { int t = hStrideNew; hStrideNew = wStrideNew; wStrideNew = t; }
  for(int h = 0; h < hSize; h++)
    for(int w = 0; w < wSize; w++) {
      int o = h * hStride + w * wStride;
      int nw = hBaseNew + h * hStrideNew + wBaseNew + w * wStrideNew;
      for(int c = 0; c < cSize; c++) dst[nw + c] = src[o + c];
    }
}

/* Trunk::apply (eigenbackend.cpp:1909-1947) up to (which==1) or including (which==0) the tip BN */
static void trunk_apply(const okmx_model* m, int n, int X, int Y, const float* input, const float* inputGlobal,
                        const float* inputMeta, const float* mask, const float* maskSum, float* trunkRaw, float* trunkOut) {
  const int S = X * Y, C = m->C;
  float* gbias = falloc((size_t)n * C);
  conv_apply(&m->initialConv, n, X, Y, input, trunkRaw, 0);
  matmul_apply(&m->initialMatMul, n, inputGlobal, gbias);
  add_nc_bias(n, S, C, trunkRaw, gbias);
  if(m->metaEncoderVersion > 0) { /* SGFMetadataEncoder::apply (eigenbackend.cpp:1848-1860), added like the global bias (:1929-1932) */
    const int C1 = m->metaMul1.oc, C2 = m->metaMul2.oc;
    float* h1 = falloc((size_t)n * C1);
    float* h2 = falloc((size_t)n * C2);
    matmul_apply(&m->metaMul1, n, inputMeta, h1);
    matbias_apply(&m->metaBias1, n, h1);
    for(size_t i = 0; i < (size_t)n * C1; i++) h1[i] = act_apply(h1[i], m->metaAct1);
    matmul_apply(&m->metaMul2, n, h1, h2);
    matbias_apply(&m->metaBias2, n, h2);
    for(size_t i = 0; i < (size_t)n * C2; i++) h2[i] = act_apply(h2[i], m->metaAct2);
    matmul_apply(&m->metaMul3, n, h2, gbias);
    add_nc_bias(n, S, C, trunkRaw, gbias);
    free(h1); free(h2);
  }
  blockstack_apply(m->blocks, m->numBlocks, n, X, Y, trunkRaw, mask, maskSum);
  if(trunkOut) {
    if(m->trunkNormKind == 0) bnact_apply(&m->trunkTipBN, n, S, trunkRaw, trunkOut, mask);
    else rms_tip_apply(m, n, S, trunkRaw, trunkOut, mask);
  }
  free(gbias);
}

static void compute_mask(int n, int S, int Cin, const float* input, float* mask, float* maskSum) {
  /* Model::apply (eigenbackend.cpp:2181-2182), computeMaskSum (:124-134) */
  for(int b = 0; b < n; b++) {
    float s = 0.0f;
    for(int p = 0; p < S; p++) {
      float v = input[((size_t)b * S + p) * Cin];
      mask[(size_t)b * S + p] = v;
      s += v;
    }
    maskSum[b] = s;
  }
}

int okmx_eval_trunk(const okmx_model* m, int X, int Y, int n, const float* spatial, const float* global, int which,
                    float* out) {
  if(!m || !spatial || !global || !out || n <= 0) return fail(KMX_ERR_INVALID_ARG, "okmx_eval_trunk: bad argument");
  if(m->metaEncoderVersion > 0) return fail(KMX_ERR_UNSUPPORTED, "okmx_eval_trunk: not available for sgf-metadata nets");
  const int S = X * Y, C = m->C;
  float* mask = falloc((size_t)n * S);
  float* maskSum = falloc(n);
  compute_mask(n, S, m->info.num_input_channels, spatial, mask, maskSum);
  float* raw = falloc((size_t)n * S * C);
  trunk_apply(m, n, X, Y, spatial, global, NULL, mask, maskSum, raw, which == 0 ? out : NULL);
  if(which != 0) memcpy(out, raw, sizeof(float) * (size_t)n * S * C);
  free(raw); free(mask); free(maskSum);
  return KMX_OK;
}

/* NeuralNet::getOutput for the Eigen backend (eigenbackend.cpp:2445-2628) */
int okmx_eval(const okmx_model* m, int X, int Y, int n, const float* const* row_spatial, const float* const* row_global,
              const int* symmetry, const float* policy_optimism, float* const* out_policy, float* out_value,
              float* out_score, float* const* out_ownership, int num_threads) {
  return okmx_eval_meta(m, X, Y, n, row_spatial, row_global, NULL, symmetry, policy_optimism, out_policy, out_value, out_score,
                        out_ownership, num_threads);
}

int okmx_eval_meta(const okmx_model* m, int X, int Y, int n, const float* const* row_spatial, const float* const* row_global,
                   const float* const* row_meta, const int* symmetry, const float* policy_optimism, float* const* out_policy,
                   float* out_value, float* out_score, float* const* out_ownership, int num_threads) {
  if(!m || n <= 0 || !row_spatial || !row_global || !out_policy || !out_value || !out_score)
    return fail(KMX_ERR_INVALID_ARG, "okmx_eval: bad argument");
  if((m->metaEncoderVersion > 0) != (row_meta != NULL)) /* eigenbackend.cpp:1929-1936 asserts the same pairing */
    return fail(KMX_ERR_INVALID_ARG, "okmx_eval: the metadata input must be given exactly for nets with an sgf-metadata encoder");
  if(X < 2 || Y < 2 || X > 19 || Y > 19) return fail(KMX_ERR_INVALID_ARG, "okmx_eval: nnXLen/nnYLen out of range");
#ifdef _OPENMP
  int prevThreads = omp_get_max_threads();
  if(num_threads > 0) omp_set_num_threads(num_threads);
#else
  (void)num_threads;
#endif
  const int S = X * Y, C = m->C, Cin = m->info.num_input_channels, G = m->info.num_input_global_channels;
  const int NP = m->info.num_policy_channels, NSV = m->info.num_score_value_channels;
  float* input = falloc((size_t)n * S * Cin);
  float* inputGlobal = falloc((size_t)n * G);
  for(int b = 0; b < n; b++) {
    memcpy(inputGlobal + (size_t)b * G, row_global[b], sizeof(float) * G);
    okmx_copy_with_symmetry(row_spatial[b], input + (size_t)b * S * Cin, Y, X, Cin, symmetry ? symmetry[b] : 0, 0);
  }
  float* mask = falloc((size_t)n * S);
  float* maskSum = falloc(n);
  compute_mask(n, S, Cin, input, mask, maskSum);
  float* trunkRaw = falloc((size_t)n * S * C);
  float* trunk = falloc((size_t)n * S * C);
  float* inputMeta = NULL;
  if(row_meta) {
    const int M = m->info.num_input_meta_channels;
    inputMeta = falloc((size_t)n * M);
    for(int b = 0; b < n; b++) memcpy(inputMeta + (size_t)b * M, row_meta[b], sizeof(float) * M);
  }
  trunk_apply(m, n, X, Y, input, inputGlobal, inputMeta, mask, maskSum, trunkRaw, trunk);
  free(trunkRaw);
  free(inputMeta);

  /* PolicyHead::apply (eigenbackend.cpp:1992-2036) */
  const int P1 = m->p1Conv.oc, G1 = m->g1Conv.oc;
  float* p1 = falloc((size_t)n * S * P1);
  float* g1 = falloc((size_t)n * S * G1);
  float* g1c = falloc((size_t)n * 3 * G1);
  float* g1b = falloc((size_t)n * P1);
  float* policy = falloc((size_t)n * S * NP);
  float* policyPass = falloc((size_t)n * NP);
  conv_apply(&m->p1Conv, n, X, Y, trunk, p1, 0);
  conv_apply(&m->g1Conv, n, X, Y, trunk, g1, 0);
  bnact_apply(&m->g1BN, n, S, g1, g1, mask);
  pool_gpool(n, S, G1, g1, g1c, mask, maskSum);
  matmul_apply(&m->gpoolToBiasMul, n, g1c, g1b);
  add_nc_bias(n, S, P1, p1, g1b);
  bnact_apply(&m->p1BN, n, S, p1, p1, mask);
  conv_apply(&m->p2Conv, n, X, Y, p1, policy, 0);
  if(m->version >= 15) {
    float* pp = falloc((size_t)n * m->gpoolToPassMul.oc);
    matmul_apply(&m->gpoolToPassMul, n, g1c, pp);
    matbias_apply(&m->gpoolToPassBias, n, pp);
    for(size_t i = 0; i < (size_t)n * m->gpoolToPassMul.oc; i++) pp[i] = act_apply(pp[i], m->passAct);
    matmul_apply(&m->gpoolToPassMul2, n, pp, policyPass);
    free(pp);
  } else {
    matmul_apply(&m->gpoolToPassMul, n, g1c, policyPass);
  }

  /* ValueHead::apply (eigenbackend.cpp:2079-2114) */
  const int V1 = m->v1Conv.oc, V2 = m->v2Mul.oc;
  float* v1 = falloc((size_t)n * S * V1);
  float* v1m = falloc((size_t)n * 3 * V1);
  float* v2 = falloc((size_t)n * V2);
  float* value = falloc((size_t)n * 3);
  float* scoreValue = falloc((size_t)n * NSV);
  float* ownership = falloc((size_t)n * S);
  conv_apply(&m->v1Conv, n, X, Y, trunk, v1, 0);
  bnact_apply(&m->v1BN, n, S, v1, v1, mask);
  pool_value(n, S, V1, v1, v1m, maskSum);
  matmul_apply(&m->v2Mul, n, v1m, v2);
  matbias_apply(&m->v2Bias, n, v2);
  for(size_t i = 0; i < (size_t)n * V2; i++) v2[i] = act_apply(v2[i], m->v2Act);
  matmul_apply(&m->v3Mul, n, v2, value);
  matbias_apply(&m->v3Bias, n, value);
  matmul_apply(&m->sv3Mul, n, v2, scoreValue);
  matbias_apply(&m->sv3Bias, n, scoreValue);
  conv_apply(&m->vOwnershipConv, n, X, Y, v1, ownership, 0);

  /* output scatter (eigenbackend.cpp:2539-2627) */
  float* tmp = falloc(S);
  for(int b = 0; b < n; b++) {
    const int sym = symmetry ? symmetry[b] : 0;
    const float opt = policy_optimism ? policy_optimism[b] : 0.0f;
    const float* ps = policy + (size_t)b * S * NP;
    const float* pps = policyPass + (size_t)b * NP;
    float* po = out_policy[b];
    if(NP == 2 || (NP == 4 && m->version >= 16)) {
      for(int i = 0; i < S; i++) {
        float p = ps[i * NP], pOpt = ps[i * NP + 1];
        tmp[i] = p + (pOpt - p) * opt;
      }
      okmx_copy_with_symmetry(tmp, po, Y, X, 1, sym, 1);
      po[S] = pps[0] + (pps[1] - pps[0]) * opt;
    } else {
      okmx_copy_with_symmetry(ps, po, Y, X, 1, sym, 1);
      po[S] = pps[0];
    }
    for(int i = 0; i < 3; i++) out_value[b * 3 + i] = value[b * 3 + i];
    for(int i = 0; i < 6; i++) out_score[b * 6 + i] = i < NSV ? scoreValue[b * NSV + i] : 0.0f;
    if(out_ownership && out_ownership[b]) okmx_copy_with_symmetry(ownership + (size_t)b * S, out_ownership[b], Y, X, 1, sym, 1);
  }
  free(tmp);
  free(input); free(inputGlobal); free(mask); free(maskSum); free(trunk);
  free(p1); free(g1); free(g1c); free(g1b); free(policy); free(policyPass);
  free(v1); free(v1m); free(v2); free(value); free(scoreValue); free(ownership);
#ifdef _OPENMP
  omp_set_num_threads(prevThreads);
#endif
  return KMX_OK;
}

/* ---------------------------------- layer test hooks --------------------------------------- */
/* kmx_conv_desc weights are [oc][ic][ky][kx] (the reference's in-memory order, desc.cpp:131-152) */
static void conv_from_desc(const kmx_conv_desc* d, OConv* c) {
  memset(c, 0, sizeof(*c));
  c->ky = d->conv_y_size; c->kx = d->conv_x_size; c->ic = d->in_channels; c->oc = d->out_channels;
  c->w = falloc((size_t)c->ky * c->kx * c->ic * c->oc);
  for(int o = 0; o < c->oc; o++)
    for(int i = 0; i < c->ic; i++)
      for(int y = 0; y < c->ky; y++)
        for(int x = 0; x < c->kx; x++)
          c->w[(((size_t)y * c->kx + x) * c->ic + i) * c->oc + o] =
            d->weights[(((size_t)o * c->ic + i) * c->ky + y) * c->kx + x];
}
static void bn_from_desc(const kmx_bnact_desc* d, OBn* b) {
  memset(b, 0, sizeof(*b));
  b->c = d->num_channels; b->act = d->activation;
  b->scale = falloc(b->c); b->bias = falloc(b->c);
  memcpy(b->scale, d->merged_scale, sizeof(float) * b->c);
  memcpy(b->bias, d->merged_bias, sizeof(float) * b->c);
}
static void matmul_from_desc(const kmx_matmul_desc* d, OMatMul* m) {
  memset(m, 0, sizeof(*m));
  m->ic = d->in_channels; m->oc = d->out_channels;
  m->w = falloc((size_t)m->ic * m->oc);
  memcpy(m->w, d->weights, sizeof(float) * (size_t)m->ic * m->oc);
}
static void mask_sums(int n, int S, const float* mask, float* maskSum) {
  for(int b = 0; b < n; b++) {
    float s = 0.0f;
    for(int p = 0; p < S; p++) s += mask[(size_t)b * S + p];
    maskSum[b] = s;
  }
}

int okmx_test_conv(const kmx_conv_desc* desc, int batch, int X, int Y, const float* in, float* out) {
  if(!desc || !in || !out) return fail(KMX_ERR_INVALID_ARG, "okmx_test_conv: null argument");
  OConv c;
  conv_from_desc(desc, &c);
  conv_apply(&c, batch, X, Y, in, out, 0);
  free_conv(&c);
  return KMX_OK;
}
int okmx_test_bnact(const kmx_bnact_desc* desc, int batch, int X, int Y, const float* in, const float* mask, float* out) {
  if(!desc || !in || !out || !mask) return fail(KMX_ERR_INVALID_ARG, "okmx_test_bnact: null argument");
  OBn b;
  bn_from_desc(desc, &b);
  bnact_apply(&b, batch, X * Y, in, out, mask);
  free_bn(&b);
  return KMX_OK;
}
int okmx_test_resblock(const kmx_resblock_desc* d, int batch, int X, int Y, const float* in, const float* mask, float* out) {
  if(!d || !in || !out || !mask) return fail(KMX_ERR_INVALID_ARG, "okmx_test_resblock: null argument");
  OBlock b;
  memset(&b, 0, sizeof(b));
  b.kind = BLK_ORDINARY;
  bn_from_desc(&d->pre_bn, &b.preBN);
  conv_from_desc(&d->regular_conv, &b.regularConv);
  bn_from_desc(&d->mid_bn, &b.midBN);
  conv_from_desc(&d->final_conv, &b.finalConv);
  const int S = X * Y;
  float* maskSum = falloc(batch);
  mask_sums(batch, S, mask, maskSum);
  memcpy(out, in, sizeof(float) * (size_t)batch * S * b.preBN.c);
  block_apply(&b, batch, X, Y, out, mask, maskSum);
  free(maskSum);
  free_block(&b);
  return KMX_OK;
}
int okmx_test_gpoolblock(const kmx_gpoolblock_desc* d, int batch, int X, int Y, const float* in, const float* mask,
                         float* out) {
  if(!d || !in || !out || !mask) return fail(KMX_ERR_INVALID_ARG, "okmx_test_gpoolblock: null argument");
  OBlock b;
  memset(&b, 0, sizeof(b));
  b.kind = BLK_GPOOL;
  bn_from_desc(&d->pre_bn, &b.preBN);
  conv_from_desc(&d->regular_conv, &b.regularConv);
  conv_from_desc(&d->gpool_conv, &b.gpoolConv);
  bn_from_desc(&d->gpool_bn, &b.gpoolBN);
  matmul_from_desc(&d->gpool_to_bias_mul, &b.gpoolToBiasMul);
  bn_from_desc(&d->mid_bn, &b.midBN);
  conv_from_desc(&d->final_conv, &b.finalConv);
  const int S = X * Y;
  float* maskSum = falloc(batch);
  mask_sums(batch, S, mask, maskSum);
  memcpy(out, in, sizeof(float) * (size_t)batch * S * b.preBN.c);
  block_apply(&b, batch, X, Y, out, mask, maskSum);
  free(maskSum);
  free_block(&b);
  return KMX_OK;
}

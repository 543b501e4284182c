/*
 * katamx.h — C ABI of the MI355X-native KataGo neural-net evaluation backend.
 *
 * This is the drop-in boundary for ONE hot path of lightvector/KataGo: the
 * NNEvaluator batch path. Every entry point replaces (is bound by) one function of
 * the reference's compile-time backend interface `namespace NeuralNet`
 * (reference: cpp/neuralnet/nninterface.h:32-182). The reference-side binding
 * (a backend TU implementing NeuralNet:: on top of this ABI) is
 * integration/katamxbackend.cpp; INTEGRATION.md shows how it is wired in.
 *
 * Conventions
 *   - plain C, opaque handles, plain pointers and sizes; no C++/torch types.
 *   - every function that can fail returns KMX_OK (0) or a negative kmx_status;
 *     kmx_last_error() returns a thread-local human-readable message. The reference
 *     reports errors by throwing StringError (cpp/core/global.h:151); the shim rethrows.
 *   - all tensors crossing the boundary are fp32, little endian, "NHWC" where spatial:
 *     index = (y*nnXLen + x)*C + c. "All outputs are logits" (nninterface.h:116):
 *     no softmax / tanh / scaling is applied by the backend.
 *   - a kmx_handle is used by exactly one host thread at a time (nninterface.h:19-21).
 *   - the implementation is HIP-only (gfx950). There is no CPU fallback: without a
 *     usable GPU kmx_handle_create fails with KMX_ERR_DEVICE.
 */
#ifndef KATAMX_H_
#define KATAMX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: + kmx_eval_meta / kmx_eval_device_meta (sgf-metadata nets), kmx_eval_packed / kmx_pack_row (bit-packed inputs),
 *    kmx_handle_set_split_min; kmx_model_info.reserved0 became meta_encoder_version. Additive over 1.
 * 3: + kmx_test_rmsnorm / kmx_test_attention / kmx_test_swiglu (experimental unit hooks). Additive over 2.
 * 4: + kmx_handle_set_graphs / kmx_handle_graph_stats (hipGraph replay of the launch schedule); kmx_handle_set_split_min
 *    accepts a negative value (restore the creation value); the handle stream orders both halves of a split batch.
 *    + kmx_test_pointwise_pair (unit hook of the fused 1x1 -> 1x1 seam kernel); + kmx_batcher_* (persistent leaf
 *    batcher). Additive over 3.
 * 5: + kmx_batcher_submit_packed (a row that was featurised as bit planes is handed over as such). Additive over 4.
 * 6: + kmx_test_conv_chain (unit hook of the chained 3x3 convolutions); kmx_batcher_create: a batch is sealed at ONE granule of the
 *    device (its CU count, 256 on MI355X) whatever max_batch_size asks for beyond it - a batcher created for 1024 rows runs batches of
 *    at most 256, and its engines and staging are sized for that (see kmx_batcher_create). Additive over 5.
 * 7: the kernel-tuning entry points (kmx_bench_*, kmx_debug_conv_cfg) left this header: they are instrumentation of this repository's
 *    tools, declared in katago_amd/csrc/katamx_tuning.h and still exported; + kmx_batcher_effective_batch (the seal size a batcher
 *    actually got); precision_mode KMX_PREC_FP32 is served (a plain fp32 device path, correctness only). Additive over 6. */
#define KMX_ABI_VERSION 7

typedef enum kmx_status {
  KMX_OK = 0,
  KMX_ERR_INVALID_ARG = -1,
  KMX_ERR_IO = -2,          /* model file unreadable / truncated */
  KMX_ERR_MODEL = -3,       /* model parse error or unsupported architecture */
  KMX_ERR_DEVICE = -4,      /* HIP runtime error / no device */
  KMX_ERR_UNSUPPORTED = -5, /* valid request this backend does not implement */
  KMX_ERR_INTERNAL = -6
} kmx_status;

/* Activation kinds — numeric values equal the reference's (cpp/neuralnet/activations.h:4-17). */
enum { KMX_ACT_IDENTITY = 0, KMX_ACT_RELU = 1, KMX_ACT_MISH = 2, KMX_ACT_SILU = 3,
       KMX_ACT_MISH_SCALE8 = 4 /* x tanh(softplus(8x)): mish of a net whose tensors carry 1/8 of their values (desc.cpp:421-445) */ };

/* Arithmetic of the device path. The reference's tri-state useFP16Mode
 * (cpp/core/commontypes.h:4-30) maps as: False -> KMX_PREC_FP32, True/Auto -> KMX_PREC_AUTO. */
enum {
  KMX_PREC_AUTO = 0, /* backend default: fp16 storage (convolutional nets run at 1/8 of their values, the reference's fp16 range
                        transform, desc.cpp:2718-2736; outputs unchanged), bf16 for nets that transform does not cover; fp32 accumulate */
  KMX_PREC_FP32 = 1, /* fp32 storage and arithmetic (slow verification mode) */
  KMX_PREC_FP16 = 2, /* fp16 storage, fp32 accumulate */
  KMX_PREC_BF16 = 3  /* bf16 storage, fp32 accumulate */
};

typedef struct kmx_model kmx_model;     /* replaces LoadedModel   (nninterface.h:27) */
typedef struct kmx_context kmx_context; /* replaces ComputeContext (nninterface.h:17) */
typedef struct kmx_handle kmx_handle;   /* replaces ComputeHandle + InputBuffers (nninterface.h:21,24) */

/* What NNEvaluator reads from ModelDesc (cpp/neuralnet/nneval.cpp:138-143,292,306,327;
 * fields of cpp/neuralnet/desc.h ModelDesc / ModelPostProcessParams). */
typedef struct kmx_model_info {
  char name[128];
  int32_t model_version;
  int32_t num_input_channels;        /* 22 for inputs v7 */
  int32_t num_input_global_channels; /* 19 for inputs v7 */
  int32_t num_input_meta_channels;   /* 0, or 192 for nets with an sgf-metadata encoder (metaEncoderVersion 1) */
  int32_t num_policy_channels;       /* 1, 2 or 4 */
  int32_t num_value_channels;        /* 3 */
  int32_t num_score_value_channels;  /* 4 (v8) or 6 (v>=9) */
  int32_t num_ownership_channels;    /* 1 */
  int32_t trunk_num_channels;
  int32_t mid_num_channels;
  int32_t num_blocks;
  int32_t meta_encoder_version;      /* 0 = none (was reserved0) */
  /* ModelPostProcessParams (desc.cpp:2477-2513); defaults for v<13 as in desc.h */
  float td_score_multiplier;
  float score_mean_multiplier;
  float score_stdev_multiplier;
  float lead_multiplier;
  float variance_time_multiplier;
  float shortterm_value_error_multiplier;
  float shortterm_score_error_multiplier;
  float output_scale_multiplier; /* always 1: this backend never applies scale-8 */
  int64_t num_parameters;
  double flops_per_position; /* 2*MAC per board point, direct-convolution count */
} kmx_model_info;

/* ---- process-wide ----------------------------------------------------------------- */
/* NeuralNet::globalInitialize / globalCleanup / printDevices  (nninterface.h:34-39) */
int kmx_abi_version(void);
int kmx_global_init(void);
void kmx_global_cleanup(void);
int kmx_device_count(void); /* <0 on error */
int kmx_device_name(int device, char* buf, size_t buflen);
const char* kmx_last_error(void);

/* ---- model ------------------------------------------------------------------------- */
/* NeuralNet::loadModelFile / freeLoadedModel / getModelDesc  (nninterface.h:43-46).
 * Reads KataGo .bin / .txt model files, optionally gzipped (format: desc.cpp:40-90,2441-2615).
 * expected_sha256 may be NULL or "" (no check). */
int kmx_model_load(const char* path, const char* expected_sha256, kmx_model** out);
void kmx_model_free(kmx_model* model);
int kmx_model_info_get(const kmx_model* model, kmx_model_info* out);

/* ---- context / handle -------------------------------------------------------------- */
/* NeuralNet::createComputeContext / freeComputeContext  (nninterface.h:50-65). */
int kmx_context_create(const int* gpu_idxs, int num_gpu_idxs, int nn_x_len, int nn_y_len,
                       int precision_mode, kmx_context** out);
void kmx_context_free(kmx_context* ctx);

/* NeuralNet::createComputeHandle (+createInputBuffers) / freeComputeHandle / isUsingFP16
 * (nninterface.h:76-99). Must be called on the thread that will use the handle;
 * gpu_idx < 0 means "default device 0". Uploads a private copy of the weights. */
int kmx_handle_create(kmx_context* ctx, const kmx_model* model, int max_batch_size,
                      int require_exact_nn_len, int gpu_idx, kmx_handle** out);
void kmx_handle_free(kmx_handle* handle);
int kmx_handle_precision(const kmx_handle* handle); /* KMX_PREC_FP32/FP16/BF16 actually in use */

/* ---- the hot path ------------------------------------------------------------------ */
/* NeuralNet::getOutput (nninterface.h:117-123; semantics: eigenbackend.cpp:2445-2628).
 *   row_spatial[i]  -> float[nnY*nnX*num_input_channels], NHWC, NOT yet symmetrised
 *                      (NNResultBuf::rowSpatialBuf, nneval.h:55)
 *   row_global[i]   -> float[num_input_global_channels]    (rowGlobalBuf)
 *   symmetry[i]     0..7: bit0 flipY, bit1 flipX, bit2 transpose (nninputs.cpp:529-597)
 *   policy_optimism[i]  blend weight p + (pOpt-p)*w (eigenbackend.cpp:2553-2562)
 * Outputs (all logits, inverse-symmetrised where spatial):
 *   out_policy[i]   -> float[nnX*nnY + 1], last element = pass logit
 *   out_value       -> float[n_rows*3]  win, loss, noResult (side to move)
 *   out_score       -> float[n_rows*6]  scoreMean, scoreStdev(pre-softplus), lead,
 *                      varTimeLeft, shorttermWinlossError, shorttermScoreError
 *                      (v8 nets: last two are 0)            (eigenbackend.cpp:2583-2606)
 *   out_ownership[i]-> float[nnX*nnY] or NULL to skip that row
 * Synchronous: on return all outputs are filled. */
int kmx_eval(kmx_handle* handle, int n_rows,
             const float* const* row_spatial, const float* const* row_global,
             const int* symmetry, const float* policy_optimism,
             float* const* out_policy, float* out_value, float* out_score,
             float* const* out_ownership);

/* Device-resident variant used by bench.py and the persistent batcher: inputs already
 * staged in HBM in the packed batch layout (see DESIGN.md "Input staging").
 *   d_spatial: float[n_rows][nnY*nnX*Cin], d_global: float[n_rows][G] (device pointers)
 *   symmetry/policy_optimism: host arrays. Outputs written to DEVICE buffers:
 *   d_policy float[n_rows][nnX*nnY+1], d_value float[n_rows][3], d_score float[n_rows][6],
 *   d_ownership float[n_rows][nnX*nnY]. Asynchronous on the handle's stream unless sync!=0. */
int kmx_eval_device(kmx_handle* handle, int n_rows, const float* d_spatial, const float* d_global,
                    const int* symmetry, const float* policy_optimism,
                    float* d_policy, float* d_value, float* d_score, float* d_ownership, int sync);
/* Nets with an sgf-metadata encoder (model header metaEncoderVersion > 0, "humanSL" nets; desc.cpp:1571-1625,
 * Trunk::apply eigenbackend.cpp:1929-1932) take one more input per row: NNResultBuf::rowMetaBuf, float[192]
 * (num_input_meta_channels). These two entry points are kmx_eval / kmx_eval_device with that input:
 *   row_meta[i] -> float[num_input_meta_channels]   /   d_meta: float[n_rows][num_input_meta_channels] on the device.
 * A net with an encoder requires it (KMX_ERR_INVALID_ARG through kmx_eval / with NULL); a net without one requires NULL,
 * as the reference asserts (eigenbackend.cpp:1929-1936). */
int kmx_eval_meta(kmx_handle* handle, int n_rows,
                  const float* const* row_spatial, const float* const* row_global, const float* const* row_meta,
                  const int* symmetry, const float* policy_optimism,
                  float* const* out_policy, float* out_value, float* out_score,
                  float* const* out_ownership);
int kmx_eval_device_meta(kmx_handle* handle, int n_rows, const float* d_spatial, const float* d_global, const float* d_meta,
                         const int* symmetry, const float* policy_optimism,
                         float* d_policy, float* d_value, float* d_score, float* d_ownership, int sync);
/* Bit-packed spatial input. All 22 spatial features of inputs v7 are 0/1 (nninputs.cpp:2321-2592); the
 * reference itself stores them bit-packed in its training rows (binaryInputNCHWPacked, dataio/trainingwrite.h:180-183;
 * packBits, dataio/trainingwrite.cpp:314-337) and that exact layout is accepted here:
 *   row_packed[i] -> uint8[num_input_channels * ceil(nnX*nnY / 8)]: plane by plane (NCHW), cells in y*nnX+x order, 8 per
 *   byte, MOST significant bit first, every plane zero-padded to a whole byte — 1012 bytes per 19x19 row instead of the
 *   31768 of the fp32 NHWC row. The device expands the bits (and applies the symmetry) in the input stage.
 * Otherwise identical to kmx_eval_meta (row_meta NULL unless the net has an sgf-metadata encoder).
 * kmx_pack_row converts one fp32 NHWC row (values 0 / 1; anything != 0 packs as 1) into that layout. */
int kmx_eval_packed(kmx_handle* handle, int n_rows,
                    const uint8_t* const* row_packed, const float* const* row_global, const float* const* row_meta,
                    const int* symmetry, const float* policy_optimism,
                    float* const* out_policy, float* out_value, float* out_score,
                    float* const* out_ownership);
int kmx_pack_row(const float* row_spatial_nhwc, int nn_x_len, int nn_y_len, int num_channels, uint8_t* out_packed);
/* hipStream_t the handle launches on. It is an ordering point for a whole batch, split or not: a handle that splits a
 * batch (kmx_handle_set_split_min below) launches the second half on a stream of its own, forked from this stream by an
 * event at entry and joined back into it after its last launch. So inputs enqueued on this stream before an asynchronous
 * kmx_eval_device (sync = 0) are complete for both halves, and work enqueued on it afterwards sees all outputs. */
void* kmx_handle_stream(kmx_handle* handle);
int kmx_handle_sync(kmx_handle* handle);

/* ---- persistent leaf batcher (north_star: "a persistent device-side leaf batcher over a thin C-ABI") ----
 * Replaces the server half of NNEvaluator — serve() popping up to maxBatch rows off a queue and calling getOutput
 * synchronously (nneval.cpp:562-752, core/threadsafequeue.h:173-189) — for callers that submit rows themselves:
 *   kmx_batcher_submit  thread-safe, called by the thread that owns the leaf (a search thread, or a server thread of the
 *                       reference's NNEvaluator with the rows it popped): reserves a row of the batch that is filling and
 *                       bit-packs the fp32 NHWC feature planes (all V7 planes are 0/1; any other value fails THIS call with
 *                       KMX_ERR_INVALID_ARG, no row is reserved) into that batch's pinned staging, outside the lock. The out_* buffers
 *                       (layout as kmx_eval: policy nn_x*nn_y+1, value 3, score 6, ownership nn_x*nn_y or NULL = not
 *                       wanted) receive the row's results and must stay valid until kmx_batcher_wait returns. Blocks only
 *                       while every staging set is filling or on the DEVICE - never on results that have not been
 *                       collected, so a thread may hold any number of tickets. Returns a ticket.
 *   kmx_batcher_wait    blocks until the ticket's row has been written to its out_* buffers (or its batch failed: the
 *                       status and kmx_last_error say why). Each ticket is waited for exactly once; a ticket that is never
 *                       waited for keeps its (small) completion record until kmx_batcher_free.
 * Batching is greedy like the reference's when the device is idle (a waiting row never waits for more rows); while a
 * batch is on the device the next one accumulates, and FULL batches are launched behind it, up to max_in_flight
 * (default 2 when <= 0) between H2D and D2H at once on their own engines and streams, so that copies and kernels of
 * consecutive batches overlap. FULL means the device's granule, not necessarily max_batch_size: a convolution gives a board to
 * a work-group and a work-group to a compute unit, so a batch is sealed at the device's CU count (MI355X: 256; max_batch_size itself
 * below that) and no batch is ever larger - engines and staging are sized for that. Larger batches were measured and lose (a caller
 * with L leaves in flight is served best by batches well below L/2; batcher.cpp has the numbers): environment
 * KMX_BATCH_GROW_AHEAD=k allows them all the same - the limit is then the largest multiple of the CU count <= max_batch_size, and a
 * batch is sealed at a smaller multiple only while fewer than k batches are launched or queued. KMX_BATCH_QUANTUM overrides the
 * granule (0: seal at max_batch_size only).
 * rows/batches as nneval.cpp:712-713. A row's outputs are bit-identical to the same row through kmx_eval. */
typedef struct kmx_batcher kmx_batcher;
int kmx_batcher_create(kmx_context* ctx, const kmx_model* model, int max_batch_size, int max_in_flight, int gpu_idx,
                       kmx_batcher** out);
void kmx_batcher_free(kmx_batcher* batcher); /* fails rows not yet launched, completes those on the device; no thread may be inside submit/wait */
int kmx_batcher_submit(kmx_batcher* batcher, const float* row_spatial, const float* row_global, const float* row_meta,
                       int symmetry, float policy_optimism, float* out_policy, float* out_value, float* out_score,
                       float* out_ownership, uint64_t* ticket);
/* The same with the spatial planes already bit-packed (layout of kmx_eval_packed: num_input_channels * ceil(nn_x*nn_y/8) bytes,
 * plane by plane, most significant bit first) - for a caller that featurises straight into bits (integration/katamx_features.h
 * does, in place of the fp32 row of NNInputs::fillRowV7, nninputs.cpp:2288-2731): 1012 bytes copied into the staging instead of
 * 31768 read and packed. Padding bits of a plane's last byte must be zero. */
int kmx_batcher_submit_packed(kmx_batcher* batcher, const uint8_t* row_packed, const float* row_global, const float* row_meta,
                              int symmetry, float policy_optimism, float* out_policy, float* out_value, float* out_score,
                              float* out_ownership, uint64_t* ticket);
int kmx_batcher_wait(kmx_batcher* batcher, uint64_t ticket);
int kmx_batcher_stats(kmx_batcher* batcher, uint64_t* rows, uint64_t* batches);
/* The largest batch this batcher ever launches: min(max_batch_size, the device's granule) unless KMX_BATCH_GROW_AHEAD / KMX_BATCH_QUANTUM
 * say otherwise - what an embedder that asked for 1024 rows actually got (ABI 7). 0 for a null batcher. */
int kmx_batcher_effective_batch(const kmx_batcher* batcher);
int kmx_batcher_precision(const kmx_batcher* batcher); /* KMX_PREC_FP16 or KMX_PREC_BF16: what its engines compute in (isUsingFP16, nninterface.h:108) */

/* NNEvaluator counters (nneval.cpp:330-347, incremented :712-713): rows = evaluated
 * positions, batches = kmx_eval calls. */
int kmx_handle_stats(const kmx_handle* handle, uint64_t* rows, uint64_t* batches);

/* ---- instrumentation --------------------------------------------------------------- */
/* Per-kernel-class timing with hipEvents recorded on the handle's own stream around every launch of the
 * schedule (what bench.py's roofline figure is computed from; the reference has no equivalent, its
 * benchmark only reports wall-clock nnEvals/s, cpp/program/playutils.cpp:834-855).
 * flops/bytes are ALGORITHMIC totals over the recorded launches: 2*MAC*cells for convolutions
 * (direct-convolution count on the real board area) and the tensor bytes a launch must read+write. */
typedef struct kmx_profile_entry {
  char name[48];
  uint64_t launches;
  double total_ms;
  double flops;
  double bytes;
} kmx_profile_entry;
/* Large batches are evaluated as two halves on two HIP streams (two engines inside the handle) when the handle was
 * created with max_batch_size >= the split threshold (default 224 rows, environment KMX_SPLIT_MIN at creation; 0 = off):
 * one half's memory-bound phases overlap the other's MFMA loops. min_rows = 0 turns splitting off for later calls (used
 * to time a kernel with the chip to itself), a positive value sets the smallest batch that is split, a negative value
 * restores the threshold the handle was created with. */
int kmx_handle_set_split_min(kmx_handle* handle, int min_rows);
int kmx_handle_set_profiling(kmx_handle* handle, int enabled); /* resets the accumulated profile */
/* hipGraph replay (default off; environment KMX_GRAPHS=1 at creation turns it on): the ~130 kernel launches of a pass are
 * captured the second time a (row count, buffer pointers) combination occurs and replayed with one hipGraphLaunch from
 * then on - what NNEvaluator's warm-up evaluations (setIsWarmup, nneval.cpp:487-560) prepare. Same kernels, arguments
 * and order as direct launches: bit-identical results. graph_launches counts replayed passes (all engines of a handle).
 * Measured on MI355X (b18c384nbt, batch 1 ... 256): the same rate as direct launches to 1 % - a pass is bound by the
 * kernels, not by the CPU's launch calls - hence opt-in. */
int kmx_handle_set_graphs(kmx_handle* handle, int enabled);
int kmx_handle_graph_stats(const kmx_handle* handle, uint64_t* graph_launches);
int kmx_handle_get_profile(kmx_handle* handle, kmx_profile_entry* entries, int max_entries, int* n_entries);
/* ---- layer test hooks -------------------------------------------------------------- */
/* NeuralNet::testEvaluateConv / BatchNorm / ResidualBlock / GlobalPoolingResidualBlock
 * (nninterface.h:134-180). Raw fp32 NHWC buffers in host memory; weights in the
 * reference's in-memory layouts (conv: [oc][ic][ky][kx] desc.cpp:131-152; matmul:
 * [ic][oc] desc.cpp:461-476; BN: merged scale/bias desc.cpp:272-279). */
typedef struct kmx_conv_desc {
  int32_t conv_y_size, conv_x_size, in_channels, out_channels;
  const float* weights; /* [oc][ic][ky][kx] */
} kmx_conv_desc;
typedef struct kmx_bnact_desc {
  int32_t num_channels;
  int32_t activation; /* KMX_ACT_* */
  const float* merged_scale;
  const float* merged_bias;
} kmx_bnact_desc;
typedef struct kmx_matmul_desc {
  int32_t in_channels, out_channels;
  const float* weights; /* [ic][oc] */
} kmx_matmul_desc;
typedef struct kmx_resblock_desc { /* ResidualBlockDesc, desc.h */
  kmx_bnact_desc pre_bn;
  kmx_conv_desc regular_conv;
  kmx_bnact_desc mid_bn;
  kmx_conv_desc final_conv;
} kmx_resblock_desc;
typedef struct kmx_gpoolblock_desc { /* GlobalPoolingResidualBlockDesc, desc.h */
  kmx_bnact_desc pre_bn;
  kmx_conv_desc regular_conv;
  kmx_conv_desc gpool_conv;
  kmx_bnact_desc gpool_bn;
  kmx_matmul_desc gpool_to_bias_mul;
  kmx_bnact_desc mid_bn;
  kmx_conv_desc final_conv;
} kmx_gpoolblock_desc;

/* precision_mode: KMX_PREC_*; returns KMX_ERR_UNSUPPORTED if that mode is not available. */
int kmx_test_conv(const kmx_conv_desc* desc, int batch, int nn_x_len, int nn_y_len,
                  int precision_mode, const float* in_nhwc, float* out_nhwc);
int kmx_test_bnact(const kmx_bnact_desc* desc, int batch, int nn_x_len, int nn_y_len,
                   int precision_mode, const float* in_nhwc, const float* mask_nhw, float* out_nhwc);
int kmx_test_resblock(const kmx_resblock_desc* desc, int batch, int nn_x_len, int nn_y_len,
                      int precision_mode, const float* in_nhwc, const float* mask_nhw, float* out_nhwc);
int kmx_test_gpoolblock(const kmx_gpoolblock_desc* desc, int batch, int nn_x_len, int nn_y_len,
                        int precision_mode, const float* in_nhwc, const float* mask_nhw, float* out_nhwc);

/* Unit hook for the fused seam of two 1x1 convolutions between nested-bottleneck blocks (NestedBottleneckResidualBlock::apply,
 * eigenbackend.cpp:1308-1314, at a block boundary): trunk = resid + W1 x; t = act1(bn1(trunk)) mask; mid = W2 t;
 * mid_act = act2(bn2(mid)) mask. fused = 1: the one-launch kernel (KMX_ERR_UNSUPPORTED when no kernel exists for the
 * channel counts), fused = 0: the two convolution launches it replaces. Weights are [out][in]. */
int kmx_test_pointwise_pair(int batch, int nn_x_len, int nn_y_len, int precision_mode, int c1, int c2, int c3,
                            const float* in_nhwc, const float* resid_nhwc, const float* w1_oi, const float* scale1,
                            const float* bias1, int act1, const float* w2_oi, const float* scale2, const float* bias2, int act2,
                            const float* mask_nhw, int fused, float* out_trunk_raw, float* out_mid_raw, float* out_mid_act);
/* Unit hook for a chain of 3x3 convolutions 192 -> 192 - the inner residual blocks of a nested-bottleneck block (ResidualBlock::apply,
 * eigenbackend.cpp:1103-1146, n_conv / 2 times): for k = 0, 2, ..:  t = act(bn_k(conv_k(x))) mask;  r += conv_{k+1}(t);
 * x = act(bn_{k+1}(r)) mask. n_conv is 2 or 4; x and r are [cells][192] fp32 NHWC (r in: the residual stream), weights
 * [n_conv][out][in][3][3], scale / bias [n_conv][192]. chained = 0: one launch per convolution; 2 / 4: launches of that many
 * convolutions with the activated image handed over inside the CU (KMX_ERR_UNSUPPORTED when no such kernel exists for the
 * activation). Both forms take the one-work-group-per-board shape whatever the batch size. Outputs: r and x after the last block. */
int kmx_test_conv_chain(int batch, int nn_x_len, int nn_y_len, int precision_mode, int n_conv, const float* x_nhwc, const float* r_nhwc,
                        const float* w_oihw, const float* scale, const float* bias, int activation, const float* mask_nhw, int chained,
                        float* out_r, float* out_x);
/* EXPERIMENTAL unit hooks for the layers of model-v17 transformer trunks that are not convolutions (the reference has no
 * test hook for them; its definitions are TransformerRMSNormLayer / RMSNormLayer, eigenbackend.cpp:867-1034, the attention
 * of TransformerAttentionBlock::apply, :1376-1600, and the SwiGLU of TransformerFFNBlock::apply, :1674-1689). fp32 NHWC
 * buffers in and out, like the hooks above. These kernels have not run on hardware yet.
 *   rmsnorm:   out = act(in / rms * weight + beta) on on-board cells, 0 elsewhere; rms per cell over channels, or
 *              (per_board) per board over on-board cells x channels; beta may be NULL.
 *   attention: q [n][S][heads*q_dim], k [n][S][kv_heads*q_dim], v [n][S][kv_heads*v_dim] -> out [n][S][heads*v_dim];
 *              rope_cos/rope_sin [rope_heads == 1 ? 1 : kv_heads][q_dim/2][S] or NULL; keys with mask 0 are ignored,
 *              queries with mask 0 give 0.
 *   swiglu:    out = silu(a) * gate, [n][S][ffn_channels]. */
int kmx_test_rmsnorm(int batch, int nn_x_len, int nn_y_len, int precision_mode, int num_channels, float epsilon,
                     const float* weight, const float* beta, int activation, int per_board,
                     const float* in_nhwc, const float* mask_nhw, float* out_nhwc);
int kmx_test_attention(int batch, int nn_x_len, int nn_y_len, int precision_mode, int num_heads, int num_kv_heads,
                       int q_head_dim, int v_head_dim, const float* rope_cos, const float* rope_sin, int rope_heads,
                       const float* q, const float* k, const float* v, const float* mask_nhw, float* out);
int kmx_test_swiglu(int batch, int nn_x_len, int nn_y_len, int precision_mode, int ffn_channels,
                    const float* a, const float* gate, float* out);

#ifdef __cplusplus
}
#endif
#endif /* KATAMX_H_ */

/*
 * kmx_oracle.h — CPU ORACLE for the katamx hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C fp32 restatement of the arithmetic of the reference's Eigen CPU backend
 * (cpp/neuralnet/eigenbackend.cpp) and of its model loader (cpp/neuralnet/desc.cpp).
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg
 * can check / time the HIP path against an independent implementation. Nothing in the
 * product (katago_amd/, include/, integration/) may link, import or call it.
 *
 * Parity is PINNED: the oracle is driven through the reference's own test harness
 * (oracle/_ref/katago_oracle = reference host code + integration/katamxbackend.cpp
 * compiled with -DKMX_USE_ORACLE) against the reference's known-answer tests:
 * runnnlayertests (cpp/tests/testnn.cpp), runtinynntests (cpp/tests/tinymodel.cpp),
 * runnnontinyboardtest (cpp/tests/results/runNNOnTinyBoardTest.txt); and against the
 * reference PyTorch model on a nested-bottleneck net (tests/golden/, tools/gen_torch_golden.py).
 *
 * The functions mirror include/katamx.h one for one with the prefix okmx_ and the same
 * argument meaning; types are shared with katamx.h.
 */
#ifndef KMX_ORACLE_H_
#define KMX_ORACLE_H_

#include "../include/katamx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct okmx_model okmx_model;

const char* okmx_last_error(void);
int okmx_model_load(const char* path, const char* expected_sha256, okmx_model** out);
void okmx_model_free(okmx_model* model);
int okmx_model_info_get(const okmx_model* model, kmx_model_info* out);

/* Same contract as kmx_eval (host buffers in, host buffers out), stateless:
 * nn_x_len/nn_y_len are passed per call. num_threads<=0: all cores (OpenMP). */
int okmx_eval(const okmx_model* model, int nn_x_len, int nn_y_len, int n_rows,
              const float* const* row_spatial, const float* const* row_global,
              const int* symmetry, const float* policy_optimism,
              float* const* out_policy, float* out_value, float* out_score,
              float* const* out_ownership, int num_threads);

/* okmx_eval for nets with an sgf-metadata encoder: row_meta[i] -> float[num_input_meta_channels] (see kmx_eval_meta) */
int okmx_eval_meta(const okmx_model* model, int nn_x_len, int nn_y_len, int n_rows,
                   const float* const* row_spatial, const float* const* row_global, const float* const* row_meta,
                   const int* symmetry, const float* policy_optimism,
                   float* const* out_policy, float* out_value, float* out_score,
                   float* const* out_ownership, int num_threads);

/* Intermediate tensors for parity bisection (the reference's DEBUG_INTERMEDIATE_VALUES idea,
 * eigenbackend.cpp:26-27). which: 0 = trunk after tip BN+act [n][S][C] NHWC,
 * 1 = raw trunk before tip BN. Evaluates with symmetry 0. */
int okmx_eval_trunk(const okmx_model* model, int nn_x_len, int nn_y_len, int n_rows,
                    const float* spatial_nhwc, const float* global_nc, int which, float* out);

int okmx_test_conv(const kmx_conv_desc* desc, int batch, int nn_x_len, int nn_y_len,
                   const float* in_nhwc, float* out_nhwc);
int okmx_test_bnact(const kmx_bnact_desc* desc, int batch, int nn_x_len, int nn_y_len,
                    const float* in_nhwc, const float* mask_nhw, float* out_nhwc);
int okmx_test_resblock(const kmx_resblock_desc* desc, int batch, int nn_x_len, int nn_y_len,
                       const float* in_nhwc, const float* mask_nhw, float* out_nhwc);
int okmx_test_gpoolblock(const kmx_gpoolblock_desc* desc, int batch, int nn_x_len, int nn_y_len,
                         const float* in_nhwc, const float* mask_nhw, float* out_nhwc);

/* copyWithSymmetry restatement (cpp/neuralnet/nninputs.cpp:529-597), single channel-last
 * image: src/dst are [h][w][c]. reverse=0: inputs, reverse=1: outputs. */
void okmx_copy_with_symmetry(const float* src, float* dst, int h_size, int w_size, int c_size,
                             int symmetry, int reverse);

#ifdef __cplusplus
}
#endif
#endif

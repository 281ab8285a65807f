/* TEST INFRASTRUCTURE ONLY — see ggml-common.h in this directory. */
#ifndef KTX_ORACLE_GGML_IMPL_IQK_SHIM_H
#define KTX_ORACLE_GGML_IMPL_IQK_SHIM_H
#include <assert.h>
#include <immintrin.h>
#include "ggml-common.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ggml_type_size / ggml_blck_size / ggml_row_size for the block formats above (ggml.c type_traits table) */
static inline size_t ggml_type_size(enum ggml_type t) {
  switch (t) {
    case GGML_TYPE_F32: return 4; case GGML_TYPE_F16: return 2; case GGML_TYPE_BF16: return 2;
    case GGML_TYPE_Q4_0: return sizeof(block_q4_0); case GGML_TYPE_Q4_1: return sizeof(block_q4_1);
    case GGML_TYPE_Q5_0: return sizeof(block_q5_0); case GGML_TYPE_Q5_1: return sizeof(block_q5_1);
    case GGML_TYPE_Q8_0: return sizeof(block_q8_0); case GGML_TYPE_Q8_1: return sizeof(block_q8_1);
    case GGML_TYPE_Q2_K: return sizeof(block_q2_K); case GGML_TYPE_Q3_K: return sizeof(block_q3_K);
    case GGML_TYPE_Q4_K: return sizeof(block_q4_K); case GGML_TYPE_Q5_K: return sizeof(block_q5_K);
    case GGML_TYPE_Q6_K: return sizeof(block_q6_K); case GGML_TYPE_Q8_K: return sizeof(block_q8_K);
    case GGML_TYPE_IQ2_XXS: return sizeof(block_iq2_xxs); case GGML_TYPE_IQ2_XS: return sizeof(block_iq2_xs);
    case GGML_TYPE_IQ2_S: return sizeof(block_iq2_s); case GGML_TYPE_IQ3_XXS: return sizeof(block_iq3_xxs);
    case GGML_TYPE_IQ3_S: return sizeof(block_iq3_s); case GGML_TYPE_IQ1_S: return sizeof(block_iq1_s);
    case GGML_TYPE_IQ4_NL: return sizeof(block_iq4_nl); case GGML_TYPE_IQ4_XS: return sizeof(block_iq4_xs);
    default: return 0;
  }
}
static inline int64_t ggml_blck_size(enum ggml_type t) {
  switch (t) {
    case GGML_TYPE_F32: case GGML_TYPE_F16: case GGML_TYPE_BF16: return 1;
    case GGML_TYPE_Q4_0: case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_0: case GGML_TYPE_Q5_1: case GGML_TYPE_Q8_0:
    case GGML_TYPE_Q8_1: case GGML_TYPE_IQ4_NL: return 32;
    default: return QK_K;
  }
}
static inline size_t ggml_row_size(enum ggml_type t, int64_t ne) { return ggml_type_size(t) * (size_t)(ne / ggml_blck_size(t)); }
#ifdef __cplusplus
}
#endif
#endif

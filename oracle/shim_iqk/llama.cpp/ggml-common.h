/* TEST INFRASTRUCTURE ONLY — stand-in for llama.cpp's ggml-common.h (the submodule third_party/llama.cpp is empty in the
 * reference checkout), so that the reference's OWN third_party/llamafile/iqk_mul_mat.inc compiles, unmodified and where it
 * lies, into oracle/_ref/libiqk_ref.so (oracle/Makefile).  Block structs are the public GGUF on-disk formats.  The i-quant
 * codebooks other than IQ1_S (which iqk_mul_mat.inc carries itself) are NOT reproduced: they are zero-filled here because
 * the oracle only ever drives Q4_K, Q6_K and IQ1_S through this library; calling it with another type is meaningless. */
#ifndef KTX_ORACLE_GGML_COMMON_SHIM_H
#define KTX_ORACLE_GGML_COMMON_SHIM_H
#include "../../shim/ggml.h"
typedef uint32_t ggml_half2;
#define QK_K 256
#define K_SCALE_SIZE 12
#define QK4_0 32
#define QK4_1 32
#define QK5_0 32
#define QK5_1 32
#define QK8_0 32
#define QK8_1 32
#define QK4_NL 32
#define KTX_DM(a, b, u) union { struct { ggml_half a; ggml_half b; }; ggml_half2 u; }
typedef struct { ggml_half d; uint8_t qs[QK4_0 / 2]; } block_q4_0;
typedef struct { KTX_DM(d, m, dm); uint8_t qs[QK4_1 / 2]; } block_q4_1;
typedef struct { ggml_half d; uint8_t qh[4]; uint8_t qs[QK5_0 / 2]; } block_q5_0;
typedef struct { KTX_DM(d, m, dm); uint8_t qh[4]; uint8_t qs[QK5_1 / 2]; } block_q5_1;
typedef struct { ggml_half d; int8_t qs[QK8_0]; } block_q8_0;
typedef struct { KTX_DM(d, s, ds); int8_t qs[QK8_1]; } block_q8_1;
typedef struct { uint8_t scales[QK_K / 16]; uint8_t qs[QK_K / 4]; KTX_DM(d, dmin, dm); } block_q2_K;
typedef struct { uint8_t hmask[QK_K / 8]; uint8_t qs[QK_K / 4]; uint8_t scales[12]; ggml_half d; } block_q3_K;
typedef struct { KTX_DM(d, dmin, dm); uint8_t scales[K_SCALE_SIZE]; uint8_t qs[QK_K / 2]; } block_q4_K;
typedef struct { KTX_DM(d, dmin, dm); uint8_t scales[K_SCALE_SIZE]; uint8_t qh[QK_K / 8]; uint8_t qs[QK_K / 2]; } block_q5_K;
typedef struct { uint8_t ql[QK_K / 2]; uint8_t qh[QK_K / 4]; int8_t scales[QK_K / 16]; ggml_half d; } block_q6_K;
typedef struct { float d; int8_t qs[QK_K]; int16_t bsums[QK_K / 16]; } block_q8_K;
typedef struct { ggml_half d; uint16_t qs[QK_K / 8]; } block_iq2_xxs;
typedef struct { ggml_half d; uint16_t qs[QK_K / 8]; uint8_t scales[QK_K / 32]; } block_iq2_xs;
typedef struct { ggml_half d; uint8_t qs[QK_K / 4]; uint8_t qh[QK_K / 32]; uint8_t scales[QK_K / 32]; } block_iq2_s;
typedef struct { ggml_half d; uint8_t qs[3 * QK_K / 8]; } block_iq3_xxs;
typedef struct { ggml_half d; uint8_t qs[QK_K / 4]; uint8_t qh[QK_K / 32]; uint8_t signs[QK_K / 8]; uint8_t scales[QK_K / 64]; } block_iq3_s;
typedef struct { ggml_half d; uint8_t qs[QK_K / 8]; uint16_t qh[QK_K / 32]; } block_iq1_s;
typedef struct { ggml_half d; uint8_t qs[QK4_NL / 2]; } block_iq4_nl;
typedef struct { ggml_half d; uint16_t scales_h; uint8_t scales_l[QK_K / 64]; uint8_t qs[QK_K / 2]; } block_iq4_xs;

#endif /* KTX_ORACLE_GGML_COMMON_SHIM_H */

#if defined(GGML_COMMON_IMPL_C) && !defined(KTX_ORACLE_GGML_COMMON_IMPL_DONE)
#define KTX_ORACLE_GGML_COMMON_IMPL_DONE
/* zero-filled: see the header comment (types never driven by the oracle) */
static const uint64_t iq2xxs_grid[256] = {0};
static const uint64_t iq2xs_grid[512] = {0};
static const uint64_t iq2s_grid[1024] = {0};
static const uint32_t iq3xxs_grid[256] = {0};
static const uint32_t iq3s_grid[512] = {0};
static const uint8_t ksigns_iq2xs[128] = {0};
static const uint8_t kmask_iq2xs[8] = {1, 2, 4, 8, 16, 32, 64, 128};
static const int8_t kvalues_iq4nl[16] = {-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113};
#endif

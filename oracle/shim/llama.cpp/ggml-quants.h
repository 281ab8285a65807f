/* TEST INFRASTRUCTURE ONLY — see ../ggml.h.  Block structs named by la/amx.hpp's dispatch macro
 * (kt-kernel/operators/amx/la/amx.hpp:103-139); layouts per the public ggml-common.h. */
#ifndef KTX_ORACLE_GGML_QUANTS_SHIM_H
#define KTX_ORACLE_GGML_QUANTS_SHIM_H
#include "../ggml.h"
#define QK4_0 32
#define QK8_0 32
#define QK_K 256
#define K_SCALE_SIZE 12
typedef struct { ggml_fp16_t d; uint8_t qs[QK4_0 / 2]; } block_q4_0;
typedef struct { ggml_fp16_t d; int8_t qs[QK8_0]; } block_q8_0;
typedef struct { ggml_fp16_t d; ggml_fp16_t dmin; uint8_t scales[K_SCALE_SIZE]; uint8_t qs[QK_K / 2]; } block_q4_K;
typedef struct { float d; int8_t qs[QK_K]; int16_t bsums[QK_K / 16]; } block_q8_K;
#endif

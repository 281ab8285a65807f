/* TEST INFRASTRUCTURE ONLY — stand-in for the un-vendored llama.cpp ggml headers.
 *
 * The reference (kt-kernel/CMakeLists.txt:477, operators/common.hpp:7) pulls ggml in only for a few
 * type names, enum values and bf16/fp16 converters on the AMX/AVX512 MoE path.  This header supplies
 * exactly those so that the reference's own *unmodified* sources under /root/reference compile into
 * oracle/_ref (see oracle/Makefile).  Nothing here is product code and nothing in ktransformers_amd/
 * includes it.  Written from the public ggml type definitions (ggml.h / ggml-common.h semantics):
 * bf16 = upper 16 bits of an IEEE fp32, round-to-nearest-even on conversion, NaN quieted.
 */
#ifndef KTX_ORACLE_GGML_SHIM_H
#define KTX_ORACLE_GGML_SHIM_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t ggml_fp16_t;
typedef uint16_t ggml_half;
typedef struct { uint16_t bits; } ggml_bf16_t;

enum ggml_type {
  GGML_TYPE_F32 = 0, GGML_TYPE_F16 = 1, GGML_TYPE_Q4_0 = 2, GGML_TYPE_Q4_1 = 3,
  GGML_TYPE_Q5_0 = 6, GGML_TYPE_Q5_1 = 7, GGML_TYPE_Q8_0 = 8, GGML_TYPE_Q8_1 = 9,
  GGML_TYPE_Q2_K = 10, GGML_TYPE_Q3_K = 11, GGML_TYPE_Q4_K = 12, GGML_TYPE_Q5_K = 13,
  GGML_TYPE_Q6_K = 14, GGML_TYPE_Q8_K = 15, GGML_TYPE_IQ2_XXS = 16, GGML_TYPE_IQ2_XS = 17,
  GGML_TYPE_IQ3_XXS = 18, GGML_TYPE_IQ1_S = 19, GGML_TYPE_IQ4_NL = 20, GGML_TYPE_IQ3_S = 21,
  GGML_TYPE_IQ2_S = 22, GGML_TYPE_IQ4_XS = 23, GGML_TYPE_I8 = 24, GGML_TYPE_I16 = 25,
  GGML_TYPE_I32 = 26, GGML_TYPE_I64 = 27, GGML_TYPE_F64 = 28, GGML_TYPE_IQ1_M = 29,
  GGML_TYPE_BF16 = 30, GGML_TYPE_COUNT = 39,
};

static inline float ggml_bf16_to_fp32(ggml_bf16_t h) {
  uint32_t u = (uint32_t)h.bits << 16; float f; memcpy(&f, &u, 4); return f;
}
static inline ggml_bf16_t ggml_fp32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4); ggml_bf16_t h;
  if ((u & 0x7fffffffu) > 0x7f800000u) { h.bits = (uint16_t)((u >> 16) | 64); return h; }
  h.bits = (uint16_t)((u + (0x7fffu + ((u >> 16) & 1u))) >> 16); return h;
}
#define GGML_BF16_TO_FP32(x) ggml_bf16_to_fp32(x)
#define GGML_FP32_TO_BF16(x) ggml_fp32_to_bf16(x)

static inline float ggml_fp16_to_fp32(ggml_fp16_t h) {
  uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
  if (e == 0) {
    if (m == 0) u = s;
    else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; ++sh; } m &= 0x3ffu; u = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112u) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}
static inline ggml_fp16_t ggml_fp32_to_fp16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  uint32_t s = (u >> 16) & 0x8000u; int32_t e = (int32_t)((u >> 23) & 0xffu) - 127 + 15; uint32_t m = u & 0x7fffffu;
  if (((u >> 23) & 0xffu) == 0xffu) return (ggml_fp16_t)(s | 0x7c00u | (m ? 0x200u : 0));
  if (e >= 31) return (ggml_fp16_t)(s | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (ggml_fp16_t)s;
    m |= 0x800000u; uint32_t shift = (uint32_t)(14 - e);
    uint32_t r = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) ++r;
    return (ggml_fp16_t)(s | r);
  }
  uint32_t r = ((uint32_t)e << 10) | (m >> 13), rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
  return (ggml_fp16_t)(s | r);
}
#define GGML_FP16_TO_FP32(x) ggml_fp16_to_fp32(x)
#define GGML_FP32_TO_FP16(x) ggml_fp32_to_fp16(x)
#define GGML_COMPUTE_FP16_TO_FP32(x) ggml_fp16_to_fp32(x)
#define GGML_COMPUTE_FP32_TO_FP16(x) ggml_fp32_to_fp16(x)

static inline void ggml_bf16_to_fp32_row(const ggml_bf16_t* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = ggml_bf16_to_fp32(x[i]); }
static inline void ggml_fp32_to_bf16_row(const float* x, ggml_bf16_t* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = ggml_fp32_to_bf16(x[i]); }
static inline void ggml_fp16_to_fp32_row(const ggml_fp16_t* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = ggml_fp16_to_fp32(x[i]); }
static inline void ggml_fp32_to_fp16_row(const float* x, ggml_fp16_t* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = ggml_fp32_to_fp16(x[i]); }

struct ggml_init_params { size_t mem_size; void* mem_buffer; bool no_alloc; };
struct ggml_context;
static inline struct ggml_context* ggml_init(struct ggml_init_params p) { (void)p; return (struct ggml_context*)(uintptr_t)1; }

#ifndef MAX
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#endif
#ifndef MIN
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#endif
#define GGML_ASSERT(x) do { if (!(x)) { __builtin_trap(); } } while (0)
#define GGML_UNUSED(x) (void)(x)
#define GGML_RESTRICT __restrict__

#ifdef __cplusplus
}
#endif
#endif

// ktx_common.h — shared device/host helpers for the gfx950 kernels.
#ifndef KTX_COMMON_H
#define KTX_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef uint16_t bf16_t;  // raw bf16 bits

#define KTX_WAVE 64

// ---- error plumbing (host) ------------------------------------------------------------------------------------
std::string& ktx_err_slot();
int ktx_fail(const std::string& msg);
#define KTX_HIP(call)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return ktx_fail(std::string(#call) + ": " + hipGetErrorString(e_));                               \
  } while (0)
#define KTX_REQUIRE(cond, msg)                                                                          \
  do {                                                                                                  \
    if (!(cond)) return ktx_fail(std::string(msg));                                                     \
  } while (0)

// ---- bf16 <-> fp32 (device + host) ----------------------------------------------------------------------------
// Reference: kt-kernel/operators/amx/la/utils.hpp:14-52.  fp32->bf16 is round-to-nearest-even with input denormals
// and denormal results flushed to (signed) zero and NaN quieted — the behaviour of VCVTNE2PS2BF16, which is what
// the reference executes on AVX512-BF16 hosts (oracle/ktx_oracle.c states the same rule).
__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t h) {
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)h << 16;
  return c.f;
}
__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7f800000u) == 0) return (bf16_t)((u >> 16) & 0x8000u);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x0040u);
  return (bf16_t)((u + (0x7fffu + ((u >> 16) & 1u))) >> 16);
}

// _mm512_cvtps_epi32 + _mm512_cvtsepi32_epi8: round-to-nearest-even then signed saturation to int8.
__device__ __forceinline__ int quant_rne_sat8(float v) {
  float r = rintf(v);
  r = fminf(fmaxf(r, -128.0f), 127.0f);
  return (int)r;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

#endif

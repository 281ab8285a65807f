// ktx_common.h — shared device/host helpers for the gfx950 kernels.
#ifndef KTX_COMMON_H
#define KTX_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <unordered_map>

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

// ---- per-launch timing (ktx_prof.hip; measurement aid, off unless bench.py turns it on) -------------------------------
int ktx_timing_mode();
std::string ktx_fmt(const char* fmt, ...);
struct KtxTimeScope {
  KtxTimeScope(hipStream_t st, double bytes, std::string label);
  ~KtxTimeScope();
  hipStream_t st_;
  long idx_;
};
// brackets the rest of the enclosing block: KTX_TIMED(stream, algorithmic_bytes, "kernel<%d> %d->%d", ...)
#define KTX_TIMED(st, bytes, ...) \
  KtxTimeScope _ktx_ts((st), (double)(bytes), ktx_timing_mode() ? ktx_fmt(__VA_ARGS__) : std::string())

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

// ---- torch-semantics bf16 helpers for the glue ops (router, RMSNorm prologues): plain IEEE round-to-nearest-even as
// torch's .to(bfloat16) — one v_cvt_pk_bf16_f32 for two values — and the packed bf16 dot product v_dot2c_f32_bf16
// (d += a.lo*b.lo + a.hi*b.hi).  NOT for the expert path, whose parity contract pins the AVX512 rounding above.
typedef __bf16 ktx_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ktx_pk_bf16(float lo, float hi) {
  ktx_bf2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float ktx_lo_f32(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float ktx_hi_f32(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }
__device__ __forceinline__ float ktx_dot2_bf16(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ktx_bf2, a), __builtin_bit_cast(ktx_bf2, b), acc, false);
}
// DeepseekV3RMSNorm.forward on a packed pair: weight * bf16(x * r), both roundings (modeling_deepseek_v3.py:98-103)
__device__ __forceinline__ uint32_t ktx_norm_pk(uint32_t x, float r, uint32_t w) {
  const uint32_t h = ktx_pk_bf16(ktx_lo_f32(x) * r, ktx_hi_f32(x) * r);
  return ktx_pk_bf16(ktx_lo_f32(h) * ktx_lo_f32(w), ktx_hi_f32(h) * ktx_hi_f32(w));
}

// _mm512_cvtps_epi32 + _mm512_cvtsepi32_epi8: round-to-nearest-even then signed saturation to int8.
__device__ __forceinline__ int quant_rne_sat8(float v) {
  float r = rintf(v);
  r = fminf(fmaxf(r, -128.0f), 127.0f);
  return (int)r;
}

// hipFuncSetAttribute acts on the CURRENT device's copy of a kernel, so "set once" is once per (call site, device): a process that
// splits a model's layers over several GPUs would otherwise launch with the default dynamic-LDS limit on every device but the first
// (ADVICE r4).  hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device); the flag is set only after the call succeeded
// (a failed first attempt must not disable every later one), under a mutex.
static inline hipError_t ktx_set_max_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::unordered_map<const void*, uint64_t> done;   // kernel -> devices it has been set on
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
  if (known) {
    std::lock_guard<std::mutex> lk(mu);
    if ((done[kernel] >> dev) & 1u) return hipSuccess;
  }
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (known && e == hipSuccess) {
    std::lock_guard<std::mutex> lk(mu);
    done[kernel] |= 1ull << dev;
  }
  return e;
}

// ---- wave-wide reductions on DPP + v_readlane ------------------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute (an LDS crossbar round trip, ~60 cycles); in the latency-bound decode kernels a single
// wavefront walks chains of such reductions, so they are done with DPP row operations (quad_perm, row_half_mirror,
// row_mirror: all-reduce inside each 16-lane row) and four v_readlane for the rows.
template <int CTRL>
__device__ __forceinline__ int ktx_dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float ktx_dpp_f(float v) { return __int_as_float(ktx_dpp_i<CTRL>(__float_as_int(v))); }
#define KTX_DPP_QUAD_1032 0xB1
#define KTX_DPP_QUAD_2301 0x4E
#define KTX_DPP_ROW_HALF_MIRROR 0x141
#define KTX_DPP_ROW_MIRROR 0x140

// all-reduce inside each row of 16 lanes (the MFMA column group): every lane gets its row's result
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, ktx_dpp_f<KTX_DPP_QUAD_1032>(v));
  v = fmaxf(v, ktx_dpp_f<KTX_DPP_QUAD_2301>(v));
  v = fmaxf(v, ktx_dpp_f<KTX_DPP_ROW_HALF_MIRROR>(v));
  return fmaxf(v, ktx_dpp_f<KTX_DPP_ROW_MIRROR>(v));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += ktx_dpp_f<KTX_DPP_QUAD_1032>(v);
  v += ktx_dpp_f<KTX_DPP_QUAD_2301>(v);
  v += ktx_dpp_f<KTX_DPP_ROW_HALF_MIRROR>(v);
  return v + ktx_dpp_f<KTX_DPP_ROW_MIRROR>(v);
}
// Four independent row all-reductions at once, as VOP2 instructions that carry the DPP permutation themselves (v = op(perm(v), v)).
// From C the compiler emits v_mov_b32_dpp + a canonicalising v_max + the operation + s_nop per step (the IEEE-mode quieting of a value
// it cannot see through the permutation keeps it from folding the DPP into the operation): 14-16 instructions per reduction where this
// takes 4.  The four values of a step are independent, so the >= 2 wait states a DPP read needs behind the VALU write of its source are
// filled by the other three; the leading s_nop covers whatever the compiler issued last (VALU write of an input: 2 wait states, EXEC
// write: 5) — the hazard recogniser does not look inside an asm — and the trailing one a DPP read the compiler may issue next.
// Same values, same order of operations as row16_max / row16_sum (max and add commute): same bits.
#define KTX_DPP4_STEP(OP, CTRL)                                                      \
  OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                            \
  OP " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                            \
  OP " %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                            \
  OP " %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define KTX_DPP4_REDUCE(OP)                                                          \
  "s_nop 4\n\t" KTX_DPP4_STEP(OP, "quad_perm:[1,0,3,2]") KTX_DPP4_STEP(OP, "quad_perm:[2,3,0,1]") \
  KTX_DPP4_STEP(OP, "row_half_mirror") KTX_DPP4_STEP(OP, "row_mirror") "s_nop 1"
__device__ __forceinline__ void row16_max4(float& a, float& b, float& c, float& d) {
  asm volatile(KTX_DPP4_REDUCE("v_max_f32_dpp") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void row16_sum4(float& a, float& b, float& c, float& d) {
  asm volatile(KTX_DPP4_REDUCE("v_add_f32_dpp") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ float wave_max(float v) {
  const int b = __float_as_int(row16_max(v));
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// fixed summation tree: pairs, quads, eights, rows, then rows 0..3 left to right
__device__ __forceinline__ float wave_sum(float v) {
  const int b = __float_as_int(row16_sum(v));
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return ((r0 + r1) + r2) + r3;
}

#endif

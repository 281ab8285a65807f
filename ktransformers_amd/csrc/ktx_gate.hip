// ktx_gate.hip — MoE router for gfx950.  C ABI in include/ktx_gate.h.
//
// Restates MoEGate.forward (archive/ktransformers/models/modeling_deepseek_v3.py:430-481 and the V2 variant
// archive/ktransformers/models/modeling_deepseek.py:413-455) as two launches: an HBM/L2-bound fp32 GEMV per (token,
// expert) and a one-wavefront-per-token selection kernel that keeps all E scores in registers.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ktx_gate.h"
#include "ktx_common.h"

#include "ktx_gate_dev.inc"

// ---- logits: one wavefront per (token, expert); 16-byte loads of both bf16 rows, fp32 FMA, butterfly reduce --------
__global__ __launch_bounds__(256) void gate_logits_kernel(const int32_t* d_bsz, int qlen, int E, int H,
                                                          const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          float* __restrict__ logits) {
  int T = qlen;
  if (d_bsz) T = min(max(*d_bsz, 0), qlen);
  const int t = blockIdx.y;
  if (t >= T) return;
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  const bf16_t* xr = x + (size_t)t * H;
  const bf16_t* wr = w + (size_t)e * H;
  float acc = 0.0f;
  for (int j = lane * 8; j < H; j += 512) {
    const uint4 a = *reinterpret_cast<const uint4*>(xr + j);
    const uint4 b = *reinterpret_cast<const uint4*>(wr + j);
    const uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int q = 0; q < 4; q++) acc = ktx_dot2_bf16(av[q], bv[q], acc);   // v_dot2c_f32_bf16: two products per instruction
  }
  acc = wave_sum(acc);   // same order and tree as gate_fused_kernel: the two paths give identical logits
  if (lane == 0) logits[(size_t)t * E + e] = acc;
}

template <int EPL>
__global__ __launch_bounds__(64) void gate_select_kernel(ktx_gate_config c, const int32_t* d_bsz, int qlen,
                                                         const float* __restrict__ logits, const float* __restrict__ bias,
                                                         int64_t* __restrict__ topk_idx, float* __restrict__ topk_w) {
  int T = qlen;
  if (d_bsz) T = min(max(*d_bsz, 0), qlen);
  const int t = blockIdx.x;
  if (t >= T) return;
  gate_select_token<EPL>(c, t, threadIdx.x, logits + (size_t)t * c.n_routed_experts, bias, topk_idx, topk_w);
}

template <int EPL, int NJ>
__global__ __launch_bounds__(256) void gate_fused_kernel(GateArgs ga) {
  extern __shared__ __attribute__((aligned(16))) uint8_t gate_smem_dyn[];   // NJ > 0: the normalised row, bf16 [H]
  gate_fused_body<EPL, NJ, 4>(ga, blockIdx.x, gridDim.x, blockIdx.y, gate_smem_dyn);
}

extern "C" int ktx_gate_logits(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x,
                               const void* d_w, float* d_logits, void* stream) {
  KTX_REQUIRE(cfg && d_x && d_w && d_logits && qlen > 0, "ktx_gate_logits: bad argument");
  KTX_REQUIRE(cfg->hidden_size % 8 == 0, "ktx_gate_logits: hidden_size must be a multiple of 8");
  const dim3 grid((cfg->n_routed_experts + 3) / 4, qlen);
  hipLaunchKernelGGL(gate_logits_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_bsz, qlen, cfg->n_routed_experts,
                     cfg->hidden_size, (const bf16_t*)d_x, (const bf16_t*)d_w, d_logits);
  KTX_HIP(hipGetLastError());
  return 0;
}

static int gate_forward_impl(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x, const void* d_w,
                             const float* d_bias, float* d_logits, int32_t* d_counters, int64_t* d_topk_idx,
                             float* d_topk_weight, const void* d_norm_w, float norm_eps, void* d_xn_out, void* stream) {
  KTX_REQUIRE(cfg && d_x && d_w && d_logits && d_counters && d_topk_idx && d_topk_weight && qlen > 0, "ktx_gate_forward: bad argument");
  const int E = cfg->n_routed_experts;
  KTX_REQUIRE(cfg->hidden_size % 8 == 0, "ktx_gate_forward: hidden_size must be a multiple of 8");
  KTX_REQUIRE(E > 0 && E <= KTX_GATE_MAX_E, "ktx_gate_forward: n_routed_experts out of range (1..1024)");
  KTX_REQUIRE(cfg->top_k > 0 && cfg->top_k <= 64 && cfg->top_k <= E, "ktx_gate_forward: top_k out of range");
  KTX_REQUIRE(cfg->n_group >= 1 && cfg->n_group <= 64 && E % cfg->n_group == 0, "ktx_gate_forward: bad n_group");
  KTX_REQUIRE(cfg->topk_group >= 1 && cfg->topk_group <= cfg->n_group, "ktx_gate_forward: bad topk_group");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((E + 3) / 4, qlen);
  const int epl = (E + 63) / 64;
  GateArgs ga{};
  ga.c = *cfg; ga.d_bsz = d_bsz; ga.qlen = qlen; ga.x = (const bf16_t*)d_x; ga.w = (const bf16_t*)d_w; ga.bias = d_bias;
  ga.logits = d_logits; ga.counters = d_counters; ga.topk_idx = d_topk_idx; ga.topk_w = d_topk_weight;
  ga.norm_w = (const bf16_t*)d_norm_w; ga.norm_eps = norm_eps; ga.xn_out = (bf16_t*)d_xn_out;
#define KTX_FUSED2(N, J) hipLaunchKernelGGL((gate_fused_kernel<N, J>), grid, dim3(256), (J) > 0 ? (size_t)cfg->hidden_size * 2 : 0, st, ga)
#define KTX_FUSED(N)                                                          \
  do {                                                                        \
    if (!d_norm_w) KTX_FUSED2(N, 0);                                          \
    else if (cfg->hidden_size <= 2048) KTX_FUSED2(N, 4);                      \
    else if (cfg->hidden_size <= 4096) KTX_FUSED2(N, 8);                      \
    else KTX_FUSED2(N, 16);                                                   \
  } while (0)
  KTX_REQUIRE(!d_norm_w || cfg->hidden_size <= 8192, "ktx_gate_forward_norm: the fused RMSNorm needs hidden_size <= 8192");
  KTX_TIMED(st, (double)E * cfg->hidden_size * 2.0 + (double)qlen * cfg->hidden_size * (d_norm_w ? 6.0 : 2.0) + E * 8.0,
            "gate_fused_kernel%s T=%d E=%d H=%d", d_norm_w ? "+rmsnorm" : "", qlen, E, cfg->hidden_size);
  if (epl <= 1) KTX_FUSED(1);
  else if (epl <= 2) KTX_FUSED(2);
  else if (epl <= 4) KTX_FUSED(4);
  else if (epl <= 6) KTX_FUSED(6);
  else if (epl <= 8) KTX_FUSED(8);
  else KTX_FUSED(16);
#undef KTX_FUSED2
#undef KTX_FUSED
  KTX_HIP(hipGetLastError());
  return 0;
}

extern "C" int ktx_gate_forward(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x, const void* d_w,
                                const float* d_bias, float* d_logits, int32_t* d_counters, int64_t* d_topk_idx,
                                float* d_topk_weight, void* stream) {
  return gate_forward_impl(cfg, d_bsz, qlen, d_x, d_w, d_bias, d_logits, d_counters, d_topk_idx, d_topk_weight, nullptr, 0.f,
                           nullptr, stream);
}

extern "C" int ktx_gate_forward_norm(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x,
                                     const void* d_norm_weight, float norm_eps, void* d_xn_out, const void* d_w,
                                     const float* d_bias, float* d_logits, int32_t* d_counters, int64_t* d_topk_idx,
                                     float* d_topk_weight, void* stream) {
  KTX_REQUIRE(d_norm_weight && d_xn_out, "ktx_gate_forward_norm: null norm weight / output");
  return gate_forward_impl(cfg, d_bsz, qlen, d_x, d_w, d_bias, d_logits, d_counters, d_topk_idx, d_topk_weight, d_norm_weight,
                           norm_eps, d_xn_out, stream);
}

extern "C" int ktx_gate_select(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const float* d_logits,
                               const float* d_bias, int64_t* d_topk_idx, float* d_topk_weight, void* stream) {
  KTX_REQUIRE(cfg && d_logits && d_topk_idx && d_topk_weight && qlen > 0, "ktx_gate_select: bad argument");
  const int E = cfg->n_routed_experts;
  KTX_REQUIRE(E > 0 && E <= KTX_GATE_MAX_E, "ktx_gate_select: n_routed_experts out of range (1..1024)");
  KTX_REQUIRE(cfg->top_k > 0 && cfg->top_k <= 64 && cfg->top_k <= E, "ktx_gate_select: top_k out of range");
  KTX_REQUIRE(cfg->n_group >= 1 && cfg->n_group <= 64 && E % cfg->n_group == 0, "ktx_gate_select: bad n_group");
  KTX_REQUIRE(cfg->topk_group >= 1 && cfg->topk_group <= cfg->n_group, "ktx_gate_select: bad topk_group");
  hipStream_t st = (hipStream_t)stream;
  const int epl = (E + 63) / 64;
#define KTX_SEL(N) hipLaunchKernelGGL(gate_select_kernel<N>, dim3(qlen), dim3(64), 0, st, *cfg, d_bsz, qlen, d_logits, d_bias, d_topk_idx, d_topk_weight)
  if (epl <= 1) KTX_SEL(1);
  else if (epl <= 2) KTX_SEL(2);
  else if (epl <= 4) KTX_SEL(4);
  else if (epl <= 6) KTX_SEL(6);
  else if (epl <= 8) KTX_SEL(8);
  else KTX_SEL(16);
#undef KTX_SEL
  KTX_HIP(hipGetLastError());
  return 0;
}

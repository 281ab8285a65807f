/* ktx_linear.h — C ABI of the quantised dense linears around the MoE/MLA hot path (libktx_hip.so).
 *
 * Replaces what the reference's injected linear operators call:
 *
 *   reference (NVIDIA GPU)                                               this library (gfx950)
 *   -------------------------------------------------------------------  -----------------------------------------
 *   KLinearMarlin.load: marlin_quantize(w, 4, 64, act_order=False)       ktx_linear_load_bf16 (format W4) — the
 *     archive/ktransformers/operators/linear.py:633-677                    quantiser of quant_utils.py:36-98 restated
 *     .../custom_marlin/quantize/utils/{marlin_utils.py:79-114,            on the GPU (same q and bf16 scales), or
 *       quant_utils.py:36-98}                                              ktx_linear_load_w4 (pre-quantised q + s)
 *   KLinearMarlin.forward: KTransformersOps.gptq_marlin_gemm(...)        ktx_linear_forward
 *     linear.py:679-711 (un-vendored CUDA; W4A16, fp32 accumulate)
 *   KLinearFP8.load / forward: act_quant(x,128) + fp8_gemm(...)          ktx_linear_load_fp8 / ktx_linear_forward
 *     linear.py:408-429, ktransformers_ext/triton/fp8gemm.py:10-193
 *   KLinearTorch.forward: x @ W (+ bias)    linear.py:173-183            ktx_linear_load_bf16 (format BF16) / forward
 *
 * Conventions as in ktx_moe.h: DEVICE pointers, kernels are only enqueued on `stream` (graph capturable, no
 * allocation inside forward), 0 = success, ktx_last_error() holds the message.  Activations and outputs are bf16
 * row-major [T][in_features] / [T][out_features].  The reference pads Marlin shapes to K%128, N%64 with zero weights
 * (linear.py:622-630, 655-658); here any in_features%8==0 and any out_features work and the padding is internal.
 */
#ifndef KTX_LINEAR_H
#define KTX_LINEAR_H
#include <stddef.h>
#include <stdint.h>
#include "ktx_gate.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ktx_linear_s* ktx_linear_t;
#ifndef KTX_STREAM_T_DEFINED
#define KTX_STREAM_T_DEFINED
typedef void* ktx_stream_t; /* hipStream_t */
#endif
const char* ktx_last_error(void);

enum ktx_linear_format {
  KTX_LIN_BF16 = 0, /* dense bf16 weights (KLinearTorch) */
  KTX_LIN_W4 = 1,   /* Marlin/GPTQ symmetric uint4, zero point 8, bf16 scale per (group of `group_size` inputs, output) */
  KTX_LIN_FP8 = 2,  /* e4m3 weights + fp32 scale_inv per 128x128 block; activations quantised to e4m3 per 128 inputs */
  KTX_LIN_W8 = 3,   /* Marlin/GPTQ symmetric uint8, zero point 128, bf16 scale per (group of `group_size` inputs, output): one byte per
                       weight in HBM; the kernels multiply bf16 activations with Marlin's own multiplicand bf16((q - 128) * s),
                       formed in registers (KLinearMarlin num_bits = 8, archive/ktransformers/operators/linear.py:608-666) */
};

typedef struct ktx_linear_config {
  int32_t in_features;  /* K (KLinearBase.in_features, linear.py:76-84) */
  int32_t out_features; /* N */
  int32_t format;       /* enum ktx_linear_format */
  int32_t group_size;   /* W4: 32, 64 (KLinearMarlin default, linear.py:607) or 128; FP8: 128; BF16: 0 */
  int32_t max_len;      /* largest T of one forward */
  int32_t device;       /* HIP device ordinal */
  int32_t batch;        /* 0/1: one matrix; B > 1: B independent [N][K] matrices (per-head absorb bmm, attention.py:414-418,467) */
} ktx_linear_config;

int ktx_linear_create(const ktx_linear_config* cfg, ktx_linear_t* out);
int ktx_linear_destroy(ktx_linear_t h);

/* d_w: bf16 [batch][out_features][in_features] (the nn.Linear weight; KLinearMarlin transposes it itself, linear.py:645),
 * d_bias: bf16 [out_features] or NULL.  BF16 handles keep the weights (re-tiled); W4 handles quantise on the GPU with
 * the arithmetic of quantize_weights (quant_utils.py:61-67) evaluated as torch evaluates it on bf16 tensors:
 * s = bf16(max|w| * (2/15)), q = clamp(rint(bf16(w / s)) + 8, 0, 15).  Synchronous. */
int ktx_linear_load_bf16(ktx_linear_t h, const void* d_w, const void* d_bias);

/* Pre-quantised Marlin-semantics weights, DEVICE pointers: d_q uint8 [in_features][out_features] with values 0..15
 * (quantize_weights' q_w, before the Marlin tile permutation), d_s bf16 [in_features/group_size][out_features]
 * (its `s`).  Synchronous. */
int ktx_linear_load_w4(ktx_linear_t h, const uint8_t* d_q, const void* d_s, const void* d_bias);

/* DeepSeek block-fp8: d_w e4m3 bytes [out_features][in_features], d_scale_inv fp32 [ceil(N/128)][ceil(K/128)]
 * (`weight`, `weight_scale_inv` of the checkpoint; KLinearFP8.load, linear.py:416-429).  in_features % 128 == 0.
 * Synchronous. */
int ktx_linear_load_fp8(ktx_linear_t h, const void* d_w, const float* d_scale_inv, const void* d_bias);

/* y[t] = x[t] · W^T (+ bias) for t < min(T, *d_bsz) (d_bsz may be NULL = T rows); rows beyond are left untouched.
 * Mirrors KLinear*.forward(x, bsz_tensor) (linear.py:174,409,679).
 * Range note (decode GEMVs whose K is split over several workgroups, lin_sk_kernel: the wide projections of a decode step): the parts
 * of an output meet in a 64-bit fixed-point word whose in-range window is |part| <= 2^18 = 262144; a NaN, an Inf or a FINITE part beyond
 * it is carried as a marker and the output becomes NaN (never a silently clamped number).  The un-split decode kernel and the prompt
 * GEMM have no such window: a layer whose fp32 partial sums legitimately exceed 2^18 (bf16 activations of that size are outside every
 * model on the list) gets NaN on the split path and the finite value on the others (stated, not widened — the word's 60
 * sum bits hold 15 biased parts plus the marker with one bit to spare). */
int ktx_linear_forward(ktx_linear_t h, const int32_t* d_bsz, int T, const void* d_x, void* d_y, ktx_stream_t stream);

/* Fusions around one linear of the decoder layer (all optional, NULL = off):
 *   norm_weight/norm_eps : RMSNorm of the input row inside the kernel (input_layernorm / post_attention_layernorm in front
 *                          of the projections, modeling_deepseek_v3.py:1207,1222) — only where the decode kernel runs
 *                          (ktx_linear_decode_eligible(h, T) != 0: T <= 4, or 5..8 rows of an unbatched handle, which run as 4-row
 *                          passes of that kernel — round 6); otherwise call ktx_rmsnorm first.
 *   add1, add2           : bf16 [T][out_features] tensors (row strides add*_ld, 0 = out_features) added to the result in
 *                          this order with torch's bf16 rounding: y = bf16(add2 + bf16(add1 + linear(x))) — the residual
 *                          adds and the routed + shared sum (modeling_deepseek_v3.py:1219,1225,529). */
typedef struct ktx_linear_fusion {
  const void* norm_weight;
  float norm_eps;
  const void* add1;
  int64_t add1_ld;
  const void* add2;
  int64_t add2_ld;
  int64_t x_ld, y_ld; /* row strides of x / y in elements (multiples of 8); 0 = in_features / out_features */
  int32_t glu;        /* the matrix is [gate | up] interleaved 8 rows / 8 rows per 16-row strip: y = act_fn(gate(x)) * up(x),
                         out_features / 2 columns (DeepseekV3MLP, modeling_deepseek_v3.py:396-398) */
  int32_t glu_in;     /* the INPUT rows are [gate | up], 2 * in_features elements (x_ld defaults to that): the linear reads
                         act_fn(gate) * up with ktx_silu_mul's roundings (include/ktx_ops.h) — down_proj of a DeepseekV3MLP whose
                         gate_proj / up_proj ran as one [gate ; up] GEMV (block-fp8 checkpoints).  KTX_LIN_FP8 handles, decode
                         kernel only (ktx_linear_decode_eligible), not together with norm_weight; otherwise call ktx_silu_mul first.
                         (Occupies what was padding after `glu`: the struct's size and the other offsets are unchanged.) */
} ktx_linear_fusion;
int ktx_linear_forward_fused(ktx_linear_t h, const int32_t* d_bsz, int T, const void* d_x, void* d_y,
                             const ktx_linear_fusion* fusion, ktx_stream_t stream);
int ktx_linear_decode_eligible(ktx_linear_t h, int T);

/* ktx_linear_forward_fused with the MoE router riding in the same launch (decode steps, T <= 4).  For the MoE block of a
 * decoder layer (KDeepseekV3MoE.forward, archive/ktransformers/operators/experts.py:974-1012: router, routed experts and the
 * shared experts all read the same post-attention hidden row): `h` is the shared experts' merged gate|up linear, `fusion`
 * carries the post_attention_layernorm (norm_weight must be set) and `glu`; the router arguments are those of
 * ktx_gate_forward_norm (include/ktx_gate.h) on the same d_x with the same norm.  Results are those of the two separate
 * calls — router workgroups and GEMV workgroups are independent and share one grid.  Shapes without a combined kernel
 * (other formats / group sizes, odd k-slices, T > 4, strided x) run the two launches instead; the call never fails for that. */
int ktx_linear_forward_fused_gate(ktx_linear_t h, const int32_t* d_bsz, int T, const void* d_x, void* d_y,
                                  const ktx_linear_fusion* fusion, const struct ktx_gate_config* gate_cfg, const void* d_gate_w,
                                  const float* d_gate_bias, float* d_logits, int32_t* d_counters, int64_t* d_topk_idx,
                                  float* d_topk_weight, void* d_xn_out, ktx_stream_t stream);

/* Batched form for the per-head absorb products of MLA (torch.matmul(q_nope, q_absorb) / matmul(attn, out_absorb.mT),
 * archive/ktransformers/operators/attention.py:414-418,465-468): batch b uses weight matrix b ([batch][N][K] at load)
 * and reads x[t*ldx + b*x_batch_stride + k], writes y[t*ldy + b*y_batch_stride + n] (strides in elements). */
int ktx_linear_forward_batched(ktx_linear_t h, const int32_t* d_bsz, int T, const void* d_x, int64_t ldx,
                               int64_t x_batch_stride, void* d_y, int64_t ldy, int64_t y_batch_stride,
                               ktx_stream_t stream);

/* ktx_linear_forward_batched for a decode step (T <= 4) with ktx_mla_prep (include/ktx_ops.h) riding in the same launch: the
 * q-absorb products need only q_nope, the prep only q_pe / the kv_a row — independent work between the projections and the
 * MLA kernel (attention.py:360-418), one kernel boundary instead of two.  Arguments after y_batch_stride are ktx_mla_prep's. */
int ktx_linear_forward_batched_prep(ktx_linear_t h, int T, const void* d_x, int64_t ldx, int64_t x_batch_stride, void* d_y,
                                    int64_t ldy, int64_t y_batch_stride, int num_heads, int nope_dim, int rope_dim, int kv_lora,
                                    const void* d_q, int64_t q_row_stride, void* d_q_pe_out, const void* d_kv,
                                    int64_t kv_row_stride, const void* d_kv_norm_w, float eps, void* d_ckv_out, void* d_kpe_out,
                                    const int64_t* d_pos, const float* d_inv_freq, float mscale, ktx_stream_t stream);

/* bytes of HBM held for the weights (tiles + scales) — the algorithmic bytes one decode launch streams */
size_t ktx_linear_weight_bytes(ktx_linear_t h);

/* tests only: read back the quantiser's result in the layout of ktx_linear_load_w4 (HOST pointers). */
int ktx_linear_debug_get_w4(ktx_linear_t h, uint8_t* q, uint16_t* s);

/* q_b_proj(q_a_layernorm(q_a)) -> [ q-absorb products of the heads' q_nope | RoPE of the heads' q_pe ], plus the kv half of
 * ktx_mla_prep, in ONE launch for a decode step (T <= 4): everything between the merged q_a|kv_a projection and the MLA kernel
 * (archive/ktransformers/operators/attention.py:360-418: q_b_proj, q_a_layernorm, torch.matmul(q_nope, q_absorb), rotary).
 * One workgroup per head streams the head's q_b rows (W4) and its W_UK block (BF16 batched handle) — both requested up front —
 * so the two GEMVs share one memory round trip instead of being two dependent launches.  Arithmetic and roundings are those
 * of ktx_linear_forward_fused(q_b, norm) / ktx_linear_forward_batched(q_absorb) / ktx_mla_prep.  d_kv may be NULL (no kv half).
 * Ask ktx_linear_qb_absorb_eligible first: the combined kernel exists for W4 g64 q_b with q_lora_rank 1536, nope 128,
 * rope <= 64, kv_lora 512 (DeepSeek-V3 / R1, Kimi-K2, DeepSeek-V2); everything else keeps the separate calls. */
int ktx_linear_qb_absorb_eligible(ktx_linear_t q_b, ktx_linear_t q_absorb, int T, int num_heads, int nope_dim, int rope_dim,
                                  int kv_lora);
int ktx_linear_forward_qb_absorb(ktx_linear_t q_b, ktx_linear_t q_absorb, int T, const void* d_q_a, int64_t q_a_row_stride,
                                 const void* d_q_a_norm_w, float q_a_norm_eps, int num_heads, int nope_dim, int rope_dim,
                                 int kv_lora, void* d_q_nope_out, void* d_q_pe_out, const void* d_kv, int64_t kv_row_stride,
                                 const void* d_kv_norm_w, float kv_norm_eps, void* d_ckv_out, void* d_kpe_out,
                                 const int64_t* d_pos, const float* d_inv_freq, float mscale, ktx_stream_t stream);

/* W4 handle -> row-major bf16 weights [out_features][ld_out] with Marlin's in-register rounding w = bf16((q - 8) * s) — what
 * gptq_marlin_gemm multiplies bf16 activations with (custom_marlin/gptq_marlin).  For prompt-sized calls (hundreds of tokens)
 * the operator de-quantises into a scratch buffer with this call and runs a plain library GEMM on it (F.linear -> hipBLASLt):
 * the weight bytes are expanded once per call instead of once per 64-token tile, and the arithmetic is the reference's. */
/* KTX_LIN_W8 handles: q uint8 [in][out] in 0..255, s bf16 [in / group][out] — the (q_w, s) of quantize_weights(w, 8, group)
 * (custom_marlin/quantize/utils/quant_utils.py:36-98), the orientation of ktx_linear_load_w4. */
int ktx_linear_load_w8(ktx_linear_t h, const uint8_t* d_q, const void* d_s, const void* d_bias);

int ktx_linear_dequant_bf16(ktx_linear_t h, void* d_out, int64_t ld_out, ktx_stream_t stream);

/* Prompt-sized calls of an FP8 handle (KLinearFP8.forward, archive/ktransformers/operators/linear.py:388-436: act_quant +
 * fp8_gemm of ktransformers_ext/triton/fp8gemm.py) in two launches:
 *   ktx_fp8_act_quant  — act_quant (fp8gemm.py:10-31) of x bf16 [T][ldx]: d_q = e4m3 bytes [T][K]; d_scales_t[kb * s_ld + t] =
 *                        amax(x[t][128 kb .. 128 kb + 127]) / 448 (block-major, so a tile's scales of one k-step are contiguous);
 *   ktx_linear_gemm_fp8 — y bf16 [T][ldy] = fp8_gemm(d_q, scales, W, scale_inv) (+ bias): 128 x 128 output tiles, k-step = one
 *                        128-block, accumulator += dot * a_s * b_s (fp8gemm.py:156).  s_ld >= ceil(T / 128) * 128, a multiple of 4.
 * Same arithmetic as ktx_linear_forward on the same handle (bit for bit); the decode layout of the weights is read in place. */
int ktx_fp8_act_quant(const void* d_x, int64_t ldx, int T, int K, void* d_q, float* d_scales_t, int64_t s_ld, ktx_stream_t stream);
int ktx_linear_gemm_fp8(ktx_linear_t h, const void* d_q, const float* d_scales_t, int64_t s_ld, int T, void* d_y, int64_t ldy,
                        ktx_stream_t stream);

/* The merge of the MLA KV splits and the per-head un-absorb products (torch.matmul(attn_output, out_absorb.mT),
 * archive/ktransformers/operators/attention.py:465-468) of a decode step (T <= 4) in ONE launch: h = the batched BF16 W_UV
 * handle (batch = heads, kv_lora 512 -> v_head_dim 128); d_part_o / d_part_ml / nsplit = what ktx_mla_decode_partials
 * (include/ktx_mla.h) left in its workspace.  y[t*ldy + head*y_batch_stride + n] = W_UV[head] . merged[t][head], with the
 * merged row rounded to bf16 exactly where ktx_mla_decode_append rounds it.  Ask ktx_linear_merge_eligible first. */
int ktx_linear_merge_eligible(ktx_linear_t h, int T, int nsplit, int num_heads);
int ktx_linear_forward_batched_merge(ktx_linear_t h, int T, const float* d_part_o, const float* d_part_ml, int nsplit,
                                     int num_heads, void* d_y, int64_t ldy, int64_t y_batch_stride, ktx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif

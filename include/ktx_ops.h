/* ktx_ops.h — C ABI of the small fused ops between the GEMMs of one decoder layer (libktx_hip.so).
 *
 * SURVEY.md §8f row 3 ("decode loop glue": RMSNorm / YaRN-RoPE / cache-append fusions) and the non-GEMM half of row a14:
 *
 *   reference                                                                          this library
 *   ---------------------------------------------------------------------------------  ---------------------------
 *   RMSNorm.forward_native   archive/ktransformers/operators/layernorm.py:79-87        ktx_rmsnorm(native_rounding=1)
 *   flashinfer.norm.rmsnorm / fused_add_rmsnorm (un-vendored)  layernorm.py:61-77      ktx_rmsnorm(0) / ktx_fused_add_rmsnorm
 *   YarnRotaryEmbeddingV3.forward + apply_rotary_pos_emb                               ktx_mla_prep
 *     operators/RoPE.py:262-275, models/modeling_deepseek.py:337-366
 *   kv_a_layernorm + split of kv_a_proj_with_mqa's output  operators/attention.py:374-381
 *   DeepseekV3MLP: act_fn(gate_proj(x)) * up_proj(x)  models/modeling_deepseek_v3.py:382-398   ktx_silu_mul
 *
 * bf16 tensors, DEVICE pointers, enqueue-only on `stream`, 0 = success (ktx_last_error() for the message).
 */
#ifndef KTX_OPS_H
#define KTX_OPS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef KTX_STREAM_T_DEFINED
#define KTX_STREAM_T_DEFINED
typedef void* ktx_stream_t; /* hipStream_t */
#endif
const char* ktx_last_error(void);

/* y[t] = w * (x[t] * rsqrt(mean(x[t]^2) + eps)) for t < min(T, *d_bsz) (d_bsz may be NULL).  native_rounding = 1
 * reproduces forward_native's two roundings (bf16(w * bf16(x*r))); 0 rounds once like flashinfer's rmsnorm.
 * dim % 8 == 0, dim <= 16384; ldx / ldy = row strides in elements (multiples of 8). y may alias x. */
int ktx_rmsnorm(const void* d_x, int64_t ldx, const void* d_w, void* d_y, int64_t ldy, int T, int dim, float eps,
                int native_rounding, const int32_t* d_bsz, ktx_stream_t stream);

/* flashinfer fused_add_rmsnorm (layernorm.py:69): residual[t] = bf16(x[t] + residual[t]);
 * x[t] = bf16((x[t] + residual[t]) * rsqrt(mean(.^2) + eps) * w), the norm taken on the unrounded fp32 sum. */
int ktx_fused_add_rmsnorm(void* d_x, void* d_residual, const void* d_w, int T, int dim, float eps, const int32_t* d_bsz,
                          ktx_stream_t stream);

/* y[t][i] = bf16(bf16(silu(g[t][i])) * u[t][i]) with g = gu[t][i], u = gu[t][inter + i] — the activation between a fused
 * [gate_proj; up_proj] GEMM and down_proj.  ldg = row stride of gu in elements. */
int ktx_silu_mul(const void* d_gu, int64_t ldg, void* d_y, int T, int inter, const int32_t* d_bsz, ktx_stream_t stream);

/* One launch for everything between the q / kv_a projections and the MLA kernel (attention.py:360-395):
 *   q      [T][num_heads][nope_dim + rope_dim]  (row stride q_row_stride)  -> q_pe_out [T][num_heads][rope_dim]
 *   kv     [T][kv_lora + rope_dim]              (row stride kv_row_stride) -> ckv_out [T][kv_lora]  = RMSNorm(kv[:, :kv_lora]) (native rounding)
 *                                                                          -> kpe_out [T][rope_dim] = RoPE(kv[:, kv_lora:])
 * RoPE = DeepSeek's: de-interleave (x[0::2] | x[1::2]), cos/sin = bf16(cos(pos * inv_freq[i]) * mscale), bf16 arithmetic
 * as torch evaluates q*cos + rotate_half(q)*sin.  d_pos: int64 [T] positions; d_inv_freq: fp32 [rope_dim/2]. */
int ktx_mla_prep(int T, int num_heads, int nope_dim, int rope_dim, int kv_lora, const void* d_q, int64_t q_row_stride,
                 void* d_q_pe_out, const void* d_kv, int64_t kv_row_stride, const void* d_kv_norm_w, float eps,
                 void* d_ckv_out, void* d_kpe_out, const int64_t* d_pos, const float* d_inv_freq, float mscale,
                 ktx_stream_t stream);

/* Greedy sampling (decode_one_tokens with do_sample = False: torch.argmax over the last position's logits,
 * archive/ktransformers/util/utils.py:483-494): out[r] = index of the first maximum of row r of x, bf16 [rows][n] with row
 * stride ldx (elements; every row 16-byte aligned).  Taken on the lm_head's bf16 output directly — the fp32 copy the reference makes first
 * (logits.float()) is exact and monotonic, so the index is the same.  One launch; d_workspace: ktx_argmax_workspace_bytes(rows)
 * bytes, zero-initialised once by the caller (the kernel leaves its arrival counters at zero).  NaN logits never win. */
size_t ktx_argmax_workspace_bytes(int rows);
int ktx_argmax_bf16(const void* d_x, int64_t ldx, int rows, int n, int64_t* d_out, void* d_workspace, ktx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif

/* ktx_mla.h — C ABI of the MLA compressed-KV paged attention (absorbed form) in libktx_hip.so.
 *
 * Replaces, for KDeepseekV2Attention.forward_linux_flashinfer (archive/ktransformers/operators/attention.py:349-523)
 * and flashinfer_attn.forward (archive/ktransformers/operators/balance_serve_attention.py:66-118):
 *   flashinfer.mla.BatchMLAPagedAttentionWrapper.plan / .run       call sites archive/.../flashinfer_wrapper.py:103-161
 *   Triton decode_attention_fwd_grouped                              archive/.../triton_attention.py:358-385
 *   StaticCache.update / KDeepSeekV3Cache.update (latent append)    archive/ktransformers/models/custom_cache.py:147-199,414-447
 *
 * Semantics (= attention_ref_torch, flashinfer_wrapper.py:30-76, with K = [ckv | k_pe], V = ckv shared by all heads):
 *   s[h][n] = sm_scale * ( q_nope[h] . ckv[n] + q_pe[h] . k_pe[n] ),  n <= kv_len - qo_len + i   (causal)
 *   out[h]  = softmax_n(s[h]) @ ckv                                     fp32 softmax / accumulation, bf16 in/out
 * All pointers are DEVICE pointers; calls only enqueue on `stream` and are HIP-graph capturable (batch size and the
 * page tables are read on the device).
 */
#ifndef KTX_MLA_H
#define KTX_MLA_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ktx_mla_config {
  int32_t num_heads;      /* query heads (128 V3, 64 K2, 16 V2-Lite); multiple of 16 */
  int32_t head_dim_ckv;   /* kv_lora_rank = 512 */
  int32_t head_dim_kpe;   /* qk_rope_head_dim = 64 */
  int32_t page_size;      /* tokens per page (64 single-request cache, 256 server cache) */
  float sm_scale;         /* softmax_scale = q_head_dim^-0.5 * mscale^2 (modeling_deepseek_v3.py:697-703) */
  int32_t max_splits;     /* upper bound on KV splits the workspace was sized for (>=1) */
  int32_t kv_len_hint;    /* host-side upper bound of the context length (0 = unknown): only steers the split count */
} ktx_mla_config;

/* bytes of scratch needed for `max_q_tokens` query tokens (fp32 partial outputs + softmax stats per KV split) */
size_t ktx_mla_workspace_bytes(const ktx_mla_config* cfg, int max_q_tokens);

/* run(q_nope[T,Hq,512], q_pe[T,Hq,64], ckv[pages,page,512], k_pe[pages,page,64]) -> out[T,Hq,512]
 * (BatchMLAPagedAttentionWrapper.run; the plan() arguments are passed here as device arrays):
 *   qo_indptr int32 [batch+1], kv_indptr int32 [batch+1], kv_indices int32 [*] (page ids; NULL = identity page table: request
 *   r owns pages kv_indptr[r] .. kv_indptr[r+1]-1 in order, which spares the kernel a dependent load), kv_len_arr int32 [batch];
 *   d_bsz int32* or NULL: number of live requests (<= batch), read on the device;
 *   ckv_token_stride / kpe_token_stride: elements between consecutive tokens of a page (576 for the fused cache view);
 *   lse float [T,Hq] or NULL (natural-log sum-exp * log2(e), like flashinfer's return_lse). */
int ktx_mla_decode(const ktx_mla_config* cfg, const void* d_q_nope, const void* d_q_pe, const void* d_ckv,
                   const void* d_k_pe, int64_t ckv_token_stride, int64_t kpe_token_stride, const int32_t* d_qo_indptr,
                   const int32_t* d_kv_indptr, const int32_t* d_kv_indices, const int32_t* d_kv_len_arr,
                   const int32_t* d_bsz, int batch, int total_q_tokens, void* d_out, float* d_lse, void* d_workspace,
                   size_t workspace_bytes, void* stream);

/* Decode step with the cache append fused in (one launch less per layer): for every request with exactly one query token,
 * the newest position kv_len-1 is read from d_new_ckv [batch][512] / d_new_kpe [batch][64] instead of the cache and is
 * written into the cache page by the kernel (StaticCache.update + run in one call).  NULL/NULL = plain ktx_mla_decode. */
int ktx_mla_decode_append(const ktx_mla_config* cfg, const void* d_q_nope, const void* d_q_pe, void* d_ckv, void* d_k_pe,
                          int64_t ckv_token_stride, int64_t kpe_token_stride, const int32_t* d_qo_indptr,
                          const int32_t* d_kv_indptr, const int32_t* d_kv_indices, const int32_t* d_kv_len_arr,
                          const int32_t* d_bsz, int batch, int total_q_tokens, const void* d_new_ckv, const void* d_new_kpe,
                          void* d_out, float* d_lse, void* d_workspace, size_t workspace_bytes, void* stream);

/* ktx_mla_decode_append without its merge launch: the split-KV kernel only.  *nsplit_out (host) receives the number of KV
 * splits S; d_workspace then holds, for the caller's own merge (ktx_linear_forward_batched_merge folds it into the un-absorb
 * products of a decode step):  part_o fp32 [T][Hq][S][512] — un-normalised partial outputs relative to the split's own max —
 * followed by part_ml fp32 [T][Hq][S][2] = (m, l) per split (l == 0: the split saw no token; its part_o row is undefined).
 *     out[t][h] = sum_s exp(m_s - m*) part_o[s] / sum_s exp(m_s - m*) l_s,   m* = max over splits with l_s > 0 */
int ktx_mla_decode_partials(const ktx_mla_config* cfg, const void* d_q_nope, const void* d_q_pe, void* d_ckv, void* d_k_pe,
                            int64_t ckv_token_stride, int64_t kpe_token_stride, const int32_t* d_qo_indptr,
                            const int32_t* d_kv_indptr, const int32_t* d_kv_indices, const int32_t* d_kv_len_arr,
                            const int32_t* d_bsz, int batch, int total_q_tokens, const void* d_new_ckv, const void* d_new_kpe,
                            void* d_workspace, size_t workspace_bytes, int* nsplit_out, void* stream);

/* cache.update(): scatter T new latent rows [ckv(512) | k_pe(64)] to cache[page_idx[t]][page_offset[t]]
 * (custom_cache.py:189-195 / :433-441).  kv_cache bf16 [pages][page_size][token_stride].  num_pages > 0: rows whose
 * page_idx / page_offset fall outside [0, num_pages) x [0, page_size) are dropped instead of written (the reference's
 * indexed assignment raises; a device-side scatter cannot, so it must not corrupt HBM); 0 = unchecked. */
int ktx_mla_cache_append(const ktx_mla_config* cfg, void* d_kv_cache, int64_t token_stride, const void* d_ckv_new,
                         const void* d_kpe_new, const int32_t* d_page_idx, const int32_t* d_page_offset,
                         const int32_t* d_ntokens, int max_tokens, int num_pages, void* stream);

/* Non-absorbed prompt attention — the prefill branch of KDeepseekV2Attention (archive/ktransformers/operators/attention.py:
 * 349-523, forward_chunck :58-164): kv_b_proj has expanded the context's latents to per-head keys and values, and a causal
 * attention runs over qk dim 192 (128 nope + 64 rope) with v dim 128 — 3.4x fewer flop per (query, key) than the absorbed
 * form, which stays the decode path.  The T query tokens are the LAST T of the kv_len keys (chunked prefill: kv_len > T).
 *   q_nope bf16 [T][H][128] and q_pe bf16 [T][H][64] (RoPE applied) with explicit element strides (views of one q tensor);
 *   k_nope bf16 [H][kv_pad][128]; k_pe bf16 [kv_len][64] rows `kpe_token_stride` apart (the latent cache itself);
 *   v_t bf16 [H][128][kv_pad] (V TRANSPOSED: keys contiguous — it is a GEMM output either way); out bf16 [T][H][128].
 * kv_pad: multiple of 64 >= kv_len; k_nope rows and v_t columns in [kv_len, kv_pad) must be ZERO (they are, when the latent
 * rows fed to the expansion GEMMs are zero-padded). */
int ktx_mla_prefill(int T, int num_heads, int kv_len, int kv_pad, float sm_scale, const void* d_q_nope, int64_t qn_token_stride,
                    int64_t qn_head_stride, const void* d_q_pe, int64_t qp_token_stride, int64_t qp_head_stride,
                    const void* d_k_nope, const void* d_k_pe, int64_t kpe_token_stride, const void* d_v_t, void* d_out,
                    void* stream);

/* Tuning aid (scripts/mla_sweep.py): while a device buffer of >= 16 * workgroups int64 entries is set, every workgroup of the
 * split-KV kernel stamps the wall clock (100 MHz) at its phase boundaries into it; NULL (the default) turns it off. */
int ktx_mla_debug_stamps(long long* d_buf);

#ifdef __cplusplus
}
#endif
#endif

/* ktx_attn.h — C ABI of the ONE-LAUNCH MLA decode step (attention half of a decoder layer) in libktx_hip.so.
 *
 * Replaces, for one decode token of KDeepseekV2Attention.forward_linux_flashinfer
 * (archive/ktransformers/operators/attention.py:349-523) plus the decoder layer's input_layernorm and residual add
 * (archive/ktransformers/models/modeling_deepseek_v3.py:1200-1219), the chain the reference runs back to back:
 *     input_layernorm -> q_a_proj | kv_a_proj_with_mqa -> q_a_layernorm -> q_b_proj -> q_nope @ W_UK (absorb), RoPE(q_pe),
 *     kv_a_layernorm + RoPE(k_pe) + cache append -> paged MQA over the latent cache -> attn @ W_UV^T -> o_proj -> residual +
 * The arithmetic of every stage is that of the library's stand-alone kernels (ktx_linear.h / ktx_mla.h: same products, same
 * summation orders, same roundings); what changes is that the stages are PHASES of one persistent launch — one workgroup per
 * CU, every phase's weights requested while the previous phase is still waiting for its input, stage outputs handed over
 * through write-through stores + epoch flags in a device workspace — instead of five dependent launches.
 * `phases` selects a subset (bit 0 = q_a|kv_a ... bit 4 = o_proj) so that the same code runs as 1..5 launches for A/B
 * measurements and stage-by-stage parity tests; the flags make any split correct.
 *
 * All pointers are DEVICE pointers; the call only enqueues on `stream` and is HIP-graph capturable.
 */
#ifndef KTX_ATTN_H
#define KTX_ATTN_H
#include <stddef.h>
#include <stdint.h>

#include "ktx_linear.h"
#ifdef __cplusplus
extern "C" {
#endif

#define KTX_ATTN_PHASE_QKV_A 1     /* input RMSNorm + merged q_a | kv_a GEMV                                   */
#define KTX_ATTN_PHASE_QB 2        /* q_a_layernorm + q_b_proj + absorb + RoPE(q_pe); kv_a_layernorm + RoPE(k_pe) + cache append */
#define KTX_ATTN_PHASE_MLA 4       /* split-KV attention over the paged latent cache                            */
#define KTX_ATTN_PHASE_MERGE 8     /* merge of the KV splits + un-absorb                                        */
#define KTX_ATTN_PHASE_OPROJ 16    /* o_proj + residual add                                                     */
#define KTX_ATTN_PHASE_ALL 31
typedef struct ktx_attn_decode_args {
  /* operators (include/ktx_linear.h handles, loaded) */
  ktx_linear_t qkv_a;      /* W4 g64, in = hidden, out = q_lora + kv_lora + rope: rows [q_a | ckv | k_pe] */
  ktx_linear_t q_b;        /* W4 g64, in = q_lora, out = heads * (nope + rope)                              */
  ktx_linear_t q_absorb;   /* BF16, batch = heads, in = nope, out = kv_lora   (W_UK)                        */
  ktx_linear_t out_absorb; /* BF16, batch = heads, in = kv_lora, out = v_dim  (W_UV)                        */
  ktx_linear_t o_proj;     /* W4 g64, in = heads * v_dim, out = hidden                                      */
  int32_t num_heads, nope_dim, rope_dim, kv_lora, v_dim, q_lora, hidden;
  /* layer input = residual stream, bf16 [hidden]; output bf16 [hidden] = x + o_proj(attention) */
  const void* d_x;
  void* d_y;
  const void* d_in_norm_w;   float in_norm_eps;    /* input_layernorm                      */
  const void* d_qa_norm_w;   float qa_norm_eps;    /* q_a_layernorm                        */
  const void* d_kv_norm_w;   float kv_norm_eps;    /* kv_a_layernorm                       */
  const int64_t* d_position;                       /* [1] position of the token            */
  const float* d_inv_freq;   float mscale;         /* rope: [rope_dim / 2] inverse frequencies, YaRN mscale */
  /* paged latent cache (ktx_mla.h conventions): one request */
  void* d_ckv; void* d_k_pe; int64_t ckv_token_stride, kpe_token_stride;
  int32_t page_size;
  const int32_t* d_kv_indptr;    /* [2]                                   */
  const int32_t* d_kv_indices;   /* page ids or NULL (identity)           */
  const int32_t* d_kv_len;       /* [1] context length INCLUDING the new token */
  int32_t kv_len_hint;           /* host-side upper bound of the context (steers the split count, as ktx_mla_config) */
  float sm_scale;
  int32_t phases;                /* KTX_ATTN_PHASE_* mask of this launch */
  int32_t last;                  /* 1: this launch ends the step (advances the workspace epoch); the last launch of a split chain */
} ktx_attn_decode_args;

/* 1 when ktx_attn_decode covers this configuration on the current device (nope 128 / rope 64 / kv_lora 512 / v 128, hidden 7168 and
 * q_lora 1536 — the DeepSeek-V3 / R1 and Kimi-K2 attention — with 128 or 64 heads, W4 g64 projections without bias, at least 256
 * CUs, a context whose KV split count fits one head group's workgroups), else 0 with the reason in ktx_last_error().  The answer
 * depends on kv_len_hint (it picks the split shape): ask per call, not once per layer. */
int ktx_attn_decode_eligible(const ktx_attn_decode_args* a);

/* The launch is PERSISTENT: 256 workgroups that wait for each other.  All of them must be resident at once, so (1) at most one such
 * launch may be in flight per device — launches of one stream are ordered; launches issued from a second stream are ordered behind the
 * device's previous one by an event the library records (eager mode: wait, launch and record happen under one per-device lock, so host
 * threads may launch concurrently; inside a stream capture the caller keeps one capture per device, and REPLAYS of a captured graph are
 * not recorded — an eager launch issued beside a running replay on another stream is the caller's to order) — and (2) a foreign kernel occupying CUs for longer than the poll bound (0.2 s) makes a hand-off give up: the launch then
 * ends with undefined results and the status word below is set. */
int ktx_attn_decode(const ktx_attn_decode_args* a, ktx_stream_t stream);

/* Status word of a device: 0, or the code of the first hand-off that timed out (the launch then finished with undefined results
 * instead of hanging).  The device writes the word into pinned host memory at the moment a poll gives up, so these calls are plain
 * host loads — no synchronisation — and a decode loop checks after EVERY token.  ktx_attn_status_any: the first device with a
 * non-zero word (device_out = -1 if none).  ktx_attn_reset synchronises the device, clears the word and re-arms the workspaces. */
int ktx_attn_status(int device, uint32_t* status_out);
int ktx_attn_status_any(int* device_out, uint32_t* status_out);
int ktx_attn_reset(int device);

/* tests: copy a workspace array of the last launch (0 q_a|kv_a row, 1 ckv_new, 2 kpe_new, 3 q_lat, 4 q_pe, 6 attn_out, 7 part_ml,
 * 8 part_o, 9 exchanged q_nope) into a device buffer.  Arrays 0, 3 and 4 travel as tagged granules inside the launch; the copy is the
 * payload alone, in row order.  (5, the merged rows, no longer pass through the workspace: refused.) */
int ktx_attn_debug_read(int device, int which, void* d_dst, size_t bytes);

/* dev probe: 64 wall-clock stamps (100 MHz) of workgroup 0 per launch, or NULL */
int ktx_attn_debug_stamps(unsigned long long* d_buf);

#ifdef __cplusplus
}
#endif
#endif

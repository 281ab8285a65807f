/* ktx_gate.h — C ABI of the MoE router (gate) kernels in libktx_hip.so.
 *
 * Replaces the ~10 small torch kernels of MoEGate.forward:
 *   DeepSeek-V3 / Kimi-K2 (sigmoid + e_score_correction_bias, "noaux_tc" group-limited top-k)
 *       archive/ktransformers/models/modeling_deepseek_v3.py:430-481
 *   DeepSeek-V2 / V2-Lite (fp32 softmax, "greedy" or "group_limited_greedy")
 *       archive/ktransformers/models/modeling_deepseek.py:413-455
 * wrapped by KMoEGate (archive/ktransformers/operators/gate.py:91-127).
 *
 * Output order: the reference calls torch.topk(sorted=False), whose index order is implementation-defined; this
 * library emits the selected experts in descending choice-score order (ties: lower index first), which is one of the
 * orders torch may return.  Parity is therefore on the index SET (and the weight attached to each index).
 * All pointers are DEVICE pointers; calls only enqueue on `stream` and are HIP-graph capturable.
 */
#ifndef KTX_GATE_H
#define KTX_GATE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum ktx_gate_scoring { KTX_GATE_SIGMOID = 0, KTX_GATE_SOFTMAX = 1 };
enum ktx_gate_topk { KTX_GATE_GREEDY = 0, KTX_GATE_GROUP_LIMITED_GREEDY = 1, KTX_GATE_NOAUX_TC = 2 };

typedef struct ktx_gate_config {
  int32_t n_routed_experts;   /* config.n_routed_experts */
  int32_t hidden_size;        /* config.hidden_size */
  int32_t top_k;              /* config.num_experts_per_tok */
  int32_t n_group;            /* config.n_group (1 = ungrouped) */
  int32_t topk_group;         /* config.topk_group */
  int32_t scoring;            /* enum ktx_gate_scoring  (config.scoring_func) */
  int32_t topk_method;        /* enum ktx_gate_topk     (config.topk_method) */
  int32_t norm_topk_prob;     /* config.norm_topk_prob */
  float routed_scaling_factor;/* config.routed_scaling_factor */
} ktx_gate_config;

/* logits[t][e] = sum_h float(x[t][h]) * float(w[e][h])  (F.linear in fp32, modeling_deepseek_v3.py:434-437).
 * x bf16 [qlen][H]; w bf16 [E][H]; logits fp32 [qlen][E]. */
int ktx_gate_logits(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x, const void* d_w,
                    float* d_logits, void* stream);

/* scores -> (+bias) -> group selection -> top-k -> gather unbiased scores -> normalise / scale
 * (modeling_deepseek_v3.py:438-481).  bias fp32 [E] or NULL.  topk_idx int64 [qlen][k], topk_weight fp32 [qlen][k]. */
int ktx_gate_select(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const float* d_logits,
                    const float* d_bias, int64_t* d_topk_idx, float* d_topk_weight, void* stream);

/* Both steps in ONE launch for decode-sized batches (bf16 x and w): the last workgroup of each token to finish its
 * logits performs the selection.  d_counters: int32 [qlen], zero-initialised once by the caller (the kernel leaves them
 * zero again).  d_logits fp32 [qlen][E] is scratch that also receives the logits. */
int ktx_gate_forward(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x, const void* d_w,
                     const float* d_bias, float* d_logits, int32_t* d_counters, int64_t* d_topk_idx,
                     float* d_topk_weight, void* stream);

/* ktx_gate_forward with the MoE block's input RMSNorm folded in (post_attention_layernorm in front of the router,
 * modeling_deepseek_v3.py:1222-1224): d_x is the UN-normalised hidden state, the logits are taken on
 * xn = norm_weight * bf16(x * rsqrt(mean(x^2) + eps)), and xn (bf16 [qlen][hidden]) is written to d_xn_out for the experts
 * that consume it next.  hidden_size <= 8192. */
int ktx_gate_forward_norm(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x,
                          const void* d_norm_weight, float norm_eps, void* d_xn_out, const void* d_w, const float* d_bias,
                          float* d_logits, int32_t* d_counters, int64_t* d_topk_idx, float* d_topk_weight, void* stream);

#ifdef __cplusplus
}
#endif
#endif

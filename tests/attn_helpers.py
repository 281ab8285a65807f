"""Shared by the attention tests: a config object, golden loading, and an HF-shaped attention module to inject into."""
import os
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "attention_golden.npz")
ROPE_SCALING = {"type": "yarn", "factor": 40, "mscale": 1.0, "mscale_all_dim": 1.0, "original_max_position_embeddings": 4096,
                "beta_fast": 32, "beta_slow": 1}


def make_cfg(hidden, H, q_lora, **kw):
    d = dict(hidden_size=hidden, num_attention_heads=H, q_lora_rank=q_lora, kv_lora_rank=512, qk_rope_head_dim=64,
             qk_nope_head_dim=128, v_head_dim=128, max_position_embeddings=4096, rope_theta=10000,
             rope_scaling=dict(ROPE_SCALING), rms_norm_eps=1e-6, num_hidden_layers=1)
    d.update(kw)
    return SimpleNamespace(**d)


def load_golden(name):
    g = np.load(GOLD)
    hidden, H, q_lora, T = (int(v) for v in g[f"{name}.meta"])
    cfg = make_cfg(hidden, H, None if q_lora < 0 else q_lora)
    w = {}
    for k in g.files:
        if k.startswith(name + ".") and k.endswith(".weight"):
            w[k[len(name) + 1:-len(".weight")]] = torch.from_numpy(g[k]).view(torch.bfloat16)
    x = torch.from_numpy(g[f"{name}.hidden"]).view(torch.bfloat16)
    y_bf16 = torch.from_numpy(g[f"{name}.y_bf16"]).view(torch.bfloat16)
    y_f32 = torch.from_numpy(g[f"{name}.y_f32"])
    return cfg, w, x, y_bf16, y_f32


class ToyNorm(nn.Module):
    def __init__(self, w, eps):
        super().__init__()
        self.weight = nn.Parameter(w, requires_grad=False)
        self.variance_epsilon = eps
        self.hidden_size = w.numel()


class ToyAttention(nn.Module):
    """Attribute names of DeepseekV2/V3Attention (models/modeling_deepseek_v3.py:635-703); weights from a dict."""

    def __init__(self, cfg, w, device, layer_idx=0):
        super().__init__()
        self.config, self.layer_idx = cfg, layer_idx
        self.num_heads, self.q_lora_rank = cfg.num_attention_heads, cfg.q_lora_rank
        self.qk_rope_head_dim, self.kv_lora_rank, self.v_head_dim = cfg.qk_rope_head_dim, cfg.kv_lora_rank, cfg.v_head_dim
        self.qk_nope_head_dim = cfg.qk_nope_head_dim
        self.q_head_dim = cfg.qk_nope_head_dim + cfg.qk_rope_head_dim

        def lin(name):
            wt = w[name].to(device)
            m = nn.Linear(wt.shape[1], wt.shape[0], bias=False, device="meta")
            m.weight = nn.Parameter(wt, requires_grad=False)
            return m

        if self.q_lora_rank is None:
            self.q_proj = lin("q_proj")
        else:
            self.q_a_proj, self.q_b_proj = lin("q_a_proj"), lin("q_b_proj")
            self.q_a_layernorm = ToyNorm(w["q_a_layernorm"].to(device), cfg.rms_norm_eps)
        self.kv_a_proj_with_mqa = lin("kv_a_proj_with_mqa")
        self.kv_a_layernorm = ToyNorm(w["kv_a_layernorm"].to(device), cfg.rms_norm_eps)
        self.kv_b_proj = lin("kv_b_proj")
        self.o_proj = lin("o_proj")
        self.rotary_emb = nn.Module()

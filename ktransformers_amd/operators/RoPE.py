"""Rotary-embedding operators — mirror of archive/ktransformers/operators/RoPE.py (RotaryEmbeddingV3 :64-113,
YarnRotaryEmbeddingV3 :222-326): `load()` derives `inv_freq` (+ YaRN `_mscale`) from the config, `forward(x, position_ids)`
returns (cos, sin) in x.dtype.  KDeepseekV2Attention does not call forward(): it hands `inv_freq` / `_mscale` to the fused
ktx_mla_prep kernel, which evaluates the same cos/sin per token."""
from __future__ import annotations

import math

import torch

from ktransformers_amd.operators.base_operator import BaseInjectedModule


def yarn_find_correction_dim(num_rotations, dim, base=10000, max_position_embeddings=2048):
    return (dim * math.log(max_position_embeddings / (num_rotations * 2 * math.pi))) / (2 * math.log(base))


def yarn_find_correction_range(low_rot, high_rot, dim, base=10000, max_position_embeddings=2048):
    low = math.floor(yarn_find_correction_dim(low_rot, dim, base, max_position_embeddings))
    high = math.ceil(yarn_find_correction_dim(high_rot, dim, base, max_position_embeddings))
    return max(low, 0), min(high, dim - 1)


def yarn_get_mscale(scale=1, mscale=1):
    return 1.0 if scale <= 1 else 0.1 * mscale * math.log(scale) + 1.0


def yarn_linear_ramp_mask(lo, hi, dim):
    if lo == hi:
        hi += 0.001
    return torch.clamp((torch.arange(dim, dtype=torch.float32) - lo) / (hi - lo), 0, 1)


def yarn_inv_freq(dim, base, scaling_factor, original_max_position_embeddings=4096, beta_fast=32, beta_slow=1, device=None):
    """models/modeling_deepseek_v3.py:292-313 / RoPE.py:301-320."""
    ar = torch.arange(0, dim, 2, dtype=torch.float32, device=device) / dim
    freq_extra = 1.0 / (base ** ar)
    freq_inter = 1.0 / (scaling_factor * base ** ar)
    low, high = yarn_find_correction_range(beta_fast, beta_slow, dim, base, original_max_position_embeddings)
    mask = 1.0 - yarn_linear_ramp_mask(low, high, dim // 2).to(device=device, dtype=torch.float32)
    return freq_inter * (1 - mask) + freq_extra * mask


class RotaryEmbeddingV3(BaseInjectedModule):
    """Plain RoPE (RoPE.py:64-113)."""

    def __init__(self, key, gguf_loader, config, orig_module, generate_device: str = "cuda", prefill_device: str = "cuda", **kwargs):
        BaseInjectedModule.__init__(self, key, gguf_loader, config, orig_module, prefill_device, generate_device, **kwargs)
        object.__setattr__(self, "inv_freq", None)
        object.__setattr__(self, "_mscale", 1.0)

    def load(self):
        # MLA models rotate the 64-wide rope part; llama-style models (Mixtral) the whole head (the replaced module's `dim`)
        dim = getattr(self.config, "qk_rope_head_dim", None) or getattr(self.orig_module, "dim", None) \
            or self.config.hidden_size // self.config.num_attention_heads
        inv = 1.0 / (self.config.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.float32, device=self.device) / dim))
        object.__setattr__(self, "inv_freq", inv)

    @torch.no_grad()
    def forward(self, x, position_ids):
        freqs = (self.inv_freq[None, :, None].float().expand(position_ids.shape[0], -1, 1)
                 @ position_ids[:, None, :].float()).transpose(1, 2)
        emb = torch.cat((freqs, freqs), dim=-1)
        return (emb.cos() * self._mscale).to(dtype=x.dtype), (emb.sin() * self._mscale).to(dtype=x.dtype)


class YarnRotaryEmbeddingV3(RotaryEmbeddingV3):
    """YaRN RoPE (RoPE.py:222-326)."""

    def load(self):
        rs = self.config.rope_scaling
        kw = {k: rs[k] for k in ("original_max_position_embeddings", "beta_fast", "beta_slow") if k in rs}
        object.__setattr__(self, "inv_freq", yarn_inv_freq(self.config.qk_rope_head_dim, self.config.rope_theta,
                                                           rs["factor"], device=self.device, **kw))
        object.__setattr__(self, "_mscale", float(yarn_get_mscale(rs["factor"], rs.get("mscale", 1))
                                                  / yarn_get_mscale(rs["factor"], rs.get("mscale_all_dim", 0))))


# names the reference's rule files use for the same two behaviours
RotaryEmbedding = RotaryEmbeddingV3
YarnRotaryEmbedding = YarnRotaryEmbeddingV3
DeepSeekV3YarnRotaryEmbedding = YarnRotaryEmbeddingV3


# archive/ktransformers/operators/RoPE.py:367-416 is RotaryEmbeddingV3 again under another name (Moonlight serve rules)
RotaryEmbeddingV4 = RotaryEmbeddingV3

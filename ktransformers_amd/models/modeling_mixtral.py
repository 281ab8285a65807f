"""Module skeleton of Mixtral-8x7B (BASELINE config C1) — the host tree the reference's `optimize_rules/Mixtral.yaml` injects into.

The reference injects into its copy of the HF modeling file (archive/ktransformers/models/modeling_mixtral.py); parameter names
and forward structure are what the rule file and the weight loaders key on, so this file keeps them: model.embed_tokens,
model.layers.N.{input_layernorm, self_attn.{q_proj, k_proj, v_proj, o_proj, rotary_emb}, post_attention_layernorm,
block_sparse_moe.{gate, experts.M.{w1, w2, w3}}}, model.norm, lm_head.

What the rule file replaces — MixtralRotaryEmbedding, every nn.Linear under model.layers.* and lm_head (KTransformersLinear),
MixtralSparseMoeBlock (KMistralSparseMoEBlock) and its experts (KTransformersExperts) — does arithmetic through the HIP
library; those leaves raise here if they were not replaced.  Mixtral's grouped-query attention core is NOT on this build's hot
path (SURVEY.md section 8 scopes the attention rows to MLA) and the reference leaves it to the HF module as well
(`MixtralSdpaAttention`, modeling_mixtral.py:560-650): it stays torch glue here too — RoPE application, a per-layer K/V buffer
and scaled_dot_product_attention between the injected q/k/v/o linears."""
from __future__ import annotations

from types import SimpleNamespace

import torch
from torch import nn

from ktransformers_amd.models.modeling_deepseek import DeepseekRMSNorm, _Leaf


def make_mixtral_config(**kw) -> SimpleNamespace:
    """Field names of the HF MixtralConfig the reference's modeling_mixtral.py reads; defaults are Mixtral-8x7B's public
    config.json."""
    d = dict(vocab_size=32000, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
             num_key_value_heads=8, num_local_experts=8, num_experts_per_tok=2, max_position_embeddings=32768, rope_theta=1e6,
             rms_norm_eps=1e-5, hidden_act="silu", sliding_window=None, router_jitter_noise=0.0, attention_dropout=0.0,
             torch_dtype=torch.bfloat16, architectures=["MixtralForCausalLM"])
    d.update(kw)
    return SimpleNamespace(**d)


class MixtralRMSNorm(DeepseekRMSNorm):
    """The library's RMSNorm kernel (see DeepseekRMSNorm: the rule file leaves the norms un-replaced)."""


class MixtralRotaryEmbedding(_Leaf):
    def __init__(self, dim=None, max_position_embeddings=2048, base=10000.0):
        super().__init__()
        self.dim, self.max_position_embeddings, self.base = dim, max_position_embeddings, base


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


class MixtralKVCache:
    """Per-layer [1, kv_heads, max_len, head_dim] K and V buffers addressed by cache_position (glue for the skeleton's
    attention; the MLA caches of models/custom_cache.py are the hot path's)."""

    def __init__(self, config, max_len: int, device, dtype=torch.bfloat16):
        hd = config.hidden_size // config.num_attention_heads
        shape = (1, config.num_key_value_heads, max_len, hd)
        self.k = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(config.num_hidden_layers)]
        self.v = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(config.num_hidden_layers)]
        self.max_len = max_len

    def update(self, k, v, layer_idx, cache_position):
        if int(cache_position[-1]) >= self.max_len:
            raise ValueError(f"MixtralKVCache: position {int(cache_position[-1])} beyond max_len {self.max_len}")
        self.k[layer_idx][:, :, cache_position] = k
        self.v[layer_idx][:, :, cache_position] = v
        n = int(cache_position[-1]) + 1
        return self.k[layer_idx][:, :, :n], self.v[layer_idx][:, :, :n]


class MixtralAttention(nn.Module):
    def __init__(self, config, layer_idx):
        super().__init__()
        c = config
        self.config, self.layer_idx = c, layer_idx
        self.hidden_size, self.num_heads, self.num_key_value_heads = c.hidden_size, c.num_attention_heads, c.num_key_value_heads
        self.head_dim = c.hidden_size // c.num_attention_heads
        self.q_proj = nn.Linear(c.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.k_proj = nn.Linear(c.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.v_proj = nn.Linear(c.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, c.hidden_size, bias=False)
        self.rotary_emb = MixtralRotaryEmbedding(self.head_dim, c.max_position_embeddings, c.rope_theta)

    def forward(self, hidden_states, position_ids=None, past_key_value=None, cache_position=None, **kwargs):
        B, T, _ = hidden_states.shape
        q = self.q_proj(hidden_states).view(B, T, self.num_heads, self.head_dim).transpose(1, 2)
        k = self.k_proj(hidden_states).view(B, T, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        v = self.v_proj(hidden_states).view(B, T, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        cos, sin = self.rotary_emb(v, position_ids)
        cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
        q, k = q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin
        if past_key_value is not None:
            k, v = past_key_value.update(k, v, self.layer_idx, cache_position)
        S = k.shape[2]
        rep = self.num_heads // self.num_key_value_heads
        k, v = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
        pos_q = (cache_position if cache_position is not None else torch.arange(T, device=q.device)).view(T, 1)
        mask = torch.arange(S, device=q.device).view(1, S) <= pos_q                     # causal over absolute positions
        out = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask[None, None])
        return self.o_proj(out.transpose(1, 2).reshape(B, T, self.num_heads * self.head_dim)), None, past_key_value


class MixtralBlockSparseTop2MLP(_Leaf):
    def __init__(self, config):
        super().__init__()
        self.w1 = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.w2 = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)
        self.w3 = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)


class MixtralSparseMoeBlock(_Leaf):
    def __init__(self, config):
        super().__init__()
        self.hidden_dim, self.ffn_dim = config.hidden_size, config.intermediate_size
        self.num_experts, self.top_k = config.num_local_experts, config.num_experts_per_tok
        self.jitter_noise = getattr(config, "router_jitter_noise", 0.0)
        self.gate = nn.Linear(self.hidden_dim, self.num_experts, bias=False)
        self.experts = nn.ModuleList([MixtralBlockSparseTop2MLP(config) for _ in range(self.num_experts)])


class MixtralDecoderLayer(nn.Module):
    def __init__(self, config, layer_idx):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.self_attn = MixtralAttention(config, layer_idx)
        self.block_sparse_moe = MixtralSparseMoeBlock(config)
        self.input_layernorm = MixtralRMSNorm(config.hidden_size, config.rms_norm_eps)
        self.post_attention_layernorm = MixtralRMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, hidden_states, position_ids=None, past_key_value=None, cache_position=None, **kwargs):
        """MixtralDecoderLayer.forward (modeling_mixtral.py:905-960)."""
        residual = hidden_states
        hidden_states = self.input_layernorm(hidden_states)
        hidden_states, _, _ = self.self_attn(hidden_states, position_ids=position_ids, past_key_value=past_key_value,
                                             cache_position=cache_position)
        hidden_states = residual + hidden_states
        residual = hidden_states
        hidden_states = self.post_attention_layernorm(hidden_states)
        hidden_states, _router_logits = self.block_sparse_moe(hidden_states)
        return residual + hidden_states


class MixtralModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([MixtralDecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = MixtralRMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, input_ids=None, position_ids=None, past_key_values=None, cache_position=None, inputs_embeds=None):
        h = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        for layer in self.layers:
            h = layer(h, position_ids=position_ids, past_key_value=past_key_values, cache_position=cache_position)
        return self.norm(h)


class MixtralForCausalLM(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = MixtralModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    def forward(self, input_ids=None, position_ids=None, past_key_values=None, cache_position=None, inputs_embeds=None,
                last_token_only: bool = False):
        h = self.model(input_ids, position_ids, past_key_values, cache_position, inputs_embeds)
        if last_token_only:
            h = h[:, -1:, :]
        return self.lm_head(h).float()

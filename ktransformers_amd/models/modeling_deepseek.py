"""Module skeleton of DeepSeek-V2 / V2-Lite / V3 / R1 / Kimi-K2 — the host tree the optimize rules inject into.

The reference injects into its copies of the HF modeling files (archive/ktransformers/models/modeling_deepseek.py,
modeling_deepseek_v3.py); their parameter names and forward structure are what the rule files and the weight loaders key
on, so this file keeps exactly those (model.embed_tokens, model.layers.N.{input_layernorm, self_attn.{q_proj | q_a_proj,
q_a_layernorm, q_b_proj, kv_a_proj_with_mqa, kv_a_layernorm, kv_b_proj, o_proj}, post_attention_layernorm,
mlp.{gate, experts.M.{gate,up,down}_proj, shared_experts.*} | mlp.{gate,up,down}_proj}, model.norm, lm_head).

Only the glue lives here (residual adds, the layer loop, embedding lookup): every module that does arithmetic is meant
to be replaced by a ktransformers_amd.operators.* class through the YAML rules (optimize_rules/DeepSeek-V3-Chat.yaml);
un-replaced leaves raise, there is no torch fallback math for the hot path.
Forward structure: DeepseekV3DecoderLayer.forward (modeling_deepseek_v3.py:1188-1262), DeepseekV3MoE.forward (:520-531),
DeepseekV3MLP.forward (:396-398), DeepseekV3Model / ForCausalLM (:1449-1608, :1669-1760)."""
from __future__ import annotations

from types import SimpleNamespace

import torch
from torch import nn


def make_config(**kw) -> SimpleNamespace:
    """Field names of DeepseekV3Config (configuration_deepseek_v3.py:106-131); defaults are DeepSeek-V2-Lite's."""
    d = dict(vocab_size=102400, hidden_size=2048, intermediate_size=10944, moe_intermediate_size=1408, num_hidden_layers=27,
             num_attention_heads=16, n_shared_experts=2, n_routed_experts=64, num_experts_per_tok=6, first_k_dense_replace=1,
             moe_layer_freq=1, n_group=1, topk_group=1, topk_method="greedy", scoring_func="softmax", norm_topk_prob=False,
             routed_scaling_factor=1.0, q_lora_rank=None, kv_lora_rank=512, qk_rope_head_dim=64, qk_nope_head_dim=128,
             v_head_dim=128, max_position_embeddings=4096, rope_theta=10000.0, rope_scaling=None, rms_norm_eps=1e-6,
             hidden_act="silu", attention_bias=False, torch_dtype=torch.bfloat16, architectures=["DeepseekV2ForCausalLM"])
    d.update(kw)
    return SimpleNamespace(**d)


class _Leaf(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} was not replaced by an injected operator (check the optimize rules): "
                           "ktransformers_amd has no torch fallback for the hot path")


class DeepseekRMSNorm(nn.Module):
    """Several of the reference's rule files leave the norms (and, for V2, the router) un-replaced and rely on the HF module's
    own forward; here those modules run the library's HIP kernels themselves — the same calls the injected operators make
    (operators/layernorm.py, operators/gate.py) — so such a rule file works unmodified.  Still no torch math."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon, self.hidden_size = eps, hidden_size

    def forward(self, x, batch_size_tensor=None, residual=None):
        from ktransformers_amd._native import fused_add_rmsnorm, rmsnorm
        w = self.weight if self.weight.dtype == torch.bfloat16 else self.weight.to(torch.bfloat16)
        if batch_size_tensor is None:
            return rmsnorm(x, w, self.variance_epsilon, native_rounding=True)
        if residual is not None:
            fused_add_rmsnorm(x, residual, w, self.variance_epsilon, batch_size_tensor)
            return x, residual
        return rmsnorm(x, w, self.variance_epsilon, native_rounding=False, bsz_tensor=batch_size_tensor)


class DeepseekRotaryEmbedding(_Leaf):
    pass


class MoEGate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.top_k, self.n_routed_experts = config.num_experts_per_tok, config.n_routed_experts
        self.weight = nn.Parameter(torch.empty((config.n_routed_experts, config.hidden_size)))
        if config.topk_method == "noaux_tc":
            self.e_score_correction_bias = nn.Parameter(torch.empty((config.n_routed_experts)))
        self._router = None

    def forward(self, hidden_states):
        """(topk_idx, topk_weight): the router kernels of csrc/ktx_gate.hip, as KMoEGate.forward (see DeepseekRMSNorm)."""
        if self._router is None:
            from ktransformers_amd._native import GateHandle
            c = self.config
            self._router = GateHandle(c.n_routed_experts, c.hidden_size, c.num_experts_per_tok, getattr(c, "n_group", 1) or 1,
                                      getattr(c, "topk_group", 1) or 1, getattr(c, "scoring_func", "softmax"),
                                      getattr(c, "topk_method", "greedy"), bool(getattr(c, "norm_topk_prob", False)),
                                      float(getattr(c, "routed_scaling_factor", 1.0)))
        x = hidden_states.reshape(-1, hidden_states.shape[-1]).to(torch.bfloat16).contiguous()
        return self._router.forward(x, self.weight, getattr(self, "e_score_correction_bias", None), norm=None)


class DeepseekMLP(nn.Module):
    """act_fn(gate_proj(x)) * up_proj(x) -> down_proj; the three linears are injection targets, the activation is
    ktx_silu_mul."""

    def __init__(self, config, hidden_size=None, intermediate_size=None):
        super().__init__()
        self.config = config
        self.hidden_size = hidden_size or config.hidden_size
        self.intermediate_size = intermediate_size or config.intermediate_size
        self.gate_proj = nn.Linear(self.hidden_size, self.intermediate_size, bias=False)
        self.up_proj = nn.Linear(self.hidden_size, self.intermediate_size, bias=False)
        self.down_proj = nn.Linear(self.intermediate_size, self.hidden_size, bias=False)

    def forward(self, x):
        from ktransformers_amd._native import silu_mul
        g, u = self.gate_proj(x), self.up_proj(x)
        return self.down_proj(silu_mul(torch.cat([g, u], dim=-1)))


class DeepseekMoE(_Leaf):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.experts = nn.ModuleList([DeepseekMLP(config, intermediate_size=config.moe_intermediate_size)
                                      for _ in range(config.n_routed_experts)])
        self.gate = MoEGate(config)
        if config.n_shared_experts is not None:
            self.shared_experts = DeepseekMLP(config, intermediate_size=config.moe_intermediate_size * config.n_shared_experts)


class DeepseekAttention(_Leaf):
    def __init__(self, config, layer_idx):
        super().__init__()
        c = config
        self.config, self.layer_idx = c, layer_idx
        self.hidden_size, self.num_heads = c.hidden_size, c.num_attention_heads
        self.q_lora_rank, self.qk_rope_head_dim, self.kv_lora_rank = c.q_lora_rank, c.qk_rope_head_dim, c.kv_lora_rank
        self.v_head_dim, self.qk_nope_head_dim = c.v_head_dim, c.qk_nope_head_dim
        self.q_head_dim = c.qk_nope_head_dim + c.qk_rope_head_dim
        if c.q_lora_rank is None:
            self.q_proj = nn.Linear(c.hidden_size, self.num_heads * self.q_head_dim, bias=False)
        else:
            self.q_a_proj = nn.Linear(c.hidden_size, c.q_lora_rank, bias=c.attention_bias)
            self.q_a_layernorm = DeepseekRMSNorm(c.q_lora_rank, c.rms_norm_eps)
            self.q_b_proj = nn.Linear(c.q_lora_rank, self.num_heads * self.q_head_dim, bias=False)
        self.kv_a_proj_with_mqa = nn.Linear(c.hidden_size, c.kv_lora_rank + c.qk_rope_head_dim, bias=c.attention_bias)
        self.kv_a_layernorm = DeepseekRMSNorm(c.kv_lora_rank, c.rms_norm_eps)
        self.kv_b_proj = nn.Linear(c.kv_lora_rank, self.num_heads * (c.qk_nope_head_dim + c.v_head_dim), bias=False)
        self.o_proj = nn.Linear(self.num_heads * c.v_head_dim, c.hidden_size, bias=c.attention_bias)
        self.rotary_emb = DeepseekRotaryEmbedding()


class DeepseekDecoderLayer(nn.Module):
    def __init__(self, config, layer_idx):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.self_attn = DeepseekAttention(config, layer_idx)
        moe = (config.n_routed_experts is not None and layer_idx >= config.first_k_dense_replace
               and layer_idx % config.moe_layer_freq == 0)
        self.mlp = DeepseekMoE(config) if moe else DeepseekMLP(config)
        self.input_layernorm = DeepseekRMSNorm(config.hidden_size, config.rms_norm_eps)
        self.post_attention_layernorm = DeepseekRMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, hidden_states, position_ids=None, past_key_value=None, cache_position=None, **kwargs):
        """DeepseekV3DecoderLayer.forward (modeling_deepseek_v3.py:1188-1262).  When the injected operators expose the
        fusion hooks, input_layernorm runs inside the attention's first GEMV and both residual adds inside the epilogues
        of o_proj / the MLP's down_proj (same roundings, fewer launches); otherwise the plain sequence."""
        attn, mlp = self.self_attn, self.mlp
        if getattr(type(attn), "SUPPORTS_FUSION", False):
            hidden_states, _, past_key_value = attn(hidden_states, position_ids=position_ids, past_key_value=past_key_value,
                                                    cache_position=cache_position, pre_norm=self.input_layernorm,
                                                    residual=hidden_states)
        else:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
            hidden_states, _, past_key_value = attn(hidden_states, position_ids=position_ids,
                                                    past_key_value=past_key_value, cache_position=cache_position)
            hidden_states = residual + hidden_states
        residual = hidden_states
        if getattr(type(mlp), "SUPPORTS_FUSION", False):   # post_attention_layernorm runs inside the MLP's first launch
            return mlp(hidden_states, **{type(mlp).RESIDUAL_KW: residual, type(mlp).PRE_NORM_KW: self.post_attention_layernorm})
        hidden_states = self.post_attention_layernorm(hidden_states)
        return residual + mlp(hidden_states)


class DeepseekModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([DeepseekDecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = DeepseekRMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, input_ids=None, position_ids=None, past_key_values=None, cache_position=None, inputs_embeds=None):
        h = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        for layer in self.layers:
            h = layer(h, position_ids=position_ids, past_key_value=past_key_values, cache_position=cache_position)
        return self.norm(h)


class DeepseekForCausalLM(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = DeepseekModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    def forward(self, input_ids=None, position_ids=None, past_key_values=None, cache_position=None, inputs_embeds=None,
                last_token_only: bool = False):
        h = self.model(input_ids, position_ids, past_key_values, cache_position, inputs_embeds)
        if last_token_only:
            h = h[:, -1:, :]
        return self.lm_head(h).float()

    def greedy_next_token(self, input_ids=None, position_ids=None, past_key_values=None, cache_position=None):
        """forward(...)[:, -1].argmax(-1) — decode_one_tokens with do_sample = False (archive/ktransformers/util/utils.py:483-494)
        — with the argmax taken on the lm_head's bf16 output by one HIP launch (ktx_argmax_bf16): the fp32 copy of the logits that
        forward() returns is exact and monotonic, so the token is the same.  Returns int64 [batch]."""
        from ktransformers_amd._native import argmax_bf16

        h = self.model(input_ids, position_ids, past_key_values, cache_position)[:, -1, :]
        logits = self.lm_head(h)
        if logits.dtype != torch.bfloat16 or not logits.is_cuda:
            return logits.float().argmax(dim=-1)
        return argmax_bf16(logits)


# the reference's rule files match on these class paths
DeepseekV2RMSNorm = DeepseekV3RMSNorm = DeepseekRMSNorm
DeepseekV2YarnRotaryEmbedding = DeepseekV3YarnRotaryEmbedding = DeepseekV2RotaryEmbedding = DeepseekV3RotaryEmbedding = DeepseekRotaryEmbedding
DeepseekV2MLP = DeepseekV3MLP = DeepseekMLP
DeepseekV2MoE = DeepseekV3MoE = DeepseekMoE
DeepseekV2Attention = DeepseekV3Attention = DeepseekAttention
DeepseekV2DecoderLayer = DeepseekV3DecoderLayer = DeepseekDecoderLayer
DeepseekV2Model = DeepseekV3Model = DeepseekModel
DeepseekV2ForCausalLM = DeepseekV3ForCausalLM = DeepseekForCausalLM

"""Batched-serving MLA attention operator — mirror of archive/ktransformers/operators/balance_serve_attention.py:32-118
(`flashinfer_attn`), the attention half of the reference's balance_serve engine seam:

    forward(hidden_states [T, H], kv_cache: KDeepSeekV3Cache, position_ids [T], wrapper, num_tokens_tensors, page_idx [T],
            page_offset [T]) -> [T, H]

T = the tokens of ALL scheduled requests, flattened; the scheduler has planned `wrapper` (here `_native.MLAWrapper`, whose
plan/run signature is the flashinfer wrapper's) with the batch's qo_indptr / kv_indptr / kv_indices / kv_len and computed where
each token's latent row goes (`page_idx`, `page_offset`; KDeepSeekV3Cache.get_page_table).  The arithmetic is
KDeepseekV2Attention's — merged projections, fused RMSNorm + YaRN RoPE (`ktx_mla_prep`), cache append, batched absorb, paged
MQA over the latent cache, batched un-absorb, o_proj — only the batching contract differs, so it re-uses that operator's
pieces and kernels (tests/test_serve_attention_gpu.py)."""
from __future__ import annotations

import torch

from ktransformers_amd.operators.attention import KDeepseekV2Attention


class flashinfer_attn(KDeepseekV2Attention):
    def forward(self, hidden_states: torch.Tensor, kv_cache, position_ids: torch.Tensor, wrapper, num_tokens_tensors: torch.Tensor,
                page_idx: torch.Tensor, page_offset: torch.Tensor) -> torch.Tensor:
        from ktransformers_amd._native import mla_prep, rmsnorm

        q_len = hidden_states.shape[0]
        H, nope, rope, lora = self.num_heads, self.qk_nope_head_dim, self.qk_rope_head_dim, self.kv_lora_rank
        x = hidden_states.reshape(q_len, -1)
        if self._qkv is not None:
            op, (n0, _) = self._qkv
            qkv = op.forward(x)
            first, kv = qkv[:, :n0], qkv[:, n0:]
        else:
            first = (self.q_proj if self.q_lora_rank is None else self.q_a_proj)(x)
            kv = self.kv_a_proj_with_mqa(x)
        if self.q_lora_rank is None:
            q = first
        else:
            ln = self.q_a_layernorm
            q = self.q_b_proj(first, norm=(ln.weight, ln.variance_epsilon)) if hasattr(self.q_b_proj, "generate_linear") \
                else self.q_b_proj(rmsnorm(first, ln.weight.to(torch.bfloat16), ln.variance_epsilon, native_rounding=True))
            q = q.reshape(q_len, H * (nope + rope))
        inv_freq, mscale = self._rope_params(x.device)
        kln = self.kv_a_layernorm
        q_pe, ckv_new, kpe_new = mla_prep(q, kv, kln.weight.to(torch.bfloat16), kln.variance_epsilon,
                                          position_ids.reshape(-1).to(torch.int64), inv_freq, mscale, H, nope, rope, lora)
        cache = kv_cache.update(ckv_new, kpe_new, self.layer_idx, page_idx, page_offset)   # [pages, page, 1, lora + rope]
        qabs, oabs = self.get_absorbed()
        q_nope = qabs.forward_batched(q.unflatten(1, (H, nope + rope))[:, :, :nope])      # [T, H, lora]
        Hp = (H + 15) // 16 * 16
        if Hp != H:
            q_nope = torch.cat([q_nope, q_nope.new_zeros(q_len, Hp - H, lora)], dim=1)
            q_pe = torch.cat([q_pe, q_pe.new_zeros(q_len, Hp - H, rope)], dim=1)
        attn = wrapper.run(q_nope, q_pe, cache[:, :, 0, :lora], cache[:, :, 0, lora:])
        out = oabs.forward_batched(attn[:, :H]).reshape(q_len, H * self.v_head_dim)
        return self.o_proj(out)

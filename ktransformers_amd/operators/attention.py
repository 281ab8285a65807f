"""MLA attention operator — mirror of archive/ktransformers/operators/attention.py:47-75,349-523
(KDeepseekV2Attention.forward_linux_flashinfer; "V3 MLA is same to V2").

HF call signature and return `(attn_output, None, past_key_value)`; the replaced module supplies the sub-modules
(q_proj | q_a_proj+q_a_layernorm+q_b_proj, kv_a_proj_with_mqa, kv_a_layernorm, kv_b_proj, o_proj, rotary_emb) which the
rule file may itself have replaced by KTransformersLinear / RMSNorm / YarnRotaryEmbeddingV3.  One forward is

    q, kv_a projections            KLinear* (csrc/ktx_linear.hip)
    latent RMSNorm + RoPE(q_pe, k_pe)   one launch, ktx_mla_prep (csrc/ktx_ops.hip)
    q_nope · W_UK  (absorb)        batched bf16 linear, one launch over heads
    paged MQA over the latent + cache append     ktx_mla_decode_append (csrc/ktx_mla.hip)
    attn · W_UV^T                  batched bf16 linear
    o_proj                         KLinear*

Like the reference (attention.py:393,470-523), prompt chunks (>= 64 tokens at consecutive positions, single-request cache)
take a NON-absorbed path unless `absorb_for_prefill` is set: kv_b_proj expands the cached latents to per-head K_nope / V^T
(two batched library GEMMs) and ktx_mla_prefill runs a causal attention over qk 192 / v 128; short chunks, paged server
caches and decode use the absorbed kernel."""
from __future__ import annotations

from typing import Optional, Tuple

import os

import torch
from torch import nn

from ktransformers_amd.operators.base_operator import BaseInjectedModule
from ktransformers_amd.operators.RoPE import yarn_get_mscale
from ktransformers_amd.util.utils import InferenceState

# One decode wrapper and one prompt wrapper per device, shared by every attention layer (the layers of a model run
# serially on one stream, so they can share the split-KV workspace).  The workspace scales with the QUERY tokens of one
# call (prefill chunk), never with max_position_embeddings: a per-layer wrapper sized by the context length would need
# ~43 GB per layer at DeepSeek-V3's 163840 positions.  The decode wrapper is never re-allocated (captured graphs hold
# its workspace pointer); the prompt wrapper grows on demand (prompts are not graph-captured).
_EXPANDED_MIN_Q = 64          # prompt chunks at least this long take the non-absorbed kernel
_EXPANDED_MAX_KV = 65536      # K_nope + V^T of the context: 2 * heads * 128 * 2 B = 64 KiB per cached token at 128 heads


def _NO_IDENTITY() -> bool:      # A/B switch for timing experiments (scripts/ab_decode.py)
    import os
    return os.environ.get("KTX_MLA_NO_IDENTITY") == "1"


_DECODE_WRAPPERS: dict = {}
_PREFILL_WRAPPERS: dict = {}
PREFILL_Q_GRANULE = 1024


def _decode_wrapper(dev: torch.device):
    from ktransformers_amd._native import MLAWrapper
    w = _DECODE_WRAPPERS.get(dev)
    if w is None:
        w = _DECODE_WRAPPERS[dev] = MLAWrapper(1, 1, use_cuda_graph=True, device=dev, max_q_tokens=1)
    return w


def _prefill_wrapper(dev: torch.device, q_len: int):
    from ktransformers_amd._native import MLAWrapper
    w = _PREFILL_WRAPPERS.get(dev)
    if w is None or w.max_q_tokens < q_len:
        need = (q_len + PREFILL_Q_GRANULE - 1) // PREFILL_Q_GRANULE * PREFILL_Q_GRANULE
        w = _PREFILL_WRAPPERS[dev] = MLAWrapper(1, 1, use_cuda_graph=False, device=dev, max_q_tokens=need)
    return w


def _cache_page_arrays(cache, layer_idx: int, dev: torch.device):
    """(kv_indptr [2], kv_indices [max_pages]) of the single-request cache passed to THIS call: the page table comes from
    the cache object itself (custom_cache.py:99-104), so a larger cache handed in later is indexed with its own table."""
    memo = getattr(cache, "_ktx_kv_indptr", None)
    if memo is None:
        memo = cache._ktx_kv_indptr = {}
    key = (dev, cache.max_pages)
    if key not in memo:
        memo[key] = torch.tensor([0, cache.max_pages], dtype=torch.int32, device=dev)
    return memo[key], cache.page_table_list[layer_idx][0]


def _native_mod():
    from ktransformers_amd import _native
    return _native


class KDeepseekV2Attention(BaseInjectedModule):
    SUPPORTS_FUSION = True     # forward(..., pre_norm=, residual=)

    def __init__(self, key: str, gguf_loader, config, orig_module: nn.Module, prefill_device: str = "cuda",
                 generate_device: str = "cuda", chunck_size: int = 1000, absorb_for_prefill: bool = False, **kwargs):
        BaseInjectedModule.__init__(self, key, gguf_loader, config, orig_module, prefill_device, generate_device, **kwargs)
        c = config
        for name, val in (("chunck_size", chunck_size), ("absorb_for_prefill", absorb_for_prefill), ("mla_wrapper", None),
                          ("_absorb", None), ("_decode_plan", None), ("_qkv", None)):
            object.__setattr__(self, name, val)
        for name in ("num_heads", "q_lora_rank", "qk_rope_head_dim", "kv_lora_rank", "v_head_dim", "qk_nope_head_dim",
                     "q_head_dim", "layer_idx"):
            if not hasattr(orig_module, name):
                derived = {"num_heads": getattr(c, "num_attention_heads", None),
                           "q_head_dim": c.qk_nope_head_dim + c.qk_rope_head_dim,
                           "layer_idx": None}.get(name, getattr(c, name, None))
                setattr(orig_module, name, derived)
        if not hasattr(orig_module, "softmax_scale"):                      # modeling_deepseek_v3.py:697-703
            scale = orig_module.q_head_dim ** (-0.5)
            rs = getattr(c, "rope_scaling", None)
            if rs is not None and rs.get("mscale_all_dim", 0):
                m = yarn_get_mscale(rs["factor"], rs["mscale_all_dim"])
                scale = scale * m * m
            orig_module.softmax_scale = scale

    def load(self):
        """Children load themselves; then the two projections that read the layer input (q_proj | q_a_proj and
        kv_a_proj_with_mqa) are re-loaded as ONE row-concatenated quantised linear: same rows, one GEMV launch."""
        BaseInjectedModule.load(self)
        from ktransformers_amd.operators.linear import KTransformersLinear, build_merged_linear

        first = "q_proj" if self.q_lora_rank is None else "q_a_proj"
        a, b = getattr(self.orig_module, first), self.orig_module.kv_a_proj_with_mqa
        if (isinstance(a, KTransformersLinear) and isinstance(b, KTransformersLinear) and a.generate_linear is not None
                and type(a.generate_linear) is type(b.generate_linear)):
            merged = build_merged_linear(a.generate_linear, [f"{self.key}.{first}", f"{self.key}.kv_a_proj_with_mqa"],
                                         self.gguf_loader, a.generate_linear.device)
            if merged is not None:
                object.__setattr__(self, "_qkv", merged)
                for m in (a, b):
                    for op in {id(m.generate_linear): m.generate_linear, id(m.prefill_linear): m.prefill_linear}.values():
                        if op is not None:
                            op.unload()
                            op.loaded = True

    # ---- absorbed kv_b_proj (attention.py:67-74): W_UK [H, nope, lora] used as q_nope @ W_UK, W_UV [H, v, lora] --------
    def get_absorbed(self):
        if self._absorb is None:
            from ktransformers_amd._native import LinearHandle

            H, nope, v, lora = self.num_heads, self.qk_nope_head_dim, self.v_head_dim, self.kv_lora_rank
            w = self.kv_b_proj.weight
            if w.shape[0] != H * (nope + v):                               # KLinearTorch exposes the transposed weight
                w = w.T
            kv_b = w.reshape(H, nope + v, lora).to(torch.bfloat16)
            dev = kv_b.device
            q_absorb = kv_b[:, :nope, :]                                   # [H, nope, lora]
            out_absorb = kv_b[:, nope:, :]                                 # [H, v, lora]
            qa = LinearHandle(nope, lora, "BF16", 0, self._max_len(), dev, batch=H)
            qa.load_bf16(q_absorb.transpose(1, 2).contiguous())            # y[lora] = sum_nope x[nope] * W_UK[nope, lora]
            oa = LinearHandle(lora, v, "BF16", 0, self._max_len(), dev, batch=H)
            oa.load_bf16(out_absorb.contiguous())                          # y[v] = sum_lora x[lora] * W_UV[v, lora]
            object.__setattr__(self, "_absorb", (qa, oa))
            object.__setattr__(self, "q_absorb", q_absorb)
            object.__setattr__(self, "out_absorb", out_absorb)
        return self._absorb

    def _max_len(self) -> int:
        return int(getattr(self.config, "max_position_embeddings", 4096) or 4096)

    def _rope_params(self, device):
        rope = self.rotary_emb
        inv = getattr(rope, "inv_freq", None)
        if inv is None and hasattr(rope, "load"):
            rope.load()
            inv = rope.inv_freq
        inv = inv.to(device=device, dtype=torch.float32).contiguous()
        return inv, float(getattr(rope, "_mscale", 1.0))

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.Tensor] = None, past_key_value=None, output_attentions: bool = False,
                use_cache: bool = False, cache_position: Optional[torch.Tensor] = None, pre_norm=None, residual=None,
                **kwargs) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[Tuple[torch.Tensor]]]:
        """`pre_norm` (an RMSNorm module: the layer's input_layernorm, applied to hidden_states here instead of by the
        caller) and `residual` (added to the output) are fusion hooks used by the decoder-layer glue; without them the
        signature and behaviour are the reference's."""
        from ktransformers_amd._native import mla_prep, rmsnorm

        bsz, q_len, _ = hidden_states.size()
        if bsz != 1:
            raise ValueError("KDeepseekV2Attention: the single-request engine path runs bsz == 1 (attention.py:421-422)")
        if past_key_value is None:
            raise ValueError("KDeepseekV2Attention needs the paged latent cache (past_key_value)")
        dev = hidden_states.device
        H, nope, rope, lora = self.num_heads, self.qk_nope_head_dim, self.qk_rope_head_dim, self.kv_lora_rank
        if q_len == 1 and pre_norm is not None and residual is not None:
            fused = self._fused_decode(hidden_states, pre_norm, residual, position_ids, past_key_value)
            if fused is not None:
                return fused.reshape(bsz, q_len, -1), None, past_key_value
        x = hidden_states.reshape(q_len, -1)

        norm = None if pre_norm is None else (pre_norm.weight, pre_norm.variance_epsilon)
        if self._qkv is not None:
            op, (n0, _) = self._qkv
            qkv = op.forward(x, norm=norm) if norm is not None else op.forward(x)   # [T, n0 + lora + rope]
            first, kv = qkv[:, :n0], qkv[:, n0:]
        else:
            if norm is not None:
                x = rmsnorm(x.contiguous(), norm[0], norm[1], native_rounding=True)
            first = (self.q_proj if self.q_lora_rank is None else self.q_a_proj)(x)
            kv = self.kv_a_proj_with_mqa(x)
        inv_freq, mscale = self._rope_params(dev)
        pos = position_ids.reshape(-1).to(torch.int64)
        kln = self.kv_a_layernorm
        qabs, oabs = self.get_absorbed()
        fused_q = None   # decode: q_b_proj + q-absorb + RoPE + the kv half of mla_prep in one launch (ktx_linear_forward_qb_absorb)
        if self.q_lora_rank is None:
            q = first
        else:
            ln = self.q_a_layernorm                                        # DeepseekV3RMSNorm.forward (native rounding)
            qb_h = self._decode_qb_handle(q_len, first, kv, qabs)
            if qb_h is not None:
                from ktransformers_amd._native import qb_absorb_and_prep
                q = None
                fused_q = qb_absorb_and_prep(qb_h, qabs, first, (ln.weight.to(torch.bfloat16), ln.variance_epsilon), kv,
                                             kln.weight.to(torch.bfloat16), kln.variance_epsilon, pos, inv_freq, mscale,
                                             H, nope, rope, lora)
            else:
                q = self.q_b_proj(first, norm=(ln.weight, ln.variance_epsilon)) if hasattr(self.q_b_proj, "generate_linear") \
                    else self.q_b_proj(rmsnorm(first, ln.weight.to(torch.bfloat16), ln.variance_epsilon, native_rounding=True))
                q = q.reshape(q_len, H * (nope + rope))
        capacity = past_key_value.max_pages * past_key_value.page_size
        expanded = None      # (first, last) position when this prompt chunk takes the non-absorbed kernel
        if self._expanded_prefill_ok(q_len, past_key_value):
            span = self._prompt_span(position_ids, pos, past_key_value)
            if span[1] - span[0] + 1 == q_len and span[1] < min(capacity, _EXPANDED_MAX_KV):
                expanded = span
        if expanded is not None:
            # prompt chunk at consecutive positions: expand the context's latents through kv_b_proj and run the causal
            # attention over qk 192 / v 128 (attention.py:58-164 forward_chunck) — 3.4x fewer flop than the absorbed form,
            # and no absorb products at all
            q_pe, ckv_new, kpe_new = mla_prep(q, kv, kln.weight.to(torch.bfloat16), kln.variance_epsilon, pos, inv_freq, mscale,
                                              H, nope, rope, lora)
            cp = cache_position if cache_position is not None else pos
            past_key_value.update(ckv_new, kpe_new, self.layer_idx, {"cache_position": cp})
            out = self._expanded_prefill(q, q_pe, past_key_value.key_cache[self.layer_idx], expanded[1] + 1, q_len)
            return self._project_out(out, residual, bsz, q_len), None, past_key_value
        if fused_q is not None:
            q_nope, q_pe, ckv_new, kpe_new = fused_q
        elif q_len <= 4 and q.stride(-1) == 1 and kv.stride(-1) == 1 and qabs.decode_eligible(q_len) \
                and not os.environ.get("KTX_MLA_SEPARATE_PREP"):     # (dev A/B switch, scripts/ab_decode.py)
            # decode: latent RMSNorm + RoPE ride in the launch of the q-absorb products (independent work, one boundary less)
            from ktransformers_amd._native import absorb_and_prep
            q_nope, q_pe, ckv_new, kpe_new = absorb_and_prep(qabs, q, kv, kln.weight.to(torch.bfloat16), kln.variance_epsilon, pos,
                                                             inv_freq, mscale, H, nope, rope, lora)
        else:
            q_pe, ckv_new, kpe_new = mla_prep(q, kv, kln.weight.to(torch.bfloat16), kln.variance_epsilon, pos, inv_freq, mscale,
                                              H, nope, rope, lora)
            q3 = q.unflatten(1, (H, nope + rope))                          # strided view when q is a slice of the merged GEMV
            q_nope = qabs.forward_batched(q3[:, :, :nope])                 # [T, H, lora]

        Hp = (H + 15) // 16 * 16          # the MLA kernel tiles heads by 16 (every shipped model: 16 / 64 / 128 heads)
        if Hp != H:
            q_nope = torch.cat([q_nope, q_nope.new_zeros(q_len, Hp - H, lora)], dim=1)
            q_pe = torch.cat([q_pe, q_pe.new_zeros(q_len, Hp - H, rope)], dim=1)
        cache = past_key_value.key_cache[self.layer_idx]                   # [pages, page, 1, lora + rope]
        ckv_pages = cache[:, :, 0, :lora]
        kpe_pages = cache[:, :, 0, lora:]
        kv_indptr, kv_indices = _cache_page_arrays(past_key_value, self.layer_idx, dev)
        object.__setattr__(self, "mla_wrapper", _decode_wrapper(dev) if q_len == 1 else _prefill_wrapper(dev, q_len))
        if q_len == 1:
            # kv_len is read on the device: positions + 1 (attention.py:430-433); the kernel appends the new row itself.
            # Every layer of a step sees the same position tensor: derive kv_len once per step, not once per layer.
            # (keyed on the capture state too: a value computed in an eager warm-up must not leak into a captured graph)
            kv_len, _ = self._kv_len_of(position_ids, past_key_value)
            # split count: a launch-grid constant of the captured graph — size it for the context seen now plus headroom
            # (longer contexts later just put several tiles in a split)
            seen = int(past_key_value.get_seq_length(self.layer_idx))
            if seen + 1 > capacity:     # the reference's indexed assignment raises here (custom_cache.py:189-195)
                raise IndexError(f"KDeepseekV2Attention: position {seen} is beyond the cache ({capacity} tokens)")
            hint = min(seen + 512, capacity)
            self.mla_wrapper.plan(None, kv_indptr, kv_indices, kv_len, None, Hp, lora, rope, past_key_value.page_size,
                                  self.softmax_scale, torch.bfloat16, torch.bfloat16, max_kv_len=hint,
                                  identity_pages=bool(getattr(past_key_value, "identity_page_table", False)) and not _NO_IDENTITY())
            fuse_merge = (Hp == H and not os.environ.get("KTX_MLA_SEPARATE_MERGE")
                          and bool(_native_mod().lib.ktx_linear_merge_eligible(oabs._h, q_len, 1, H)))
            if fuse_merge:
                # the merge of the KV splits rides in the launch of the un-absorb products (one workgroup per head)
                parts = self.mla_wrapper.run_partials(q_nope, q_pe, ckv_pages, kpe_pages, new_ckv=ckv_new, new_kpe=kpe_new)
                past_key_value.note_appended(self.layer_idx, 1)
                object.__setattr__(self, "_decode_plan", kv_len)
                out = _native_mod().merge_and_unabsorb(oabs, parts, q_len, H)
                return self._project_out(out.reshape(q_len, H * self.v_head_dim), residual, bsz, q_len), None, past_key_value
            attn = self.mla_wrapper.run(q_nope, q_pe, ckv_pages, kpe_pages, new_ckv=ckv_new, new_kpe=kpe_new)
            past_key_value.note_appended(self.layer_idx, 1)
            object.__setattr__(self, "_decode_plan", kv_len)
        else:
            cp = cache_position if cache_position is not None else pos
            past_key_value.update(ckv_new, kpe_new, self.layer_idx, {"cache_position": cp})
            qo_indptr = torch.tensor([0, q_len], dtype=torch.int32, device=dev)
            kv_len = (pos[-1:] + 1).to(torch.int32)
            self.mla_wrapper.plan(qo_indptr, kv_indptr, kv_indices, kv_len, None, Hp, lora, rope, past_key_value.page_size,
                                  self.softmax_scale, torch.bfloat16, torch.bfloat16,
                                  identity_pages=bool(getattr(past_key_value, "identity_page_table", False)) and not _NO_IDENTITY())
            attn = self.mla_wrapper.run(q_nope, q_pe, ckv_pages, kpe_pages)
        out = oabs.forward_batched(attn[:, :H])                             # [T, H, v]
        return self._project_out(out.reshape(q_len, H * self.v_head_dim), residual, bsz, q_len), None, past_key_value

    # ---- the whole decode step as ONE launch (csrc/ktx_attn.hip) -------------------------------------------------------------
    @staticmethod
    def _gen_handle(mod):
        """The W4 / block-FP8 LinearHandle a KTransformersLinear (or a merged operator) decodes with, or None."""
        lin = mod
        if hasattr(mod, "generate_linear"):
            lin = mod.generate_linear if getattr(mod, "mode", None) == InferenceState.GENERATE else getattr(mod, "prefill_linear", None)
        h = getattr(lin, "_h", None)
        return h if h is not None and getattr(h, "fmt", None) in ("W4", "FP8") else None

    def _kv_len_of(self, position_ids, past_key_value):
        """kv_len = positions + 1 on the device, derived once per step (every layer sees the same position tensor)."""
        key = (id(position_ids), position_ids._version, torch.cuda.is_current_stream_capturing())
        memo = getattr(past_key_value, "_kv_len_memo", None)
        if memo is not None and memo[0] == key and memo[1] is position_ids:
            return memo[2], memo[3]
        pos = position_ids.reshape(-1).to(torch.int64)
        kv_len = (pos + 1).to(torch.int32)
        past_key_value._kv_len_memo = (key, position_ids, kv_len, pos)
        return kv_len, pos

    def _fused_decode(self, hidden_states, pre_norm, residual, position_ids, past_key_value):
        """q_a|kv_a -> q_b + absorb + RoPE + latent norm + cache append -> split-KV attention -> merge + un-absorb -> o_proj +
        residual as ONE persistent launch (include/ktx_attn.h) when this layer has the covered geometry (DeepSeek-V3 / R1 / Kimi-K2
        attention dimensions: 128 or 64 heads; W4 g64 or block-FP8 projections, all three alike; the identity / paged single-request
        cache); None otherwise — the caller then takes the
        five-launch path, whose kernels this launch restates bit for bit.  KTX_ATTN_SEPARATE=1 forces the five launches (A/B)."""
        ok = getattr(self, "_fused_ok", None)
        if ok is False or os.environ.get("KTX_ATTN_SEPARATE") or self.q_lora_rank is None or self._qkv is None:
            return None
        if (hidden_states.dtype != torch.bfloat16 or not hidden_states.is_contiguous() or residual.data_ptr() != hidden_states.data_ptr()
                or residual.shape != hidden_states.shape or past_key_value is None):
            return None
        from ktransformers_amd import _native as N
        dev = hidden_states.device
        H, nope, rope, lora = self.num_heads, self.qk_nope_head_dim, self.qk_rope_head_dim, self.kv_lora_rank
        handles = (self._gen_handle(self._qkv[0]), self._gen_handle(self.q_b_proj), self._gen_handle(self.o_proj))
        if any(h is None for h in handles):
            object.__setattr__(self, "_fused_ok", False)
            return None
        qabs, oabs = self.get_absorbed()
        capacity = past_key_value.max_pages * past_key_value.page_size
        seen = int(past_key_value.get_seq_length(self.layer_idx))
        if seen + 1 > capacity:     # the reference's indexed assignment raises here (custom_cache.py:189-195)
            raise IndexError(f"KDeepseekV2Attention: position {seen} is beyond the cache ({capacity} tokens)")
        hint = min(seen + 512, capacity)
        kv_len, pos = self._kv_len_of(position_ids, past_key_value)
        inv_freq, mscale = self._rope_params(dev)
        cache = past_key_value.key_cache[self.layer_idx]                   # [pages, page, 1, lora + rope]
        kv_indptr, kv_indices = _cache_page_arrays(past_key_value, self.layer_idx, dev)
        identity = bool(getattr(past_key_value, "identity_page_table", False)) and not _NO_IDENTITY()
        ln, kln = self.q_a_layernorm, self.kv_a_layernorm
        out = torch.empty_like(hidden_states)
        keep = (pre_norm.weight.to(torch.bfloat16), ln.weight.to(torch.bfloat16), kln.weight.to(torch.bfloat16))
        args = N.attn_decode_args(handles[0], handles[1], qabs, oabs, handles[2], hidden_states.reshape(-1), out.reshape(-1),
                                  (keep[0], pre_norm.variance_epsilon), (keep[1], ln.variance_epsilon), (keep[2], kln.variance_epsilon),
                                  pos, inv_freq, mscale, H, nope, rope, lora, self.v_head_dim, cache[:, :, 0, :lora], cache[:, :, 0, lora:],
                                  past_key_value.page_size, kv_indptr, None if identity else kv_indices, kv_len, hint, self.softmax_scale)
        # eligibility depends on the per-call context bound (`hint` picks the split shape), so it is asked on every call — a
        # host-only check.  A context that leaves the covered range mid-generation falls back to the five launches (same arithmetic),
        # and comes back when a later request is short again.  The STATIC part (formats, dimensions, CU count) is settled once
        # (ADVICE r5): a layer that is not covered even at the smallest context bound never will be, and stops building arguments.
        if not N.attn_decode_eligible(args):
            if ok is None:
                args.kv_len_hint = min(512, capacity)
                if not N.attn_decode_eligible(args):
                    object.__setattr__(self, "_fused_ok", False)
            return None
        if ok is None:
            object.__setattr__(self, "_fused_ok", True)
        N.attn_decode(args, dev)
        past_key_value.note_appended(self.layer_idx, 1)
        object.__setattr__(self, "_decode_plan", kv_len)
        return out

    def _decode_qb_handle(self, q_len, q_a, kv, qabs):
        """The q_b_proj LinearHandle when this call can take the combined q_b + q-absorb launch: a decode step through the
        absorbed kernel, W4 q_b_proj, the shapes ktx_linear_qb_absorb_eligible knows."""
        if q_len > 4 or os.environ.get("KTX_MLA_SEPARATE_QB") or q_a.stride(-1) != 1 or kv.stride(-1) != 1:
            return None
        if q_a.dtype != torch.bfloat16 or q_a.stride(0) % 8 != 0 or kv.stride(0) % 8 != 0:
            return None
        qb = self.q_b_proj
        lin = getattr(qb, "generate_linear", None) if getattr(qb, "mode", None) == InferenceState.GENERATE else \
            getattr(qb, "prefill_linear", None)
        h = getattr(lin, "_h", None)
        if h is None or getattr(h, "fmt", None) != "W4":
            return None
        from ktransformers_amd._native import qb_absorb_eligible
        return h if qb_absorb_eligible(h, qabs, q_len, self.num_heads, self.qk_nope_head_dim, self.qk_rope_head_dim,
                                       self.kv_lora_rank) else None

    def _project_out(self, out, residual, bsz, q_len):
        if residual is not None and hasattr(self.o_proj, "generate_linear"):
            out = self.o_proj(out, add1=residual.reshape(q_len, -1))        # hidden = residual + attn (o_proj epilogue)
        else:
            out = self.o_proj(out)
            if residual is not None:
                out = residual.reshape(q_len, -1) + out
        return out.reshape(bsz, q_len, -1)

    # ---- non-absorbed prompt attention ------------------------------------------------------------------------------------
    def _expanded_prefill_ok(self, q_len, cache) -> bool:
        return (q_len >= _EXPANDED_MIN_Q and self.qk_nope_head_dim == 128 and self.v_head_dim == 128 and self.qk_rope_head_dim == 64
                and bool(getattr(cache, "identity_page_table", False)) and not self.absorb_for_prefill
                and not os.environ.get("KTX_MLA_ABSORBED_PREFILL"))

    @staticmethod
    def _prompt_span(position_ids, pos, cache):
        """(first, last) position of the chunk on the host: one device read per forward pass (every layer sees the same
        position tensor), prompt processing is not graph-captured."""
        key = (id(position_ids), position_ids._version)
        memo = getattr(cache, "_span_memo", None)
        if memo is not None and memo[0] == key and memo[1] is position_ids:
            return memo[2]
        first, last = (int(v) for v in torch.stack((pos[0], pos[-1])).tolist())
        cache._span_memo = (key, position_ids, (first, last))
        return first, last

    def _expanded_prefill(self, q, q_pe, cache, kv_len, q_len):
        from ktransformers_amd._native import mla_prefill

        H, nope, rope, lora = self.num_heads, self.qk_nope_head_dim, self.qk_rope_head_dim, self.kv_lora_rank
        rows = cache.view(-1, lora + rope)                                  # the single-request cache is in token order
        kv_pad = (kv_len + 63) // 64 * 64
        lat = rows.new_zeros((kv_pad, lora))
        lat[:kv_len] = rows[:kv_len, :lora]                                 # zero rows past the context: exact-zero K / V there
        # kv_b_proj as two batched GEMMs of the library's own kernel (csrc/ktx_gemm.hip; the latent rows are the shared operand):
        # K_nope [H, kv, 128] = latent @ W_UK[h]^T and V^T [H, 128, kv] = W_UV[h] @ latent^T.  KTX_VENDOR_GEMM=1: torch.matmul (A/B)
        if os.environ.get("KTX_VENDOR_GEMM"):
            k_nope = torch.matmul(lat.unsqueeze(0), self.q_absorb.transpose(1, 2))
            v_t = torch.matmul(self.out_absorb, lat.t().unsqueeze(0))
        else:
            from ktransformers_amd._native import gemm_bf16_nt
            k_nope = gemm_bf16_nt(lat, self.q_absorb)
            v_t = gemm_bf16_nt(self.out_absorb, lat)
        q3 = q.unflatten(1, (H, nope + rope))
        attn = mla_prefill(q3[:, :, :nope], q_pe, k_nope.contiguous(), rows[:, lora:], v_t.contiguous(), kv_len, self.softmax_scale)
        return attn.reshape(q_len, H * self.v_head_dim)


KDeepseekV3Attention = KDeepseekV2Attention

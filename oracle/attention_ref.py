"""TEST INFRASTRUCTURE ONLY — torch (CPU) restatement of the reference's absorbed MLA attention operator, op for op and in
the model dtype (bf16) like the reference runs it:

  KDeepseekV2Attention.forward_linux_flashinfer  archive/ktransformers/operators/attention.py:349-469
  DeepseekV3RMSNorm.forward                      archive/ktransformers/models/modeling_deepseek_v3.py:111-116
  YarnRotaryEmbeddingV3.forward / _init          archive/ktransformers/operators/RoPE.py:262-326
  apply_rotary_pos_emb (DeepSeek de-interleave)  archive/ktransformers/models/modeling_deepseek.py:337-366
  softmax_scale                                  archive/ktransformers/models/modeling_deepseek_v3.py:697-703
  the MLA core = attention_ref_torch             archive/ktransformers/operators/flashinfer_wrapper.py:30-76

Linears are dense bf16 `F.linear` (what KLinearTorch computes); the latent cache is a plain [n, 576] history tensor."""
import math

import torch
import torch.nn.functional as F

from oracle.mla_ref import attention_ref_torch


def rmsnorm_ref(x, w, eps):
    dt = x.dtype
    h = x.to(torch.float32)
    h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + eps)
    return w * h.to(dt)


def _yarn_find_correction_dim(num_rotations, dim, base, max_pos):
    return (dim * math.log(max_pos / (num_rotations * 2 * math.pi))) / (2 * math.log(base))


def yarn_get_mscale(scale=1, mscale=1):
    return 1.0 if scale <= 1 else 0.1 * mscale * math.log(scale) + 1.0


def rope_tables(cfg, position_ids, dtype):
    """(cos, sin) [T, rope_dim] as YarnRotaryEmbeddingV3.forward / RotaryEmbeddingV3.forward return them."""
    dim, base = cfg.qk_rope_head_dim, cfg.rope_theta
    rs = getattr(cfg, "rope_scaling", None)
    ar = torch.arange(0, dim, 2, dtype=torch.float32) / dim
    if rs is None:
        inv_freq, mscale = 1.0 / (base ** ar), 1.0
    else:
        f = rs["factor"]
        freq_extra, freq_inter = 1.0 / (base ** ar), 1.0 / (f * base ** ar)
        low = max(math.floor(_yarn_find_correction_dim(rs.get("beta_fast", 32), dim, base, rs.get("original_max_position_embeddings", 4096))), 0)
        high = min(math.ceil(_yarn_find_correction_dim(rs.get("beta_slow", 1), dim, base, rs.get("original_max_position_embeddings", 4096))), dim - 1)
        hi = high + 0.001 if low == high else high
        ramp = torch.clamp((torch.arange(dim // 2, dtype=torch.float32) - low) / (hi - low), 0, 1)
        mask = 1.0 - ramp
        inv_freq = freq_inter * (1 - mask) + freq_extra * mask
        mscale = float(yarn_get_mscale(f, rs.get("mscale", 1)) / yarn_get_mscale(f, rs.get("mscale_all_dim", 0)))
    freqs = position_ids.reshape(-1, 1).float() * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return (emb.cos() * mscale).to(dtype), (emb.sin() * mscale).to(dtype), inv_freq, mscale


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(x, cos, sin):
    """x [T, H, d]; cos/sin [T, d].  De-interleave then x*cos + rotate_half(x)*sin in x.dtype."""
    T, H, d = x.shape
    x = x.view(T, H, d // 2, 2).transpose(3, 2).reshape(T, H, d)
    return (x * cos[:, None, :]) + (rotate_half(x) * sin[:, None, :])


def softmax_scale(cfg):
    s = (cfg.qk_nope_head_dim + cfg.qk_rope_head_dim) ** (-0.5)
    rs = getattr(cfg, "rope_scaling", None)
    if rs is not None and rs.get("mscale_all_dim", 0):
        m = yarn_get_mscale(rs["factor"], rs["mscale_all_dim"])
        s = s * m * m
    return s


def mla_attention_ref(cfg, w, hidden, position_ids, history, lin=None):
    """hidden bf16 [T, hidden]; position_ids int64 [T]; history bf16 [n_past, 576] (already rotated/normalised rows).
    w: dict of bf16 weights (q_proj | q_a_proj,q_a_layernorm,q_b_proj; kv_a_proj_with_mqa; kv_a_layernorm; kv_b_proj; o_proj).
    lin(name, x): the quantised projection `name` applied to x (default: dense bf16 F.linear on w[name]) — KLinearFP8 layers pass
    oracle.linear_ref.linear_fp8_ref on their e4m3 blocks (activation quantisation included, linear.py:408-413).
    Returns (out bf16 [T, hidden], new history rows bf16 [T, 576])."""
    if lin is None:
        lin = lambda name, x: F.linear(x, w[name])   # noqa: E731
    T = hidden.shape[0]
    H, nope, rope, lora, v = cfg.num_attention_heads, cfg.qk_nope_head_dim, cfg.qk_rope_head_dim, cfg.kv_lora_rank, cfg.v_head_dim
    eps = getattr(cfg, "rms_norm_eps", 1e-6)
    if getattr(cfg, "q_lora_rank", None) is None:
        q = lin("q_proj", hidden)
    else:
        q = lin("q_b_proj", rmsnorm_ref(lin("q_a_proj", hidden), w["q_a_layernorm"], eps))
    q = q.view(T, H, nope + rope)
    q_nope, q_pe = torch.split(q, [nope, rope], dim=-1)
    ckv = lin("kv_a_proj_with_mqa", hidden)
    ckv, k_pe = torch.split(ckv, [lora, rope], dim=-1)
    ckv = rmsnorm_ref(ckv, w["kv_a_layernorm"], eps)
    cos, sin, _, _ = rope_tables(cfg, position_ids, hidden.dtype)
    q_pe = apply_rope(q_pe, cos, sin)
    k_pe = apply_rope(k_pe.view(T, 1, rope), cos, sin).view(T, rope)
    new_rows = torch.cat([ckv, k_pe], dim=-1)
    lat = torch.cat([history, new_rows], dim=0)                      # cache after update()
    kv_b = w["kv_b_proj"].view(H, nope + v, lora)
    q_absorb, out_absorb = kv_b[:, :nope, :], kv_b[:, nope:, :]
    q_nope = torch.matmul(q_nope.transpose(0, 1), q_absorb).transpose(0, 1).contiguous()      # [T, H, lora]
    n = lat.shape[0]
    k = lat.view(n, 1, lora + rope).repeat_interleave(H, dim=1)
    vv = lat[:, :lora].reshape(n, 1, lora).repeat_interleave(H, dim=1)
    attn, _ = attention_ref_torch(1, torch.cat([q_nope, q_pe], dim=-1), k, vv, True, softmax_scale(cfg))
    attn = attn.to(hidden.dtype)                                     # the wrapper returns bf16
    attn = torch.matmul(attn.transpose(0, 1), out_absorb.mT).transpose(0, 1).contiguous()      # [T, H, v]
    out = lin("o_proj", attn.reshape(T, H * v))
    return out, new_rows

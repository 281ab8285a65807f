"""GPU: KDeepseekV2Attention (operators/attention.py) against the reference's eager attention (golden fixture) and the
restated absorbed operator (oracle/attention_ref.py).  bf16 pipeline, fp32 accumulation: the bound is bf16 noise —
norm-wise <= 2e-2 vs the reference's fp32 run (its own bf16 run is ~1e-2 away), <= 1.5e-2 vs the bf16 oracle; the latent
rows written to the cache are compared against the oracle to 1 bf16 ulp."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from attn_helpers import ToyAttention, load_golden, make_cfg  # noqa: E402
from oracle.attention_ref import mla_attention_ref  # noqa: E402


def build(cfg, w, linear_op=None, absorb_for_prefill=True):
    from ktransformers_amd.models.custom_cache import StaticCache
    from ktransformers_amd.operators.attention import KDeepseekV2Attention
    from ktransformers_amd.operators.linear import KTransformersLinear
    from ktransformers_amd.operators.RoPE import YarnRotaryEmbeddingV3
    from ktransformers_amd.util.loader import DictLoader
    from ktransformers_amd.util.utils import InferenceState

    dev = "cuda:0"
    orig = ToyAttention(cfg, w, dev)
    loader = DictLoader({f"attn.{k}.weight": v for k, v in w.items()})
    rope = YarnRotaryEmbeddingV3("attn.rotary_emb", loader, cfg, orig.rotary_emb, dev, dev)
    rope.load()
    orig.rotary_emb = rope
    if linear_op is not None:
        for name in ("q_proj", "q_a_proj", "q_b_proj", "kv_a_proj_with_mqa", "o_proj"):
            if hasattr(orig, name):
                lin = KTransformersLinear(f"attn.{name}", loader, cfg, getattr(orig, name), dev, linear_op, dev, linear_op)
                lin.load(mode=InferenceState.GENERATE)
                setattr(orig, name, lin)
    attn = KDeepseekV2Attention("attn", loader, cfg, orig, dev, dev, absorb_for_prefill=absorb_for_prefill)
    cache = StaticCache(cfg, 1, 4096, dev, torch.bfloat16)
    return attn, cache


def rows_close(rows, ref):
    """cache rows vs the oracle: 1 bf16 ulp, plus the absolute slack of a rope sum that cancels (|a|+|b| ~ 4)."""
    d = (rows - ref).abs()
    bad = d > 2.0 ** -7 * ref.abs() + 2.0 ** -6
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} cache elements off, max diff {float(d.max())} at {bad.nonzero()[:4].tolist()}"


def rel(a, b):
    return float((a.float().cpu() - b.float()).norm() / b.float().norm())


@pytest.mark.parametrize("name", ["v3", "v2lite"])
@pytest.mark.parametrize("linear_op", [None, "KLinearTorch"])
def test_prefill_and_decode_match_reference(name, linear_op):
    cfg, w, x, y_bf16, y_f32 = load_golden(name)
    T = x.shape[0]
    oracle_out, oracle_rows = mla_attention_ref(cfg, w, x, torch.arange(T), torch.zeros(0, 576, dtype=torch.bfloat16))
    attn, cache = build(cfg, w, linear_op)
    xg = x.cuda()
    pos = torch.arange(T, device="cuda")
    out, _, _ = attn(xg[None], position_ids=pos[None], past_key_value=cache, cache_position=pos)
    assert out.shape == (1, T, cfg.hidden_size)
    assert rel(out[0], y_f32) < 2e-2
    assert rel(out[0], oracle_out) < 1.5e-2
    rows = cache.key_cache[0].reshape(-1, 576)[:T].float().cpu()
    rows_close(rows, oracle_rows.float())
    assert cache.get_seq_length(0) == T
    # token-by-token decode (kernel-appended cache rows) reproduces the prompt pass
    cache.reset()
    outs = []
    for t in range(T):
        o, _, _ = attn(xg[None, t:t + 1], position_ids=pos[None, t:t + 1], past_key_value=cache, cache_position=pos[t:t + 1])
        outs.append(o[0])
    dec = torch.cat(outs, 0)
    assert rel(dec, y_f32) < 2e-2
    assert rel(dec, oracle_out) < 1.5e-2
    rows2 = cache.key_cache[0].reshape(-1, 576)[:T].float().cpu()
    rows_close(rows2, oracle_rows.float())


def test_non_absorbed_prompt_path_matches_reference_and_absorbed():
    """The 70-token golden prompt through the expanded (kv_b_proj + causal qk-192 attention) path, which is what the reference's
    eager module computes: same bounds as the absorbed path, and the two paths agree with each other to bf16 noise."""
    cfg, w, x, y_bf16, y_f32 = load_golden("v3")
    T = x.shape[0]
    assert T >= 64
    oracle_out, oracle_rows = mla_attention_ref(cfg, w, x, torch.arange(T), torch.zeros(0, 576, dtype=torch.bfloat16))
    pos = torch.arange(T, device="cuda")
    outs = {}
    for absorbed in (True, False):
        attn, cache = build(cfg, w, None, absorb_for_prefill=absorbed)
        out, _, _ = attn(x.cuda()[None], position_ids=pos[None], past_key_value=cache, cache_position=pos)
        outs[absorbed] = out[0]
        assert rel(out[0], y_f32) < 2e-2
        assert rel(out[0], oracle_out) < 1.5e-2
        rows_close(cache.key_cache[0].reshape(-1, 576)[:T].float().cpu(), oracle_rows.float())
    assert rel(outs[False], outs[True].float().cpu()) < 1.5e-2


def test_non_absorbed_chunked_prompt_against_the_oracle():
    """Two prompt chunks of 100 tokens: the second attends over the first chunk's cached latents too (kv_len 200 > q_len 100),
    and a decode step afterwards reads the rows both chunks wrote."""
    cfg, w, _, _, _ = load_golden("v3")
    torch.manual_seed(3)
    x = (torch.randn(201, cfg.hidden_size) * 0.5).to(torch.bfloat16)
    attn, cache = build(cfg, w, None, absorb_for_prefill=False)
    hist = torch.zeros(0, 576, dtype=torch.bfloat16)
    for a, b in ((0, 100), (100, 200), (200, 201)):
        want, rows = mla_attention_ref(cfg, w, x[a:b], torch.arange(a, b), hist)
        hist = torch.cat([hist, rows], 0)
        pos = torch.arange(a, b, device="cuda")
        got, _, _ = attn(x[a:b].cuda()[None], position_ids=pos[None], past_key_value=cache, cache_position=pos)
        assert rel(got[0], want) < 1.5e-2, (a, b, rel(got[0], want))
    rows_close(cache.key_cache[0].reshape(-1, 576)[:201].float().cpu(), hist.float())


def test_marlin_linears_track_the_quantised_oracle():
    """With KLinearMarlin projections the operator follows the oracle evaluated on the de-quantised weights."""
    from oracle.linear_ref import dequant_w4, quantize_weights_ref

    cfg, w, x, _, _ = load_golden("v3")
    wq = dict(w)
    for name in ("q_a_proj", "q_b_proj", "kv_a_proj_with_mqa", "o_proj"):
        if w[name].shape[1] % 64 == 0:
            q, s = quantize_weights_ref(w[name].T.contiguous(), 64)
            wq[name] = dequant_w4(q, s, 64, False).T.contiguous().to(torch.bfloat16)
    T = 5
    oracle_out, _ = mla_attention_ref(cfg, wq, x[:T], torch.arange(T), torch.zeros(0, 576, dtype=torch.bfloat16))
    attn, cache = build(cfg, w, "KLinearMarlin")
    pos = torch.arange(T, device="cuda")
    out, _, _ = attn(x[:T].cuda()[None], position_ids=pos[None], past_key_value=cache, cache_position=pos)
    assert rel(out[0], oracle_out) < 2e-2


def test_requires_cache_and_single_request():
    cfg, w, x, _, _ = load_golden("v2lite")
    attn, cache = build(cfg, w)
    with pytest.raises(ValueError):
        attn(x[None, :1].cuda(), position_ids=torch.zeros(1, 1, dtype=torch.long, device="cuda"), past_key_value=None)
    with pytest.raises(ValueError):
        attn(x[None, :1].cuda().repeat(2, 1, 1), position_ids=torch.zeros(2, 1, dtype=torch.long, device="cuda"),
             past_key_value=cache)


def test_real_context_length_and_a_larger_cache_later():
    """DeepSeek's real max_position_embeddings (163840) must not size any per-layer workspace, and a second, LARGER cache
    handed to the same operator is indexed with its own page table (the reference allocates a StaticCache per request)."""
    from ktransformers_amd.models.custom_cache import StaticCache

    cfg, w, x, _, y_f32 = load_golden("v2lite")
    cfg.max_position_embeddings = 163840
    attn, _ = build(cfg, w)
    T = x.shape[0]
    xg, pos = x.cuda(), torch.arange(T, device="cuda")
    before = torch.cuda.memory_allocated()
    small = StaticCache(cfg, 1, 128, "cuda:0", torch.bfloat16)
    out1, _, _ = attn(xg[None], position_ids=pos[None], past_key_value=small, cache_position=pos)
    assert rel(out1[0], y_f32) < 2e-2
    big = StaticCache(cfg, 1, 4096, "cuda:0", torch.bfloat16)     # more pages than the first cache had
    shift = 1000                                                   # rows land in pages the small cache never had
    pos2 = pos + shift
    outs = []
    big.past_tokens[0] = shift
    for t in range(T):      # decode path: kernel-side append through the big cache's own page table
        o, _, _ = attn(xg[None, t:t + 1], position_ids=pos2[None, t:t + 1], past_key_value=big, cache_position=pos2[t:t + 1])
        outs.append(o[0])
    torch.cuda.synchronize()
    rows = big.key_cache[0].reshape(-1, 576)
    assert rows[shift:shift + T].abs().sum() > 0 and rows[shift + T:].abs().sum() == 0
    assert torch.isfinite(torch.cat(outs).float()).all()
    # everything this test allocated beyond the two caches stays far below "one workspace per layer sized by 163840 positions"
    assert torch.cuda.memory_allocated() - before < (1 << 30)


def test_cache_bounds_raise_instead_of_writing_out_of_range():
    from ktransformers_amd.models.custom_cache import StaticCache

    cfg, w, x, _, _ = load_golden("v2lite")
    attn, _ = build(cfg, w)
    cache = StaticCache(cfg, 1, 64, "cuda:0", torch.bfloat16)      # one page
    xg = x.cuda()
    pos = torch.arange(70, device="cuda")
    with pytest.raises(IndexError):                                # 70-token prompt into a 64-token cache
        attn(xg[None, :1].repeat(1, 70, 1), position_ids=pos[None], past_key_value=cache, cache_position=pos)
    cache.reset()
    cache.past_tokens[0] = 64
    with pytest.raises(IndexError):                                # decode step past the last page
        attn(xg[None, :1], position_ids=pos[None, 64:65], past_key_value=cache, cache_position=pos[64:65])


def test_decode_combined_launches_at_v3_head_dims():
    """The decode step of the operator at DeepSeek-V3's attention dimensions (128 heads, q_lora 1536, W4 projections) takes the
    combined launches — q_b_proj + q-absorb + RoPE + latent norm (ktx_linear_forward_qb_absorb) and KV-split merge + un-absorb
    (ktx_linear_forward_batched_merge): same outputs as the separate launches up to fp32 re-association, identical cache rows."""
    import os
    from attn_helpers import make_cfg
    from ktransformers_amd import _native
    cfg = make_cfg(2048, 128, 1536)
    g = torch.Generator().manual_seed(3)
    H, qh = 128, 192

    def rnd(n, k, s):
        return (torch.randn((n, k), generator=g) * s).to(torch.bfloat16)
    w = {"q_a_proj": rnd(1536, 2048, 2048 ** -0.5), "q_b_proj": rnd(H * qh, 1536, 1536 ** -0.5),
         "kv_a_proj_with_mqa": rnd(576, 2048, 2048 ** -0.5), "kv_b_proj": rnd(H * 256, 512, 512 ** -0.5),
         "o_proj": rnd(2048, H * 128, (H * 128) ** -0.5),
         "q_a_layernorm": (1 + 0.1 * torch.randn(1536, generator=g)).to(torch.bfloat16),
         "kv_a_layernorm": (1 + 0.1 * torch.randn(512, generator=g)).to(torch.bfloat16)}
    x = torch.randn((44, 2048), generator=g).to(torch.bfloat16).cuda()
    pos = torch.arange(44, device="cuda")
    outs, rows = {}, {}
    for mode in ("combined", "separate"):
        for k in ("KTX_MLA_SEPARATE_QB", "KTX_MLA_SEPARATE_MERGE"):
            os.environ.pop(k, None)
            if mode == "separate":
                os.environ[k] = "1"
        try:
            attn, cache = build(cfg, w, "KLinearMarlin")
            attn(x[None, :40], position_ids=pos[None, :40], past_key_value=cache, cache_position=pos[:40])
            dec = []
            _native.timing_enable(2)        # labels only: which kernels did the decode steps launch?
            _native.timing_collect()
            for t in range(40, 44):
                o, _, _ = attn(x[None, t:t + 1], position_ids=pos[None, t:t + 1], past_key_value=cache, cache_position=pos[t:t + 1])
                dec.append(o[0])
            torch.cuda.synchronize()
            labels = [lab for lab, _, _ in _native.timing_collect()]
            _native.timing_enable(0)
            combined = sum("lin_qb_absorb_kernel" in lab for lab in labels), sum("lin_merge_unabsorb_kernel" in lab for lab in labels)
            assert combined == ((4, 4) if mode == "combined" else (0, 0)), (mode, labels[:12])
            # launches per decode step of the operator (q_a and kv_a are not merged in this build): 6 combined, more separate
            assert (len(labels) == 6 * 4) if mode == "combined" else (len(labels) >= 8 * 4), (mode, len(labels), labels[:12])
            outs[mode] = torch.cat(dec, 0).float().cpu()
            rows[mode] = cache.key_cache[0].reshape(-1, 576)[:44].clone().cpu()
        finally:
            for k in ("KTX_MLA_SEPARATE_QB", "KTX_MLA_SEPARATE_MERGE"):
                os.environ.pop(k, None)
    assert torch.isfinite(outs["combined"]).all() and outs["combined"].abs().max() > 0
    assert torch.equal(rows["combined"], rows["separate"]), "latent rows: same code in both launches"
    assert rel(outs["combined"].cuda(), outs["separate"]) < 5e-3

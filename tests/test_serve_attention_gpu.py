"""GPU: the batched-serving attention operator `flashinfer_attn` — two requests flattened into one call, paged latent cache
with a scattered page table — against the single-request operator run on each request alone."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_model_gpu import CFG, model_and_gold  # noqa: E402,F401  (fixture)


def test_two_requests_match_single_request_operator(model_and_gold):
    from ktransformers_amd._native import MLAWrapper
    from ktransformers_amd.models.custom_cache import KDeepSeekV3Cache, StaticCache
    from ktransformers_amd.models.modeling_deepseek import make_config
    from ktransformers_amd.operators.balance_serve_attention import flashinfer_attn
    from ktransformers_amd.util.generate import set_inference_mode
    from ktransformers_amd.util.utils import InferenceState
    model, _, _ = model_and_gold
    set_inference_mode(model, InferenceState.PREFILL)
    attn = model.model.layers[1].self_attn
    cfg = make_config(**CFG)
    dev = torch.device("cuda", 0)
    lens = [37, 70]                                         # request lengths (second spans two 64-token pages)
    g = torch.Generator().manual_seed(0)
    xs = [(torch.randn(n, CFG["hidden_size"], generator=g) * 0.5).to(torch.bfloat16).to(dev) for n in lens]

    # reference: each request alone through the single-request operator and its own StaticCache
    want = []
    for x in xs:
        cache = StaticCache(cfg, 1, 256, "cuda:0", torch.bfloat16)
        pos = torch.arange(x.shape[0], device=dev)[None]
        with torch.no_grad():
            want.append(attn(x[None], None, pos, cache, cache_position=pos[0])[0][0].clone())

    # serving call: both requests flattened, pages 5 | 2, 7 handed out by a pretend scheduler
    kv = KDeepSeekV3Cache(cfg, page_size=64, device="cuda:0")
    kv.allocate(8)
    q_indptr = torch.tensor([0, lens[0], lens[0] + lens[1]], dtype=torch.int32, device=dev)
    kv_indptr = torch.tensor([0, 1, 3], dtype=torch.int32, device=dev)
    kv_indices = torch.tensor([5, 2, 7], dtype=torch.int32, device=dev)
    kv_len = torch.tensor(lens, dtype=torch.int32, device=dev)
    pos = torch.cat([torch.arange(n, device=dev) for n in lens])
    bsz = torch.tensor([sum(lens)], dtype=torch.int32, device=dev)
    page_idx, page_offset = kv.get_page_table(pos, q_indptr, kv_indptr, kv_indices, bsz)
    H = attn.num_heads
    Hp = (H + 15) // 16 * 16
    wrapper = MLAWrapper(2, 8, use_cuda_graph=False, device=dev, max_q_tokens=256)
    wrapper.plan(q_indptr, kv_indptr, kv_indices, kv_len, torch.tensor([2], dtype=torch.int32, device=dev), Hp, attn.kv_lora_rank,
                 attn.qk_rope_head_dim, 64, attn.softmax_scale, torch.bfloat16, torch.bfloat16)
    with torch.no_grad():
        got = flashinfer_attn.forward(attn, torch.cat(xs), kv, pos, wrapper, bsz, page_idx.to(torch.int32), page_offset.to(torch.int32))
    torch.cuda.synchronize()
    got = got.split(lens)
    for a, b in zip(got, want):
        assert torch.isfinite(a.float()).all()
        assert float((a.float() - b.float()).norm() / b.float().norm()) < 5e-3

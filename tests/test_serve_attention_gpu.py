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


def _serve_call(attn, kv, wrapper, xs, starts, pages, dev):
    """One flattened serving call: request r contributes xs[r] at positions starts[r].. on the pages pages[r]."""
    from ktransformers_amd.operators.balance_serve_attention import flashinfer_attn
    lens = [int(x.shape[0]) for x in xs]

    def indptr(counts):
        return torch.tensor([0] + torch.tensor(counts).cumsum(0).tolist(), dtype=torch.int32, device=dev)
    q_indptr, kv_indptr = indptr(lens), indptr([len(p) for p in pages])
    kv_indices = torch.tensor([p for ps in pages for p in ps], dtype=torch.int32, device=dev)
    kv_len = torch.tensor([s + n for s, n in zip(starts, lens)], dtype=torch.int32, device=dev)
    pos = torch.cat([torch.arange(s, s + n, device=dev) for s, n in zip(starts, lens)])
    bsz = torch.tensor([sum(lens)], dtype=torch.int32, device=dev)
    page_idx, page_offset = kv.get_page_table(pos, q_indptr, kv_indptr, kv_indices, bsz)
    H = attn.num_heads
    wrapper.plan(q_indptr, kv_indptr, kv_indices, kv_len, torch.tensor([len(xs)], dtype=torch.int32, device=dev), (H + 15) // 16 * 16,
                 attn.kv_lora_rank, attn.qk_rope_head_dim, 64, attn.softmax_scale, torch.bfloat16, torch.bfloat16)
    with torch.no_grad():
        out = flashinfer_attn.forward(attn, torch.cat(xs), kv, pos, wrapper, bsz, page_idx.to(torch.int32), page_offset.to(torch.int32))
    torch.cuda.synchronize()
    return out.split(lens)


def test_serving_operator_against_the_oracle():
    """`flashinfer_attn.forward` (reference: operators/balance_serve_attention.py:66-118) checked DIRECTLY against the restated
    absorbed MLA operator (oracle/attention_ref.py, fp32) on the reference-generated golden weights: a mixed call (two prompts
    of different length on scattered pages), then a batched decode step for both requests whose history is what the first call
    left in the paged cache — outputs and the cache rows themselves.  (The test above ties the serving operator to the
    single-request operator; this one goes through no other product operator.)"""
    from attn_helpers import load_golden
    from oracle.attention_ref import mla_attention_ref
    from test_attention_gpu import build, rel, rows_close
    from ktransformers_amd._native import MLAWrapper
    from ktransformers_amd.models.custom_cache import KDeepSeekV3Cache
    cfg, w, x, _, _ = load_golden("v3")
    attn, _ = build(cfg, w)
    dev = torch.device("cuda", 0)
    lens = [23, 46]                                          # x rows 0..22 -> request 0, rows 23..68 -> request 1
    xs = [x[:23], x[23:69]]
    pages = [[6], [1, 4]]
    kv = KDeepSeekV3Cache(cfg, page_size=64, device="cuda:0")
    kv.allocate(8)
    wrapper = MLAWrapper(2, 8, use_cuda_graph=False, device=dev, max_q_tokens=256)
    empty = torch.zeros(0, 576, dtype=torch.bfloat16)
    got = _serve_call(attn, kv, wrapper, [t.cuda() for t in xs], [0, 0], pages, dev)
    hist = []
    for r in range(2):
        want, rows = mla_attention_ref(cfg, w, xs[r], torch.arange(lens[r]), empty)
        assert rel(got[r], want) < 1.5e-2, (r, rel(got[r], want))
        hist.append(rows)
    flat = kv.k_caches[attn.layer_idx].reshape(-1, 64, 576)   # the rows the call scattered to (page, offset)
    rows_close(flat[6, :23].float().cpu(), hist[0].float())
    rows_close(torch.cat([flat[1], flat[4]])[:46].float().cpu(), hist[1].float())
    assert float(flat[[0, 2, 3, 5, 7]].float().abs().sum()) == 0 and float(flat[6, 23:].float().abs().sum()) == 0   # nothing else was written
    # batched decode: one new token per request on top of the cached history (x rows 69 and 0 serve as the new tokens)
    xd = [x[69:70], x[0:1]]
    got = _serve_call(attn, kv, wrapper, [t.cuda() for t in xd], lens, pages, dev)
    for r in range(2):
        want, _ = mla_attention_ref(cfg, w, xd[r], torch.arange(lens[r], lens[r] + 1), hist[r])
        assert rel(got[r], want) < 1.5e-2, (r, rel(got[r], want))

"""The one-launch MLA decode step (csrc/ktx_attn.hip, include/ktx_attn.h) against the five-launch path it restates, at the
published DeepSeek-V3 attention dimensions (128 heads, hidden 7168, q_lora 1536, kv_lora 512): every phase's output — the
q_a|kv_a row, the normalised latent row and roped k_pe, the absorbed q rows and roped q_pe, the split partials, the merged
rows, the un-absorbed attention rows, the layer output — must be BIT-IDENTICAL to what lin_sk_kernel / lin_qb_absorb_kernel /
mla_decode_kernel / lin_merge_unabsorb_kernel / lin_sk_kernel produce on the same inputs (those kernels are the ones held
against the oracles in tests/test_linear_gpu.py, test_mla_gpu.py, test_attention_gpu.py, test_v3_layer_gpu.py), run as one
launch, as five launches of one phase each, and as replays of a captured HIP graph; the cache row of the new token must
be written, and the workspace status word must stay 0."""
import pytest
import torch

pytestmark = pytest.mark.gpu

H, NOPE, ROPE, LORA, VDIM, QLORA, HIDDEN = 128, 128, 64, 512, 128, 1536, 7168
PAGE = 64


def _u(shape, gen, dev, scale):
    return ((torch.rand(shape, generator=gen, device=dev, dtype=torch.float32) * 2 - 1) * scale).to(torch.bfloat16)


@pytest.fixture(scope="module")
def layer():
    from ktransformers_amd._native import LinearHandle

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    ops = {}
    ops["qkv_a"] = LinearHandle(HIDDEN, QLORA + LORA + ROPE, "W4", 64, 8, dev)
    ops["qkv_a"].load_bf16(_u((QLORA + LORA + ROPE, HIDDEN), g, dev, 0.03))
    ops["q_b"] = LinearHandle(QLORA, H * (NOPE + ROPE), "W4", 64, 8, dev)
    ops["q_b"].load_bf16(_u((H * (NOPE + ROPE), QLORA), g, dev, 0.05))
    ops["qabs"] = LinearHandle(NOPE, LORA, "BF16", 0, 8, dev, batch=H)
    ops["qabs"].load_bf16(_u((H, LORA, NOPE), g, dev, 0.08))
    ops["oabs"] = LinearHandle(LORA, VDIM, "BF16", 0, 8, dev, batch=H)
    ops["oabs"].load_bf16(_u((H, VDIM, LORA), g, dev, 0.05))
    ops["o_proj"] = LinearHandle(H * VDIM, HIDDEN, "W4", 64, 8, dev)
    ops["o_proj"].load_bf16(_u((HIDDEN, H * VDIM), g, dev, 0.02))
    ops["in_norm"] = (1 + _u((HIDDEN,), g, dev, 0.2).float()).to(torch.bfloat16)
    ops["qa_norm"] = (1 + _u((QLORA,), g, dev, 0.2).float()).to(torch.bfloat16)
    ops["kv_norm"] = (1 + _u((LORA,), g, dev, 0.2).float()).to(torch.bfloat16)
    ops["inv_freq"] = (1.0 / (10000.0 ** (torch.arange(0, ROPE, 2, device=dev, dtype=torch.float32) / ROPE))).contiguous()
    ops["gen"], ops["dev"] = g, dev
    return ops


def _case(layer, ctx, pages, permute):
    """Fresh cache of `pages` pages with ctx - 1 cached latent rows, the token at position ctx - 1, a random residual row."""
    dev, g = layer["dev"], layer["gen"]
    cache = torch.zeros((pages, PAGE, 1, LORA + ROPE), dtype=torch.bfloat16, device=dev)
    table = torch.randperm(pages, generator=g, device=dev).to(torch.int32) if permute else torch.arange(pages, dtype=torch.int32, device=dev)
    rows = _u((ctx - 1, LORA + ROPE), g, dev, 1.0)
    pos = torch.arange(ctx - 1, device=dev)
    cache.view(-1, LORA + ROPE)[table[pos // PAGE].long() * PAGE + pos % PAGE] = rows
    x = _u((1, HIDDEN), g, dev, 1.0)
    return {"cache": cache, "table": table, "x": x, "ctx": ctx, "pages": pages, "identity": not permute,
            "position": torch.tensor([ctx - 1], dtype=torch.int64, device=dev),
            "kv_len": torch.tensor([ctx], dtype=torch.int32, device=dev),
            "kv_indptr": torch.tensor([0, pages], dtype=torch.int32, device=dev)}


def _five_launches(layer, c, cache):
    """The product's current decode path (operators/attention.py), launch by launch, keeping every intermediate."""
    from ktransformers_amd._native import MLAWrapper, merge_and_unabsorb, qb_absorb_and_prep

    dev = layer["dev"]
    eps = 1e-6
    qkv = layer["qkv_a"].forward(c["x"], norm=(layer["in_norm"], eps))
    q_a, kv = qkv[:, :QLORA], qkv[:, QLORA:]
    q_nope, q_pe, ckv_new, kpe_new = qb_absorb_and_prep(layer["q_b"], layer["qabs"], q_a, (layer["qa_norm"], eps), kv, layer["kv_norm"], eps,
                                                        c["position"], layer["inv_freq"], 1.0, H, NOPE, ROPE, LORA)
    w = MLAWrapper(1, c["pages"], use_cuda_graph=False, device=dev, max_q_tokens=1)
    hint = min(c["ctx"] - 1 + 512, c["pages"] * PAGE)
    w.plan(None, c["kv_indptr"], c["table"], c["kv_len"], None, H, LORA, ROPE, PAGE, 0.1147, torch.bfloat16, torch.bfloat16,
           max_kv_len=hint, identity_pages=c["identity"])
    ws, nsplit = w.run_partials(q_nope, q_pe, cache[:, :, 0, :LORA], cache[:, :, 0, LORA:], new_ckv=ckv_new, new_kpe=kpe_new)
    torch.cuda.synchronize()
    part_o = ws[: H * nsplit * LORA * 4].view(torch.float32).reshape(H, nsplit, LORA).clone()
    part_ml = ws[H * nsplit * LORA * 4: H * nsplit * (LORA + 2) * 4].view(torch.float32).reshape(H, nsplit, 2).clone()
    out = merge_and_unabsorb(layer["oabs"], (ws, nsplit), 1, H)
    y = layer["o_proj"].forward(out.reshape(1, H * VDIM), add1=c["x"])
    torch.cuda.synchronize()
    return {"qkv": qkv.clone(), "q_lat": q_nope.clone(), "q_pe": q_pe.clone(), "ckv_new": ckv_new.clone(), "kpe_new": kpe_new.clone(),
            "part_o": part_o, "part_ml": part_ml, "attn_out": out.clone(), "y": y.clone(), "nsplit": nsplit, "hint": hint}


def _args(layer, c, cache, out, phases=31, last=True, hint=0):
    from ktransformers_amd._native import attn_decode_args

    eps = 1e-6
    return attn_decode_args(layer["qkv_a"], layer["q_b"], layer["qabs"], layer["oabs"], layer["o_proj"], c["x"].reshape(-1), out,
                            (layer["in_norm"], eps), (layer["qa_norm"], eps), (layer["kv_norm"], eps), c["position"], layer["inv_freq"], 1.0,
                            H, NOPE, ROPE, LORA, VDIM, cache[:, :, 0, :LORA], cache[:, :, 0, LORA:], PAGE, c["kv_indptr"],
                            None if c["identity"] else c["table"], c["kv_len"], hint, 0.1147, phases, last)


def _compare(layer, ref, y, tag):
    from ktransformers_amd._native import attn_debug_read, attn_status

    dev = layer["dev"]
    S = ref["nsplit"]
    got = {"qkv": attn_debug_read(dev, "qkv", (1, QLORA + LORA + ROPE)), "ckv_new": attn_debug_read(dev, "ckv_new", (1, LORA)),
           "kpe_new": attn_debug_read(dev, "kpe_new", (1, ROPE)), "q_lat": attn_debug_read(dev, "q_lat", (1, H, LORA)),
           "q_pe": attn_debug_read(dev, "q_pe", (1, H, ROPE)),
           "part_ml": attn_debug_read(dev, "part_ml", (H, S, 2), torch.float32),
           "part_o": attn_debug_read(dev, "part_o", (H, S, LORA), torch.float32),
           "attn_out": attn_debug_read(dev, "attn_out", (1, H, VDIM)), "y": y}
    assert attn_status(dev) == 0, f"{tag}: a hand-off timed out (status {attn_status(dev):#x})"
    for name in ("qkv", "ckv_new", "kpe_new", "q_lat", "q_pe", "part_ml", "attn_out", "y"):
        a, b = got[name].reshape(-1), ref[name].reshape(-1)
        if a.dtype == torch.float32:
            live = torch.isfinite(b)
            bad = int(((a != b) & live).sum()) + int((torch.isfinite(a) != live).sum())
        else:
            bad = int((a.view(torch.int16) != b.view(torch.int16)).sum())
        assert bad == 0, f"{tag}: {name}: {bad} of {a.numel()} values differ from the five-launch path"
    live = (ref["part_ml"][:, :, 1] > 0).unsqueeze(-1).expand_as(ref["part_o"])   # a dead split's row is undefined in both paths
    bad = int(((got["part_o"] != ref["part_o"]) & live).sum())
    assert bad == 0, f"{tag}: part_o: {bad} live values differ"


@pytest.mark.parametrize("ctx,pages,permute", [(4096, 80, False), (1000, 32, True), (33, 4, False), (1, 2, False)])
def test_one_launch_equals_five_launches(layer, ctx, pages, permute):
    from ktransformers_amd._native import attn_decode, attn_decode_eligible

    c = _case(layer, ctx, pages, permute)
    cache_ref, cache_new = c["cache"].clone(), c["cache"].clone()
    ref = _five_launches(layer, c, cache_ref)
    y = torch.empty((1, HIDDEN), dtype=torch.bfloat16, device=layer["dev"])
    a = _args(layer, c, cache_new, y.reshape(-1), hint=ref["hint"])
    assert attn_decode_eligible(a)
    attn_decode(a, layer["dev"])
    torch.cuda.synchronize()
    _compare(layer, ref, y, f"one launch, ctx {ctx}")
    assert torch.equal(cache_new.view(torch.int16), cache_ref.view(torch.int16)), "the new token's cache row differs"


def test_phase_by_phase_launches(layer):
    """The same device code as five launches of one phase each (and as 3 + 4 + 24): every split of the chain is correct because
    the hand-offs are flags, not launch order."""
    from ktransformers_amd._native import attn_decode

    c = _case(layer, 2500, 48, True)
    cache_ref = c["cache"].clone()
    ref = _five_launches(layer, c, cache_ref)
    for chain in ((1, 2, 4, 8, 16), (3, 4, 24), (3, 28)):
        cache_new = c["cache"].clone()
        y = torch.zeros((1, HIDDEN), dtype=torch.bfloat16, device=layer["dev"])
        a = _args(layer, c, cache_new, y.reshape(-1), hint=ref["hint"])
        for i, ph in enumerate(chain):
            attn_decode(a, layer["dev"], phases=ph, last=(i == len(chain) - 1))
        torch.cuda.synchronize()
        _compare(layer, ref, y, f"chain {chain}")
        assert torch.equal(cache_new.view(torch.int16), cache_ref.view(torch.int16))


def test_graph_replay_and_changing_inputs(layer):
    """Captured once, replayed with new residual rows, positions and cache contents: the epoch in the workspace advances on
    the device, so every replay hands over fresh data (a stale flag would reproduce the previous step's rows)."""
    from ktransformers_amd._native import attn_decode

    dev = layer["dev"]
    c = _case(layer, 700, 16, False)
    y = torch.zeros((1, HIDDEN), dtype=torch.bfloat16, device=dev)
    cache_new = c["cache"].clone()
    refs = []
    hint = min(700 - 1 + 512, 16 * PAGE)
    a = _args(layer, c, cache_new, y.reshape(-1), hint=hint)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        attn_decode(a, dev)   # warm-up outside the capture (allocates the workspace)
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    cache_new.copy_(c["cache"])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        attn_decode(a, dev)
    for step in range(4):
        c["x"].copy_(_u((1, HIDDEN), layer["gen"], dev, 1.0))
        c["position"].fill_(699 + step)
        c["kv_len"].fill_(700 + step)
        cache_ref = cache_new.clone()
        ref = _five_launches(layer, c, cache_ref)
        assert ref["hint"] >= 0
        g.replay()
        torch.cuda.synchronize()
        refs.append(ref["nsplit"])   # (the hint is capped by the cache capacity here, so every step plans the same split count)
        assert refs[-1] == refs[0]
        assert torch.equal(y.view(torch.int16), ref["y"].view(torch.int16)), f"replay {step}: layer output differs"
        assert torch.equal(cache_new.view(torch.int16), cache_ref.view(torch.int16)), f"replay {step}: cache row differs"


def test_moe_front_rides_in_the_launch(layer):
    """KTX_ATTN_PHASE_MOE_FRONT: the MoE block's front — post_attention_layernorm, router (256 experts, top-8 in 4 of 8 groups),
    shared experts' merged gate|up with SiLU*up — as a sixth phase on the row the o_proj phase produces: normalised row, selected
    experts (same order), routing weights and shared activations bit-identical to ktx_linear_forward_fused_gate on that row; the
    attention outputs unchanged; eager and as replays of one captured launch (the arrival ticket must be left at zero)."""
    from ktransformers_amd._native import GateHandle, LinearHandle, attn_decode, attn_decode_args, attn_status, gate_with_linear

    dev, g = layer["dev"], layer["gen"]
    E, K, IS = 256, 8, 2048
    sgu = LinearHandle(HIDDEN, 2 * IS, "W4", 64, 8, dev)
    sgu.load_bf16(_u((2 * IS, HIDDEN), g, dev, 0.02))
    gate = GateHandle(E, HIDDEN, K, 8, 4, "sigmoid", "noaux_tc", True, 2.5)
    gate_w = _u((E, HIDDEN), g, dev, 0.05)
    gate_b = ((torch.rand(E, generator=g, device=dev) - 0.5) * 0.2).float().contiguous()
    post_w = (1 + _u((HIDDEN,), g, dev, 0.2).float()).to(torch.bfloat16)
    c = _case(layer, 1500, 32, False)
    cache_ref = c["cache"].clone()
    ref = _five_launches(layer, c, cache_ref)
    idx0, wt0, xn0, act0 = gate_with_linear(gate, sgu, ref["y"], gate_w, gate_b, (post_w, 1e-6), glu=True)
    torch.cuda.synchronize()
    eps = 1e-6
    y = torch.zeros((1, HIDDEN), dtype=torch.bfloat16, device=dev)
    front = {"shared_gate_up": sgu, "gate": gate, "gate_weight": gate_w, "gate_bias": gate_b, "norm": (post_w, eps),
             "xn": torch.zeros((1, HIDDEN), dtype=torch.bfloat16, device=dev), "shared_act": torch.zeros((1, IS), dtype=torch.bfloat16, device=dev),
             "topk_idx": torch.zeros((1, K), dtype=torch.int64, device=dev), "topk_w": torch.zeros((1, K), dtype=torch.float32, device=dev)}
    cache_new = c["cache"].clone()
    a = attn_decode_args(layer["qkv_a"], layer["q_b"], layer["qabs"], layer["oabs"], layer["o_proj"], c["x"].reshape(-1), y.reshape(-1),
                         (layer["in_norm"], eps), (layer["qa_norm"], eps), (layer["kv_norm"], eps), c["position"], layer["inv_freq"], 1.0,
                         H, NOPE, ROPE, LORA, VDIM, cache_new[:, :, 0, :LORA], cache_new[:, :, 0, LORA:], PAGE, c["kv_indptr"], None,
                         c["kv_len"], ref["hint"], 0.1147, moe_front=front)

    def check(tag):
        assert attn_status(dev) == 0, tag
        assert torch.equal(y.view(torch.int16), ref["y"].view(torch.int16)), f"{tag}: layer output differs"
        assert torch.equal(front["xn"].view(torch.int16), xn0.view(torch.int16)), f"{tag}: normalised row differs"
        assert torch.equal(front["topk_idx"], idx0), f"{tag}: experts {front['topk_idx'].tolist()} vs {idx0.tolist()}"
        assert torch.equal(front["topk_w"], wt0), f"{tag}: routing weights differ"
        assert torch.equal(front["shared_act"].view(torch.int16), act0.view(torch.int16)), f"{tag}: shared activations differ"

    attn_decode(a, dev)
    torch.cuda.synchronize()
    check("eager")
    cache_new.copy_(c["cache"])
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        attn_decode(a, dev)
    for rep in range(3):
        for t in (y, front["xn"], front["shared_act"], front["topk_w"]):
            t.zero_()
        cache_new.copy_(c["cache"])
        gr.replay()
        torch.cuda.synchronize()
        check(f"replay {rep}")

"""The one-launch MLA decode step (csrc/ktx_attn.hip, include/ktx_attn.h) against the five-launch path it restates, at the
published DeepSeek-V3 / R1 attention dimensions (128 heads, hidden 7168, q_lora 1536, kv_lora 512; W4 g64 and block-fp8
projections) and Kimi-K2's (64 heads): every phase's output — the
q_a|kv_a row, the normalised latent row and roped k_pe, the absorbed q rows and roped q_pe, the split partials, the merged
rows, the un-absorbed attention rows, the layer output — must be BIT-IDENTICAL to what lin_sk_kernel / lin_qb_absorb_kernel /
mla_decode_kernel / lin_merge_unabsorb_kernel / lin_sk_kernel produce on the same inputs (those kernels are the ones held
against the oracles in tests/test_linear_gpu.py, test_mla_gpu.py, test_attention_gpu.py, test_v3_layer_gpu.py), run as one
launch, as five launches of one phase each, and as replays of a captured HIP graph; the cache row of the new token must
be written, and the workspace status word must stay 0."""
import pytest
import torch

pytestmark = pytest.mark.gpu

NOPE, ROPE, LORA, VDIM, QLORA, HIDDEN = 128, 64, 512, 128, 1536, 7168
PAGE = 64


def _u(shape, gen, dev, scale):
    return ((torch.rand(shape, generator=gen, device=dev, dtype=torch.float32) * 2 - 1) * scale).to(torch.bfloat16)


def _fp8_blocks(w):
    """bf16 [N, K] -> (e4m3 bytes, fp32 scale_inv [ceil(N/128), K/128]): the DeepSeek block-fp8 checkpoint format."""
    N, K = w.shape
    Np = (N + 127) // 128 * 128
    wf = torch.zeros((Np, K), dtype=torch.float32, device=w.device)
    wf[:N] = w.float()
    blk = wf.view(Np // 128, 128, K // 128, 128)
    sc = blk.abs().amax(dim=(1, 3)).clamp_min(1e-12) / 448.0
    q = (blk / sc[:, None, :, None]).reshape(Np, K)[:N].to(torch.float8_e4m3fn)
    return q.contiguous(), sc.contiguous()


def _quantised(K, N, fmt, w, dev):
    from ktransformers_amd._native import LinearHandle

    h = LinearHandle(K, N, fmt, 64, 8, dev)
    if fmt == "FP8":
        q, sc = _fp8_blocks(w)
        h.load_fp8(q, sc)
        h.test_fp8 = (q.cpu(), sc.cpu())     # the checkpoint tensors, for the oracle case
    else:
        h.load_bf16(w)
    return h


@pytest.fixture(scope="module", params=[("W4", 128), ("W4", 64), ("FP8", 128)], ids=["v3_w4_128_heads", "k2_w4_64_heads", "v3_fp8_128_heads"])
def layer(request):
    """One attention layer at the published DeepSeek-V3 / R1 (128 heads; W4 g64 or block-fp8 projections) or Kimi-K2 (64 heads)
    dimensions."""
    from ktransformers_amd._native import LinearHandle

    fmt, H = request.param
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    ops = {"H": H, "fmt": fmt}
    ops["qkv_a"] = _quantised(HIDDEN, QLORA + LORA + ROPE, fmt, _u((QLORA + LORA + ROPE, HIDDEN), g, dev, 0.03), dev)
    ops["q_b"] = _quantised(QLORA, H * (NOPE + ROPE), fmt, _u((H * (NOPE + ROPE), QLORA), g, dev, 0.05), dev)
    ops["w_uk"], ops["w_uv"] = _u((H, LORA, NOPE), g, dev, 0.08), _u((H, VDIM, LORA), g, dev, 0.05)   # [h][lora, nope], [h][v, lora]
    ops["qabs"] = LinearHandle(NOPE, LORA, "BF16", 0, 8, dev, batch=H)
    ops["qabs"].load_bf16(ops["w_uk"])
    ops["oabs"] = LinearHandle(LORA, VDIM, "BF16", 0, 8, dev, batch=H)
    ops["oabs"].load_bf16(ops["w_uv"])
    ops["o_proj"] = _quantised(H * VDIM, HIDDEN, fmt, _u((HIDDEN, H * VDIM), g, dev, 0.02), dev)
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
    return {"cache": cache, "table": table, "x": x, "ctx": ctx, "pages": pages, "identity": not permute, "rows": rows,
            "position": torch.tensor([ctx - 1], dtype=torch.int64, device=dev),
            "kv_len": torch.tensor([ctx], dtype=torch.int32, device=dev),
            "kv_indptr": torch.tensor([0, pages], dtype=torch.int32, device=dev)}


def _five_launches(layer, c, cache):
    """The product's current decode path (operators/attention.py), launch by launch, keeping every intermediate."""
    H = layer["H"]
    from ktransformers_amd._native import MLAWrapper, merge_and_unabsorb, qb_absorb_and_prep

    dev = layer["dev"]
    eps = 1e-6
    qkv = layer["qkv_a"].forward(c["x"], norm=(layer["in_norm"], eps))
    q_a, kv = qkv[:, :QLORA], qkv[:, QLORA:]
    if layer["fmt"] == "W4":
        q_nope, q_pe, ckv_new, kpe_new = qb_absorb_and_prep(layer["q_b"], layer["qabs"], q_a, (layer["qa_norm"], eps), kv, layer["kv_norm"], eps,
                                                            c["position"], layer["inv_freq"], 1.0, H, NOPE, ROPE, LORA)
    else:   # block-fp8 projections: q_b_proj (q_a_layernorm in its prologue) in its own launch, then absorb + prep (the operator's path)
        from ktransformers_amd._native import absorb_and_prep
        q = layer["q_b"].forward(q_a, norm=(layer["qa_norm"], eps))
        q_nope, q_pe, ckv_new, kpe_new = absorb_and_prep(layer["qabs"], q, kv, layer["kv_norm"], eps, c["position"], layer["inv_freq"], 1.0,
                                                         H, NOPE, ROPE, LORA)
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


def _args(layer, c, cache, out, phases=31, last=True, hint=0, sm_scale=0.1147):
    H = layer["H"]
    from ktransformers_amd._native import attn_decode_args

    eps = 1e-6
    return attn_decode_args(layer["qkv_a"], layer["q_b"], layer["qabs"], layer["oabs"], layer["o_proj"], c["x"].reshape(-1), out,
                            (layer["in_norm"], eps), (layer["qa_norm"], eps), (layer["kv_norm"], eps), c["position"], layer["inv_freq"], 1.0,
                            H, NOPE, ROPE, LORA, VDIM, cache[:, :, 0, :LORA], cache[:, :, 0, LORA:], PAGE, c["kv_indptr"],
                            None if c["identity"] else c["table"], c["kv_len"], hint, sm_scale, phases, last)


def _compare(layer, ref, y, tag):
    H = layer["H"]
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
    exact = ("qkv", "ckv_new", "kpe_new", "q_lat", "q_pe", "part_ml", "attn_out", "y")
    if layer["fmt"] == "FP8":
        # o_proj: the launch deals 8-k-step groups of a strip to its wavefronts, lin_dec_kernel<FP8> sums four k-slices of 32 — the same
        # products in another fp32 association (every other phase is bit-identical: q_a|kv_a = 8 slices of 7, q_b = one chain)
        exact = exact[:-1]
        a, b = got["y"].float().reshape(-1), ref["y"].float().reshape(-1)
        err = (a - b).abs()
        bound = 2.0 ** -7 * b.abs() + 2.0 ** -9 * float(b.abs().max())
        assert bool((err <= bound).all()), f"{tag}: y: max |diff| {float(err.max()):.4g} beyond one bf16 ulp of the five-launch row"
        assert float((a != b).float().mean()) < 0.25, f"{tag}: y: {float((a != b).float().mean()):.3f} of the outputs differ"
    for name in exact:
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
    chains = ((1, 2, 4, 8, 16),) if layer["fmt"] == "FP8" else ((1, 2, 4, 8, 16), (3, 4, 24), (3, 28))   # (the mixed subsets are W4 builds)
    for chain in chains:
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
        if layer["fmt"] == "FP8":   # (o_proj's fp32 association differs from lin_dec_kernel<FP8>'s: see _compare)
            assert torch.allclose(y.float(), ref["y"].float(), rtol=2.0 ** -7, atol=2.0 ** -9 * float(ref["y"].float().abs().max()))
        else:
            assert torch.equal(y.view(torch.int16), ref["y"].view(torch.int16)), f"replay {step}: layer output differs"
        assert torch.equal(cache_new.view(torch.int16), cache_ref.view(torch.int16)), f"replay {step}: cache row differs"


@pytest.mark.parametrize("ctx,pages,permute", [(1000, 32, True), (3000, 64, False)])
def test_one_launch_against_the_oracle(layer, ctx, pages, permute):
    """The launch held DIRECTLY against the restated reference operator (oracle/attention_ref.py: forward_linux_flashinfer op for op in
    bf16, dense F.linear on the de-quantised weights — what KLinearMarlin multiplies with) at the published dimensions, not only
    against the library's own five launches: layer output norm-wise within the bound tests/test_attention_gpu.py uses for the
    operator (2e-2: bf16 pipeline noise), the appended cache row within one bf16 ulp (+ the rope sum's absolute slack)."""
    import types
    from ktransformers_amd._native import attn_decode, attn_status
    from oracle.attention_ref import mla_attention_ref, rmsnorm_ref, softmax_scale

    H, dev = layer["H"], layer["dev"]
    cfg = types.SimpleNamespace(num_attention_heads=H, qk_nope_head_dim=NOPE, qk_rope_head_dim=ROPE, kv_lora_rank=LORA, v_head_dim=VDIM,
                                q_lora_rank=QLORA, rms_norm_eps=1e-6, rope_theta=10000.0, rope_scaling=None)
    w = {"q_a_layernorm": layer["qa_norm"].cpu(), "kv_a_layernorm": layer["kv_norm"].cpu(),
         "kv_b_proj": torch.cat([layer["w_uk"].transpose(1, 2), layer["w_uv"]], dim=1).reshape(H * (NOPE + VDIM), LORA).cpu()}
    lin = None
    if layer["fmt"] == "W4":
        qkv = layer["qkv_a"].dequant_bf16().cpu()
        w.update({"q_a_proj": qkv[:QLORA], "kv_a_proj_with_mqa": qkv[QLORA:], "q_b_proj": layer["q_b"].dequant_bf16().cpu(),
                  "o_proj": layer["o_proj"].dequant_bf16().cpu()})
    else:
        # block-fp8 projections (round 6; was skipped): KLinearFP8.forward = act_quant + fp8_gemm (operators/linear.py:408-413,
        # ktransformers_ext/triton/fp8gemm.py:10-55,117-193) restated by oracle/linear_ref.py — per-128 e4m3 activation blocks against
        # the checkpoint's e4m3 weight blocks, block dots scaled by a_s * b_s, fp32 sum, bf16 out.  q_a and kv_a are the two row
        # ranges of the fused projection (same input blocks, so the same activation codes as two separate KLinearFP8 calls).
        from oracle.linear_ref import linear_fp8_ref
        (qa_q, qa_s), (qb_q, qb_s), (o_q, o_s) = layer["qkv_a"].test_fp8, layer["q_b"].test_fp8, layer["o_proj"].test_fp8
        full = {}

        def lin(name, x):
            if name in ("q_a_proj", "kv_a_proj_with_mqa"):
                if "qkv" not in full:
                    full["qkv"] = linear_fp8_ref(x, qa_q, qa_s)
                return full["qkv"][:, :QLORA] if name == "q_a_proj" else full["qkv"][:, QLORA:]
            return linear_fp8_ref(x, qb_q, qb_s) if name == "q_b_proj" else linear_fp8_ref(x, o_q, o_s)
    c = _case(layer, ctx, pages, permute)
    # a small residual row (the layer's input passes through RMSNorm, so its scale does not matter to the attention): with |x| ~ 1 the
    # bf16 rounding of `x + attn` (half an ulp of ~1) would be several per cent of the attention part this test looks at
    c["x"].mul_(2.0 ** -6)
    x = c["x"].cpu()
    hidden = rmsnorm_ref(x, layer["in_norm"].cpu(), 1e-6)                       # input_layernorm (modeling_deepseek_v3.py:1200-1205)
    out_ref, new_row = mla_attention_ref(cfg, w, hidden, torch.tensor([ctx - 1]), c["rows"].cpu(), lin=lin)
    y_ref = x + out_ref                                                         # residual add in bf16 (:1219)
    cache = c["cache"].clone()
    y = torch.empty((1, HIDDEN), dtype=torch.bfloat16, device=dev)
    a = _args(layer, c, cache, y.reshape(-1), hint=min(ctx - 1 + 512, pages * PAGE), sm_scale=softmax_scale(cfg))
    attn_decode(a, dev)
    torch.cuda.synchronize()
    assert attn_status(dev) == 0
    attn_part, ref_part = (y.cpu().float() - x.float()), out_ref.float()
    rel = float((attn_part - ref_part).norm() / ref_part.norm())
    assert rel < 2e-2, f"attention output {rel:.4f} away from the oracle (norm-wise)"
    # element-wise: one bf16 ulp of the residual sum + 2^-6 of the largest attention output (round 5 allowed 0.1 of it — not a bound)
    err = float((y.cpu().float() - y_ref.float()).abs().max())
    assert err <= 2.0 ** -7 * float(y_ref.float().abs().max()) + 2.0 ** -6 * float(ref_part.abs().max()), \
        f"max |y - oracle| = {err:.3e} (max |attn| {float(ref_part.abs().max()):.3e}, max |y| {float(y_ref.float().abs().max()):.3e})"
    pos = ctx - 1
    row = cache.view(-1, LORA + ROPE)[int(c["table"][pos // PAGE]) * PAGE + pos % PAGE].cpu().float()
    d = (row - new_row[0].float()).abs()
    assert bool((d <= 2.0 ** -7 * new_row[0].float().abs() + 2.0 ** -6).all()), f"appended cache row: max diff {float(d.max())}"


def test_eligibility_follows_the_context_bound(layer):
    """whether the launch covers a call depends on kv_len_hint (it picks the KV split shape) — a context bound of
    8192 tokens or more takes the 4x2 'long context' shape of the stand-alone kernel, which this launch does not restate; the answer
    must therefore be asked per call.  Short -> long -> short flips the answer both ways on the same handles."""
    from ktransformers_amd._native import attn_decode_eligible

    c = _case(layer, 100, 160, False)
    y = torch.empty((1, HIDDEN), dtype=torch.bfloat16, device=layer["dev"])
    short = _args(layer, c, c["cache"], y.reshape(-1), hint=612)
    mid = _args(layer, c, c["cache"], y.reshape(-1), hint=32768)        # round 6: covered with deeper 2x4 splits up to 40 K tokens
    long_ = _args(layer, c, c["cache"], y.reshape(-1), hint=40960)
    assert attn_decode_eligible(short) and attn_decode_eligible(mid) and not attn_decode_eligible(long_) and attn_decode_eligible(short)


@pytest.mark.parametrize("ctx,pages", [(8500, 144), (20000, 320)])
def test_one_launch_past_8192_tokens(layer, ctx, pages):
    """Round 6: between 8192 and 40960 tokens the launch keeps the 2x4 workgroup shape (<= 64 splits, several tiles per split) where
    the stand-alone kernel switches to 4x2 — so the five-launch reference is pinned to the 2x4 shape (dev knob 6 = 2) and then every
    intermediate is bit-identical again."""
    from ktransformers_amd import _native
    from ktransformers_amd._native import attn_decode, attn_decode_eligible

    c = _case(layer, ctx, pages, False)
    cache_ref, cache_new = c["cache"].clone(), c["cache"].clone()
    _native.check(_native.lib.ktx_debug_set(6, 2))
    try:
        ref = _five_launches(layer, c, cache_ref)
    finally:
        _native.check(_native.lib.ktx_debug_set(6, 0))
    y = torch.empty((1, HIDDEN), dtype=torch.bfloat16, device=layer["dev"])
    a = _args(layer, c, cache_new, y.reshape(-1), hint=ref["hint"])
    assert attn_decode_eligible(a)
    attn_decode(a, layer["dev"])
    torch.cuda.synchronize()
    _compare(layer, ref, y, f"one launch, ctx {ctx}")
    assert torch.equal(cache_new.view(torch.int16), cache_ref.view(torch.int16)), "the new token's cache row differs"


def test_two_streams_decode_on_one_device(layer):
    """Persistent-launch hygiene: two 'models' (two caches, two residual rows) stepping on two streams of one device.  The launch
    needs every workgroup resident, so the library orders a launch behind the device's previous one when it comes from another
    stream (include/ktx_attn.h): interleaved issue from two streams gives exactly the serial results, status word 0."""
    from ktransformers_amd._native import attn_decode, attn_status

    dev = layer["dev"]
    cases = [_case(layer, 900, 32, False), _case(layer, 2100, 48, True)]
    serial = []
    for c in cases:
        cache = c["cache"].clone()
        y = torch.zeros((1, HIDDEN), dtype=torch.bfloat16, device=dev)
        ys = []
        for step in range(3):
            c["position"].fill_(c["ctx"] - 1 + step)
            c["kv_len"].fill_(c["ctx"] + step)
            attn_decode(_args(layer, c, cache, y.reshape(-1), hint=c["pages"] * PAGE), dev)
            torch.cuda.synchronize()
            ys.append(y.clone())
        serial.append((ys, cache))
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    caches = [c["cache"].clone() for c in cases]
    outs = [[torch.zeros((1, HIDDEN), dtype=torch.bfloat16, device=dev) for _ in range(3)] for _ in cases]
    poss = [[torch.tensor([c["ctx"] - 1 + s], dtype=torch.int64, device=dev) for s in range(3)] for c in cases]
    lens = [[torch.tensor([c["ctx"] + s], dtype=torch.int32, device=dev) for s in range(3)] for c in cases]
    torch.cuda.synchronize()
    keep = []
    for step in range(3):
        for i, c in enumerate(cases):
            with torch.cuda.stream(streams[i]):
                ci = dict(c, position=poss[i][step], kv_len=lens[i][step])
                a = _args(layer, ci, caches[i], outs[i][step].reshape(-1), hint=c["pages"] * PAGE)
                keep.append(a)
                attn_decode(a, dev)
    torch.cuda.synchronize()
    assert attn_status(dev) == 0
    for i in range(2):
        for step in range(3):
            assert torch.equal(outs[i][step].view(torch.int16), serial[i][0][step].view(torch.int16)), f"model {i} step {step} differs from serial"
        assert torch.equal(caches[i].view(torch.int16), serial[i][1].view(torch.int16)), f"model {i}: cache rows differ from serial"

"""Router parity on the GPU against the torch restatement of MoEGate.forward (oracle/router_ref.py).  Index SETS must be
identical (torch.topk(sorted=False) leaves the order unspecified); weights agree to 2e-6 relative — the only difference
is the fp32 summation order of the 7168-long logit dot products, as between any two fp32 GEMM implementations.
Batches above 64 tokens take the library's own MFMA GEMM on exact bf16 planes of the weight (csrc/ktx_gemm.hip): every product
is exact, the MFMA's fp32 accumulation measured against fp64 math is 1.0e-7 * sum|x w| at worst (torch's fp32 GEMM on the same
operands: 5.7e-8; K = 5120 / 7168, 300 tokens) — same class, twice the constant, so those cases get 8e-6."""
import numpy as np
import pytest
import torch

from oracle.router_ref import moe_gate_ref

pytestmark = pytest.mark.gpu

CONFIGS = {
    "deepseek_v3": dict(E=256, H=7168, top_k=8, n_group=8, topk_group=4, scoring_func="sigmoid", topk_method="noaux_tc",
                        norm_topk_prob=True, routed_scaling_factor=2.5),
    "kimi_k2": dict(E=384, H=7168, top_k=8, n_group=1, topk_group=1, scoring_func="sigmoid", topk_method="noaux_tc",
                    norm_topk_prob=True, routed_scaling_factor=2.827),
    "deepseek_v2_lite": dict(E=64, H=2048, top_k=6, n_group=1, topk_group=1, scoring_func="softmax", topk_method="greedy",
                             norm_topk_prob=False, routed_scaling_factor=1.0),
    "deepseek_v2": dict(E=160, H=5120, top_k=6, n_group=8, topk_group=3, scoring_func="softmax",
                        topk_method="group_limited_greedy", norm_topk_prob=False, routed_scaling_factor=16.0),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
@pytest.mark.parametrize("T", [1, 7, 300])
def test_router_matches_reference_math(name, T):
    from ktransformers_amd._native import GateHandle
    cfg = dict(CONFIGS[name])
    E, H = cfg.pop("E"), cfg.pop("H")
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((T, H), generator=g).to(torch.bfloat16)
    w = (torch.randn((E, H), generator=g) * H ** -0.5).to(torch.bfloat16)
    bias = torch.randn((E,), generator=g) * 0.1 if cfg["topk_method"] == "noaux_tc" else None
    ridx, rw = moe_gate_ref(x, w, bias, **cfg)
    gh = GateHandle(E, H, cfg["top_k"], cfg["n_group"], cfg["topk_group"], cfg["scoring_func"], cfg["topk_method"],
                    cfg["norm_topk_prob"], cfg["routed_scaling_factor"])
    idx, wt = gh.forward(x.to(dev), w.to(dev), None if bias is None else bias.to(dev))
    torch.cuda.synchronize()
    idx, wt = idx.cpu(), wt.cpu()
    assert idx.dtype == torch.int64 and wt.dtype == torch.float32
    for t in range(T):
        assert set(idx[t].tolist()) == set(ridx[t].tolist()), f"token {t}: routed expert set differs"
        ref = dict(zip(ridx[t].tolist(), rw[t].tolist()))
        for e, v in zip(idx[t].tolist(), wt[t].tolist()):
            assert abs(v - ref[e]) <= (2e-6 if T <= 64 else 8e-6) * max(abs(ref[e]), 1e-6) + 1e-9, (t, e, v, ref[e])


def test_fused_router_handoff_is_never_stale_under_load():
    """The fused router hands the logits from 16+ workgroups to the last arriver with sc1 stores/loads and no fences:
    hammer it (changing inputs every launch, a bandwidth hog on another stream) and require every launch to equal the
    two-kernel logits + select path exactly."""
    import ctypes as C
    from ktransformers_amd import _native as n
    torch.manual_seed(0)
    for (E, H, k, ng, tg, sc, tm) in ((64, 2048, 6, 1, 1, "softmax", "greedy"), (256, 7168, 8, 8, 4, "sigmoid", "noaux_tc")):
        g = n.GateHandle(E, H, k, ng, tg, sc, tm, True, 2.5)
        w = (torch.randn(E, H, device="cuda") * H ** -0.5).to(torch.bfloat16)
        bias = (torch.randn(E, device="cuda") * 0.1) if tm == "noaux_tc" else None
        hog = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        side = torch.cuda.Stream()
        xs = (torch.randn(300, 3, H, device="cuda") / 10).to(torch.bfloat16)
        outs = []
        for i in range(300):
            if i % 10 == 0:
                with torch.cuda.stream(side):
                    hog.add_(1.0)
            outs.append(g.forward(xs[i], w, bias))
        torch.cuda.synchronize()
        for i in range(300):
            logits = torch.empty((3, E), dtype=torch.float32, device="cuda")
            n.check(n.lib.ktx_gate_logits(C.byref(g.cfg), None, 3, xs[i].data_ptr(), w.data_ptr(), logits.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream))
            idx = torch.empty((3, k), dtype=torch.int64, device="cuda")
            wt = torch.empty((3, k), dtype=torch.float32, device="cuda")
            n.check(n.lib.ktx_gate_select(C.byref(g.cfg), None, 3, logits.data_ptr(), bias.data_ptr() if bias is not None else None,
                                          idx.data_ptr(), wt.data_ptr(), torch.cuda.current_stream().cuda_stream))
            assert torch.equal(outs[i][0], idx) and torch.equal(outs[i][1], wt), f"launch {i} differs"


@pytest.mark.parametrize("name", ["v3", "k2", "v3_ties", "v2lite", "v2", "v2_ties"])
def test_router_matches_reference_module_golden(name):
    """The HIP router against outputs of the reference's OWN MoEGate modules (tests/golden/make_router_golden.py imports
    modeling_deepseek_v3.py / modeling_deepseek.py), incl. duplicated router rows whose scores tie exactly."""
    import os
    from ktransformers_amd._native import GateHandle
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "router_golden.npz"))
    E, H, k, ng, tg, norm = (int(v) for v in g[f"{name}.cfg"])
    scoring, method = (str(v) for v in g[f"{name}.func"])
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(g[f"{name}.x"]).view(torch.bfloat16).to(dev)
    w = torch.from_numpy(g[f"{name}.w"]).view(torch.bfloat16).to(dev)
    bias = torch.from_numpy(g[f"{name}.bias"]).to(dev) if f"{name}.bias" in g.files else None
    gh = GateHandle(E, H, k, ng, tg, scoring, method, bool(norm), float(g[f"{name}.scale"]))
    idx, wt = gh.forward(x, w, bias)
    torch.cuda.synchronize()
    idx, wt, ridx, rwt = idx.cpu().numpy(), wt.cpu().numpy(), g[f"{name}.idx"], g[f"{name}.wt"]
    for t in range(idx.shape[0]):
        if name.endswith("ties"):    # whichever tied expert was picked, the score multiset is the reference's
            np.testing.assert_allclose(np.sort(wt[t]), np.sort(rwt[t]), rtol=2e-6, atol=1e-9)
        else:
            assert set(idx[t].tolist()) == set(ridx[t].tolist()), f"token {t}: routed expert set differs"
            ref = dict(zip(ridx[t].tolist(), rwt[t].tolist()))
            np.testing.assert_allclose(wt[t], np.array([ref[e] for e in idx[t].tolist()], np.float32), rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("name,I_shared", [("deepseek_v3", 2048), ("deepseek_v2_lite", 2816), ("kimi_k2", 2048), ("deepseek_v2", 3072)])
@pytest.mark.parametrize("T", [1, 3])
def test_router_riding_in_the_shared_gate_up_launch(name, I_shared, T):
    """ktx_linear_forward_fused_gate (decode step of a MoE block: router + the shared experts' merged gate|up GEMV on the same
    un-normalised row, one launch) against the two separate calls.  With the combined kernel switched off (knob 13) the
    library issues exactly those two launches: bit-identical.  The combined kernel runs the same device code with 8 router
    wavefronts per workgroup and possibly another k-slice split of the GEMV, i.e. other fp32 summation orders: the
    normalised row within one bf16 ulp, identical expert sets, weights and GEMV outputs within rounding noise."""
    from ktransformers_amd import _native as n
    cfg = dict(CONFIGS[name])
    E, H = cfg.pop("E"), cfg.pop("H")
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11 + T)
    x = torch.randn((T, H), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((E, H), generator=g) * H ** -0.5).to(torch.bfloat16).to(dev)
    nw = (1 + 0.1 * torch.randn((H,), generator=g)).to(torch.bfloat16).to(dev)
    bias = (torch.randn((E,), generator=g) * 0.1).to(dev) if cfg["topk_method"] == "noaux_tc" else None
    wl = (torch.randn((2 * I_shared, H), generator=g) / 10).to(torch.bfloat16).to(dev)
    gh = n.GateHandle(E, H, cfg["top_k"], cfg["n_group"], cfg["topk_group"], cfg["scoring_func"], cfg["topk_method"],
                      cfg["norm_topk_prob"], cfg["routed_scaling_factor"])
    lin = n.LinearHandle(H, 2 * I_shared, "W4", 64, 64, dev)
    lin.load_bf16(wl)
    idx0, wt0, xn0 = gh.forward(x, w, bias, norm=(nw, 1e-6))
    y0 = lin.forward(x, norm=(nw, 1e-6), glu=True)
    try:
        n.lib.ktx_debug_set(13, 1)
        idx1, wt1, xn1, y1 = n.gate_with_linear(gh, lin, x, w, bias, (nw, 1e-6))
    finally:
        n.lib.ktx_debug_set(13, 0)
    assert torch.equal(idx0, idx1) and torch.equal(wt0, wt1) and torch.equal(xn0, xn1) and torch.equal(y0, y1)
    for rep in range(3):                                  # replays: the arrival counters must be left at zero
        idx2, wt2, xn2, y2 = n.gate_with_linear(gh, lin, x, w, bias, (nw, 1e-6))
        torch.cuda.synchronize()
        ulp = (xn2.float() - xn0.float()).abs() <= xn0.float().abs() * 2.0 ** -7 + 1e-30
        assert bool(ulp.all()), "normalised row differs by more than one bf16 ulp"
        for t in range(T):
            assert set(idx2[t].tolist()) == set(idx0[t].tolist()), f"token {t}: routed expert set differs"
            ref = dict(zip(idx0[t].tolist(), wt0[t].tolist()))
            for e, v in zip(idx2[t].tolist(), wt2[t].tolist()):
                assert abs(v - ref[e]) <= 2e-3 * abs(ref[e]) + 1e-9, (t, e, v, ref[e])
        err = (y2.float() - y0.float()).abs().max().item()
        assert err <= 2e-2 * y0.float().abs().max().item(), err
    # router workgroups of 4 experts (default where the grid has room) vs 8 (dev knob 25 = 1): the same dot product per expert,
    # the same selection code, the same GEMV workgroups — every output bit for bit
    try:
        n.lib.ktx_debug_set(25, 1)
        for rep in range(2):
            idx3, wt3, xn3, y3 = n.gate_with_linear(gh, lin, x, w, bias, (nw, 1e-6))
            torch.cuda.synchronize()
            assert torch.equal(idx3, idx2) and torch.equal(wt3, wt2) and torch.equal(xn3, xn2) and torch.equal(y3, y2)
    finally:
        n.lib.ktx_debug_set(25, 0)


@pytest.mark.parametrize("T", [1, 3])
def test_router_riding_in_the_fp8_shared_gate_up_launch(T):
    """The same combined launch for block-FP8 shared experts (DeepSeek-V3 / R1 fp8 checkpoints: [gate ; up] rows of one GEMV, no
    GLU epilogue — operators/mlp.py keeps every 128-row scale block whole): knob 13 = the two separate launches, bit-identical;
    the combined kernel = same device code, possibly another k-slice split (fp32 summation order) — and the profile label shows
    that it was the combined kernel that ran."""
    from ktransformers_amd import _native as n
    cfg = dict(CONFIGS["deepseek_v3"])
    E, H = cfg.pop("E"), cfg.pop("H")
    I_shared = 2048
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(23 + T)
    x = torch.randn((T, H), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((E, H), generator=g) * H ** -0.5).to(torch.bfloat16).to(dev)
    nw = (1 + 0.1 * torch.randn((H,), generator=g)).to(torch.bfloat16).to(dev)
    bias = (torch.randn((E,), generator=g) * 0.1).to(dev)
    wl = (torch.randn((2 * I_shared, H), generator=g) / 4).to(torch.float8_e4m3fn).to(dev)
    sc = ((torch.rand((2 * I_shared // 128, H // 128), generator=g) + 0.5) / 32).to(dev)
    gh = n.GateHandle(E, H, cfg["top_k"], cfg["n_group"], cfg["topk_group"], cfg["scoring_func"], cfg["topk_method"],
                      cfg["norm_topk_prob"], cfg["routed_scaling_factor"])
    lin = n.LinearHandle(H, 2 * I_shared, "FP8", 128, 64, dev)
    lin.load_fp8(wl, sc)
    idx0, wt0, xn0 = gh.forward(x, w, bias, norm=(nw, 1e-6))
    y0 = lin.forward(x, norm=(nw, 1e-6))
    try:
        n.lib.ktx_debug_set(13, 1)
        idx1, wt1, xn1, y1 = n.gate_with_linear(gh, lin, x, w, bias, (nw, 1e-6), glu=False)
    finally:
        n.lib.ktx_debug_set(13, 0)
    assert torch.equal(idx0, idx1) and torch.equal(wt0, wt1) and torch.equal(xn0, xn1) and torch.equal(y0, y1)
    for rep in range(3):
        idx2, wt2, xn2, y2 = n.gate_with_linear(gh, lin, x, w, bias, (nw, 1e-6), glu=False)
        torch.cuda.synchronize()
        ulp = (xn2.float() - xn0.float()).abs() <= xn0.float().abs() * 2.0 ** -7 + 1e-30
        assert bool(ulp.all()), "normalised row differs by more than one bf16 ulp"
        for t in range(T):
            assert set(idx2[t].tolist()) == set(idx0[t].tolist()), f"token {t}: routed expert set differs"
            ref = dict(zip(idx0[t].tolist(), wt0[t].tolist()))
            for e, v in zip(idx2[t].tolist(), wt2[t].tolist()):
                assert abs(v - ref[e]) <= 2e-3 * abs(ref[e]) + 1e-9, (t, e, v, ref[e])
        err = (y2.float() - y0.float()).abs().max().item()
        assert err <= 2e-2 * y0.float().abs().max().item(), err
    n.timing_collect()
    n.timing_enable(2)
    try:
        n.gate_with_linear(gh, lin, x, w, bias, (nw, 1e-6), glu=False)
        torch.cuda.synchronize()
        names = [r[0] for r in n.timing_collect()]
    finally:
        n.timing_enable(0)
    assert any(nm.startswith("lin_dec_gate_kernel<FP8>") for nm in names), names


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("name,I_shared", [("deepseek_v3", 2048), ("deepseek_v2_lite", 2816), ("kimi_k2", 2048)])
@pytest.mark.parametrize("T", [1, 3])
def test_router_riding_in_the_all_cu_gate_up_kernel(name, I_shared, T, mode):
    """The two opt-in placements of the router inside lin_sk_gate_kernel (dev knob 19: 3 = round 2's router workgroups in
    front of the all-CU GEMV's grid, 2 = wavefront 7 of every GEMV workgroup + one selector workgroup sweeping {tag, logit}
    granules) against the separate router + GEMV calls; replayed, so tickets / granules must come back to zero."""
    from ktransformers_amd import _native as n
    cfg = dict(CONFIGS[name])
    E, H = cfg.pop("E"), cfg.pop("H")
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(17 + T)
    x = torch.randn((T, H), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((E, H), generator=g) * H ** -0.5).to(torch.bfloat16).to(dev)
    nw = (1 + 0.1 * torch.randn((H,), generator=g)).to(torch.bfloat16).to(dev)
    bias = (torch.randn((E,), generator=g) * 0.1).to(dev) if cfg["topk_method"] == "noaux_tc" else None
    wl = (torch.randn((2 * I_shared, H), generator=g) / 10).to(torch.bfloat16).to(dev)
    gh = n.GateHandle(E, H, cfg["top_k"], cfg["n_group"], cfg["topk_group"], cfg["scoring_func"], cfg["topk_method"],
                      cfg["norm_topk_prob"], cfg["routed_scaling_factor"])
    lin = n.LinearHandle(H, 2 * I_shared, "W4", 64, 64, dev)
    lin.load_bf16(wl)
    idx0, wt0, xn0 = gh.forward(x, w, bias, norm=(nw, 1e-6))
    y0 = lin.forward(x, norm=(nw, 1e-6), glu=True)
    try:
        n.lib.ktx_debug_set(19, mode)
        for rep in range(4):
            idx2, wt2, xn2, y2 = n.gate_with_linear(gh, lin, x, w, bias, (nw, 1e-6))
            torch.cuda.synchronize()
            assert bool(((xn2.float() - xn0.float()).abs() <= xn0.float().abs() * 2.0 ** -7 + 1e-30).all())
            for t in range(T):
                assert set(idx2[t].tolist()) == set(idx0[t].tolist()), f"token {t}: routed expert set differs (rep {rep})"
                ref = dict(zip(idx0[t].tolist(), wt0[t].tolist()))
                for e, v in zip(idx2[t].tolist(), wt2[t].tolist()):
                    assert abs(v - ref[e]) <= 2e-3 * abs(ref[e]) + 1e-9, (t, e, v, ref[e])
            assert (y2.float() - y0.float()).abs().max().item() <= 2e-2 * y0.float().abs().max().item()
    finally:
        n.lib.ktx_debug_set(19, 0)


def test_two_models_route_on_two_streams_of_one_device():
    """the {tag, logit} granules of the router that rides in the shared gate|up launch were ONE buffer per device —
    two models (or two streams) on one GPU could sweep each other's logits.  Every handle now owns its granules: two MoE fronts of
    DeepSeek-V3 dimensions issued alternately on two streams, many times, give exactly what each gives alone."""
    from ktransformers_amd import _native as n
    cfg = dict(CONFIGS["deepseek_v3"])
    E, H = cfg.pop("E"), cfg.pop("H")
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    models = []
    for m in range(2):
        w = (torch.randn((E, H), generator=g) * H ** -0.5).to(torch.bfloat16).to(dev)
        nw = (1 + 0.1 * torch.randn((H,), generator=g)).to(torch.bfloat16).to(dev)
        bias = (torch.randn((E,), generator=g) * 0.1).to(dev)
        wl = (torch.randn((4096, H), generator=g) / 10).to(torch.bfloat16).to(dev)
        gh = n.GateHandle(E, H, cfg["top_k"], cfg["n_group"], cfg["topk_group"], cfg["scoring_func"], cfg["topk_method"],
                          cfg["norm_topk_prob"], cfg["routed_scaling_factor"])
        lin = n.LinearHandle(H, 4096, "W4", 64, 8, dev)
        lin.load_bf16(wl)
        xs = [torch.randn((1, H), generator=g).to(torch.bfloat16).to(dev) for _ in range(6)]
        alone = [tuple(t.clone() for t in n.gate_with_linear(gh, lin, x, w, bias, (nw, 1e-6))) for x in xs]
        models.append((gh, lin, w, bias, nw, xs, alone))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for rnd in range(4):
        got = [[], []]
        for i in range(6):
            for m, (gh, lin, w, bias, nw, xs, _) in enumerate(models):
                with torch.cuda.stream(streams[m]):
                    got[m].append(tuple(t.clone() for t in n.gate_with_linear(gh, lin, xs[i], w, bias, (nw, 1e-6))))
        torch.cuda.synchronize()
        for m in range(2):
            for i in range(6):
                for a, b in zip(got[m][i], models[m][6][i]):
                    assert torch.equal(a, b), f"round {rnd}, model {m}, step {i}: differs from the model running alone"

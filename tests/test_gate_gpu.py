"""Router parity on the GPU against the torch restatement of MoEGate.forward (oracle/router_ref.py).  Index SETS must be
identical (torch.topk(sorted=False) leaves the order unspecified); weights agree to 2e-6 relative — the only difference
is the fp32 summation order of the 7168-long logit dot products, as between any two fp32 GEMM implementations."""
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
            assert abs(v - ref[e]) <= 2e-6 * max(abs(ref[e]), 1e-6) + 1e-9, (t, e, v, ref[e])


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

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

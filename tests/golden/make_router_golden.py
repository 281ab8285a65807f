"""Generate tests/golden/router_golden.npz: the REFERENCE's own router modules — MoEGate of
archive/ktransformers/models/modeling_deepseek_v3.py:401-481 (V3 / R1 / Kimi-K2: sigmoid + noaux_tc) and of
archive/ktransformers/models/modeling_deepseek.py:381-461 (V2 / V2-Lite: softmax + greedy / group_limited_greedy) — imported
and run on CPU over seeded inputs, including rows whose scores TIE (duplicated router rows).  Stored: inputs (bf16 bits),
router weights, bias, and the module's (topk_idx, topk_weight).  Pins oracle/router_ref.py (tests/test_router_pin_cpu.py).

    python tests/golden/make_router_golden.py        (needs /root/reference)
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_import import reference_models  # noqa: E402

v3, _ = reference_models()
from ktransformers.models import modeling_deepseek as v2  # noqa: E402

CASES = {
    # name: (module, config fields, T, tie)
    "v3": (v3, dict(n_routed_experts=64, hidden_size=256, num_experts_per_tok=8, n_group=8, topk_group=4, scoring_func="sigmoid",
                    topk_method="noaux_tc", norm_topk_prob=True, routed_scaling_factor=2.5), 33, False),
    "k2": (v3, dict(n_routed_experts=48, hidden_size=128, num_experts_per_tok=8, n_group=1, topk_group=1, scoring_func="sigmoid",
                    topk_method="noaux_tc", norm_topk_prob=True, routed_scaling_factor=2.827), 17, False),
    "v3_ties": (v3, dict(n_routed_experts=32, hidden_size=128, num_experts_per_tok=4, n_group=4, topk_group=2, scoring_func="sigmoid",
                         topk_method="noaux_tc", norm_topk_prob=True, routed_scaling_factor=2.5), 21, True),
    "v2lite": (v2, dict(n_routed_experts=64, hidden_size=256, num_experts_per_tok=6, n_group=1, topk_group=1, scoring_func="softmax",
                        topk_method="greedy", norm_topk_prob=False, routed_scaling_factor=1.0), 33, False),
    "v2": (v2, dict(n_routed_experts=40, hidden_size=128, num_experts_per_tok=6, n_group=8, topk_group=3, scoring_func="softmax",
                    topk_method="group_limited_greedy", norm_topk_prob=False, routed_scaling_factor=16.0), 19, False),
    "v2_ties": (v2, dict(n_routed_experts=32, hidden_size=128, num_experts_per_tok=6, n_group=4, topk_group=2, scoring_func="softmax",
                         topk_method="group_limited_greedy", norm_topk_prob=True, routed_scaling_factor=1.0), 15, True),
}
out = {}
for name, (mod, fields, T, tie) in CASES.items():
    cfg = SimpleNamespace(**fields, aux_loss_alpha=0.001, seq_aux=True)
    gate = mod.MoEGate(cfg).eval()
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    E, H = fields["n_routed_experts"], fields["hidden_size"]
    w = (torch.randn((E, H), generator=g) * H ** -0.5).to(torch.bfloat16)
    if tie:     # experts 3 and 5 (same group) and expert E-1 / E-2 share their router rows: their scores tie on every token
        w[5] = w[3]
        w[E - 1] = w[E - 2]
    x = torch.randn((T, H), generator=g).to(torch.bfloat16)
    bias = None
    with torch.no_grad():
        gate.weight.copy_(w.float())
        if hasattr(gate, "e_score_correction_bias"):
            bias = torch.randn((E,), generator=g) * 0.1
            if tie:
                bias[5] = bias[3]
                bias[E - 1] = bias[E - 2]
            gate.e_score_correction_bias.copy_(bias)
        idx, wt = gate(x[None])[:2]      # the V2 module also returns its aux loss (None in eval mode)
    out[f"{name}.cfg"] = np.array([E, H, fields["num_experts_per_tok"], fields["n_group"], fields["topk_group"],
                                   int(fields["norm_topk_prob"])], np.int64)
    out[f"{name}.scale"] = np.float32(fields["routed_scaling_factor"])
    out[f"{name}.func"] = np.array([fields["scoring_func"], fields["topk_method"]])
    out[f"{name}.x"] = x.view(torch.uint16).numpy()
    out[f"{name}.w"] = w.view(torch.uint16).numpy()
    if bias is not None:
        out[f"{name}.bias"] = bias.numpy()
    out[f"{name}.idx"] = idx.numpy()
    out[f"{name}.wt"] = wt.float().numpy()
    print(name, idx.shape, wt.dtype)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "router_golden.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path))

"""The injected operators end to end on the GPU: YAML injection on a toy DeepSeek-shaped model, weights through the
loader protocol, KDeepseekV3MoE.forward = gate -> KTransformersExperts, checked against oracle router + oracle experts."""
import os

import numpy as np
import pytest
import torch

from helpers import f32_to_bf16, numpy_u16
from oracle.oracle import FMT_AMXINT4
from oracle.router_ref import moe_gate_ref
from toy_model import ToyConfig, ToyModel

pytestmark = pytest.mark.gpu
RULES = os.path.join(os.path.dirname(__file__), "toy_rules.yaml")


def test_injected_moe_block_matches_oracle(oracle):
    from ktransformers_amd.optimize.optimize import optimize_and_load
    from ktransformers_amd.util.loader import DictLoader
    from ktransformers_amd.util.utils import InferenceState

    cfg = ToyConfig()
    E, H, I, k = cfg.n_routed_experts, cfg.hidden_size, cfg.moe_intermediate_size, cfg.num_experts_per_tok
    g = torch.Generator().manual_seed(0)
    state = {}
    with torch.device("meta"):
        model = ToyModel(cfg)
    for name, p in list(model.named_parameters()):
        if ".experts." in name:
            state[name] = (torch.randn(p.shape, generator=g) / 10).to(torch.bfloat16)
        elif name.endswith("gate.weight"):
            state[name] = (torch.randn(p.shape, generator=g) * H ** -0.5).to(torch.bfloat16)
        elif name.endswith("e_score_correction_bias"):
            state[name] = torch.randn(p.shape, generator=g) * 0.1
        else:
            state[name] = torch.zeros(p.shape, dtype=torch.bfloat16)
    optimize_and_load(model, RULES, DictLoader(state), cfg, default_device="cuda:0", load=True)
    mlp = model.model.layers[1].mlp
    assert mlp.experts.mode == InferenceState.GENERATE and mlp.experts.generate_experts.handle is not None

    T = 5
    x = (torch.randn((1, T, H), generator=g) / 100).to(torch.bfloat16)
    y = mlp(x.to("cuda:0"))
    torch.cuda.synchronize()
    assert y.shape == (1, T, H)

    pre = "model.layers.1.mlp."
    ridx, rw = moe_gate_ref(x.view(T, H), state[pre + "gate.weight"], state[pre + "gate.e_score_correction_bias"],
                            top_k=k, n_group=cfg.n_group, topk_group=cfg.topk_group, scoring_func=cfg.scoring_func,
                            topk_method=cfg.topk_method, norm_topk_prob=cfg.norm_topk_prob,
                            routed_scaling_factor=cfg.routed_scaling_factor)

    def stack(proj):
        return numpy_u16(torch.stack([state[f"{pre}experts.{e}.{proj}_proj.weight"] for e in range(E)]))
    mo = oracle.make_moe(FMT_AMXINT4, stack("gate"), stack("up"), stack("down"))
    # feed the oracle the GPU router's own (idx, weight) order so the slot-order sum is comparable bit for bit,
    # after checking the routed sets agree with the reference math
    idx, wt = mlp.gate(x.to("cuda:0"))
    for t in range(T):
        assert set(idx[t].tolist()) == set(ridx[t].tolist())
    want = oracle.moe_forward(mo, idx.cpu().numpy(), wt.cpu().numpy(), numpy_u16(x.view(T, H)))
    assert np.array_equal(numpy_u16(y.view(T, H)), want)

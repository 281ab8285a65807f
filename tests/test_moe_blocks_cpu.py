"""CPU: orchestration of the MoE block operators with stand-in experts (the HIP experts are exercised in the GPU tests):
the Mixtral block's routing arithmetic (archive/ktransformers/operators/experts.py:1074-1128) and the batched-serving
variants' (bsz_tensor, cuda_graph_idx) plumbing (:1172-1213, :1331-1339)."""
import types

import torch
from torch import nn

from ktransformers_amd.operators.experts import (KDeepseekV3MoEV2, KMistralSparseMoEBlock, KTransformersExperts,
                                                 KTransformersExpertsV2)
from ktransformers_amd.optimize.optimize import resolve_class
from ktransformers_amd.util.utils import InferenceState


class DenseExperts(nn.Module):
    """y[t] = sum_j w[t, j] * W[ids[t, j]] @ x[t], in fp32; records the serving arguments."""

    def __init__(self, E, H):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.W = torch.randn(E, H, H, generator=g) / H ** 0.5
        self.seen = None

    def forward(self, x, ids, w, bsz_tensor=None, cuda_graph_idx=0):
        self.seen = (bsz_tensor, cuda_graph_idx, w.dtype)
        y = torch.einsum("tj,tjo->to", w.float(), torch.einsum("tjoh,th->tjo", self.W[ids], x.float()))
        return y.to(x.dtype)


def test_mixtral_block_routing():
    E, H, k = 8, 32, 2
    orig = nn.Module()
    orig.gate = nn.Linear(H, E, bias=False).to(torch.bfloat16)
    orig.experts = DenseExperts(E, H)
    orig.top_k, orig.num_experts, orig.jitter_noise = k, E, 0.0
    blk = KMistralSparseMoEBlock("model.layers.0.block_sparse_moe", None, types.SimpleNamespace(), orig)
    x = torch.randn(2, 5, H).to(torch.bfloat16)
    y, logits = blk(x)
    assert y.shape == x.shape and logits.shape == (10, E) and y.dtype == torch.bfloat16
    flat = x.view(-1, H)
    want_logits = orig.gate(flat)
    assert torch.equal(logits, want_logits)
    p = torch.softmax(want_logits.float(), dim=1)
    pw, pi = torch.topk(p, k, dim=-1)
    pw = (pw / pw.sum(-1, keepdim=True)).to(torch.bfloat16)          # the reference casts the weights to the input dtype
    want = orig.experts(flat, pi, pw)
    assert torch.equal(y.view(-1, H), want)
    assert orig.experts.seen[2] == torch.bfloat16


def test_serving_variants_forward_bsz_tensor():
    E, H, k, T = 8, 32, 2, 6

    class Gate(nn.Module):
        def forward(self, h):
            g = torch.Generator().manual_seed(1)
            n = h.view(-1, h.shape[-1]).shape[0]
            return torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(n)]), torch.rand(n, k, generator=g)

    class Shared(nn.Module):
        def forward(self, x):
            return x * 2

    orig = nn.Module()
    orig.gate, orig.experts, orig.shared_experts = Gate(), DenseExperts(E, H), Shared()
    blk = KDeepseekV3MoEV2("model.layers.1.mlp", None, types.SimpleNamespace(n_shared_experts=1), orig)
    x = torch.randn(1, T, H).to(torch.bfloat16)
    bsz = torch.tensor([4], dtype=torch.int32)
    y = blk(x, bsz, 3)
    assert orig.experts.seen[0] is bsz and orig.experts.seen[1] == 3
    ids, w = Gate()(x)
    want = orig.experts(x.view(-1, H), ids, w).view(1, T, H) + x * 2
    assert torch.equal(y, want)
    assert resolve_class("ktransformers.operators.experts.KDeepseekV3MoEV2") is KDeepseekV3MoEV2
    assert resolve_class("ktransformers.operators.experts.KTransformersExpertsV2") is KTransformersExpertsV2
    assert resolve_class("ktransformers.operators.experts.KMistralSparseMoEBlock") is KMistralSparseMoEBlock


def test_experts_switch_passes_serving_arguments():
    cfg = types.SimpleNamespace(n_routed_experts=4, num_experts_per_tok=2, hidden_size=256, moe_intermediate_size=256)
    for cls in (KTransformersExperts, KTransformersExpertsV2):
        ke = cls("model.layers.1.mlp.experts", None, cfg, nn.ModuleList([nn.Identity() for _ in range(4)]), prefill_device="cuda",
                 generate_device="cuda", prefill_op="KExpertsTorch", generate_op="KExpertsCPU", backend="BF16")
        seen = {}
        ke.generate_experts.forward = lambda *a: seen.setdefault("args", a) and "out"
        object.__setattr__(ke, "mode", InferenceState.GENERATE)
        bsz = torch.tensor([2], dtype=torch.int32)
        assert ke.forward("x", "ids", "w", bsz, 5) == "out" and seen["args"] == ("x", "ids", "w", bsz, 5)
        seen.clear()
        ke.forward("x", "ids", "w")
        assert seen["args"] == ("x", "ids", "w", None, 0)


def test_decode_step_with_the_router_in_the_shared_gate_up_launch(monkeypatch):
    """KDeepseekV3MoE._forward with the combined router || shared gate|up launch (operators/experts.py), on stand-ins: a W4 merged
    operator comes back activated (GLU epilogue) and its down projection may ride with the routed experts (_tail_side is asked with it);
    a block-fp8 [gate ; up] operator comes back RAW — `down` must be told (glu_in=True: SiLU * up in its prologue, round 5) and the raw
    rows must never be offered to the routed experts' side strip."""
    from ktransformers_amd.operators.experts import KDeepseekV3MoE
    E, H, k, I = 8, 32, 2, 16
    seen = {}

    class Gate(nn.Module):
        def forward_with_linear(self, h, norm, handle, glu=True):
            seen["glu"] = glu
            g = torch.Generator().manual_seed(3)
            idx = torch.randperm(E, generator=g)[:k][None]
            w = torch.rand(1, k, generator=g)
            act = torch.full((1, I if glu else 2 * I), 1.0 if glu else -1.0, dtype=torch.bfloat16)
            return idx, w, h.reshape(-1, H) * 2, act

    class Shared(nn.Module):
        def down(self, a, shape, add1=None, add2=None, glu_in=False):
            seen["down"] = (tuple(a.shape), float(a.flatten()[0]), glu_in)
            return add1 + add2

    norm = types.SimpleNamespace(weight=torch.ones(H, dtype=torch.bfloat16), variance_epsilon=1e-6)
    x = torch.randn(1, 1, H).to(torch.bfloat16)
    res = torch.randn(1, 1, H).to(torch.bfloat16)
    for fmt in ("W4", "FP8"):
        orig = nn.Module()
        orig.gate, orig.experts, orig.shared_experts = Gate(), DenseExperts(E, H), Shared()
        blk = KDeepseekV3MoE("model.layers.1.mlp", None, types.SimpleNamespace(n_shared_experts=1), orig)
        monkeypatch.setattr(blk, "_router_side_linear", lambda h, n, allow_cat=False, fmt=fmt: types.SimpleNamespace(fmt=fmt) if allow_cat else None,
                            raising=False)
        asked = []
        monkeypatch.setattr(blk, "_tail_side", lambda act, r, op: asked.append(act) or None, raising=False)
        seen.clear()
        y = blk._forward(x, residual=res, pre_norm=norm)
        assert seen["glu"] == (fmt == "W4")
        assert seen["down"] == ((1, I), 1.0, False) if fmt == "W4" else seen["down"] == ((1, 2 * I), -1.0, True)
        assert asked and (asked[-1] is not None) == (fmt == "W4")       # raw [gate | up] rows are never a side strip
        idx, w, xn, _ = Gate().forward_with_linear(x, None, None)
        want = orig.experts(xn, idx, w).view(1, 1, H) + res            # Shared.down returned add1 + add2: routed output + residual
        assert torch.equal(y, want)

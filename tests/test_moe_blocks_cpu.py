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

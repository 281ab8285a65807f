"""CPU: the rotary-embedding operators against the reference's own rotary modules (tests/golden/rope_golden.npz, made by
tests/golden/make_rope_golden.py from models/modeling_deepseek_v3.py): same inv_freq / YaRN mscale => same cos / sin at
positions up to 160K.  The fused ktx_mla_prep kernel consumes exactly these `inv_freq` / `_mscale` values."""
import os
import types

import numpy as np
import pytest
import torch

from ktransformers_amd.operators.RoPE import RotaryEmbeddingV3, RotaryEmbeddingV4, YarnRotaryEmbedding, YarnRotaryEmbeddingV3

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rope_golden.npz"))


@pytest.mark.parametrize("name", ["v3", "v2lite", "odd"])
def test_yarn_rope_matches_reference_module(name):
    dim, base, factor, orig, bf, bs, ms, msa = G[f"{name}_cfg"]
    cfg = types.SimpleNamespace(qk_rope_head_dim=int(dim), rope_theta=float(base), max_position_embeddings=163840,
                                rope_scaling={"type": "yarn", "factor": factor, "original_max_position_embeddings": int(orig),
                                              "beta_fast": bf, "beta_slow": bs, "mscale": ms, "mscale_all_dim": msa})
    op = YarnRotaryEmbeddingV3("model.layers.0.self_attn.rotary_emb", None, cfg, torch.nn.Module(), generate_device="cpu", prefill_device="cpu")
    op.load()
    cos, sin = op(torch.zeros(1, dtype=torch.float32), torch.from_numpy(G["pos"])[None])
    # the reference builds a table with fp32 arange positions; positions here are exact integers: agreement to fp32 rounding
    assert np.allclose(cos[0].numpy(), G[f"{name}_cos"], rtol=0, atol=3e-4 * max(1.0, float(op._mscale)))
    assert np.allclose(sin[0].numpy(), G[f"{name}_sin"], rtol=0, atol=3e-4 * max(1.0, float(op._mscale)))
    small = G["pos"] < 5000                                    # where fp32 angle rounding is negligible: tight
    assert np.abs(cos[0].numpy()[small] - G[f"{name}_cos"][small]).max() < 2e-6
    assert YarnRotaryEmbedding is YarnRotaryEmbeddingV3


def test_plain_rope_matches_reference_module():
    cfg = types.SimpleNamespace(qk_rope_head_dim=64, rope_theta=10000.0, max_position_embeddings=8192)
    for cls in (RotaryEmbeddingV3, RotaryEmbeddingV4):
        op = cls("k", None, cfg, torch.nn.Module(), generate_device="cpu", prefill_device="cpu")
        op.load()
        cos, sin = op(torch.zeros(1, dtype=torch.bfloat16), torch.from_numpy(G["plain_pos"])[None])
        assert cos.dtype == torch.bfloat16
        assert np.abs(cos[0].float().numpy() - G["plain_cos"]).max() < 8e-3 and np.abs(sin[0].float().numpy() - G["plain_sin"]).max() < 8e-3
        cos, sin = op(torch.zeros(1, dtype=torch.float32), torch.from_numpy(G["plain_pos"])[None])
        assert np.abs(cos[0].numpy() - G["plain_cos"]).max() < 3e-4

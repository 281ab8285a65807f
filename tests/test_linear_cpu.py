"""CPU: the restated Marlin quantiser equals the reference's own quantize_weights on the committed golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle.linear_ref import act_quant_ref, dequant_w4, linear_fp8_ref, linear_w4_ref, quantize_weights_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "linear_w4_golden.npz")


@pytest.mark.parametrize("name", ["a", "b"])
@pytest.mark.parametrize("G", [32, 64, 128])
def test_quantizer_restatement_matches_reference_golden(name, G):
    g = np.load(GOLD)
    w = torch.from_numpy(g[f"{name}_w"]).view(torch.bfloat16)
    q, s = quantize_weights_ref(w.T.contiguous(), G)
    s_bits = s.view(torch.uint16).numpy()
    assert np.array_equal(s_bits, g[f"{name}_s{G}"])
    live = np.repeat(g[f"{name}_s{G}"] != 0, G, axis=0)     # s == 0: int(NaN) is platform-defined in the reference
    assert np.array_equal(q.numpy().astype(np.uint8)[live], g[f"{name}_q{G}"][live])


def test_w4_reference_math_is_close_to_dense():
    torch.manual_seed(0)
    w = (torch.randn(64, 256) / 10).to(torch.bfloat16)
    x = (torch.randn(5, 256) / 10).to(torch.bfloat16)
    q, s = quantize_weights_ref(w.T.contiguous(), 64)
    y = linear_w4_ref(x, q, s, 64).float()
    d = (x.float() @ w.float().T)
    assert (y - d).norm() / d.norm() < 0.15                  # 4-bit quantisation error, not a parity bound
    y2 = linear_w4_ref(x, q, s, 64, round_weights=True).float()
    assert (y - y2).norm() / y.norm() < 4e-3                 # Marlin's bf16 dequant rounding vs exact (q-8)*s
    assert dequant_w4(q, s, 64, False).shape == (256, 64)


def test_fp8_act_quant_and_linear():
    torch.manual_seed(1)
    x = (torch.randn(3, 256)).to(torch.bfloat16)
    xq, s = act_quant_ref(x)
    assert xq.dtype == torch.float8_e4m3fn and s.shape == (3, 2)
    assert torch.all(xq.float().abs().amax(dim=-1) <= 448)
    w = (torch.randn(32, 256) / 4).to(torch.float8_e4m3fn)
    sc = torch.rand(1, 2) + 0.5
    y = linear_fp8_ref(x, w, sc).float()
    d = x.float() @ (w.float() * sc.repeat_interleave(128, 1)).T
    assert (y - d).norm() / d.norm() < 0.05


@pytest.mark.parametrize("name", ["a", "b"])
@pytest.mark.parametrize("G,act", [(64, False), (128, True)])
def test_marlin_8bit_multiplicand_matches_the_reference_quantiser(name, G, act):
    """operators/linear.marlin_multiplicand (what KLinearMarlin(num_bits=8) loads) against the REFERENCE's quantize_weights at
    8 bits (golden from its own quant_utils.py): bf16((q - 128) * s) element for element.  With act_order the reference stores
    the rows of q_w permuted (rand_perm) and its g_idx / sort_indices undo that inside the kernel: undoing it here gives the very
    same (q, s) as without act_order — the flag changes storage, not the product."""
    from ktransformers_amd.operators.linear import marlin_multiplicand
    g = np.load(GOLD)
    w = torch.from_numpy(g[f"{name}_w"]).view(torch.bfloat16)
    tag = f"{name}_8b{G}{'_act' if act else ''}"
    q, s = torch.from_numpy(g[tag + "_q"].astype(np.int32)), torch.from_numpy(g[tag + "_s"]).view(torch.bfloat16)
    if act:
        perm = torch.from_numpy(g[tag + "_perm"])
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel())
        q = q[inv]                                               # stored row r holds original row perm[r]
        q_plain, s_plain = quantize_weights_ref(w.T.contiguous(), G, num_bits=8)
        assert torch.equal(s_plain.view(torch.uint16), s.view(torch.uint16))
        live = torch.repeat_interleave(s != 0, G, dim=0)
        assert torch.equal(q[live], q_plain[live])
    want = ((q.float() - 128.0) * s.float().repeat_interleave(G, dim=0)).to(torch.bfloat16).T     # [N, K]
    got = marlin_multiplicand(w, 8, G)
    live = torch.repeat_interleave(s != 0, G, dim=0).T
    assert torch.equal(got.view(torch.uint16)[live], want.contiguous().view(torch.uint16)[live])
    assert float(got.float()[~live].abs().max() if (~live).any() else 0.0) == 0.0                 # an all-zero group stays zero
    assert float((got.float() - w.float()).abs().max()) < float(w.float().abs().max()) / 100      # 8-bit grid: < 1 % of full scale
    # round 4: the W8 handle is loaded with (q, s) themselves (ktx_linear_load_w8): marlin_quantize must hand out the reference's
    from ktransformers_amd.operators.linear import marlin_quantize
    qs, sc = marlin_quantize(w, 8, G)                                                              # [N, K/G, G] signed, [N, K/G, 1]
    q8 = (qs + 128).to(torch.int32).view(w.shape[0], -1).T                                         # [K, N] like quantize_weights' q_w
    assert torch.equal(sc.view(w.shape[0], -1).T.contiguous().view(torch.uint16), s.view(torch.uint16))
    assert torch.equal(q8[live.T], q[live.T])

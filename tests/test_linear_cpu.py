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

"""CPU: GGUF k-quant block codec vs the reference's own numpy dequantisers (golden fixture), Q8_K quantiser properties,
and the restated llamafile expert forward (oracle/ktx_oracle_gguf.c; its GEMM kernels are pinned against the reference's iqk
kernels in test_gguf_ref_pin_cpu.py, its Q8_K quantiser is restated from ggml's published algorithm) against fp64 math
on the de-quantised weights (the only kind of check the reference itself has for this path: kt-kernel/examples/test_moe.py)."""
import os

import numpy as np
import pytest

from helpers import bf16_to_f32
from oracle.gguf_ref import (DEQUANT, GGML_TYPE_IQ1_S, GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K, QUANT, GgufOracle,
                             dequantize_iq1_s, dequantize_q4_k, dequantize_q5_k, dequantize_q6_k, iq1s_grid, quantize_iq1_s)
from oracle.oracle import f32_to_bf16

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gguf_blocks_golden.npz")


def test_block_layouts_match_reference_dequantisers():
    g = np.load(GOLD)
    assert np.array_equal(dequantize_q4_k(g["q4k_blocks"]), g["q4k_values"])
    assert np.array_equal(dequantize_q6_k(g["q6k_blocks"]), g["q6k_values"])
    assert np.array_equal(dequantize_q5_k(g["t13_blocks"]), g["t13_values"])


@pytest.mark.parametrize("t", [GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K])
def test_test_quantisers_round_trip(t):
    rng = np.random.default_rng(t)
    w = (rng.standard_normal((6, 512)) / 10).astype(np.float32)
    wq = DEQUANT[t](QUANT[t](w))
    rel = np.linalg.norm(wq - w) / np.linalg.norm(w)
    assert rel < {GGML_TYPE_Q4_K: 0.12, GGML_TYPE_Q5_K: 0.06, GGML_TYPE_Q6_K: 0.03}[t], rel


def test_q8k_quantiser():
    o = GgufOracle()
    rng = np.random.default_rng(0)
    x = rng.standard_normal(512).astype(np.float32)
    x[256:] = 0                                                   # an all-zero block
    x[7] = -9.0                                                   # largest magnitude is negative -> iscale positive
    q, d, bs = o.quantize_row_q8_K(x)
    assert d[1] == 0 and not q[256:].any()
    assert q[7] == -127 and d[0] == np.float32(1) / (np.float32(-127) / np.float32(-9.0))
    assert np.array_equal(bs, q.astype(np.int32).reshape(-1, 16).sum(1).astype(np.int16))
    assert np.abs(q[:256] * d[0] - x[:256]).max() <= abs(d[0]) * 0.5 + 1e-6


@pytest.mark.parametrize("types", [(GGML_TYPE_Q4_K, GGML_TYPE_Q4_K, GGML_TYPE_Q6_K), (GGML_TYPE_Q6_K, GGML_TYPE_Q4_K, GGML_TYPE_Q4_K),
                                   (GGML_TYPE_IQ1_S, GGML_TYPE_IQ1_S, GGML_TYPE_IQ1_S),
                                   (GGML_TYPE_Q5_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K)])     # the q5_k_m mix (oracle side only so far)
def test_gguf_forward_tracks_dequantised_math(types):
    o = GgufOracle()
    E, k, H, I, T = 4, 2, 256, 512, 3
    rng = np.random.default_rng(1)
    gate_f, up_f = (rng.standard_normal((E, I, H)) / 10).astype(np.float32), (rng.standard_normal((E, I, H)) / 10).astype(np.float32)
    down_f = (rng.standard_normal((E, H, I)) / 10).astype(np.float32)
    def blocks(t, w):
        if t != GGML_TYPE_IQ1_S:
            return QUANT[t](w)
        # any bit pattern is a valid IQ1_S block: random ones (small d) instead of the slow nearest-grid encoder, whose own
        # round trip is covered at the end of test_iq1s_codebook_and_block_layout
        b = rng.integers(0, 256, w.shape[:-1] + (w.shape[-1] // 256, 50), dtype=np.uint8)
        d = rng.random(w.shape[:-1] + (w.shape[-1] // 256,)).astype(np.float16) * np.float16(0.02) + np.float16(0.005)
        b[..., 0:2] = d.view(np.uint8).reshape(d.shape + (2,))
        return b.reshape(w.shape[:-1] + (-1,))
    gate, up, down = blocks(types[0], gate_f), blocks(types[1], up_f), blocks(types[2], down_f)
    x = f32_to_bf16(rng.standard_normal((T, H)).astype(np.float32))
    ids = np.stack([rng.permutation(E)[:k] for _ in range(T)]).astype(np.int64)
    ids[1, 0] = -1                                                # skipped slot
    w = rng.random((T, k)).astype(np.float32)
    yb, inter = o.moe_forward(gate, up, down, types, E, H, I, ids, w, x, want_inter=True)
    y = bf16_to_f32(yb)
    gd, ud, dd = (DEQUANT[t](a).astype(np.float64) for t, a in zip(types, (gate, up, down)))
    xf = bf16_to_f32(x).astype(np.float64)
    ref = np.zeros((T, H))
    for t in range(T):
        for j in range(k):
            e = ids[t, j]
            if e < 0:
                continue
            g, u = gd[e] @ xf[t], ud[e] @ xf[t]
            ref[t] += w[t, j] * (dd[e] @ ((g / (1 + np.exp(-g))) * u))
    rel = np.linalg.norm(y - ref) / np.linalg.norm(ref)
    assert rel < 3e-2, rel                                        # two Q8_K activation quantisations (~1-2 %) + bf16 output
    # white box: with the SAME Q8_K codes the integer block arithmetic must reproduce de-quantised math to fp32 rounding
    q, d8, _ = o.quantize_row_q8_K(bf16_to_f32(x[0]))
    xq = (q.astype(np.float64).reshape(-1, 256) * d8.astype(np.float64)[:, None]).reshape(-1)
    for j in range(k):
        e = ids[0, j]
        g, u = gd[e] @ xq, ud[e] @ xq
        want = (g / (1 + np.exp(-g))) * u
        assert np.abs(inter[j] - want).max() <= 2e-5 * np.abs(want).max() + 1e-7


def test_iq1s_codebook_and_block_layout():
    """The IQ1_S codebook transcribed from the reference (tests/golden/make_iq1s_grid.py): 2048 distinct ternary points; the
    numpy de-quantiser (w = d * (2s+1) * (g +- 1/8)) and the C restatement of the reference's mul_mat_iq1_s_q8_K agree."""
    import ctypes as C
    g = iq1s_grid()
    assert g.shape == (2048, 8) and set(np.unique(g)) == {-1, 0, 1} and np.unique(g, axis=0).shape[0] == 2048
    rng = np.random.default_rng(4)
    blocks = rng.integers(0, 256, (6, 50), dtype=np.uint8)
    blocks[:, 0:2] = (rng.random(6).astype(np.float16) * np.float16(0.02)).view(np.uint8).reshape(6, 2)
    want = dequantize_iq1_s(blocks)
    lib = GgufOracle().lib
    out = np.empty(6 * 256, np.float32)
    lib.ktxo_dequant_iq1_s(blocks.ctypes.data_as(C.c_void_p), C.c_int(6), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out.reshape(6, 256), want)
    # dot product of a block row with Q8_K codes == de-quantised math (fp32 rounding only)
    o = GgufOracle()
    x = rng.standard_normal(256 * 6).astype(np.float32)
    q, d8, bs = o.quantize_row_q8_K(x)
    lib.ktxo_vec_dot_iq1_s.restype = C.c_float
    got = lib.ktxo_vec_dot_iq1_s(blocks.ctypes.data_as(C.c_void_p), C.c_int(256 * 6), q.ctypes.data_as(C.c_void_p),
                                 d8.ctypes.data_as(C.c_void_p), bs.ctypes.data_as(C.c_void_p))
    xq = (q.astype(np.float64).reshape(-1, 256) * d8.astype(np.float64)[:, None]).reshape(-1)
    ref = float(want.reshape(-1).astype(np.float64) @ xq)
    assert abs(got - ref) <= 2e-5 * abs(ref) + 1e-6
    w = (rng.standard_normal((4, 512)) / 10).astype(np.float32)
    assert np.linalg.norm(dequantize_iq1_s(quantize_iq1_s(w)) - w) / np.linalg.norm(w) < 0.75     # ~1.6 bits per weight

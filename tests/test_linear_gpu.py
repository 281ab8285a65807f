"""GPU parity of the quantised linears (include/ktx_linear.h) against oracle/linear_ref.py, through the C ABI.

Tolerances (floating point path, fp32 accumulation in a different order than the un-vendored CUDA/Triton kernels):
  W4/BF16: |y - ref| <= 2^-7*|ref| + 2e-3*max|ref|  (one bf16 ulp of the output + accumulation-order noise);
           norm-wise relative error <= 1e-3 against exact (q-8)*s math in fp64.
  FP8    : same bound; activations that sit on an e4m3 rounding tie after x/s may quantise one step apart, which the
           norm-wise bound (2e-3) absorbs.
The quantiser is integer work: bit-exact against the reference's own quantize_weights (golden fixture)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.linear_ref import linear_bf16_ref, linear_fp8_ref, linear_w4_ref, quantize_weights_ref  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "linear_w4_golden.npz")


def native():
    from ktransformers_amd import _native
    return _native


def close(y, ref, rel=1e-3):
    y, ref = y.float().cpu(), ref.float()
    tol = 2.0 ** -7 * ref.abs() + 2e-3 * ref.abs().max()
    bad = (y - ref).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} elements out of tolerance, max diff {(y - ref).abs().max()}"
    assert (y - ref).norm() / ref.norm().clamp_min(1e-30) <= max(rel, 2.0 ** -8), float((y - ref).norm() / ref.norm())


@pytest.mark.parametrize("name", ["a", "b"])
@pytest.mark.parametrize("G", [32, 64, 128])
def test_w4_quantizer_bit_exact_vs_reference_golden(name, G):
    n = native()
    g = np.load(GOLD)
    w = torch.from_numpy(g[f"{name}_w"]).view(torch.bfloat16).cuda()
    N, K = w.shape
    h = n.LinearHandle(K, N, "W4", G, 16)
    h.load_bf16(w)
    q, s = h.debug_get_w4()
    assert np.array_equal(s, g[f"{name}_s{G}"])
    live = np.repeat(g[f"{name}_s{G}"] != 0, G, axis=0)
    assert np.array_equal(q[live], g[f"{name}_q{G}"][live])


SHAPES = [(256, 64), (2048, 576), (2048, 3072), (1536, 200), (7168, 1536), (384, 48)]


# every shape at the group size the models use (64); the other group sizes on two shapes
W4_CASES = [(K, N, T, G) for G in (64, 128, 32) for T in (1, 2, 3, 4, 7, 16, 33, 130) for (K, N) in SHAPES
            if G == 64 or (K, N) in ((256, 64), (2048, 576))]


@pytest.mark.parametrize("K,N,T,G", W4_CASES)
def test_w4_forward(K, N, T, G):
    n = native()
    torch.manual_seed(K + N + T)
    w = (torch.randn(N, K) / 10).to(torch.bfloat16)
    x = (torch.randn(T, K) / 10).to(torch.bfloat16)
    bias = (torch.randn(N) / 10).to(torch.bfloat16) if (T % 2 == 0) else None
    q, s = quantize_weights_ref(w.T.contiguous(), G)
    h = n.LinearHandle(K, N, "W4", G, 256)
    h.load_bf16(w.cuda(), bias.cuda() if bias is not None else None)
    y = h.forward(x.cuda())
    close(y, linear_w4_ref(x, q, s, G, bias))
    if T <= 4:   # the same rows through the general kernel
        n.linear_force_gemm(True)
        try:
            y2 = h.forward(x.cuda())
        finally:
            n.linear_force_gemm(False)
        close(y2, linear_w4_ref(x, q, s, G, bias))
    # pre-quantised upload gives the same tiles
    h2 = n.LinearHandle(K, N, "W4", G, 256)
    h2.load_w4(q.to(torch.uint8).cuda(), s.cuda(), bias.cuda() if bias is not None else None)
    assert torch.equal(h2.forward(x.cuda()), y)


@pytest.mark.parametrize("K,N", SHAPES)
@pytest.mark.parametrize("T", [1, 4, 5, 40])
def test_bf16_forward(K, N, T):
    n = native()
    torch.manual_seed(K + N + T)
    w = (torch.randn(N, K) / 10).to(torch.bfloat16)
    x = (torch.randn(T, K) / 10).to(torch.bfloat16)
    h = n.LinearHandle(K, N, "BF16", 0, 64)
    h.load_bf16(w.cuda())
    close(h.forward(x.cuda()), linear_bf16_ref(x, w), rel=1e-3)


@pytest.mark.parametrize("K,N", [(256, 64), (2048, 576), (7168, 1536), (1536, 200), (384, 48)])
@pytest.mark.parametrize("T", [1, 3, 4, 9, 40])
def test_fp8_forward(K, N, T):
    n = native()
    torch.manual_seed(K + N + T)
    w = (torch.randn(N, K) / 4).to(torch.float8_e4m3fn)
    sc = (torch.rand((N + 127) // 128, K // 128) + 0.5) / 32
    x = (torch.randn(T, K) / 10).to(torch.bfloat16)
    bias = (torch.randn(N) / 10).to(torch.bfloat16) if T == 4 else None
    h = n.LinearHandle(K, N, "FP8", 128, 64)
    h.load_fp8(w.cuda(), sc.cuda(), bias.cuda() if bias is not None else None)
    y = h.forward(x.cuda())
    close(y, linear_fp8_ref(x, w, sc, bias), rel=2e-3)


@pytest.mark.parametrize("K,N,T", [(256, 64, 128), (1536, 200, 300), (2048, 576, 513), (7168, 2112, 2048), (512, 4096, 1000)])
def test_fp8_prompt_gemm_against_the_strip_kernel(K, N, T, monkeypatch):
    """Prompt-sized FP8 calls (T >= 128) run act_quant once + lin_fp8_gemm_kernel (csrc/ktx_linear_fp8gemm.inc): the same four MFMAs
    per 128-block in the same order, the same (dot * a_s) * b_s accumulation and the same output rounding as the strip kernel every
    other test of this file pins to the reference (lin_gemm_kernel<FP8>) — so the two paths must agree bit for bit: ragged T and N
    (partial 128-tiles), bias, the GLU epilogue, the decoder layer's adds, a fused input norm, strided rows; and against
    oracle/linear_ref.py at the file's FP8 bound."""
    n = native()
    torch.manual_seed(K + N + T)
    w = (torch.randn(N, K) / 4).to(torch.float8_e4m3fn)
    sc = (torch.rand((N + 127) // 128, K // 128) + 0.5) / 32
    x = (torch.randn(T, K) / 10).to(torch.bfloat16).cuda()
    bias = (torch.randn(N) / 10).to(torch.bfloat16) if N % 3 == 0 else None
    h = n.LinearHandle(K, N, "FP8", 128, T)
    h.load_fp8(w.cuda(), sc.cuda(), bias.cuda() if bias is not None else None)
    add1 = (torch.randn(T, N) / 10).to(torch.bfloat16).cuda()
    nw = (1 + torch.randn(K) / 10).to(torch.bfloat16).cuda()
    xs = torch.zeros(T, K + 64, dtype=torch.bfloat16, device="cuda")[:, :K]
    xs.copy_(x)
    calls = [dict(), dict(add1=add1), dict(norm=(nw, 1e-6)), dict(add1=add1, add2=add1)]
    if N % 16 == 0 and bias is None:
        calls.append(dict(glu=True))
    # round 6: the default GEMM sums a 128-block in ONE K = 128 MFMA (twice the rate; the hardware's own order inside the block) —
    # within one bf16 ulp of the strip kernel; dev knob 27 = 1 keeps the four K = 32 MFMAs, which ARE the strip kernel's bits
    fast = [h.forward(x, **kw) for kw in calls] + [h.forward(xs)]
    n.check(n.lib.ktx_debug_set(27, 1))
    try:
        got = [h.forward(x, **kw) for kw in calls] + [h.forward(xs)]
    finally:
        n.check(n.lib.ktx_debug_set(27, 0))
    monkeypatch.setenv("KTX_FP8_PROMPT_KERNEL", "1")
    want = [h.forward(x, **kw) for kw in calls] + [h.forward(xs)]
    for a, b, f, kw in zip(got, want, fast, calls + [dict(strided=True)]):
        assert a.shape == b.shape and torch.equal(a, b), f"{kw.keys()}: {(a != b).sum().item()} of {a.numel()} outputs differ"
        d = (f.float() - b.float()).abs()
        bound = 2.0 ** -7 * b.float().abs() + 2.0 ** -10 * float(b.float().abs().max())
        assert f.shape == b.shape and bool((d <= bound).all()), f"{kw.keys()}: K = 128 MFMA path beyond one bf16 ulp of the strip kernel"
        assert float((f != b).float().mean()) < 0.02, f"{kw.keys()}: {float((f != b).float().mean()):.4f} of the outputs differ"
    close(got[0], linear_fp8_ref(x.cpu(), w, sc, bias), rel=2e-3)
    close(fast[0], linear_fp8_ref(x.cpu(), w, sc, bias), rel=2e-3)


@pytest.mark.parametrize("K,N,G", [(2048, 576, 64), (1536, 200, 32), (7168, 2112, 128), (1024, 64, -1), (384, 48, 128)])
def test_w8_format_is_the_bf16_kernel_on_marlins_multiplicand_bit_for_bit(K, N, G):
    """The W8 format (include/ktx_linear.h: KTX_LIN_W8, round 4): one byte per weight in HBM, bf16((q - 128) * s) formed in registers
    in front of the BF16 format's MFMAs — so a W8 handle and a BF16 handle loaded with that very matrix must agree bit for bit, for
    decode rows, small and prompt-sized batches, with bias, fused norm, adds and the GLU epilogue; per-channel scales (G = -1) ride
    as groups of 128.  (The multiplicand itself is pinned to the reference's quantize_weights in tests/test_linear_cpu.py.)"""
    n = native()
    from ktransformers_amd.operators.linear import marlin_multiplicand, marlin_quantize
    torch.manual_seed(K + N + G)
    w = (torch.randn(N, K) / 10).to(torch.bfloat16).cuda()
    g = K if G == -1 else G
    hg = min(g, 128)
    bias = (torch.randn(N) / 10).to(torch.bfloat16).cuda() if N % 3 == 0 else None
    q, sc = marlin_quantize(w, 8, g)
    h8 = n.LinearHandle(K, N, "W8", hg, 512)
    h8.load_w8((q + 128).to(torch.uint8).view(N, K).T.contiguous(), sc.view(N, -1).repeat_interleave(g // hg, dim=1).T.contiguous(), bias)
    hb = n.LinearHandle(K, N, "BF16", 0, 512)
    hb.load_bf16(marlin_multiplicand(w, 8, g).contiguous(), bias)
    assert h8.weight_bytes() < 0.6 * hb.weight_bytes()
    nw = (1 + torch.randn(K) / 10).to(torch.bfloat16).cuda()
    for T in (1, 2, 4, 5, 33, 300):
        x = (torch.randn(T, K) / 10).to(torch.bfloat16).cuda()
        add1 = (torch.randn(T, N) / 10).to(torch.bfloat16).cuda()
        calls = [dict(), dict(norm=(nw, 1e-6)), dict(add1=add1)]
        if N % 16 == 0 and bias is None:
            calls.append(dict(glu=True))
        for kw in calls:
            a, b = h8.forward(x, **kw), hb.forward(x, **kw)
            assert torch.equal(a, b), f"T={T} {list(kw)}: {(a != b).sum().item()} of {a.numel()} outputs differ"


def test_fp8_linears_merge_on_block_boundaries():
    """build_merged_linear for block-fp8 checkpoints (round 4): q_a_proj (1536 rows = 12 whole 128-row scale blocks) | kv_a_proj_with_mqa
    (576 rows: the last block is partial) as ONE operator gives exactly the rows the two operators give, decode and prompt sized; a first
    matrix that does not end on a block boundary is refused (None: the separate operators stay)."""
    from types import SimpleNamespace
    from ktransformers_amd.operators.linear import KLinearFP8, build_merged_linear
    from ktransformers_amd.util.loader import DictLoader
    torch.manual_seed(3)
    K = 1024
    def mk(N):
        return (torch.randn(N, K) / 4).to(torch.float8_e4m3fn), (torch.rand((N + 127) // 128, K // 128) + 0.5) / 32
    (wa, sa), (wb, sb), (wc, sc) = mk(1536), mk(576), mk(200)
    ld = DictLoader({"a.weight": wa, "a.weight_scale_inv": sa, "b.weight": wb, "b.weight_scale_inv": sb, "c.weight": wc, "c.weight_scale_inv": sc})
    ops = {}
    for key, N in (("a", 1536), ("b", 576), ("c", 200)):
        ops[key] = KLinearFP8(key, ld, SimpleNamespace(), torch.nn.Linear(K, N, bias=False, device="meta"), device="cuda")
        ops[key].load()
    merged = build_merged_linear(ops["a"], ["a", "b"], ld, "cuda")
    assert merged is not None and merged[1] == [1536, 576]
    assert build_merged_linear(ops["c"], ["c", "b"], ld, "cuda") is None          # 200 rows: not a block boundary
    for T in (1, 3, 200):
        x = (torch.randn(T, K) / 10).to(torch.bfloat16).cuda()
        y = merged[0].forward(x)
        assert torch.equal(y[:, :1536], ops["a"].forward(x)) and torch.equal(y[:, 1536:], ops["b"].forward(x))


def test_fp8_mlp_runs_gate_and_up_as_one_operator():
    """KDeepseekV3MLP on a block-fp8 checkpoint (round 4): gate_proj and up_proj are loaded as ONE operator over the concatenation
    [gate ; up] (whole 128-row scale blocks; the W4 GLU interleave would split blocks) — the SiLU * up input, and so the block's
    output, must be bit for bit what the separate operators give, with and without the fused input norm, decode and prompt sized."""
    from types import SimpleNamespace
    from ktransformers_amd._native import rmsnorm, silu_mul
    from ktransformers_amd.operators.linear import KLinearFP8, KTransformersLinear
    from ktransformers_amd.operators.mlp import KDeepseekV3MLP
    from ktransformers_amd.util.loader import DictLoader
    torch.manual_seed(5)
    H, I = 1024, 512
    def mk(N, K):
        return (torch.randn(N, K) / 4).to(torch.float8_e4m3fn), (torch.rand((N + 127) // 128, K // 128) + 0.5) / 32
    st = {}
    for nm, (N, K) in (("gate_proj", (I, H)), ("up_proj", (I, H)), ("down_proj", (H, I))):
        w, sc = mk(N, K)
        st[f"mlp.{nm}.weight"], st[f"mlp.{nm}.weight_scale_inv"] = w, sc
    ld, cfg = DictLoader(st), SimpleNamespace()
    class Orig(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for nm, (N, K) in (("gate_proj", (I, H)), ("up_proj", (I, H)), ("down_proj", (H, I))):
                setattr(self, nm, KTransformersLinear(f"mlp.{nm}", ld, cfg, torch.nn.Linear(K, N, bias=False, device="meta"), "cuda",
                                                      "KLinearFP8", "cuda", "KLinearFP8"))
    mlp = KDeepseekV3MLP("mlp", ld, cfg, Orig(), "cuda", "cuda")
    mlp.load()
    assert mlp._gate_up is None and mlp._gate_up_cat is not None and mlp._gate_up_cat._h.N == 2 * I
    ops = {nm: KLinearFP8(f"mlp.{nm}", ld, cfg, torch.nn.Linear(H, I, bias=False, device="meta"), device="cuda") for nm in ("gate_proj", "up_proj")}
    for op in ops.values():
        op.load()
    nw = (1 + torch.randn(H) / 10).to(torch.bfloat16).cuda()
    for T in (1, 3, 130):
        x = (torch.randn(T, H) / 10).to(torch.bfloat16).cuda()
        for norm in (None, (nw, 1e-6)):
            xn = x if norm is None else rmsnorm(x, norm[0], norm[1], native_rounding=True)
            want = silu_mul(torch.cat([ops["gate_proj"].forward(xn), ops["up_proj"].forward(xn)], dim=-1))
            assert torch.equal(mlp.gate_up(x, norm), want), f"T={T} norm={norm is not None}"
            # the whole block: SiLU * up rides in down_proj's prologue on decode-sized calls (glu_in) — same bits as the three launches
            res = (torch.randn(T, H) / 10).to(torch.bfloat16).cuda()
            assert torch.equal(mlp.forward(x, add1=res, norm=norm), mlp.down(want, x.shape, add1=res)), f"T={T} block"


@pytest.mark.parametrize("fmt,K,N", [("FP8", 2048, 7168), ("FP8", 1536, 576), ("W4", 2048, 7168), ("W4", 1408, 2048), ("BF16", 512, 200)])
def test_silu_mul_in_the_prologue_of_a_decode_linear(fmt, K, N):
    """ktx_linear_fusion.glu_in (round 5): x rows are [gate | up] and the block-fp8 decode kernel stages silu(gate) * up itself — the
    same roundings as ktx_silu_mul, so bit for bit the two-launch result, with the epilogue adds, the bsz tensor and for every decode
    row count; a prompt-sized call, and the other formats (their MLPs have the GLU epilogue instead), take the separate launch inside
    LinearHandle.forward — same call, same bits."""
    n = native()
    torch.manual_seed(K + N)
    h = n.LinearHandle(K, N, fmt, 128 if fmt == "FP8" else 64, 64)
    if fmt == "FP8":
        h.load_fp8((torch.randn(N, K) / 4).to(torch.float8_e4m3fn).cuda(), ((torch.rand((N + 127) // 128, K // 128) + 0.5) / 32).cuda())
    else:
        h.load_bf16((torch.randn(N, K) / 10).to(torch.bfloat16).cuda())
    for T in (1, 2, 3, 4, 9):
        gu = (torch.randn(T, 2 * K) * 2).to(torch.bfloat16).cuda()
        a1 = torch.randn(T, N).to(torch.bfloat16).cuda()
        a2 = torch.randn(T, N).to(torch.bfloat16).cuda()
        act = n.silu_mul(gu)
        n.timing_collect()
        n.timing_enable(2)
        try:
            got = h.forward(gu, glu_in=True)
            torch.cuda.synchronize()
            labels = [lab for lab, _, _ in n.timing_collect()]
        finally:
            n.timing_enable(0)
        assert torch.equal(got, h.forward(act)), (fmt, T)
        if fmt == "FP8" and T <= 4:       # one launch, and it is the decode kernel with the prologue
            assert len(labels) == 1 and "silu*up in" in labels[0], labels
        assert torch.equal(h.forward(gu, add1=a1, add2=a2, glu_in=True), h.forward(act, add1=a1, add2=a2)), (fmt, T, "adds")
        if T == 3:
            bsz = torch.tensor([2], dtype=torch.int32, device="cuda")
            got, want = h.forward(gu, bsz, glu_in=True), h.forward(n.silu_mul(gu, bsz)[:T], bsz)
            assert torch.equal(got[:2], want[:2]) and bool((got[2:] == 0).all())
    with pytest.raises(n.KtxError):
        h.forward(torch.zeros(1, 2 * K, dtype=torch.bfloat16, device="cuda"), norm=(torch.ones(K, dtype=torch.bfloat16, device="cuda"), 1e-6), glu_in=True)


def test_bsz_tensor_and_graph_capture():
    n = native()
    torch.manual_seed(0)
    K, N = 2048, 576
    w = (torch.randn(N, K) / 10).to(torch.bfloat16)
    x = (torch.randn(4, K) / 10).to(torch.bfloat16).cuda()
    q, s = quantize_weights_ref(w.T.contiguous(), 64)
    h = n.LinearHandle(K, N, "W4", 64, 16)
    h.load_bf16(w.cuda())
    bsz = torch.tensor([2], dtype=torch.int32, device="cuda")
    out = torch.full((4, N), 7.0, dtype=torch.bfloat16, device="cuda")
    h.forward(x, bsz, out)
    close(out[:2], linear_w4_ref(x[:2].cpu(), q, s, 64))
    assert torch.all(out[2:] == 7.0)
    g = torch.cuda.CUDAGraph()
    y = torch.empty((4, N), dtype=torch.bfloat16, device="cuda")
    s0 = torch.cuda.Stream()
    with torch.cuda.stream(s0):
        h.forward(x, None, y)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s0):
            h.forward(x, None, y)
    y.zero_()
    g.replay()
    torch.cuda.synchronize()
    close(y, linear_w4_ref(x.cpu(), q, s, 64))


def test_errors_are_loud():
    n = native()
    with pytest.raises(n.KtxError):
        n.LinearHandle(100, 64, "W4", 64, 16)          # in_features % 8
    with pytest.raises(n.KtxError):
        n.LinearHandle(256, 64, "W4", 48, 16)          # group size
    h = n.LinearHandle(256, 64, "W4", 64, 4)
    x = torch.zeros(2, 256, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(n.KtxError):
        h.forward(x)                                    # not loaded
    h.load_bf16(torch.zeros(64, 256, dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(n.KtxError):
        h.forward(torch.zeros(8, 256, dtype=torch.bfloat16, device="cuda"))   # T > max_len
    assert torch.all(h.forward(x) == 0)


@pytest.mark.parametrize("T", [1, 3, 4, 9, 40])
@pytest.mark.parametrize("fmt", ["W4", "BF16"])
def test_fused_norm_glu_and_addends(T, fmt):
    """ktx_linear_forward_fused: RMSNorm prologue, [gate|up] GLU epilogue and two addends equal the unfused sequence
    (same kernels, same roundings) bit for bit."""
    n = native()
    torch.manual_seed(T)
    K, I, N = 512, 256, 384
    x = (torch.randn(T, K)).to(torch.bfloat16).cuda()
    nw = (1 + 0.1 * torch.randn(K)).to(torch.bfloat16).cuda()
    g = (torch.randn(I, K) / 10).to(torch.bfloat16).cuda()
    u = (torch.randn(I, K) / 10).to(torch.bfloat16).cuda()
    hg, hu = n.LinearHandle(K, I, fmt, 64, 64), n.LinearHandle(K, I, fmt, 64, 64)
    hg.load_bf16(g); hu.load_bf16(u)
    inter = torch.stack([g.view(-1, 8, K), u.view(-1, 8, K)], dim=1).reshape(-1, K).contiguous()
    hm = n.LinearHandle(K, 2 * I, fmt, 64, 64)
    hm.load_bf16(inter)
    xn = n.rmsnorm(x, nw, 1e-6, native_rounding=True)
    ref = n.silu_mul(torch.cat([hg.forward(xn), hu.forward(xn)], dim=-1))
    got = hm.forward(x, norm=(nw, 1e-6), glu=True)
    assert got.shape == (T, I)
    # the fused prologue sums the squares in a different order: allow the rare 1-ulp flip of a normalised activation
    assert (got != ref).float().mean() < 0.02 and torch.allclose(got.float(), ref.float(), rtol=2 ** -6, atol=1e-3)
    assert torch.equal(hm.forward(xn, glu=True), ref)
    # addends
    w = (torch.randn(N, I) / 10).to(torch.bfloat16).cuda()
    hd = n.LinearHandle(I, N, fmt, 64, 64)
    hd.load_bf16(w)
    a1 = torch.randn(T, N).to(torch.bfloat16).cuda()
    a2 = torch.randn(T, N).to(torch.bfloat16).cuda()
    y = hd.forward(ref)
    want = a2 + (a1 + y)
    assert torch.equal(hd.forward(ref, add1=a1, add2=a2), want)
    # strided input rows (a slice of a wider buffer)
    wide = torch.zeros(T, I + 64, dtype=torch.bfloat16, device="cuda")
    wide[:, :I] = ref
    assert torch.equal(hd.forward(wide[:, :I]), y)


@pytest.mark.parametrize("K,N,G", [(2048, 576, 64), (7168, 1536, 64), (1536, 200, 64), (2048, 576, 128), (2048, 576, 32)])
@pytest.mark.parametrize("T", [33, 130, 257])
def test_w4_prompt_kernel_with_several_strips_per_wavefront(K, N, G, T):
    """lin_gemm_w4n_kernel (prompt-sized W4, 2 / 4 strips per wavefront, packed-fp32 dequantisation epilogue) performs the
    one-strip kernel's operations element for element: bit-identical outputs, with bias, addends and the GLU epilogue,
    ragged feature counts (N % 64 != 0) and ragged token tiles."""
    n = native()
    torch.manual_seed(K + N + T + G)
    w = (torch.randn(N, K) / 10).to(torch.bfloat16).cuda()
    x = (torch.randn(T, K) / 10).to(torch.bfloat16).cuda()
    bias = (torch.randn(N) / 10).to(torch.bfloat16).cuda()
    a1 = torch.randn(T, N).to(torch.bfloat16).cuda()
    h = n.LinearHandle(K, N, "W4", G, 512)
    h.load_bf16(w, bias)
    hg = None
    if N % 16 == 0:
        hg = n.LinearHandle(K, N, "W4", G, 512)
        hg.load_bf16(w)
    outs = {}
    try:
        for knob in (1, 2, 4):
            n.lib.ktx_debug_set(12, knob)
            outs[knob] = (h.forward(x), h.forward(x, add1=a1), hg.forward(x, glu=True) if hg is not None else None)
    finally:
        n.lib.ktx_debug_set(12, 0)
    q, s = quantize_weights_ref(w.cpu().T.contiguous(), G)
    close(outs[1][0], linear_w4_ref(x.cpu(), q, s, G, bias.cpu()))
    for knob in (2, 4):
        for a, b in zip(outs[1], outs[knob]):
            if a is not None:
                assert torch.equal(a, b), f"strips/wavefront = {knob}: {int((a != b).sum())} of {a.numel()} outputs differ"


@pytest.mark.parametrize("H,T", [(128, 1), (64, 3), (16, 4)])
def test_qb_absorb_and_prep_in_one_launch(H, T):
    """ktx_linear_forward_qb_absorb (q_b_proj with q_a_layernorm, the per-head q-absorb products, RoPE of q_pe and the kv half of
    mla_prep, one launch) against the separate calls it replaces: q_b (fused norm) -> absorb_and_prep.  RoPE, the latent norm and
    the absorb products run the same arithmetic on the same q values; q_b's 12 k-steps are summed as two halves instead of one
    run, so q agrees within fp32 re-association (a bf16 ulp here and there), which the comparison allows for."""
    n = native()
    nope, rope, lora, qlora = 128, 64, 512, 1536
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(H * 10 + T)
    wqb = (torch.randn((H * (nope + rope), qlora), generator=g) / 20).to(torch.bfloat16).to(dev)
    wuk = (torch.randn((H, lora, nope), generator=g) / 10).to(torch.bfloat16).to(dev)
    qa_full = torch.randn((T, qlora + lora + rope), generator=g).to(torch.bfloat16).to(dev)      # the merged q_a | kv_a GEMV output
    q_a, kv = qa_full[:, :qlora], qa_full[:, qlora:]
    nw = (1 + 0.1 * torch.randn((qlora,), generator=g)).to(torch.bfloat16).to(dev)
    knw = (1 + 0.1 * torch.randn((lora,), generator=g)).to(torch.bfloat16).to(dev)
    pos = torch.tensor([5, 77, 4000, 123456][:T], dtype=torch.int64, device=dev)
    inv_freq = (1.0 / (10000 ** (torch.arange(0, rope, 2).float() / rope))).to(dev)
    qb = n.LinearHandle(qlora, H * (nope + rope), "W4", 64, 8, dev)
    qb.load_bf16(wqb)
    qabs = n.LinearHandle(nope, lora, "BF16", 0, 8, dev, batch=H)
    qabs.load_bf16(wuk)
    assert n.qb_absorb_eligible(qb, qabs, T, H, nope, rope, lora)
    q = qb.forward(q_a, norm=(nw, 1e-6)).reshape(T, H * (nope + rope))
    ref = n.absorb_and_prep(qabs, q, kv, knw, 1e-6, pos, inv_freq, 1.3, H, nope, rope, lora)
    for rep in range(2):
        got = n.qb_absorb_and_prep(qb, qabs, q_a, (nw, 1e-6), kv, knw, 1e-6, pos, inv_freq, 1.3, H, nope, rope, lora)
        torch.cuda.synchronize()
        assert torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3]), "kv half of mla_prep: same code, same bits"
        for a, b, what in ((got[0], ref[0], "q_nope (absorbed)"), (got[1], ref[1], "q_pe")):
            a, b = a.float(), b.float()
            assert torch.isfinite(a).all()
            rel = float((a - b).norm() / b.norm())
            assert rel < 2e-3, (what, rel)
            assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()), what


@pytest.mark.parametrize("K,N,G", [(2048, 576, 64), (7168, 1536, 64), (1536, 208, 128)])
def test_prompt_sized_w4_dequantises_once_and_runs_a_library_gemm(K, N, G):
    """T >= LinearHandle.PROMPT_MIN_T: the W4 weights are expanded once with Marlin's in-register rounding bf16((q - 8) * s)
    (ktx_linear_dequant_bf16: bit-identical to the torch expression on the reference quantiser's q and s) and a plain library
    GEMM runs on them — against fp64 math on the same rounded weights, against the hand-written W4 GEMM, and with the GLU /
    addend epilogues of the decoder layer."""
    import os
    n = native()
    torch.manual_seed(K + N)
    T = n.LinearHandle.PROMPT_MIN_T + 40
    w = (torch.randn(N, K) / 10).to(torch.bfloat16).cuda()
    x = (torch.randn(T, K) / 10).to(torch.bfloat16).cuda()
    a1 = torch.randn(T, N).to(torch.bfloat16).cuda()
    a2 = torch.randn(T, N).to(torch.bfloat16).cuda()
    h = n.LinearHandle(K, N, "W4", G, 1024)
    h.load_bf16(w)
    q, s = quantize_weights_ref(w.cpu().T.contiguous(), G)
    wdq = ((q.float() - 8.0) * s.float().repeat_interleave(G, dim=0)).T.contiguous().to(torch.bfloat16)      # [N, K]
    assert torch.equal(h.dequant_bf16().cpu(), wdq)
    y = h.forward(x)
    close(y, linear_w4_ref(x.cpu(), q, s, G, None, round_weights=True), rel=2e-3 ** 1)
    os.environ["KTX_W4_PROMPT_KERNEL"] = "1"
    try:
        y_k = h.forward(x)
        y_k_add = h.forward(x, add1=a1, add2=a2)
    finally:
        os.environ.pop("KTX_W4_PROMPT_KERNEL")
    assert float((y.float() - y_k.float()).norm() / y_k.float().norm()) < 3e-3      # exact-integer weights vs bf16-rounded ones
    y_add = h.forward(x, add1=a1, add2=a2)
    assert float((y_add.float() - y_k_add.float()).norm() / y_k_add.float().norm()) < 3e-3
    assert torch.equal(y_add, a2 + (a1 + y))
    if N % 16 == 0:
        hg = n.LinearHandle(K, N, "W4", G, 1024)
        hg.load_bf16(w)
        got = hg.forward(x, glu=True)
        os.environ["KTX_W4_PROMPT_KERNEL"] = "1"
        try:
            ref = hg.forward(x, glu=True)
        finally:
            os.environ.pop("KTX_W4_PROMPT_KERNEL")
        assert got.shape == ref.shape == (T, N // 2)
        assert float((got.float() - ref.float()).norm() / ref.float().norm()) < 5e-3


# ---- lin_sk_kernel (round 3): the decode GEMV dealt over all CUs ----------------------------------------------------------
# Shapes: DeepSeek-V3 linears (k-steps 56 / 128 / 144 / 12 / 16 -> ring depths 7 / 8 / 6 / 4), a glu pair, ragged N.
SK_SHAPES = [(7168, 2112), (16384, 7168), (18432, 1024), (1536, 3072), (2048, 7168), (7168, 4096), (896, 200), (1408, 512)]


def _knobs(n, **kv):
    class K:
        def __enter__(self):
            for k, v in kv.items():
                n.lib.ktx_debug_set(int(k[1:]), v)

        def __exit__(self, *a):
            for k in kv:
                n.lib.ktx_debug_set(int(k[1:]), 0)
    return K()


@pytest.mark.parametrize("K,N", SK_SHAPES)
@pytest.mark.parametrize("T", [1, 2, 4])
def test_all_cu_decode_gemv_both_deals_against_the_reference_math(K, N, T):
    """The all-CU decode GEMV with whole strips per workgroup (default) and with single groups dealt (knob 17 = 1: strips
    shared between workgroups meet through the fixed-point words) against exact (q - 8) * s math, against round 2's
    lin_dec_kernel (knob 16 = 1), and against itself: the shared-strip path adds integers, so replays are bit-identical
    whatever the arrival order, and the meeting words must be back at zero after every launch."""
    n = native()
    torch.manual_seed(K * 3 + N + T)
    w = (torch.randn(N, K) / 10).to(torch.bfloat16)
    x = (torch.randn(T, K) / 10).to(torch.bfloat16)
    q, s = quantize_weights_ref(w.T.contiguous(), 64)
    ref = linear_w4_ref(x, q, s, 64, None)
    h = n.LinearHandle(K, N, "W4", 64, 16)
    h.load_bf16(w.cuda())
    xg = x.cuda()
    y_strips = h.forward(xg)
    close(y_strips, ref)
    with _knobs(n, k17=1):
        y_groups = [h.forward(xg).clone() for _ in range(4)]
    close(y_groups[0], ref)
    for y in y_groups[1:]:
        assert torch.equal(y, y_groups[0]), "shared strips: replays differ (arrival-order dependence or words not reset)"
    with _knobs(n, k16=1):
        y_old = h.forward(xg)
    close(y_old, ref)
    # the two deals differ only in fp32 association (+ the 2^-32 grid of the meeting words): within one bf16 ulp of each other
    d = (y_strips.float() - y_groups[0].float()).abs()
    assert bool((d <= y_strips.float().abs() * 2.0 ** -7 + 1e-6).all())


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), 3.0e38])
def test_shared_strip_words_do_not_swallow_non_finite_parts(bad):
    """ADVICE r3 / VERDICT r4 #11: a strip shared between workgroups meets through fixed-point words; a part that is NaN, Inf or
    beyond the words' range used to be clamped into a NUMBER (fminf / fmaxf drop a NaN).  It now travels as a marker no sum of
    in-range parts can reach and the finisher writes NaN: a poisoned input row gives non-finite outputs under BOTH deals, the
    words are back at zero afterwards (a clean row right behind it is exact again)."""
    n = native()
    torch.manual_seed(11)
    K, N = 7168, 2112
    w = (torch.randn(N, K) / 10).to(torch.bfloat16)
    h = n.LinearHandle(K, N, "W4", 64, 16)
    h.load_bf16(w.cuda())
    x = (torch.randn(1, K) / 10).to(torch.bfloat16)
    xb = x.clone()
    xb[0, 5] = bad                      # (3e38 * a weight of ~0.1 over a 1 KiB tile stays finite in fp32 but leaves the words' +-2^18 range)
    q, s = quantize_weights_ref(w.T.contiguous(), 64)
    ref = linear_w4_ref(x, q, s, 64, None)
    for knobs in ({}, {"k17": 1}):
        with _knobs(n, **knobs):
            y_bad = h.forward(xb.cuda())
            y_ok = h.forward(x.cuda())
        assert not bool(torch.isfinite(y_bad.float()).all()), f"deal {knobs or 'whole strips'}: the poisoned row came out finite"
        if knobs:   # every output of the row depends on x[5] (one k-step of every strip): all of them must be flagged
            assert float(torch.isfinite(y_bad.float()).float().mean()) < 0.01
        close(y_ok, ref)


@pytest.mark.parametrize("T", [1, 3])
def test_all_cu_decode_gemv_epilogues(T):
    """Fused RMSNorm prologue, glu, bias and both addends through lin_sk_kernel == the same call through lin_dec_kernel up to
    fp32 association (the arithmetic of the prologue and the epilogues is shared expression by expression)."""
    n = native()
    K, N = 7168, 4096
    torch.manual_seed(5 + T)
    w = (torch.randn(N, K) / 10).to(torch.bfloat16).cuda()
    x = torch.randn(T, K).to(torch.bfloat16).cuda()
    nw = (1 + 0.1 * torch.randn(K)).to(torch.bfloat16).cuda()
    a1 = torch.randn(T, N).to(torch.bfloat16).cuda()
    a2 = torch.randn(T, N).to(torch.bfloat16).cuda()
    bias = torch.randn(N).to(torch.bfloat16).cuda()
    h = n.LinearHandle(K, N, "W4", 64, 16)
    h.load_bf16(w, bias)
    hg = n.LinearHandle(K, N, "W4", 64, 16)
    hg.load_bf16(w)
    for kw, hh in ((dict(norm=(nw, 1e-6), add1=a1, add2=a2), h), (dict(norm=(nw, 1e-6), glu=True), hg), (dict(add1=a1), h)):
        y_new = hh.forward(x, **kw)
        with _knobs(n, k16=1):
            y_old = hh.forward(x, **kw)
        with _knobs(n, k17=1):
            y_grp = hh.forward(x, **kw)
        for y in (y_new, y_grp):
            d = (y.float() - y_old.float()).abs()
            assert bool((d <= y_old.float().abs() * 2.0 ** -6 + 2e-2 * y_old.float().abs().max()).all()), float(d.max())
            assert float((y.float() - y_old.float()).norm() / y_old.float().norm()) < 4e-3


def test_all_cu_decode_gemv_under_a_replayed_graph_and_many_handles():
    """Meeting words, slab arena and launch geometry under graph replay: 12 handles created and destroyed in shuffled order
    (slabs are returned when their last piece goes), a captured chain replayed 20 times gives the same bits every time."""
    n = native()
    torch.manual_seed(3)
    K, N = 2048, 1536
    hs, ws = [], []
    for i in range(12):
        w = (torch.randn(N, K) / 10).to(torch.bfloat16).cuda()
        h = n.LinearHandle(K, N, "W4", 64, 8)
        h.load_bf16(w)
        hs.append(h); ws.append(w)
    for i in (1, 5, 7, 2):
        hs[i].close()
    live = [h for i, h in enumerate(hs) if i not in (1, 5, 7, 2)]
    x = (torch.randn(1, K) / 10).to(torch.bfloat16).cuda()
    ys = [torch.empty(1, N, dtype=torch.bfloat16, device="cuda") for _ in live]
    with _knobs(n, k17=1):
        for h, y in zip(live, ys):
            h.forward(x, out=y)
        torch.cuda.synchronize()
        first = [y.clone() for y in ys]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for h, y in zip(live, ys):
                h.forward(x, out=y)
        for _ in range(20):
            for y in ys:
                y.zero_()
            g.replay()
            torch.cuda.synchronize()
            for y, f in zip(ys, first):
                assert torch.equal(y, f)


@pytest.mark.parametrize("act_order", [False, True])
@pytest.mark.parametrize("T", [1, 5, 70])
def test_marlin_8bit_operator_multiplies_with_the_reference_multiplicand(T, act_order):
    """KLinearMarlin(num_bits=8) (linear.py:608-666): Marlin's 8-bit multiplicand bf16((q - 128) * s) — pinned to the reference's
    own quantize_weights in tests/test_linear_cpu.py — now formed in registers from the W8 format (one byte per weight); forward
    against fp64 math on that matrix, at the bound of the BF16 kernels (one bf16 rounding of an fp32-accumulated sum)."""
    from types import SimpleNamespace
    from ktransformers_amd.operators.linear import KLinearMarlin, marlin_multiplicand
    from ktransformers_amd.util.loader import DictLoader
    K, N, G = 2048, 576, 64
    torch.manual_seed(T + 31 * act_order)
    w = (torch.randn(N, K) / 10).to(torch.bfloat16)
    x = (torch.randn(T, K) / 10).to(torch.bfloat16)
    op = KLinearMarlin("k", DictLoader({"k.weight": w}), SimpleNamespace(), torch.nn.Linear(K, N, bias=False, device="meta"),
                       device="cuda", num_bits=8, group_size=G, act_order=act_order)
    op.load()
    assert op._h.fmt == "W8" and N * K <= op._h.weight_bytes() < 1.1 * N * K
    y = op.forward(x.cuda())
    torch.cuda.synchronize()
    close(y.cpu(), linear_bf16_ref(x, marlin_multiplicand(w, 8, G)), rel=1e-3)
    # the 8-bit grid is 16x finer than the 4-bit one
    dense = x.double() @ w.double().T
    assert float((y.cpu().double() - dense).norm() / dense.norm()) < 1e-2


@pytest.mark.parametrize("fmt", ["W4", "FP8", "BF16"])
@pytest.mark.parametrize("K,N,T", [(2048, 576, 5), (7168, 2112, 8), (1536, 3072, 7), (16384, 7168, 8)])
def test_five_to_eight_rows_run_as_four_row_passes_of_the_decode_kernel(fmt, K, N, T):
    """Round 6: a 5..8-row call of an unbatched handle is two launches of the decode GEMV (rows 0-3, rows 4..), so every row carries the
    decode kernel's bits — with the fused RMSNorm, the addends (offset by the pass), the GLU epilogue, strided rows and a device-side row
    count — and stays within the file's bound of the 16-row-strip kernel it replaces (dev knob 3 = 1)."""
    n = native()
    torch.manual_seed(K + N + T)
    w = (torch.randn(N, K) / 10).to(torch.bfloat16).cuda()
    if fmt == "FP8":
        h = n.LinearHandle(K, N, "FP8", 128, 16)
        h.load_fp8((w.float() * 4).to(torch.float8_e4m3fn), ((torch.rand((N + 127) // 128, K // 128) + 0.5) / 32).cuda())
    else:
        h = n.LinearHandle(K, N, fmt, 64 if fmt == "W4" else 0, 16)
        h.load_bf16(w)
    x = (torch.randn(T, K) / 4).to(torch.bfloat16).cuda()
    add1 = (torch.randn(T, N) / 10).to(torch.bfloat16).cuda()
    nw = (1 + torch.randn(K) / 10).to(torch.bfloat16).cuda()
    xs = torch.zeros(T, K + 64, dtype=torch.bfloat16, device="cuda")[:, :K]
    xs.copy_(x)
    assert h.decode_eligible(T) and h.decode_eligible(4)
    calls = [dict(), dict(add1=add1), dict(norm=(nw, 1e-6)), dict(add1=add1, add2=add1), dict(norm=(nw, 1e-6), add1=add1)]
    if N % 16 == 0 and fmt == "W4":
        calls.append(dict(glu=True))

    def rows(kw, a, b):
        kw2 = {k: (v[a:b] if k in ("add1", "add2") else v) for k, v in kw.items()}
        return h.forward(x[a:b].contiguous(), **kw2)

    for kw in calls:
        got = h.forward(x, **kw)
        want = torch.cat([rows(kw, 0, 4), rows(kw, 4, T)], dim=0)
        assert got.shape == want.shape and torch.equal(got, want), f"{fmt} {tuple(kw)}: the passes are not the decode kernel's rows"
    assert torch.equal(h.forward(xs), h.forward(x)), "strided rows"
    bsz = torch.tensor([T - 2], dtype=torch.int32, device="cuda")
    yb = h.forward(x, bsz_tensor=bsz)
    assert torch.equal(yb[: T - 2], h.forward(x)[: T - 2]) and not bool(yb[T - 2:].any()), "device-side row count across the passes"
    fast = [h.forward(x, **kw) for kw in calls[:2]]
    n.check(n.lib.ktx_debug_set(3, 1))
    try:
        assert not h.decode_eligible(T)
        slow = [h.forward(x, **kw) for kw in calls[:2]]
    finally:
        n.check(n.lib.ktx_debug_set(3, 0))
    for a, b in zip(fast, slow):
        close(a, b.cpu())

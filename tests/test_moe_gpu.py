"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on identical seeded inputs, against the
reference-generated golden fixtures, and size-independent properties at full model shapes.  Bit-exact: every rounding
step of the reference is reproduced and all accumulations are either exact (int32) or in the reference's order."""
import os

import numpy as np
import pytest
import torch

from helpers import bf16_to_f32, f32_to_bf16, make_case, numpy_u16, torch_bf16
from oracle.oracle import FMT_AMXINT4, FMT_AMXINT8

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "moe_amx_golden.npz")
FMT = {"AMXINT4": FMT_AMXINT4, "AMXINT8": FMT_AMXINT8}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def make_handle(method, c, E, k, H, I, max_len, dev, **kw):
    from ktransformers_amd._native import MoEHandle
    h = MoEHandle(E, k, H, I, max_len=max_len, method=method, device=0, **kw)
    h.load_bf16(torch_bf16(c["gate"], dev), torch_bf16(c["up"], dev), torch_bf16(c["down"], dev))
    return h


def run(h, c, dev, **kw):
    y = h.forward(torch_bf16(c["x"], dev), torch.from_numpy(c["ids"]).to(dev), torch.from_numpy(c["w"]).to(dev), **kw)
    torch.cuda.synchronize()
    return numpy_u16(y)


@pytest.mark.parametrize("method", ["AMXINT4", "AMXINT8"])
@pytest.mark.parametrize("shape", [
    (8, 2, 512, 256, 1),      # decode path, one token
    (8, 2, 512, 256, 5),      # decode path, ragged routing with invalid ids
    (8, 6, 2048, 1408, 1),    # DeepSeek-V2-Lite layer shape (K=1408 -> 11 k-steps)
    (16, 8, 7168, 2048, 2),   # DeepSeek-V3 layer shape, 16 of the 256 experts
    (8, 2, 256, 512, 40),     # grouped path, MT=1
    (8, 2, 256, 512, 300),    # grouped path, MT=4, several tiles per expert
    (4, 2, 256, 256, 700),    # grouped path, ragged last tiles
])
def test_parity_with_oracle_both_paths(oracle, dev, method, shape):
    from ktransformers_amd import _native
    E, k, H, I, T = shape
    c = make_case(1, E, k, H, I, T, invalid_ids=T >= 5)
    mo = oracle.make_moe(FMT[method], c["gate"], c["up"], c["down"])
    want = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    want_inc = oracle.moe_forward(mo, c["ids"], c["w"], c["x"], y_prev=want)
    h = make_handle(method, c, E, k, H, I, max(T, 8), dev)
    try:
        for force_generic in (False, True):
            _native.force_generic_path(force_generic)
            got = run(h, c, dev)
            assert np.array_equal(got, want), f"{int((got != want).sum())} bf16 outputs differ (generic={force_generic})"
            got_inc = run(h, c, dev, out=torch_bf16(want, dev), incremental=True)
            assert np.array_equal(got_inc, want_inc)
    finally:
        _native.force_generic_path(False)
        h.close()


@pytest.mark.parametrize("method", ["AMXINT4", "AMXINT8"])
@pytest.mark.parametrize("shape", [
    (4, 2, 256, 256, 700),     # one k-chunk, K a multiple of the ring depth, ragged last tiles + invalid ids
    (4, 2, 1152, 640, 300),    # 9 / 5 k-steps: ring tails; 40 strips: half-empty last strip group
    (2, 2, 4352, 384, 200),    # K = 4352 > 2048: three LDS chunks (16 + 16 + 2 k-steps)
])
def test_streaming_prompt_kernels(oracle, dev, method, shape):
    """The LDS-resident / register-ring grouped GEMM for prompts (moe_gemm_stream_kernel) forced on (dev knob 4 = 2), against
    the oracle and against the chunk-pipelined kernels (knob 4 = 1) on the same input: bit-exact."""
    from ktransformers_amd import _native
    E, k, H, I, T = shape
    c = make_case(3, E, k, H, I, T, invalid_ids=True)
    mo = oracle.make_moe(FMT[method], c["gate"], c["up"], c["down"])
    want = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    h = make_handle(method, c, E, k, H, I, T, dev)
    try:
        got = {}
        for knob in (1, 2):
            _native.lib.ktx_debug_set(4, knob)
            got[knob] = run(h, c, dev)
        assert np.array_equal(got[2], want), f"{int((got[2] != want).sum())} bf16 outputs differ (streaming kernels)"
        assert np.array_equal(got[1], want)
    finally:
        _native.lib.ktx_debug_set(4, 0)
        h.close()


@pytest.mark.parametrize("method", ["AMXINT4", "AMXINT8"])
@pytest.mark.parametrize("shape", [
    (4, 2, 256, 256, 700),     # 350 rows per expert: one full 256-row tile + a ragged one
    (4, 2, 1152, 640, 300),    # 9 / 5 k-steps; 40 strips: half-empty last strip group
    (8, 6, 2048, 1408, 600),   # DeepSeek-V2-Lite expert shape
])
def test_register_tile_prompt_kernels(oracle, dev, method, shape):
    """moe_gemm_rt_kernel (256-row register tiles; dev knob 4 = 3 — not selected by default until it has been timed) against the
    chunk-pipelined kernels (knob 4 = 1) and the oracle: bit-exact."""
    from ktransformers_amd import _native
    E, k, H, I, T = shape
    c = make_case(5, E, k, H, I, T, invalid_ids=True)
    h = make_handle(method, c, E, k, H, I, T, dev)
    try:
        got = {}
        for knob in (1, 3):
            _native.lib.ktx_debug_set(4, knob)
            got[knob] = run(h, c, dev)
        assert np.array_equal(got[3], got[1]), f"{int((got[3] != got[1]).sum())} bf16 outputs differ (register-tile kernels)"
        if T * k * H * I <= 2 ** 31:
            mo = oracle.make_moe(FMT[method], c["gate"], c["up"], c["down"])
            assert np.array_equal(got[1], oracle.moe_forward(mo, c["ids"], c["w"], c["x"]))
    finally:
        _native.lib.ktx_debug_set(4, 0)
        h.close()


@pytest.mark.parametrize("fname,method", [("int4", "AMXINT4"), ("int8", "AMXINT8")])
@pytest.mark.parametrize("case", ["t1", "t7_invalid", "t33_prefill"])
def test_parity_with_reference_golden(dev, fname, method, case):
    g = np.load(GOLDEN)
    E, k, H, I = int(g["E"]), int(g["k"]), int(g["H"]), int(g["I"])
    c = dict(gate=g["gate"], up=g["up"], down=g["down"], x=g[f"{fname}_{case}_x"], ids=g[f"{fname}_{case}_ids"],
             w=g[f"{fname}_{case}_w"])
    h = make_handle(method, c, E, k, H, I, 64, dev)
    try:
        got = run(h, c, dev)
        assert np.array_equal(got, g[f"{fname}_{case}_y"]), "HIP path differs from the reference kernels' own output"
        got_inc = run(h, c, dev, out=torch_bf16(g[f"{fname}_{case}_y"], dev), incremental=True)
        assert np.array_equal(got_inc, g[f"{fname}_{case}_yinc"])
    finally:
        h.close()


def test_prequantised_load_equals_online_quant(oracle, dev):
    """load_quantized(host q, scale) (reference pre-quantised branch, moe.hpp:266-300) == load_bf16 (online quant)."""
    from ktransformers_amd._native import MAT_DOWN, MAT_GATE, MAT_UP, MoEHandle
    E, k, H, I, T = 4, 2, 256, 128, 6
    c = make_case(9, E, k, H, I, T)
    mo = oracle.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"])
    h1 = make_handle("AMXINT4", c, E, k, H, I, 8, dev)
    h2 = MoEHandle(E, k, H, I, max_len=8, method="AMXINT4", device=0)
    for e in range(E):
        h2.load_quantized(e, MAT_GATE, mo["gate_q"][e], mo["gate_d"][e])
        h2.load_quantized(e, MAT_UP, mo["up_q"][e], mo["up_d"][e])
        h2.load_quantized(e, MAT_DOWN, mo["down_q"][e], mo["down_d"][e])
    try:
        assert np.array_equal(run(h1, c, dev), run(h2, c, dev))
    finally:
        h1.close(); h2.close()


def test_expert_mask_and_bsz_tensor(oracle, dev):
    E, k, H, I, T = 8, 2, 256, 256, 6
    c = make_case(4, E, k, H, I, T)
    mask = np.zeros(E, np.uint8); mask[[1, 5]] = 1
    mo = oracle.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"], mask=mask)
    want = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    h = make_handle("AMXINT4", c, E, k, H, I, 8, dev)
    try:
        h.set_expert_mask(mask)
        assert np.array_equal(run(h, c, dev), want)
        # masking an expert == routing to an invalid id (should_skip_expert, common.hpp:256-258)
        h.set_expert_mask(None)
        c2 = dict(c, ids=np.where(np.isin(c["ids"], [1, 5]), -1, c["ids"]))
        assert np.array_equal(run(h, c2, dev), want)
        # device-side batch size (moe-tp.hpp:209): only the first bsz tokens are touched
        bsz = torch.tensor([4], dtype=torch.int32, device=dev)
        sentinel = f32_to_bf16(np.full((T, H), 7.0, np.float32))
        out = torch_bf16(sentinel, dev)
        got = run(h, c, dev, out=out, bsz_tensor=bsz)
        full = oracle.moe_forward(oracle.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"]), c["ids"], c["w"], c["x"])
        assert np.array_equal(got[:4], full[:4]) and np.array_equal(got[4:], sentinel[4:])
    finally:
        h.close()


def test_expert_parallel_partials_sum_to_the_whole(oracle, dev):
    """Two handles owning half the experts each: fp32 partials add up to the single-handle result (<= 1 bf16 ulp)."""
    from ktransformers_amd._native import MoEHandle
    E, k, H, I, T = 8, 4, 256, 256, 5
    c = make_case(12, E, k, H, I, T)
    whole = make_handle("AMXINT4", c, E, k, H, I, 8, dev)
    parts = []
    for r in range(2):
        h = MoEHandle(E // 2, k, H, I, max_len=8, method="AMXINT4", device=0, expert_begin=r * E // 2, global_expert_num=E)
        sl = slice(r * E // 2, (r + 1) * E // 2)
        h.load_bf16(torch_bf16(c["gate"][sl], dev), torch_bf16(c["up"][sl], dev), torch_bf16(c["down"][sl], dev))
        parts.append(h)
    try:
        x, ids, w = torch_bf16(c["x"], dev), torch.from_numpy(c["ids"]).to(dev), torch.from_numpy(c["w"]).to(dev)
        s = parts[0].forward_partial(x, ids, w) + parts[1].forward_partial(x, ids, w)
        ref = bf16_to_f32(run(whole, c, dev))
        got = s.to(torch.bfloat16).float().cpu().numpy()
        assert np.all(np.abs(got - ref) <= np.abs(ref) * 2.0 ** -7 + 1e-5 * np.abs(ref).max())
    finally:
        whole.close(); [p.close() for p in parts]


# ---- size-independent properties at full layer shapes ------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(64, 6, 2048, 1408), (32, 8, 7168, 2048)])  # V2-Lite full layer; V3 shape, 32 experts
def test_properties_full_shapes(dev, shape):
    from ktransformers_amd import _native
    from ktransformers_amd._native import MoEHandle
    E, k, H, I = shape
    g = torch.Generator(device=dev); g.manual_seed(3)
    gate = (torch.randn((E, I, H), generator=g, device=dev) / 10).to(torch.bfloat16)
    up = (torch.randn((E, I, H), generator=g, device=dev) / 10).to(torch.bfloat16)
    down = (torch.randn((E, H, I), generator=g, device=dev) / 10).to(torch.bfloat16)
    T = 24
    h = MoEHandle(E, k, H, I, max_len=T, method="AMXINT4", device=0)
    h.load_bf16(gate, up, down)
    del gate, up, down
    try:
        x = (torch.randn((T, H), generator=g, device=dev) / 100).to(torch.bfloat16)
        ids = torch.rand((T, E), generator=g, device=dev).topk(k, dim=-1).indices.to(torch.int64).contiguous()
        w = torch.rand((T, k), generator=g, device=dev)
        y = h.forward(x, ids, w)                                 # grouped path (T*k > 64)
        # (1) every row equals the same token pushed alone through the decode path
        for t in (0, 7, T - 1):
            y1 = h.forward(x[t:t + 1].contiguous(), ids[t:t + 1].contiguous(), w[t:t + 1].contiguous())
            assert torch.equal(y1[0], y[t]), f"token {t}: batched and single-token results differ"
        # (2) token permutation equivariance
        perm = torch.randperm(T, generator=torch.Generator().manual_seed(0)).to(dev)
        yp = h.forward(x[perm].contiguous(), ids[perm].contiguous(), w[perm].contiguous())
        assert torch.equal(yp, y[perm])
        # (3) scaling the routing weights by 2 scales the output by exactly 2 (power of two: no rounding change)
        y2 = h.forward(x, ids, (w * 2).contiguous())
        assert torch.equal(y2.float(), y.float() * 2)
        # (4) slot-order permutation of one token changes at most fp32 summation order: <= 1 bf16 ulp
        idsr, wr = ids.flip(-1).contiguous(), w.flip(-1).contiguous()
        yr = h.forward(x, idsr, wr).float()
        # (a different fp32 order changes the sum by ~1e-7 of the largest partial sum, then at most one bf16 ulp)
        yf = y.float()
        assert torch.all((yr - yf).abs() <= yf.abs() * 2.0 ** -7 + 1e-5 * yf.abs().max())
        # (5) incremental == previous + new, rounded once
        yi = h.forward(x, ids, w, out=y.clone(), incremental=True)
        assert torch.all((yi.float() - 2 * yf).abs() <= (2 * yf).abs() * 2.0 ** -7 + 1e-5 * yf.abs().max())
        # (6) both implementations agree on the same small batch
        xs, idss, ws = x[:4].contiguous(), ids[:4].contiguous(), w[:4].contiguous()
        a = h.forward(xs, idss, ws)
        _native.force_generic_path(True)
        b = h.forward(xs, idss, ws)
        _native.force_generic_path(False)
        assert torch.equal(a, b)
    finally:
        _native.force_generic_path(False)
        h.close()


def test_full_v2lite_layer_vs_oracle(oracle, dev):
    """One full DeepSeek-V2-Lite MoE layer shape, 8 of the 64 experts routed, against the CPU oracle."""
    E, k, H, I, T = 8, 6, 2048, 1408, 3
    c = make_case(21, E, k, H, I, T)
    mo = oracle.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"])
    want = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    h = make_handle("AMXINT4", c, E, k, H, I, 8, dev)
    try:
        assert np.array_equal(run(h, c, dev), want)
    finally:
        h.close()


def test_hip_graph_capture_and_variable_batch(oracle, dev):
    """The forward is capturable and the captured graph honours a device-side batch size (reference: CUDAGraphRunner +
    bsz_tensor, experts.py:293-318)."""
    E, k, H, I, T = 8, 2, 256, 256, 4
    c = make_case(2, E, k, H, I, T)
    mo = oracle.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"])
    want = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    h = make_handle("AMXINT4", c, E, k, H, I, 8, dev)
    try:
        x, ids, w = torch_bf16(c["x"], dev), torch.from_numpy(c["ids"]).to(dev), torch.from_numpy(c["w"]).to(dev)
        out = torch.zeros_like(x)
        bsz = torch.tensor([T], dtype=torch.int32, device=dev)
        h.forward(x, ids, w, out=out, bsz_tensor=bsz)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            h.forward(x, ids, w, out=out, bsz_tensor=bsz)
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        assert np.array_equal(numpy_u16(out), want)
        out.zero_(); bsz.fill_(2)
        g.replay(); torch.cuda.synchronize()
        got = numpy_u16(out)
        assert np.array_equal(got[:2], want[:2]) and not got[2:].any()
    finally:
        h.close()


def test_error_behaviour(dev):
    from ktransformers_amd._native import KtxError, MoEHandle
    h = MoEHandle(4, 2, 256, 128, max_len=4, method="AMXINT4", device=0)
    try:
        x = torch.zeros((8, 256), dtype=torch.bfloat16, device=dev)
        with pytest.raises(KtxError):   # qlen > max_len
            h.forward(x, torch.zeros((8, 2), dtype=torch.int64, device=dev), torch.zeros((8, 2), device=dev))
        with pytest.raises(KtxError):   # wrong dtype
            h.forward(x[:2].float(), torch.zeros((2, 2), dtype=torch.int64, device=dev), torch.zeros((2, 2), device=dev))
        with pytest.raises(KtxError):
            MoEHandle(4, 2, 200, 128, max_len=4, method="AMXINT4", device=0)
        with pytest.raises(KtxError):
            MoEHandle(4, 2, 256, 128, max_len=4, method="Q9_9", device=0)
    finally:
        h.close()


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()


# ---- FP8 (DeepSeek block scales) and BF16 experts ----------------------------------------------------------------------
# The reference accumulates each 128-K group as a sequential fp32 VDPBF16PS chain (and in tile order on its AMX path);
# the MFMA adds the same exact products in another order.  Outputs therefore agree to fp32 rounding of the partial
# sums: the test allows one bf16 ulp (2^-7 relative) plus 2^-9 of the row scale, on a small fraction of elements,
# and requires the aggregate relative error (the reference tests' own metric) to be far below north_star's 1e-3.
def _check_fp(got_u16, want_u16):
    a, b = bf16_to_f32(got_u16), bf16_to_f32(want_u16)
    assert np.all(np.abs(a - b) <= np.abs(b) * 2.0 ** -7 + 2.0 ** -9 * np.abs(b).max())
    assert (got_u16 != want_u16).mean() < 0.05
    assert np.abs(a - b).mean() / max(np.abs(b).mean(), 1e-30) < 1e-3


def test_fp8_decode_at_deepseek_v3_expert_shape(oracle, dev):
    """BASELINE config C3 (DeepSeek-V3 fp8 experts, 7168 x 2048, top-8): the two-launch decode path against the oracle that is
    pinned to the reference's own AMX_FP8_MOE_TP, and against the grouped path on the same input."""
    from helpers import fp8_block_quant
    from ktransformers_amd._native import MoEHandle, force_generic_path
    E, k, H, I, T = 8, 8, 7168, 2048, 2
    c = make_case(11, E, k, H, I, T)
    q = [fp8_block_quant(bf16_to_f32(c[n])) for n in ("gate", "up", "down")]
    mo = oracle.make_moe_fp8(q[0][0], q[1][0], q[2][0], q[0][1], q[1][1], q[2][1])
    want = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    h = MoEHandle(E, k, H, I, max_len=8, method="FP8", device=0, group_size=128)
    try:
        h.load_fp8(*[torch.from_numpy(x[0]).to(dev) for x in q], *[torch.from_numpy(x[1]).to(dev) for x in q])
        got = run(h, c, dev)
        _check_fp(got, want)
        force_generic_path(True)
        try:
            grouped = run(h, c, dev)
        finally:
            force_generic_path(False)
        _check_fp(got, grouped)
    finally:
        h.close()


@pytest.mark.parametrize("fmt", ["FP8", "BF16"])
@pytest.mark.parametrize("shape", [(8, 2, 512, 256, 1), (8, 2, 512, 256, 5), (8, 6, 2048, 1408, 2), (8, 2, 256, 512, 40),
                                   (8, 2, 256, 512, 300), (4, 2, 256, 256, 700)])
def test_fp_formats_against_oracle(oracle, dev, fmt, shape):
    from helpers import fp8_block_quant
    from ktransformers_amd._native import MoEHandle
    E, k, H, I, T = shape
    c = make_case(6, E, k, H, I, T, invalid_ids=T >= 5)
    h = MoEHandle(E, k, H, I, max_len=max(T, 8), method=fmt, device=0, group_size=128 if fmt == "FP8" else 0)
    try:
        if fmt == "FP8":
            q = [fp8_block_quant(bf16_to_f32(c[n])) for n in ("gate", "up", "down")]
            mo = oracle.make_moe_fp8(q[0][0], q[1][0], q[2][0], q[0][1], q[1][1], q[2][1])
            h.load_fp8(*[torch.from_numpy(x[0]).to(dev) for x in q], *[torch.from_numpy(x[1]).to(dev) for x in q])
        else:
            mo = oracle.make_moe_bf16(c["gate"], c["up"], c["down"])
            h.load_bf16(torch_bf16(c["gate"], dev), torch_bf16(c["up"], dev), torch_bf16(c["down"], dev))
        want = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
        _check_fp(run(h, c, dev), want)
        want_inc = oracle.moe_forward(mo, c["ids"], c["w"], c["x"], y_prev=want)
        _check_fp(run(h, c, dev, out=torch_bf16(want, dev), incremental=True), want_inc)
    finally:
        h.close()


@pytest.mark.parametrize("shape", [(8, 2, 512, 256, 1), (8, 2, 512, 256, 5), (8, 8, 7168, 2048, 2), (8, 2, 256, 512, 40),
                                   (4, 2, 256, 256, 700)])
def test_fp8_perchannel_against_oracle(oracle, dev, shape):
    """FP8_PERCHANNEL (e4m3 weights, one fp32 scale per output row; the method the reference serves GLM-4.7-FP8 style
    checkpoints with) against the oracle that tests/test_oracle_cpu.py pins bit for bit to the reference's own
    AMX_FP8_PERCHANNEL_MOE_TP: decode kernels (T*k <= 64), the grouped path on the same input, invalid ids, incremental.
    Same bound as the block-scaled format: the fp32 summation order over K is the MFMA's, not the AVX512 chain's."""
    from helpers import fp8_perchannel_quant
    from ktransformers_amd import _native
    from ktransformers_amd._native import MoEHandle
    E, k, H, I, T = shape
    c = make_case(12, E, k, H, I, T, invalid_ids=T >= 5)
    q = [fp8_perchannel_quant(bf16_to_f32(c[n])) for n in ("gate", "up", "down")]
    mo = oracle.make_moe_fp8_perchannel(q[0][0], q[1][0], q[2][0], q[0][1], q[1][1], q[2][1])
    want = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    want_inc = oracle.moe_forward(mo, c["ids"], c["w"], c["x"], y_prev=want)
    h = MoEHandle(E, k, H, I, max_len=max(T, 8), method="FP8_PERCHANNEL", device=0)
    try:
        h.load_fp8_perchannel(*[torch.from_numpy(x[0]).to(dev) for x in q], *[torch.from_numpy(x[1]).to(dev) for x in q])
        for generic in (False, True):
            _native.force_generic_path(generic)
            _check_fp(run(h, c, dev), want)
            _check_fp(run(h, c, dev, out=torch_bf16(want, dev), incremental=True), want_inc)
    finally:
        _native.force_generic_path(False)
        h.close()


@pytest.mark.parametrize("fname,fmt", [("fp8", "FP8"), ("fp8pc", "FP8_PERCHANNEL"), ("bf16", "BF16")])
@pytest.mark.parametrize("case", ["t1", "t7_invalid", "t33_prefill"])
def test_fp_formats_against_reference_golden(dev, fname, fmt, case):
    from ktransformers_amd._native import MoEHandle
    g = np.load(GOLDEN)
    E, k, H, I = int(g["E"]), int(g["k"]), int(g["H"]), int(g["I"])
    c = dict(x=g[f"int4_{case}_x"], ids=g[f"int4_{case}_ids"], w=g[f"int4_{case}_w"])
    h = MoEHandle(E, k, H, I, max_len=64, method=fmt, device=0, group_size=128 if fmt == "FP8" else 0)
    try:
        if fmt == "FP8":
            h.load_fp8(*[torch.from_numpy(g[f"fp8_{n}"]).to(dev) for n in ("gate", "up", "down")],
                       *[torch.from_numpy(g[f"fp8_{n}_s"]).to(dev) for n in ("gate", "up", "down")])
        elif fmt == "FP8_PERCHANNEL":
            h.load_fp8_perchannel(*[torch.from_numpy(g[f"fp8pc_{n}"]).to(dev) for n in ("gate", "up", "down")],
                                  *[torch.from_numpy(g[f"fp8pc_{n}_s"]).to(dev) for n in ("gate", "up", "down")])
        else:
            h.load_bf16(torch_bf16(g["gate"], dev), torch_bf16(g["up"], dev), torch_bf16(g["down"], dev))
        _check_fp(run(h, c, dev), g[f"{fname}_{case}_y"])
    finally:
        h.close()


# ---- RAWINT4 (Kimi-K2 native int4, group 32): bit-exact --------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(8, 2, 512, 512, 1), (8, 2, 512, 512, 5), (8, 3, 1024, 512, 2), (16, 8, 7168, 2048, 1),
                                   (8, 2, 512, 1024, 40), (4, 2, 512, 512, 300)])
def test_rawint4_bit_exact(oracle, dev, shape):
    from helpers import rawint4_quantize
    from ktransformers_amd import _native
    from ktransformers_amd._native import MoEHandle
    E, k, H, I, T = shape
    c = make_case(8, E, k, H, I, T, invalid_ids=T >= 5)
    q = [rawint4_quantize(bf16_to_f32(c[n])) for n in ("gate", "up", "down")]
    mo = oracle.make_moe_rawint4(q[0][0], q[1][0], q[2][0], q[0][1], q[1][1], q[2][1])
    want = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    h = MoEHandle(E, k, H, I, max_len=max(T, 8), method="RAWINT4", device=0, group_size=32)
    try:
        h.load_rawint4(*[torch.from_numpy(x[0]).to(dev) for x in q], *[torch_bf16(x[1], dev) for x in q])
        want_inc = oracle.moe_forward(mo, c["ids"], c["w"], c["x"], y_prev=want)
        _native.lib.ktx_debug_set(29, 1)  # the exact kernels at every size (prompt chunks of >= 64 tokens otherwise take the
        #                                   re-associating chunk kernel: test_rawint4_prompt_chunks)
        for generic in (False, True):     # T*k <= 64: the two-launch decode kernels, then the grouped path on the same input
            _native.force_generic_path(generic)
            got = run(h, c, dev)
            assert np.array_equal(got, want), f"{int((got != want).sum())} of {want.size} bf16 outputs differ (generic={generic})"
            assert np.array_equal(run(h, c, dev, out=torch_bf16(want, dev), incremental=True), want_inc)
    finally:
        _native.lib.ktx_debug_set(29, 0)
        _native.force_generic_path(False)
        h.close()


@pytest.mark.parametrize("shape", [(4, 2, 512, 512, 300), (8, 3, 1024, 512, 130), (16, 8, 7168, 2048, 64), (6, 2, 1536, 1024, 257)])
def test_rawint4_prompt_chunks(oracle, dev, shape):
    """Prompt chunks (qlen >= 64) of the RAWINT4 format go through moe_rawint4_chunk_kernel: the same int8 x int4 products and the
    same per-group scale products as the reference's GemmKernel224Int4SmallKGroup, summed as ONE fp32 chain over the 32-k groups
    instead of sixteen interleaved chains + a tree (csrc/ktx_moe.hip).  Bound: that of the FP8 / BF16 formats, whose prompt kernels
    re-associate the same way (element-wise 2^-7 |ref| + 2^-9 max|ref|, < 5 % of the bf16 outputs different at all, mean error
    < 1e-3 of the mean magnitude); and against the exact kernels of this library on the same input (MoEHandle.set_exact), Kimi-K2's
    expert shape included, ragged tiles and invalid ids included."""
    from helpers import rawint4_quantize
    from ktransformers_amd import _native
    from ktransformers_amd._native import MoEHandle
    E, k, H, I, T = shape
    c = make_case(21, E, k, H, I, T, invalid_ids=True)
    q = [rawint4_quantize(bf16_to_f32(c[n])) for n in ("gate", "up", "down")]
    mo = oracle.make_moe_rawint4(q[0][0], q[1][0], q[2][0], q[0][1], q[1][1], q[2][1])
    want = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    h = MoEHandle(E, k, H, I, max_len=T, method="RAWINT4", device=0, group_size=32)
    try:
        h.load_rawint4(*[torch.from_numpy(x[0]).to(dev) for x in q], *[torch_bf16(x[1], dev) for x in q])
        got = run(h, c, dev)
        _check_fp(got, want)
        h.set_exact(True)                 # the PUBLIC switch (include/ktx_moe.h ktx_moe_set_exact): bit-identical at any size
        exact = run(h, c, dev)
        assert np.array_equal(exact, want)
        assert (got != exact).mean() < 0.05
        h.set_exact(False)
        assert np.array_equal(run(h, c, dev), got)        # ... and back to the fast kernel
        if H * I <= (1 << 22):            # (the oracle's second pass over the Kimi-K2 shape would cost another ~15 s of CPU)
            want_inc = oracle.moe_forward(mo, c["ids"], c["w"], c["x"], y_prev=want)
            _check_fp(run(h, c, dev, out=torch_bf16(want, dev), incremental=True), want_inc)
    finally:
        _native.lib.ktx_debug_set(29, 0)
        h.close()


@pytest.mark.parametrize("case", ["t1", "t7_invalid", "t33_prefill"])
def test_rawint4_against_reference_golden(dev, case):
    """Bit-exact against outputs of the reference's OWN Kimi-K2 class (TP_MOE<AMX_K2_MOE_TP<GemmKernel224Int4SmallKGroup>>,
    k2-moe.hpp:124-191) on weights quantised by the reference test's own rawint4_quantize (tests/golden/make_golden.py)."""
    from ktransformers_amd import _native
    from ktransformers_amd._native import MoEHandle
    g = np.load(GOLDEN)
    E, k, H, I = int(g["k2_E"]), int(g["k2_k"]), int(g["k2_H"]), int(g["k2_I"])
    c = dict(x=g[f"k2_{case}_x"], ids=g[f"k2_{case}_ids"], w=g[f"k2_{case}_w"])
    h = MoEHandle(E, k, H, I, max_len=64, method="RAWINT4", device=0, group_size=32)
    try:
        h.load_rawint4(*[torch.from_numpy(g[f"k2_{n}_p"]).to(dev) for n in ("gate", "up", "down")],
                       *[torch_bf16(g[f"k2_{n}_s"], dev) for n in ("gate", "up", "down")])
        want = g[f"k2_{case}_y"]
        for generic in (False, True):
            _native.force_generic_path(generic)
            got = run(h, c, dev)
            assert np.array_equal(got, want), f"{int((got != want).sum())} of {want.size} bf16 outputs differ (generic={generic})"
            assert np.array_equal(run(h, c, dev, out=torch_bf16(want, dev), incremental=True), g[f"k2_{case}_yinc"])
    finally:
        _native.force_generic_path(False)
        h.close()


@pytest.mark.parametrize("method", ["AMXINT4", "AMXINT8"])
@pytest.mark.parametrize("shape", [
    (16, 8, 7168, 2048, 2048, 1),   # DeepSeek-V3 / Kimi-K2 layer shape: shared intermediate 2048 -> 2 side k-steps per wavefront
    (16, 8, 7168, 2048, 2048, 3),
    (8, 6, 2048, 1408, 2816, 1),    # DeepSeek-V2-Lite: two shared experts (2816 = 22 k-steps over 6 wavefronts, ragged)
    (8, 6, 2048, 1408, 2816, 4),
    (8, 2, 512, 256, 384, 5),       # invalid routing ids beside the side strip
])
def test_forward_side_equals_the_separate_launches(dev, method, shape):
    """ktx_moe_forward_side (tail of a MoE block: routed experts + the shared experts' W4 down_proj + the two closing bf16 adds)
    against ktx_moe_forward followed by ktx_linear_forward_fused(add1 = routed, add2 = residual).  With the combined kernel
    switched off (knob 14) the library runs exactly those launches: bit-identical.  The combined kernel deals the side strip's
    k-steps out to the routed wavefronts (k-slices of other lengths than the stand-alone GEMV's, the same k-step arithmetic):
    the routed part stays bit-exact, the sum agrees within the fp32 re-association of the side GEMV (<= 1 bf16 ulp of the
    largest operand per element)."""
    from ktransformers_amd import _native as n
    E, k, H, I, Ks, T = shape
    c = make_case(3, E, k, H, I, T, invalid_ids=T >= 5)
    h = make_handle(method, c, E, k, H, I, 8, dev)
    g = torch.Generator().manual_seed(E + H + T)
    wd = (torch.randn((H, Ks), generator=g) / 16).to(torch.bfloat16).to(dev)
    sx = (torch.randn((T, Ks), generator=g) / 4).to(torch.bfloat16).to(dev)
    res = torch.randn((T, H), generator=g).to(torch.bfloat16).to(dev)
    lin = n.LinearHandle(Ks, H, "W4", 64, 8, dev)
    lin.load_bf16(wd)
    x, ids, w = torch_bf16(c["x"], dev), torch.from_numpy(c["ids"]).to(dev), torch.from_numpy(c["w"]).to(dev)
    routed = h.forward(x, ids, w)
    want = lin.forward(sx, add1=routed, add2=res)
    want_nores = lin.forward(sx, add1=routed)
    try:
        n.lib.ktx_debug_set(14, 1)
        assert torch.equal(h.forward_side(x, ids, w, lin, sx, res), want)
        assert torch.equal(h.forward_side(x, ids, w, lin, sx, None), want_nores)
    finally:
        n.lib.ktx_debug_set(14, 0)
    side_only = lin.forward(sx).float()
    for r, ref in ((res, want), (None, want_nores)):
        got = h.forward_side(x, ids, w, lin, sx, r)
        torch.cuda.synchronize()
        scale = torch.maximum(torch.maximum(routed.float().abs(), side_only.abs()), ref.float().abs())
        bad = (got.float() - ref.float()).abs() > scale * 2.0 ** -7 + 1e-30
        assert not bool(bad.any()), f"{int(bad.sum())} of {bad.numel()} outputs differ by more than one bf16 ulp of the operands"
        assert float((got != ref).float().mean()) < 0.2, "re-association noise should touch a minority of the outputs"


@pytest.mark.parametrize("dup", [False, True])
@pytest.mark.parametrize("shape", [(64, 6, 256, 128, 700), (256, 8, 256, 128, 2048), (8, 2, 256, 256, 1500), (16, 4, 256, 128, 100)])
def test_bucketing_is_the_reference_stable_order(oracle, dev, shape, dup):
    """a5: the (token, slot) -> sorted-row map of the grouped path is the reference's m_local_pos_ (token-major arrival rank
    inside each expert, experts in ascending order; moe_base.hpp:208-227) — the wavefront-ballot counting sort of
    moe_prep_kernel is deterministic, so the map can be compared entry by entry with the oracle's bucketing, twice."""
    import ctypes as C
    from ktransformers_amd import _native as n
    E, k, H, I, T = shape
    c = make_case(7, E, k, H, I, T, invalid_ids=True)
    if dup:   # a token that names one expert twice: the bitmap scatter hands over to the ballot counting sort
        c["ids"][T // 2, 1] = c["ids"][T // 2, 0]
        c["ids"][T - 2, k - 1] = c["ids"][T - 2, 0]
    h = make_handle("AMXINT4", c, E, k, H, I, T, dev)
    num, pos, _ = oracle.bucket(E, c["ids"])
    off = np.concatenate([[0], np.cumsum(num)[:-1]]).astype(np.int64)
    ids = c["ids"].astype(np.int64)
    ok = (ids >= 0) & (ids < E)
    want = np.where(ok, off[np.clip(ids, 0, E - 1)] + pos, -1).astype(np.int32)
    for rep in range(2):
        run(h, c, dev)
        rp = C.c_void_p()
        n.check(n.lib.ktx_moe_debug_ptrs(h._h, None, None, C.byref(rp)))
        torch.cuda.synchronize()
        got = torch.empty(T * k, dtype=torch.int32, device=dev)
        assert _memcpy_d2d(got.data_ptr(), rp.value, T * k * 4) == 0
        assert np.array_equal(got.cpu().numpy().reshape(T, k), want), rep


def _memcpy_d2d(dst, src, nbytes):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return hip.hipMemcpy(dst, src, nbytes, 3)

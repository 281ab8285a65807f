"""CPU tests of the oracle (the checker itself): golden vectors produced by the reference's own kernels, and — when
oracle/_ref is built and the host has AVX512-VNNI/BF16 — a live bit-for-bit comparison against those kernels."""
import os

import numpy as np
import pytest

from helpers import bf16_to_f32, f32_to_bf16, make_case
from oracle.oracle import (FMT_AMXINT4, FMT_AMXINT8, FMT_BF16, FMT_FP8, FMT_FP8_PERCHANNEL, FMT_RAWINT4, Reference,
                           reference_available)

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "moe_amx_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.mark.parametrize("fname,fmt", [("int4", FMT_AMXINT4), ("int8", FMT_AMXINT8)])
@pytest.mark.parametrize("case", ["t1", "t7_invalid", "t33_prefill"])
def test_oracle_matches_reference_golden(oracle, golden, fname, fmt, case):
    g = golden
    moe = oracle.make_moe(fmt, g["gate"], g["up"], g["down"])
    x, ids, w = g[f"{fname}_{case}_x"], g[f"{fname}_{case}_ids"], g[f"{fname}_{case}_w"]
    y = oracle.moe_forward(moe, ids, w, x)
    assert np.array_equal(y, g[f"{fname}_{case}_y"]), "oracle differs from the reference kernels' golden output"
    y2 = oracle.moe_forward(moe, ids, w, x, y_prev=y)
    assert np.array_equal(y2, g[f"{fname}_{case}_yinc"]), "incremental merge differs from the reference"


@pytest.mark.parametrize("fname", ["fp8", "fp8pc", "bf16"])
@pytest.mark.parametrize("case", ["t1", "t7_invalid", "t33_prefill"])
def test_oracle_fp_formats_match_reference_golden(oracle, golden, fname, case):
    g = golden
    if fname == "fp8":
        moe = oracle.make_moe_fp8(g["fp8_gate"], g["fp8_up"], g["fp8_down"], g["fp8_gate_s"], g["fp8_up_s"], g["fp8_down_s"])
    elif fname == "fp8pc":      # one fp32 scale per output row (the reference's AMX_FP8_PERCHANNEL_MOE_TP)
        moe = oracle.make_moe_fp8_perchannel(g["fp8pc_gate"], g["fp8pc_up"], g["fp8pc_down"], g["fp8pc_gate_s"], g["fp8pc_up_s"],
                                             g["fp8pc_down_s"])
    else:
        moe = oracle.make_moe_bf16(g["gate"], g["up"], g["down"])
    x, ids, w = g[f"int4_{case}_x"], g[f"int4_{case}_ids"], g[f"int4_{case}_w"]   # same seeded inputs for all formats
    y = oracle.moe_forward(moe, ids, w, x)
    assert np.array_equal(y, g[f"{fname}_{case}_y"])
    assert np.array_equal(oracle.moe_forward(moe, ids, w, x, y_prev=y), g[f"{fname}_{case}_yinc"])


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref not built or host lacks AVX512-VNNI/BF16")
@pytest.mark.parametrize("shape", [(8, 2, 512, 256, 40), (8, 2, 1024, 512, 3)])
def test_oracle_fp_formats_match_live_reference(oracle, shape):
    from helpers import fp8_block_quant
    E, k, H, I, T = shape
    c = make_case(3, E, k, H, I, T)
    ref = Reference(threads=4)
    gq, gs = fp8_block_quant(bf16_to_f32(c["gate"])); uq, us = fp8_block_quant(bf16_to_f32(c["up"]))
    dq, ds = fp8_block_quant(bf16_to_f32(c["down"]))
    mr = ref.make_moe_quant(FMT_FP8, E, H, I, k, gq, uq, dq, gs, us, ds, max_len=64, group_size=128)
    assert np.array_equal(oracle.moe_forward(oracle.make_moe_fp8(gq, uq, dq, gs, us, ds), c["ids"], c["w"], c["x"]),
                          ref.moe_forward(mr, c["ids"], c["w"], c["x"]))
    mb = ref.make_moe(FMT_BF16, c["gate"], c["up"], c["down"], k=k, max_len=64)
    assert np.array_equal(oracle.moe_forward(oracle.make_moe_bf16(c["gate"], c["up"], c["down"]), c["ids"], c["w"], c["x"]),
                          ref.moe_forward(mb, c["ids"], c["w"], c["x"]))


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref not built or host lacks AVX512-VNNI/BF16")
@pytest.mark.parametrize("shape", [(8, 2, 512, 256, 40), (8, 2, 1024, 512, 3), (4, 2, 256, 512, 1)])
def test_oracle_fp8_perchannel_matches_live_reference(oracle, shape):
    """FP8_PERCHANNEL restatement (one fp32 chain over K, then * scale[n]) against the reference's own
    TP_MOE<AMX_FP8_PERCHANNEL_MOE_TP<GemmKernel224FP8PerChannel>> (fp8-perchannel-moe.hpp), vector and matrix paths."""
    from helpers import fp8_perchannel_quant
    E, k, H, I, T = shape
    c = make_case(5, E, k, H, I, T, invalid_ids=T >= 5)
    ref = Reference(threads=4)
    gq, gs = fp8_perchannel_quant(bf16_to_f32(c["gate"])); uq, us = fp8_perchannel_quant(bf16_to_f32(c["up"]))
    dq, ds = fp8_perchannel_quant(bf16_to_f32(c["down"]))
    mr = ref.make_moe_quant(FMT_FP8_PERCHANNEL, E, H, I, k, gq, uq, dq, gs, us, ds, max_len=64)
    mo = oracle.make_moe_fp8_perchannel(gq, uq, dq, gs, us, ds)
    yr = ref.moe_forward(mr, c["ids"], c["w"], c["x"])
    yo = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    assert np.array_equal(yo, yr)
    assert np.array_equal(oracle.moe_forward(mo, c["ids"], c["w"], c["x"], y_prev=yo),
                          ref.moe_forward(mr, c["ids"], c["w"], c["x"], y_prev=yr))


@pytest.mark.parametrize("case", ["t1", "t7_invalid", "t33_prefill"])
def test_oracle_rawint4_matches_reference_golden(oracle, golden, case):
    """RAWINT4 (Kimi-K2) restatement against outputs of the reference's own K2 class (k2-moe.hpp:124-191) — vec_mul path
    (t1, t7), mat_mul path (t33), invalid ids, incremental merge."""
    g = golden
    moe = oracle.make_moe_rawint4(g["k2_gate_p"], g["k2_up_p"], g["k2_down_p"], g["k2_gate_s"], g["k2_up_s"], g["k2_down_s"])
    x, ids, w = g[f"k2_{case}_x"], g[f"k2_{case}_ids"], g[f"k2_{case}_w"]
    y = oracle.moe_forward(moe, ids, w, x)
    assert np.array_equal(y, g[f"k2_{case}_y"]), "RAWINT4 oracle differs from the reference K2 kernels' golden output"
    assert np.array_equal(oracle.moe_forward(moe, ids, w, x, y_prev=y), g[f"k2_{case}_yinc"])


def test_rawint4_quantiser_helper_matches_reference_golden(golden):
    """tests/helpers.rawint4_quantize (vectorised) reproduces the packed bytes / bf16 scales stored by make_golden.py, whose
    first expert was checked there against the reference test's own scalar rawint4_quantize."""
    from helpers import rawint4_quantize
    base = make_case(20260922, int(golden["k2_E"]), int(golden["k2_k"]), int(golden["k2_H"]), int(golden["k2_I"]), 1)
    for nm in ("gate", "up", "down"):
        p, s = rawint4_quantize(bf16_to_f32(base[nm]))
        assert np.array_equal(p, golden[f"k2_{nm}_p"]) and np.array_equal(s, golden[f"k2_{nm}_s"])


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref not built or host lacks AVX512-VNNI/BF16")
@pytest.mark.parametrize("shape", [(8, 2, 512, 512, 1), (8, 2, 512, 512, 5), (8, 3, 1024, 512, 40), (8, 2, 512, 1024, 70)])
def test_oracle_rawint4_matches_live_reference(oracle, shape):
    from helpers import rawint4_quantize
    E, k, H, I, T = shape
    c = make_case(3, E, k, H, I, T, invalid_ids=T >= 5)
    q = [rawint4_quantize(bf16_to_f32(c[n])) for n in ("gate", "up", "down")]
    ref = Reference(threads=4)
    mr = ref.make_moe_quant(FMT_RAWINT4, E, H, I, k, q[0][0], q[1][0], q[2][0], q[0][1], q[1][1], q[2][1], max_len=128, group_size=32)
    mo = oracle.make_moe_rawint4(q[0][0], q[1][0], q[2][0], q[0][1], q[1][1], q[2][1])
    yr = ref.moe_forward(mr, c["ids"], c["w"], c["x"])
    yo = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    assert np.array_equal(yo, yr)
    assert np.array_equal(oracle.moe_forward(mo, c["ids"], c["w"], c["x"], y_prev=yo), ref.moe_forward(mr, c["ids"], c["w"], c["x"], y_prev=yr))
    ref.free_moe(mr)


def test_oracle_int4_quantiser_matches_reference_golden(oracle, golden):
    q, d = oracle.quant_weight(FMT_AMXINT4, golden["gate"][0])
    assert np.array_equal(d, golden["int4_gate0_scale"])
    assert np.array_equal(oracle.dequant_amxint4(q, d), golden["int4_gate0_dequant"])
    assert set(np.unique(q)).issubset(set(range(-112, 113, 16)))


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref not built or host lacks AVX512-VNNI/BF16")
@pytest.mark.parametrize("fmt", [FMT_AMXINT4, FMT_AMXINT8])
@pytest.mark.parametrize("shape", [(8, 2, 512, 256, 1), (8, 2, 256, 512, 5), (8, 3, 256, 384, 40), (16, 6, 2048, 1408, 2)])
def test_oracle_matches_live_reference(oracle, fmt, shape):
    E, k, H, I, T = shape
    c = make_case(3, E, k, H, I, T, invalid_ids=T >= 5)
    ref = Reference(threads=4)
    mr = ref.make_moe(fmt, c["gate"], c["up"], c["down"], k=k, max_len=64)
    mo = oracle.make_moe(fmt, c["gate"], c["up"], c["down"])
    yr = ref.moe_forward(mr, c["ids"], c["w"], c["x"])
    yo = oracle.moe_forward(mo, c["ids"], c["w"], c["x"])
    assert np.array_equal(yo, yr)
    ref.free_moe(mr)


def test_bf16_rounding_rule():
    # round-to-nearest-even, ties to even, denormals flushed (VCVTNE2PS2BF16 semantics)
    vals = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.0e-39, -1.0e-40, 65504.0], np.float32)
    b = f32_to_bf16(vals)
    assert b[0] == 0x3F80
    assert b[1] == 0x3F80            # 1 + 2^-8: tie -> even (down)
    assert b[2] == 0x3F82            # 1 + 3*2^-8: tie -> even (up)
    assert b[4] == 0x0000 and b[5] == 0x8000
    assert np.allclose(bf16_to_f32(b)[[0, 3]], [1.0, -2.5])


def test_act_fn_restatement(oracle):
    # silu(g)*u with the polynomial exp: close to the exact function, exact at 0, saturating for large |g|
    for g, u in [(0.0, 3.0), (1.0, 1.0), (-2.0, 0.5), (20.0, 2.0), (-100.0, 1.0), (100.0, 1.0)]:
        exact = g / (1.0 + np.exp(-g)) * u
        got = oracle.act_fn(g, u)
        assert abs(got - exact) <= 2e-6 * max(1.0, abs(exact)), (g, u, got, exact)


def test_bucket_restatement(oracle):
    rng = np.random.default_rng(0)
    E, T, k = 8, 13, 3
    ids = rng.integers(-1, E + 1, size=(T, k)).astype(np.int64)
    num, pos, emap = oracle.bucket(E, ids)
    cnt = np.zeros(E, np.int64)
    for t in range(T):
        for j in range(k):
            e = ids[t, j]
            if 0 <= e < E:
                assert pos[t, j] == cnt[e]
                cnt[e] += 1
            else:
                assert pos[t, j] == -1
    assert np.array_equal(num, cnt)
    assert list(emap) == [e for e in range(E) if cnt[e] > 0]


def test_empty_and_all_invalid(oracle):
    c = make_case(5, 4, 2, 128, 128, 3)
    moe = oracle.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"])
    ids = np.full((3, 2), -1, np.int64)
    y = oracle.moe_forward(moe, ids, c["w"], c["x"])
    assert not y.any()
    prev = f32_to_bf16(np.ones((3, 128), np.float32))
    assert np.array_equal(oracle.moe_forward(moe, ids, c["w"], c["x"], y_prev=prev), prev)

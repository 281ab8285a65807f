"""CPU: pins the GGUF restatement (oracle/ktx_oracle_gguf.c) against the reference's OWN k-/i-quant GEMM kernels —
third_party/llamafile/iqk_mul_mat.inc compiled unmodified into oracle/_ref/libiqk_ref_{avx2,zen4}.so (oracle/Makefile `iqk`);
these are the kernels LLAMA_MOE_TP reaches through llamafile_sgemm (tinyblas_cpu_sgemm.inc:331-335 ->
iqk_mul_mat(m, n, k*blck, Atype, A, lda, Btype=Q8_K, B, ldb, C, ldc, ith, nth)).

The integer parts are identical by construction; the fp32 combination differs in ORDER only (iqk keeps 8 AVX lanes of
partial sums per output and adds them at the end, the restatement — like the HIP kernel — folds each 256-block with one
fma).  So the bar is: both agree with the exact (float64) value of the same quantised product to within the fp32
reordering bound `4 ulp_f32 x sum_b |block term|`, and with each other to twice that."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.gguf_ref import DEQUANT, GGML_TYPE_IQ1_S, GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K, QUANT, GgufOracle

REF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
GGML_TYPE_Q8_K = 15
GGML_TYPE_Q2_K, GGML_TYPE_Q3_K, GGML_TYPE_IQ4_XS = 10, 11, 23
BLOCK_BYTES = {GGML_TYPE_Q4_K: 144, GGML_TYPE_Q5_K: 176, GGML_TYPE_Q6_K: 210, GGML_TYPE_IQ1_S: 50, GGML_TYPE_Q2_K: 84,
               GGML_TYPE_Q3_K: 110, GGML_TYPE_IQ4_XS: 136}
# types restated for the checker only (no HIP kernel yet): random bit patterns are valid blocks; fp16 fields kept small
RANDOM_BITS = {GGML_TYPE_IQ1_S: (0,), GGML_TYPE_Q2_K: (80, 82), GGML_TYPE_Q3_K: (108,), GGML_TYPE_IQ4_XS: (0,)}


def dequant(t, wb):
    """float32 values of the blocks: the oracle's codec where it has one, else the loader's (pinned bit for bit to the
    reference's numpy dequantisers by tests/test_gguf_loader_cpu.py)."""
    if t in DEQUANT:
        return DEQUANT[t](wb)
    from ktransformers_amd.util.gguf_loader import _dequant
    return _dequant(t, np.ascontiguousarray(wb).reshape(-1)).reshape(wb.shape[0], -1)
VARIANTS = {"avx2": ("libiqk_ref_avx2.so", "iqk_mul_mat", "avx2"), "zen4": ("libiqk_ref_zen4.so", "iqk_mul_mat_zen4", "avx512_vnni")}


def cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            return any(flag in line.split() for line in f if line.startswith("flags"))
    except OSError:
        return False


def load(variant):
    so, sym, flag = VARIANTS[variant]
    path = os.path.join(REF_DIR, so)
    if not os.path.exists(path):
        pytest.skip(f"{so} not built (needs /root/reference at build time: make -C oracle iqk)")
    if not cpu_has(flag):
        pytest.skip(f"host CPU lacks {flag}")
    fn = getattr(C.CDLL(path), sym)
    fn.restype = C.c_bool
    fn.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p,
                   C.c_long, C.c_int, C.c_int]
    return fn


def random_blocks(t, N, K, rng):
    """Valid blocks of type t: Q4_K / Q6_K through the test quantisers; IQ1_S as random bit patterns with a small d."""
    if t not in RANDOM_BITS:
        return QUANT[t]((rng.standard_normal((N, K)) / 10).astype(np.float32))
    b = rng.integers(0, 256, (N, K // 256, BLOCK_BYTES[t]), dtype=np.uint8)
    for c in RANDOM_BITS[t]:
        d = rng.random((N, K // 256)).astype(np.float16) * np.float16(0.004) + np.float16(0.001)
        b[..., c:c + 2] = d.view(np.uint8).reshape(N, K // 256, 2)
    return b.reshape(N, -1)


def q8k_rows(o, x):
    """x fp32 [T][K] -> (block_q8_K bytes [T][K/256*292], q int8 [T][K], d fp32 [T][K/256], bsums int16 [T][K/16])."""
    T, K = x.shape
    nb = K // 256
    rec = np.zeros((T, nb), dtype=np.dtype([("d", "<f4"), ("qs", "i1", 256), ("bsums", "<i2", 16)]))
    assert rec.dtype.itemsize == 292
    qs, ds, bss = [], [], []
    for t in range(T):
        q, d, bs = o.quantize_row_q8_K(x[t])
        rec["d"][t], rec["qs"][t], rec["bsums"][t] = d, q.reshape(nb, 256), bs.reshape(nb, 16)
        qs.append(q); ds.append(d); bss.append(bs)
    return rec, np.stack(qs), np.stack(ds), np.stack(bss)


@pytest.mark.parametrize("variant", ["avx2", "zen4"])
@pytest.mark.parametrize("t", [GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K, GGML_TYPE_IQ1_S, GGML_TYPE_Q2_K, GGML_TYPE_Q3_K,
                               GGML_TYPE_IQ4_XS])
@pytest.mark.parametrize("shape", [(48, 512, 1), (40, 2048, 5), (24, 1536, 19)])
def test_restated_vec_dot_against_reference_iqk_kernels(variant, t, shape):
    iqk = load(variant)
    N, K, T = shape
    o = GgufOracle()
    rng = np.random.default_rng(N * 1000 + K + T + t)
    wb = np.ascontiguousarray(random_blocks(t, N, K, rng))
    x = rng.standard_normal((T, K)).astype(np.float32)
    if t == GGML_TYPE_IQ4_XS:
        # the reference's IQ4_XS kernel forms (value + 128) * q8 pair sums in 16 bits (maddubs), which SATURATE when a large
        # code sits next to the block's +-127 element — on uniformly random blocks and normal activations a few outputs per
        # hundred differ from the exact product by 10-40 % (measured here).  That is data-dependent behaviour of the CPU kernel,
        # not the format: the pin uses activations whose pair sums stay below 2^15 (one outlier per block, its neighbour 0).
        x = np.clip(x * 0.1, -0.25, 0.25)
        x[:, 0::256] = np.where(rng.random((T, K // 256)) < 0.5, -1.0, 1.0).astype(np.float32)
        x[:, 1::256] = 0
    x[0, :256] = 0  # an all-zero activation block (d = 0)
    rec, q8, d8, bs = q8k_rows(o, x)

    out = np.full((T, N), np.nan, np.float32)
    ok = iqk(N, T, K, t, wb.ctypes.data, K // 256, GGML_TYPE_Q8_K, rec.ctypes.data, K // 256, out.ctypes.data, N, 0, 1)
    if not ok:
        pytest.skip(f"the reference's {variant} build has no kernel for ggml type {t}")

    vec_dot = {GGML_TYPE_Q4_K: o.lib.ktxo_vec_dot_q4_K, GGML_TYPE_Q5_K: o.lib.ktxo_vec_dot_q5_K, GGML_TYPE_Q6_K: o.lib.ktxo_vec_dot_q6_K,
               GGML_TYPE_IQ1_S: o.lib.ktxo_vec_dot_iq1_s, GGML_TYPE_Q2_K: o.lib.ktxo_vec_dot_q2_K, GGML_TYPE_Q3_K: o.lib.ktxo_vec_dot_q3_K,
               GGML_TYPE_IQ4_XS: o.lib.ktxo_vec_dot_iq4_xs}[t]
    vec_dot.restype = C.c_float
    mine = np.empty((T, N), np.float32)
    rb = BLOCK_BYTES[t] * (K // 256)
    for ti in range(T):
        for n in range(N):
            mine[ti, n] = vec_dot(C.c_void_p(wb.ctypes.data + n * rb), C.c_int(K), C.c_void_p(q8[ti].ctypes.data),
                                  C.c_void_p(d8[ti].ctypes.data), C.c_void_p(bs[ti].ctypes.data))

    # exact value of the quantised product and the size of its per-block terms, in float64
    wd = dequant(t, wb).astype(np.float64).reshape(N, K // 256, 256)
    xd = (q8.astype(np.float64).reshape(T, K // 256, 256) * d8.astype(np.float64)[:, :, None])
    terms = np.einsum("nbk,tbk->tnb", wd, xd)
    exact = terms.sum(-1)
    # fp32 re-association bound: 16 ulp of the per-block terms, plus 1 ulp of the largest magnitudes an implementation may
    # round separately before they cancel (Q4_K: d*sum(s*dot) and dmin*sum(m*bsum); Q6_K: the -32 offset; IQ1_S: the delta term)
    bound = 2.0 ** -23 * (16 * np.abs(terms).sum(-1) + np.einsum("nb,tb->tn", np.abs(wd).max(-1), np.abs(xd).sum(-1))) + 1e-30
    assert np.isfinite(out).all()
    assert (np.abs(out.astype(np.float64) - exact) <= bound).all(), f"reference kernel off by {np.abs(out - exact).max()}"
    assert (np.abs(mine.astype(np.float64) - exact) <= bound).all(), f"restatement off by {np.abs(mine - exact).max()}"
    assert (np.abs(mine.astype(np.float64) - out) <= 2 * bound).all()
    # measured agreement: 2-3e-7 of the output scale, median 2-3 ulp
    assert np.abs(mine - out).max() <= 1e-6 * np.abs(out).max()
    assert np.mean(np.abs(mine - out) <= 4 * np.spacing(np.abs(out).astype(np.float32))) > 0.5


# ---- legacy types (round 5): Q4_0 / Q5_0 weights x Q8_0 activations through the reference's mul_mat_qX_0_q8_0_T --------------------
GGML_TYPE_Q4_0, GGML_TYPE_Q5_0, GGML_TYPE_Q8_0 = 2, 6, 8
LEGACY_BYTES = {GGML_TYPE_Q4_0: 18, GGML_TYPE_Q5_0: 22, GGML_TYPE_Q8_0: 34}


def legacy_blocks(t, N, K, rng):
    """Random valid blocks (every bit pattern is a block; the fp16 scale kept small and positive)."""
    b = rng.integers(0, 256, (N, K // 32, LEGACY_BYTES[t]), dtype=np.uint8)
    d = rng.random((N, K // 32)).astype(np.float16) * np.float16(0.004) + np.float16(0.001)
    b[..., 0:2] = d.view(np.uint8).reshape(N, K // 32, 2)
    return b.reshape(N, -1)


def legacy_values(t, wb, N, K):
    """float64 values of legacy blocks (ggml dequantize_row_q4_0 / q5_0 / q8_0)."""
    b = wb.reshape(N, K // 32, LEGACY_BYTES[t])
    d = b[..., 0:2].copy().view(np.float16).astype(np.float64)[..., 0]
    if t == GGML_TYPE_Q8_0:
        q = b[..., 2:].copy().view(np.int8).astype(np.float64)
    else:
        qs = b[..., (2 if t == GGML_TYPE_Q4_0 else 6):].astype(np.int32)
        lo, hi = qs & 15, qs >> 4
        if t == GGML_TYPE_Q5_0:
            qh = b[..., 2:6].copy().view(np.uint32)[..., 0].astype(np.int64)
            j = np.arange(16)
            lo = lo | (((qh[..., None] >> j) & 1) << 4)
            hi = hi | (((qh[..., None] >> (j + 16)) & 1) << 4)
        q = (np.concatenate([lo, hi], axis=-1) - (8 if t == GGML_TYPE_Q4_0 else 16)).astype(np.float64)
    return (q * d[..., None]).reshape(N, K)


def q80_rows(o, x):
    T, K = x.shape
    rec = np.zeros((T, K // 32), dtype=np.dtype([("d", "<f2"), ("qs", "i1", 32)]))
    assert rec.dtype.itemsize == 34
    qs, ds = [], []
    for t in range(T):
        q = np.empty(K, np.int8); d = np.empty(K // 32, np.float32)
        o.lib.ktxo_quantize_row_q8_0(x[t].ctypes.data_as(C.c_void_p), C.c_int(K), q.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
        rec["d"][t], rec["qs"][t] = d.astype(np.float16), q.reshape(K // 32, 32)
        assert np.array_equal(rec["d"][t].astype(np.float32), d)        # d IS an fp16 value
        qs.append(q); ds.append(d)
    return rec, np.stack(qs), np.stack(ds)


def test_quantize_row_q8_0_is_ggml_x86_arithmetic():
    """d = fp16(amax / 127), q = rne(x * (127 / amax)) per 32 — against the same expressions in numpy, over scales that reach fp16
    subnormals and zero blocks."""
    o = GgufOracle()
    rng = np.random.default_rng(3)
    for scale in (1e-7, 1e-5, 3e-4, 1.0, 300.0):
        x = (rng.standard_normal((3, 256)) * scale).astype(np.float32)
        x[1, 32:64] = 0
        rec, q, d = q80_rows(o, x)
        am = np.abs(x.reshape(3, 8, 32)).max(-1)
        with np.errstate(divide="ignore"):
            idv = np.where(am != 0, np.float32(127) / am, np.float32(0)).astype(np.float32)
        assert np.array_equal(d, (am / np.float32(127)).astype(np.float32).astype(np.float16).astype(np.float32))
        assert np.array_equal(q.reshape(3, 8, 32), np.rint(x.reshape(3, 8, 32) * idv[..., None]).astype(np.int8))


@pytest.mark.parametrize("variant", ["avx2", "zen4"])
@pytest.mark.parametrize("t", [GGML_TYPE_Q4_0, GGML_TYPE_Q5_0])
@pytest.mark.parametrize("shape", [(48, 512, 1), (40, 2048, 5), (24, 1536, 19)])
def test_restated_legacy_vec_dot_against_reference_iqk_kernels(variant, t, shape):
    """ktxo_vec_dot_q4_0 / q5_0 (one fma per 32-block in k order — the HIP kernel's order) against iqk's mul_mat_qX_0_q8_0_T
    (iqk_mul_mat.inc:2201-2213: 8 AVX lanes of (block mod 4, half-block) partial sums, added at the end): identical integer dots, fp32
    association differs; both within the re-association bound of the exact value.  (Q8_0 x Q8_0 is declined by iqk — set_mul_mat
    answers false, checked below — and goes to tinyBLAS in the reference.)"""
    iqk = load(variant)
    N, K, T = shape
    o = GgufOracle()
    rng = np.random.default_rng(N * 1000 + K + T + t)
    wb = np.ascontiguousarray(legacy_blocks(t, N, K, rng))
    x = rng.standard_normal((T, K)).astype(np.float32)
    x[0, :32] = 0
    rec, q8, d8 = q80_rows(o, x)
    out = np.full((T, N), np.nan, np.float32)
    ok = iqk(N, T, K, t, wb.ctypes.data, K // 32, GGML_TYPE_Q8_0, rec.ctypes.data, K // 32, out.ctypes.data, N, 0, 1)
    assert ok, "the reference's iqk build has no Q4_0 / Q5_0 x Q8_0 kernel"
    vec_dot = {GGML_TYPE_Q4_0: o.lib.ktxo_vec_dot_q4_0, GGML_TYPE_Q5_0: o.lib.ktxo_vec_dot_q5_0}[t]
    vec_dot.restype = C.c_float
    mine = np.empty((T, N), np.float32)
    rb = LEGACY_BYTES[t] * (K // 32)
    for ti in range(T):
        for n in range(N):
            mine[ti, n] = vec_dot(C.c_void_p(wb.ctypes.data + n * rb), C.c_int(K), C.c_void_p(q8[ti].ctypes.data), C.c_void_p(d8[ti].ctypes.data))
    wd = legacy_values(t, wb, N, K).reshape(N, K // 32, 32)
    xd = q8.astype(np.float64).reshape(T, K // 32, 32) * d8.astype(np.float64)[:, :, None]
    terms = np.einsum("nbk,tbk->tnb", wd, xd)
    exact = terms.sum(-1)
    bound = 2.0 ** -23 * 16 * np.abs(terms).sum(-1) + 1e-30
    assert np.isfinite(out).all()
    assert (np.abs(out.astype(np.float64) - exact) <= bound).all(), f"reference kernel off by {np.abs(out - exact).max()}"
    assert (np.abs(mine.astype(np.float64) - exact) <= bound).all(), f"restatement off by {np.abs(mine - exact).max()}"
    assert (np.abs(mine.astype(np.float64) - out) <= 2 * bound).all()
    assert np.abs(mine - out).max() <= 1e-6 * np.abs(out).max()
    # Q8_0 weights with Q8_0 activations: iqk declines (it expects its own interleaved Q8 formats), the reference falls through to tinyBLAS
    w8 = np.ascontiguousarray(legacy_blocks(GGML_TYPE_Q8_0, N, K, rng))
    assert not iqk(N, T, K, GGML_TYPE_Q8_0, w8.ctypes.data, K // 32, GGML_TYPE_Q8_0, rec.ctypes.data, K // 32, out.ctypes.data, N, 0, 1)


def test_restated_q8_0_vec_dot_is_the_exact_block_sum():
    """Q8_0 x Q8_0 (tinyBLAS_Q0_AVX2 in the reference, tinyblas_cpu.h:828-1010: per block d_w * d_x times the exact 32-element int dot,
    fp32 madd into 8 lanes): the restatement against float64."""
    o = GgufOracle()
    rng = np.random.default_rng(17)
    N, K, T = 24, 1024, 3
    wb = np.ascontiguousarray(legacy_blocks(GGML_TYPE_Q8_0, N, K, rng))
    x = rng.standard_normal((T, K)).astype(np.float32)
    rec, q8, d8 = q80_rows(o, x)
    o.lib.ktxo_vec_dot_q8_0.restype = C.c_float
    wd = legacy_values(GGML_TYPE_Q8_0, wb, N, K).reshape(N, K // 32, 32)
    xd = q8.astype(np.float64).reshape(T, K // 32, 32) * d8.astype(np.float64)[:, :, None]
    terms = np.einsum("nbk,tbk->tnb", wd, xd)
    for ti in range(T):
        for n in range(N):
            v = o.lib.ktxo_vec_dot_q8_0(C.c_void_p(wb.ctypes.data + n * 34 * (K // 32)), C.c_int(K), C.c_void_p(q8[ti].ctypes.data),
                                        C.c_void_p(d8[ti].ctypes.data))
            assert abs(v - terms[ti, n].sum()) <= 2.0 ** -23 * 16 * np.abs(terms[ti, n]).sum() + 1e-30

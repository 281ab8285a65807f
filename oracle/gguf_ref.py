"""TEST INFRASTRUCTURE ONLY — GGUF k-quant block codec (numpy) + ctypes front-end of oracle/ktx_oracle_gguf.c.

dequantize_q4_k / dequantize_q6_k restate archive/ktransformers/util/custom_gguf.py:326-343,... (which itself restates
ggml-quants.c); tests/golden/make_gguf_golden.py pins them against the reference's own numpy functions on random blocks.
quantize_q4_k / quantize_q6_k are simple encoders written for the tests (any valid encoding is a legal weight file; they are
NOT ggml's search-based quantisers).  Block layouts:
  Q4_K (type 12, 144 B / 256 w): fp16 d, fp16 dmin, 12 B packed 6-bit (scale, min) x 8, 128 B nibbles (4 x [32 low | 32 high])
  Q6_K (type 14, 210 B / 256 w): 128 B low nibbles, 64 B high 2-bit, 16 x int8 scales, fp16 d
  Q5_K (type 13, 176 B / 256 w): Q4_K's header, then 32 B of fifth bits (bit j of byte l = sub-block j, element l), 128 B nibbles
       (oracle side only so far: no HIP kernel reads Q5_K natively yet)
"""
import ctypes as C

import numpy as np

GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K, GGML_TYPE_IQ1_S = 12, 13, 14, 19
BLOCK_BYTES = {GGML_TYPE_Q4_K: 144, GGML_TYPE_Q5_K: 176, GGML_TYPE_Q6_K: 210, GGML_TYPE_IQ1_S: 50}


def dequantize_q4_k(data: np.ndarray) -> np.ndarray:
    """uint8 [..., nblk*144] -> float32 [..., nblk*256]."""
    lead = data.shape[:-1]
    b = np.ascontiguousarray(data).reshape(-1, 144)
    nb = b.shape[0]
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32).reshape(nb, 1, 1)
    dmin = b[:, 2:4].copy().view(np.float16).astype(np.float32).reshape(nb, 1, 1)
    qs1 = b[:, 4:16].reshape(nb, 12, 1)
    qs2 = b[:, 16:].reshape(nb, 4, 32)
    factors = d * np.concatenate([qs1[:, 0:4] & 0b111111, (qs1[:, 8:] & 15) | ((qs1[:, 0:4] >> 6) << 4)], axis=1)
    offsets = dmin * np.concatenate([qs1[:, 4:8] & 0b111111, (qs1[:, 8:] >> 4) | ((qs1[:, 4:8] >> 6) << 4)], axis=1)
    q = np.stack([qs2 & 0xF, qs2 >> 4], axis=2).reshape(nb, 8, 32)
    return (factors * q - offsets).astype(np.float32).reshape(*lead, -1)


def dequantize_q6_k(data: np.ndarray) -> np.ndarray:
    lead = data.shape[:-1]
    b = np.ascontiguousarray(data).reshape(-1, 210)
    nb = b.shape[0]
    ql = b[:, :128].reshape(nb, 2, 64).astype(np.int16)
    qh = b[:, 128:192].reshape(nb, 2, 32).astype(np.int16)
    sc = b[:, 192:208].copy().view(np.int8).astype(np.float32).reshape(nb, 2, 8)
    d = b[:, 208:210].copy().view(np.float16).astype(np.float32).reshape(nb, 1, 1)
    q1 = ((ql[:, :, :32] & 0xF) | (((qh >> 0) & 3) << 4)) - 32
    q2 = ((ql[:, :, 32:] & 0xF) | (((qh >> 2) & 3) << 4)) - 32
    q3 = ((ql[:, :, :32] >> 4) | (((qh >> 4) & 3) << 4)) - 32
    q4 = ((ql[:, :, 32:] >> 4) | (((qh >> 6) & 3) << 4)) - 32
    q = np.stack([q1, q2, q3, q4], axis=2).reshape(nb, 2, 128).astype(np.float32)     # element = half*128 + quarter*32 + l
    scale = np.repeat(sc.reshape(nb, 2, 8), 16, axis=2)                                # sub-block of 16 -> scale
    return (d * scale * q).astype(np.float32).reshape(*lead, -1)


def quantize_q4_k(w: np.ndarray) -> np.ndarray:
    """float32 [..., K] (K % 256 == 0) -> uint8 [..., K/256*144]."""
    lead = w.shape[:-1]
    x = np.ascontiguousarray(w, dtype=np.float32).reshape(-1, 8, 32)
    nb = x.shape[0]
    mn = np.minimum(x.min(axis=2), 0.0)
    mx = np.maximum(x.max(axis=2), mn + 1e-30)
    sc_f, m_f = (mx - mn) / 15.0, -mn
    d = (sc_f.max(axis=1) / 63.0).astype(np.float16)
    dmin = (m_f.max(axis=1) / 63.0).astype(np.float16)
    df, dmf = d.astype(np.float32)[:, None], dmin.astype(np.float32)[:, None]
    sc = np.clip(np.rint(np.divide(sc_f, df, out=np.zeros_like(sc_f), where=df > 0)), 0, 63).astype(np.uint8)
    m = np.clip(np.rint(np.divide(m_f, dmf, out=np.zeros_like(m_f), where=dmf > 0)), 0, 63).astype(np.uint8)
    eff = (df * sc)[:, :, None]
    q = np.clip(np.rint(np.divide(x + (dmf * m)[:, :, None], eff, out=np.zeros_like(x), where=eff > 0)), 0, 15).astype(np.uint8)
    out = np.zeros((nb, 144), np.uint8)
    out[:, 0:2] = d.view(np.uint8).reshape(nb, 2)
    out[:, 2:4] = dmin.view(np.uint8).reshape(nb, 2)
    out[:, 4:8] = (sc[:, 0:4] & 63) | ((sc[:, 4:8] >> 4) << 6)
    out[:, 8:12] = (m[:, 0:4] & 63) | ((m[:, 4:8] >> 4) << 6)
    out[:, 12:16] = (sc[:, 4:8] & 15) | ((m[:, 4:8] & 15) << 4)
    qq = q.reshape(nb, 4, 2, 32)
    out[:, 16:] = (qq[:, :, 0] | (qq[:, :, 1] << 4)).reshape(nb, 128)
    return out.reshape(*lead, -1)


def dequantize_q5_k(data: np.ndarray) -> np.ndarray:
    """uint8 [..., nblk*176] -> float32 [..., nblk*256] (custom_gguf.py:356-410: d*scale*q - dmin*min, q in 0..31)."""
    lead = data.shape[:-1]
    b = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, 176)
    nb = b.shape[0]
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32).reshape(nb, 1, 1)
    dmin = b[:, 2:4].copy().view(np.float16).astype(np.float32).reshape(nb, 1, 1)
    s1, qs = b[:, 4:16].reshape(nb, 12, 1), b[:, 48:].reshape(nb, 4, 32)
    fac = d * np.concatenate([s1[:, 0:4] & 63, (s1[:, 8:] & 15) | ((s1[:, 0:4] >> 6) << 4)], axis=1).astype(np.float32)
    off = dmin * np.concatenate([s1[:, 4:8] & 63, (s1[:, 8:] >> 4) | ((s1[:, 4:8] >> 6) << 4)], axis=1).astype(np.float32)
    hbit = np.unpackbits(b[:, 16:48].reshape(nb, 32, 1), axis=2, bitorder="little").transpose(0, 2, 1)     # [nb, sub-block, l]
    q = (np.stack([qs & 0xF, qs >> 4], axis=2).reshape(nb, 8, 32) + (hbit << 4)).astype(np.float32)
    return (fac * q - off).astype(np.float32).reshape(*lead, -1)


def quantize_q5_k(w: np.ndarray) -> np.ndarray:
    """float32 [..., K] (K % 256 == 0) -> uint8 [..., K/256*176]; the Q4_K test encoder with 32 levels."""
    lead = w.shape[:-1]
    x = np.ascontiguousarray(w, dtype=np.float32).reshape(-1, 8, 32)
    nb = x.shape[0]
    mn = np.minimum(x.min(axis=2), 0.0)
    mx = np.maximum(x.max(axis=2), mn + 1e-30)
    sc_f, m_f = (mx - mn) / 31.0, -mn
    d = (sc_f.max(axis=1) / 63.0).astype(np.float16)
    dmin = (m_f.max(axis=1) / 63.0).astype(np.float16)
    df, dmf = d.astype(np.float32)[:, None], dmin.astype(np.float32)[:, None]
    sc = np.clip(np.rint(np.divide(sc_f, df, out=np.zeros_like(sc_f), where=df > 0)), 0, 63).astype(np.uint8)
    m = np.clip(np.rint(np.divide(m_f, dmf, out=np.zeros_like(m_f), where=dmf > 0)), 0, 63).astype(np.uint8)
    eff = (df * sc)[:, :, None]
    q = np.clip(np.rint(np.divide(x + (dmf * m)[:, :, None], eff, out=np.zeros_like(x), where=eff > 0)), 0, 31).astype(np.uint8)
    out = np.zeros((nb, 176), np.uint8)
    out[:, 0:2] = d.view(np.uint8).reshape(nb, 2)
    out[:, 2:4] = dmin.view(np.uint8).reshape(nb, 2)
    out[:, 4:8] = (sc[:, 0:4] & 63) | ((sc[:, 4:8] >> 4) << 6)
    out[:, 8:12] = (m[:, 0:4] & 63) | ((m[:, 4:8] >> 4) << 6)
    out[:, 12:16] = (sc[:, 4:8] & 15) | ((m[:, 4:8] & 15) << 4)
    out[:, 16:48] = np.packbits((q >> 4).transpose(0, 2, 1), axis=2, bitorder="little").reshape(nb, 32)   # bit j of byte l
    qq = (q & 0xF).reshape(nb, 4, 2, 32)
    out[:, 48:] = (qq[:, :, 0] | (qq[:, :, 1] << 4)).reshape(nb, 128)
    return out.reshape(*lead, -1)


def quantize_q6_k(w: np.ndarray) -> np.ndarray:
    lead = w.shape[:-1]
    x = np.ascontiguousarray(w, dtype=np.float32).reshape(-1, 16, 16)
    nb = x.shape[0]
    amax = np.abs(x).max(axis=2)
    sc_f = amax / 31.0
    d = (sc_f.max(axis=1) / 127.0).astype(np.float16)
    df = d.astype(np.float32)[:, None]
    sc = np.clip(np.rint(np.divide(sc_f, df, out=np.zeros_like(sc_f), where=df > 0)), 1, 127).astype(np.int8)
    eff = (df * sc)[:, :, None]
    q = (np.clip(np.rint(np.divide(x, eff, out=np.zeros_like(x), where=eff > 0)), -32, 31) + 32).astype(np.uint8)
    e = q.reshape(nb, 2, 4, 32)                               # [half][quarter][l]
    ql = np.zeros((nb, 2, 64), np.uint8)
    ql[:, :, :32] = (e[:, :, 0] & 0xF) | ((e[:, :, 2] & 0xF) << 4)
    ql[:, :, 32:] = (e[:, :, 1] & 0xF) | ((e[:, :, 3] & 0xF) << 4)
    qh = (e[:, :, 0] >> 4) | ((e[:, :, 1] >> 4) << 2) | ((e[:, :, 2] >> 4) << 4) | ((e[:, :, 3] >> 4) << 6)
    out = np.zeros((nb, 210), np.uint8)
    out[:, :128] = ql.reshape(nb, 128)
    out[:, 128:192] = qh.reshape(nb, 64)
    out[:, 192:208] = sc.view(np.uint8)
    out[:, 208:210] = d.view(np.uint8).reshape(nb, 2)
    return out.reshape(*lead, -1)


_GRID = None


def iq1s_grid() -> np.ndarray:
    """[2048, 8] int8 grid points in {-1, 0, +1} (the codebook the C oracle compiles in; format constant)."""
    global _GRID
    if _GRID is None:
        from oracle.oracle import Oracle
        lib = Oracle().lib
        lib.ktxo_iq1s_grid.restype = C.POINTER(C.c_uint16)
        packed = np.ctypeslib.as_array(lib.ktxo_iq1s_grid(), shape=(2048,)).astype(np.int32)
        _GRID = np.stack([((packed >> (2 * e)) & 3) - 1 for e in range(8)], axis=1).astype(np.int8)
    return _GRID


def dequantize_iq1_s(data: np.ndarray) -> np.ndarray:
    """uint8 [..., nblk*50] -> float32 [..., nblk*256]:  w = d * (2s+1) * (g +- 0.125)."""
    lead = data.shape[:-1]
    b = np.ascontiguousarray(data).reshape(-1, 50)
    nb = b.shape[0]
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32).reshape(nb, 1, 1, 1)
    qs = b[:, 2:34].reshape(nb, 8, 4).astype(np.int32)
    qh = b[:, 34:50].copy().view(np.uint16).astype(np.int32).reshape(nb, 8)
    idx = qs | (((qh[:, :, None] >> (3 * np.arange(4))[None, None, :]) & 7) << 8)
    g = iq1s_grid()[idx].astype(np.float32)                                 # [nb, 8, 4, 8]
    scale = (2 * ((qh >> 12) & 7) + 1).astype(np.float32)[:, :, None, None]
    delta = np.where(qh & 0x8000, -0.125, 0.125).astype(np.float32)[:, :, None, None]
    return (d * scale * (g + delta)).astype(np.float32).reshape(*lead, -1)


def quantize_iq1_s(w: np.ndarray) -> np.ndarray:
    """Simple test encoder: per 32-weight sub-block pick scale s in 0..7 and delta sign, per 8 weights the nearest grid
    point (brute force over the 2048 entries)."""
    lead = w.shape[:-1]
    x = np.ascontiguousarray(w, dtype=np.float32).reshape(-1, 8, 4, 8)        # [nb, ib, l, e]
    nb = x.shape[0]
    grid = iq1s_grid().astype(np.float32)                                       # [2048, 8]
    amax = np.abs(x).reshape(nb, -1).max(axis=1)
    d = (amax / 15.0 / 1.125).astype(np.float16)
    df = np.maximum(d.astype(np.float32), 1e-30)
    sub = np.abs(x).reshape(nb, 8, -1).max(axis=2) / 1.125                      # wanted d*(2s+1)
    s = np.clip(np.rint((sub / df[:, None] - 1) / 2), 0, 7).astype(np.int32)
    dl = df[:, None] * (2 * s + 1)
    out = np.zeros((nb, 50), np.uint8)
    out[:, 0:2] = d.view(np.uint8).reshape(nb, 2)
    qh = (s << 12).astype(np.int32)
    choice = {}
    g2 = (grid ** 2).sum(-1)                                                    # [2048]
    for sign, delta in ((0, 0.125), (1, -0.125)):
        t = (x / dl[:, :, None, None] - delta).reshape(-1, 8)                   # target grid values, one row per 8 weights
        idx = np.empty(t.shape[0], np.int64)
        err = np.empty(t.shape[0], np.float32)
        for s0 in range(0, t.shape[0], 1 << 16):                                # nearest grid point: |t-g|^2 = |t|^2 - 2 t.g + |g|^2
            tt = t[s0:s0 + (1 << 16)]
            dist = g2[None, :] - 2.0 * (tt @ grid.T)
            ii = dist.argmin(-1)
            idx[s0:s0 + len(tt)] = ii
            err[s0:s0 + len(tt)] = dist[np.arange(len(tt)), ii] + (tt ** 2).sum(-1)
        choice[sign] = (idx.reshape(nb, 8, 4), err.reshape(nb, 8, 4).sum(-1))
    use_neg = choice[1][1] < choice[0][1]
    idx = np.where(use_neg[:, :, None], choice[1][0], choice[0][0])              # [nb, 8, 4]
    qh |= np.where(use_neg, 0x8000, 0)
    for l in range(4):
        qh |= ((idx[:, :, l] >> 8) & 7) << (3 * l)
    out[:, 2:34] = (idx & 0xFF).astype(np.uint8).reshape(nb, 32)
    out[:, 34:50] = qh.astype(np.uint16).view(np.uint8).reshape(nb, 16)
    return out.reshape(*lead, -1)


QUANT = {GGML_TYPE_Q4_K: quantize_q4_k, GGML_TYPE_Q5_K: quantize_q5_k, GGML_TYPE_Q6_K: quantize_q6_k, GGML_TYPE_IQ1_S: quantize_iq1_s}
DEQUANT = {GGML_TYPE_Q4_K: dequantize_q4_k, GGML_TYPE_Q5_K: dequantize_q5_k, GGML_TYPE_Q6_K: dequantize_q6_k,
           GGML_TYPE_IQ1_S: dequantize_iq1_s}


class _GgufMoe(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("E", "H", "I", "gate_type", "up_type", "down_type")] + [
        ("gate", C.c_void_p), ("up", C.c_void_p), ("down", C.c_void_p), ("mask", C.c_void_p)]


class GgufOracle:
    def __init__(self):
        from oracle.oracle import Oracle
        self.lib = Oracle().lib
        self.lib.ktxo_moe_forward_gguf.restype = C.c_int

    def quantize_row_q8_K(self, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float32)
        K = x.shape[-1]
        q = np.empty(K, np.int8); d = np.empty(K // 256, np.float32); bs = np.empty(K // 16, np.int16)
        self.lib.ktxo_quantize_row_q8_K(x.ctypes.data_as(C.c_void_p), C.c_int(K), q.ctypes.data_as(C.c_void_p),
                                        d.ctypes.data_as(C.c_void_p), bs.ctypes.data_as(C.c_void_p))
        return q, d, bs

    def moe_forward(self, gate, up, down, types, E, H, I, ids, w, x_bf16, mask=None, want_inter=False):
        """gate/up/down: uint8 raw GGUF blocks [E, N, K/256*blockbytes]; x_bf16 uint16 [T, H] -> uint16 [T, H]."""
        gate, up, down = (np.ascontiguousarray(a) for a in (gate, up, down))
        ids = np.ascontiguousarray(ids, dtype=np.int64); w = np.ascontiguousarray(w, dtype=np.float32)
        x = np.ascontiguousarray(x_bf16, dtype=np.uint16)
        T, k = ids.shape
        m = _GgufMoe(E, H, I, types[0], types[1], types[2], gate.ctypes.data, up.ctypes.data, down.ctypes.data,
                     None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).ctypes.data)
        y = np.empty((T, H), np.uint16)
        inter = np.empty((k, I), np.float32) if want_inter else None
        rc = self.lib.ktxo_moe_forward_gguf(C.byref(m), C.c_int(T), C.c_int(k), ids.ctypes.data_as(C.c_void_p),
                                            w.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                                            y.ctypes.data_as(C.c_void_p),
                                            None if inter is None else inter.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError("ktxo_moe_forward_gguf: unsupported ggml type")
        return (y, inter) if want_inter else y


# ---- CPU leg of bench.py's q4_k_m workload: the reference's OWN llamafile kernels, timed ------------------------------------
def iqk_forward_bench(H: int, I: int, k: int, types, n_layers: int, budget_s: float = 10.0, threads: int | None = None):
    """bs=1 forward of `k` routed experts per layer through the reference's unmodified iqk GEMM kernels
    (third_party/llamafile/iqk_mul_mat.inc in oracle/_ref/libiqk_ref_*.so — what LLAMA_MOE_TP::forward_one reaches through
    llamafile_sgemm, kt-kernel/operators/llamafile/moe.hpp:271-460): x -> Q8_K, gate / up GEMVs split over the threads by
    output rows (iqk's own ith / nth partition), fp32 silu(gate) * up, -> Q8_K, down GEMV.  Weights: random valid blocks, two
    distinct layers' worth (>> the host L3) used in rotation.  Returns bench.py's cpu-leg dict; tok/s = 1 / (n_layers x t_layer).
    Bounded by `budget_s`.  Test infrastructure: only bench.py's CPU leg and tests call this."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))

    def cpu_has(flag):
        try:
            with open("/proc/cpuinfo") as f:
                return any(flag in line.split() for line in f if line.startswith("flags"))
        except OSError:
            return False
    so, sym, isa = (("libiqk_ref_zen4.so", "iqk_mul_mat_zen4", "AVX512-VNNI") if cpu_has("avx512_vnni") else
                    ("libiqk_ref_avx2.so", "iqk_mul_mat", "AVX2"))
    path = os.path.join(here, "_ref", so)
    if not os.path.exists(path):
        return {"value": None, "unit": "tok/s", "cores": 0, "kind": "reference", "sample": f"{so} not built (needs /root/reference at build time)"}
    fn = getattr(C.CDLL(path), sym)
    fn.restype = C.c_bool
    fn.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p,
                   C.c_long, C.c_int, C.c_int]
    BB = {12: 144, 14: 210, 19: 50}
    nth = threads or max(1, min(64, (os.cpu_count() or 8) // 2))
    rng = np.random.default_rng(0)

    def blocks(N, K, ty):
        b = rng.integers(0, 256, (N, K // 256, BB[ty]), dtype=np.uint8)
        d = (rng.random((N, K // 256)) * 0.0006 + 0.0006).astype(np.float16).view(np.uint8).reshape(N, K // 256, 2)
        off = {12: 0, 14: 208, 19: 0}[ty]
        b[..., off:off + 2] = d
        if ty == 12:
            b[..., 2:4] = d
        return b.reshape(N, -1)

    nsets = 2
    W = [[(blocks(I, H, types[0]), blocks(I, H, types[1]), blocks(H, I, types[2])) for _ in range(k)] for _ in range(nsets)]
    o = GgufOracle()
    lib = o.lib
    lib.ktxo_iqk_bench.restype = C.c_double
    lib.ktxo_iqk_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    ptrs = (C.c_void_p * (nsets * k * 3))(*[m.ctypes.data for s_ in W for e_ in s_ for m in e_])
    ty = (C.c_int * 3)(*types)
    x = (rng.standard_normal(H) / 100).astype(np.float32)
    fptr = C.cast(fn, C.c_void_p)
    run = lambda it: lib.ktxo_iqk_bench(fptr, H, I, k, ty, ptrs, nsets, x.ctypes.data, nth, it)
    t1 = run(4)
    n = int(max(8, min(5000, budget_s / max(t1 / 4, 1e-6))))
    dt = run(n)
    t_layer = dt / n
    bpw = ((BB[types[0]] + BB[types[1]]) * H * I + BB[types[2]] * H * I) / 256
    return {"value": round(1.0 / (n_layers * t_layer), 3), "unit": "tok/s", "cores": nth, "kind": "reference",
            "us_per_layer": round(t_layer * 1e6, 1), "GBs": round(k * bpw / t_layer / 1e9, 1),
            "sample": f"{n} bs=1 layer forwards of the reference's iqk_mul_mat kernels ({isa} build of third_party/llamafile/iqk_mul_mat.inc), "
                      f"H={H} I={I} top-{k}, ggml types {tuple(types)}, {nth} OpenMP threads (iqk's own row partition, barriers between the "
                      f"three GEMV stages; oracle/iqk_bench.c), rotating over {nsets} distinct layers; "
                      f"tok/s = 1/({n_layers} layers x t_layer)"}

"""TEST INFRASTRUCTURE ONLY — GGUF k-quant block codec (numpy) + ctypes front-end of oracle/ktx_oracle_gguf.c.

dequantize_q4_k / dequantize_q6_k restate archive/ktransformers/util/custom_gguf.py:326-343,... (which itself restates
ggml-quants.c); tests/golden/make_gguf_golden.py pins them against the reference's own numpy functions on random blocks.
quantize_q4_k / quantize_q6_k are simple encoders written for the tests (any valid encoding is a legal weight file; they are
NOT ggml's search-based quantisers).  Block layouts:
  Q4_K (type 12, 144 B / 256 w): fp16 d, fp16 dmin, 12 B packed 6-bit (scale, min) x 8, 128 B nibbles (4 x [32 low | 32 high])
  Q6_K (type 14, 210 B / 256 w): 128 B low nibbles, 64 B high 2-bit, 16 x int8 scales, fp16 d
"""
import ctypes as C

import numpy as np

GGML_TYPE_Q4_K, GGML_TYPE_Q6_K = 12, 14
BLOCK_BYTES = {GGML_TYPE_Q4_K: 144, GGML_TYPE_Q6_K: 210}


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


QUANT = {GGML_TYPE_Q4_K: quantize_q4_k, GGML_TYPE_Q6_K: quantize_q6_k}
DEQUANT = {GGML_TYPE_Q4_K: dequantize_q4_k, GGML_TYPE_Q6_K: dequantize_q6_k}


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

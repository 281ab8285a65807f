"""TEST INFRASTRUCTURE ONLY — ctypes front-ends for the CPU oracle.

* ``Oracle``     : oracle/libktx_oracle.so, the plain-C restatement (ktx_oracle.c).
* ``Reference``  : oracle/_ref/libkt_ref.so, the reference's own unmodified kernels (ref_driver.cpp), when built.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package ``ktransformers_amd`` must never do so.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libktx_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libkt_ref.so")

FMT_AMXINT4, FMT_AMXINT8, FMT_RAWINT4, FMT_FP8, FMT_BF16, FMT_FP8_PERCHANNEL = 0, 1, 2, 3, 4, 5


def build(ref: bool = True) -> None:
    """Compile the C restatement (always) and oracle/_ref (when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", _HERE, "oracle"], check=True)
    if ref and os.path.isdir("/root/reference/kt-kernel"):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)
    if ref and os.path.isfile("/root/reference/third_party/llamafile/iqk_mul_mat.inc"):
        subprocess.run(["make", "-s", "-C", _HERE, "iqk"], check=True)


def host_has_avx512_vnni() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            flags = f.read()
    except OSError:
        return False
    return all(x in flags for x in ("avx512f", "avx512bw", "avx512_vnni", "avx512_bf16", "avx512vbmi"))


def reference_available() -> bool:
    return os.path.exists(REF_SO) and host_has_avx512_vnni()


# ---------------------------------------------------------------------------------------------------
# bf16 helpers on numpy (uint16 carriers)
# ---------------------------------------------------------------------------------------------------

def f32_to_bf16(a: np.ndarray) -> np.ndarray:
    """RNE fp32 -> bf16 bits with the VCVTNE2PS2BF16 denormal flush (see ktx_oracle.c)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    out = ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)
    den = (u & 0x7F800000) == 0
    out[den] = ((u[den] >> 16) & 0x8000).astype(np.uint16)
    return out


def bf16_to_f32(a: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(a, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _KtxoMoe(C.Structure):
    _fields_ = [
        ("fmt", C.c_int), ("E", C.c_int), ("H", C.c_int), ("I", C.c_int), ("group", C.c_int),
        ("gate_q", C.c_void_p), ("gate_d", C.c_void_p),
        ("up_q", C.c_void_p), ("up_d", C.c_void_p),
        ("down_q", C.c_void_p), ("down_d", C.c_void_p),
        ("gpu_experts_mask", C.c_void_p), ("dp_even_first", C.c_int),
    ]


class Oracle:
    """Plain-C restatement.  Weights are quantised with the restated reference quantiser."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.ktxo_act_fn.restype = C.c_float
        self.lib.ktxo_act_fn.argtypes = [C.c_float, C.c_float]
        self.lib.ktxo_moe_forward.restype = C.c_int
        self.lib.ktxo_bucket.restype = C.c_int

    # -- a6 / a7 ---------------------------------------------------------------------------------
    def quant_act_row(self, x_bf16: np.ndarray):
        K = x_bf16.shape[-1]
        x2 = np.ascontiguousarray(x_bf16.reshape(-1, K))
        q = np.empty(x2.shape, np.int8)
        d = np.empty(x2.shape[0], np.float32)
        for r in range(x2.shape[0]):
            dd = C.c_float()
            self.lib.ktxo_quant_act_row(_p(x2[r]), C.c_int(K), _p(q[r]), C.byref(dd))
            d[r] = dd.value
        return q, d

    def quant_weight(self, fmt: int, w_bf16: np.ndarray):
        """w_bf16 [..., N, K] uint16 -> (q int8 same shape, d fp32 [..., N])."""
        N, K = w_bf16.shape[-2:]
        w2 = np.ascontiguousarray(w_bf16.reshape(-1, K))
        q = np.empty(w2.shape, np.int8)
        d = np.empty(w2.shape[0], np.float32)
        fn = {FMT_AMXINT4: self.lib.ktxo_quant_weight_amxint4, FMT_AMXINT8: self.lib.ktxo_quant_weight_amxint8}[fmt]
        fn(_p(w2), C.c_int(w2.shape[0]), C.c_int(K), _p(q), _p(d))
        return q.reshape(w_bf16.shape), d.reshape(w_bf16.shape[:-1])

    def dequant_amxint4(self, q16: np.ndarray, d: np.ndarray) -> np.ndarray:
        N, K = q16.shape[-2:]
        out = np.empty(q16.shape, np.uint16)
        self.lib.ktxo_dequant_weight_amxint4(_p(np.ascontiguousarray(q16)), _p(np.ascontiguousarray(d)),
                                             C.c_int(int(np.prod(q16.shape[:-1]))), C.c_int(K), _p(out))
        return out

    def act_fn(self, g: float, u: float) -> float:
        return float(self.lib.ktxo_act_fn(C.c_float(g), C.c_float(u)))

    # -- the expert forward ------------------------------------------------------------------------
    def make_moe(self, fmt: int, gate_bf16, up_bf16, down_bf16, mask=None):
        """gate/up [E, I, H], down [E, H, I] bf16 bits -> dict of quantised weights."""
        E, I, H = gate_bf16.shape
        gq, gd = self.quant_weight(fmt, gate_bf16)
        uq, ud = self.quant_weight(fmt, up_bf16)
        dq, dd = self.quant_weight(fmt, down_bf16)
        return dict(fmt=fmt, E=E, H=H, I=I, gate_q=gq, gate_d=gd, up_q=uq, up_d=ud, down_q=dq, down_d=dd,
                    mask=None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8))

    def moe_forward(self, moe: dict, ids: np.ndarray, weights: np.ndarray, x_bf16: np.ndarray,
                    y_prev: np.ndarray | None = None, trace: bool = False):
        T, k = ids.shape
        H, I = moe["H"], moe["I"]
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        weights = np.ascontiguousarray(weights, dtype=np.float32)
        x_bf16 = np.ascontiguousarray(x_bf16, dtype=np.uint16)
        y = np.zeros((T, H), np.uint16) if y_prev is None else np.ascontiguousarray(y_prev, dtype=np.uint16).copy()
        s = _KtxoMoe(moe["fmt"], moe["E"], H, I, int(moe.get("group", 0)),
                     moe["gate_q"].ctypes.data, moe["gate_d"].ctypes.data if moe["gate_d"] is not None else None,
                     moe["up_q"].ctypes.data, moe["up_d"].ctypes.data if moe["up_d"] is not None else None,
                     moe["down_q"].ctypes.data, moe["down_d"].ctypes.data if moe["down_d"] is not None else None,
                     moe["mask"].ctypes.data if moe.get("mask") is not None else None, int(moe.get("dp_even_first", 0)))
        if moe["fmt"] == FMT_RAWINT4:
            rc = self.lib.ktxo_moe_forward_rawint4(C.byref(s), C.c_int(T), C.c_int(k), _p(ids), _p(weights), _p(x_bf16),
                                                   _p(y), C.c_int(0 if y_prev is None else 1))
            if rc != 0:
                raise RuntimeError("ktxo_moe_forward_rawint4 failed")
            return y
        if moe["fmt"] in (FMT_FP8, FMT_BF16, FMT_FP8_PERCHANNEL):
            if trace:
                raise NotImplementedError("traces are only kept for the integer formats")
            rc = self.lib.ktxo_moe_forward_fp(C.byref(s), C.c_int(T), C.c_int(k), _p(ids), _p(weights), _p(x_bf16), _p(y),
                                              C.c_int(0 if y_prev is None else 1))
            if rc != 0:
                raise RuntimeError("ktxo_moe_forward_fp failed")
            return y
        tr = None
        if trace:
            tr = dict(gate=np.zeros((T, k, I), np.uint16), up=np.zeros((T, k, I), np.uint16),
                      act=np.zeros((T, k, I), np.uint16), down=np.zeros((T, k, H), np.uint16))
        rc = self.lib.ktxo_moe_forward(C.byref(s), C.c_int(T), C.c_int(k), _p(ids), _p(weights), _p(x_bf16), _p(y),
                                       C.c_int(0 if y_prev is None else 1),
                                       _p(tr["gate"]) if tr else None, _p(tr["up"]) if tr else None,
                                       _p(tr["act"]) if tr else None, _p(tr["down"]) if tr else None)
        if rc != 0:
            raise RuntimeError("ktxo_moe_forward: unsupported format")
        return (y, tr) if trace else y

    def make_moe_fp8(self, gate_fp8, up_fp8, down_fp8, gate_s, up_s, down_s, mask=None, dp_even_first=0):
        """e4m3 bytes gate/up [E,I,H], down [E,H,I]; fp32 scale_inv [E, N/128, K/128]."""
        E, I, H = gate_fp8.shape
        c = np.ascontiguousarray
        return dict(fmt=FMT_FP8, E=E, H=H, I=I, gate_q=c(gate_fp8, dtype=np.uint8), up_q=c(up_fp8, dtype=np.uint8),
                    down_q=c(down_fp8, dtype=np.uint8), gate_d=c(gate_s, dtype=np.float32), up_d=c(up_s, dtype=np.float32),
                    down_d=c(down_s, dtype=np.float32), mask=mask, dp_even_first=dp_even_first)

    def make_moe_fp8_perchannel(self, gate_fp8, up_fp8, down_fp8, gate_s, up_s, down_s, mask=None, dp_even_first=0):
        """e4m3 bytes gate/up [E,I,H], down [E,H,I]; fp32 scale per output row: gate/up [E,I], down [E,H] (FP8_PERCHANNEL)."""
        E, I, H = gate_fp8.shape
        c = np.ascontiguousarray
        return dict(fmt=FMT_FP8_PERCHANNEL, E=E, H=H, I=I, gate_q=c(gate_fp8, dtype=np.uint8), up_q=c(up_fp8, dtype=np.uint8),
                    down_q=c(down_fp8, dtype=np.uint8), gate_d=c(gate_s, dtype=np.float32).reshape(E, I),
                    up_d=c(up_s, dtype=np.float32).reshape(E, I), down_d=c(down_s, dtype=np.float32).reshape(E, H), mask=mask,
                    dp_even_first=dp_even_first)

    def make_moe_rawint4(self, gate_p, up_p, down_p, gate_s, up_s, down_s, mask=None, uncontracted=0):
        """packed nibbles gate/up [E,I,H/2], down [E,H,I/2] uint8; scales bf16 bits (uint16) or fp32 [E,N,K/32]."""
        E, I, H2 = gate_p.shape
        c = np.ascontiguousarray

        def sc(a):
            return c(bf16_to_f32(a) if a.dtype == np.uint16 else a, dtype=np.float32)
        return dict(fmt=FMT_RAWINT4, E=E, H=H2 * 2, I=I, group=32, gate_q=c(gate_p, dtype=np.uint8), up_q=c(up_p, dtype=np.uint8),
                    down_q=c(down_p, dtype=np.uint8), gate_d=sc(gate_s), up_d=sc(up_s), down_d=sc(down_s), mask=mask,
                    dp_even_first=uncontracted)

    def make_moe_bf16(self, gate, up, down, mask=None, dp_even_first=0):
        E, I, H = gate.shape
        c = np.ascontiguousarray
        return dict(fmt=FMT_BF16, E=E, H=H, I=I, gate_q=c(gate, dtype=np.uint16), up_q=c(up, dtype=np.uint16),
                    down_q=c(down, dtype=np.uint16), gate_d=None, up_d=None, down_d=None, mask=mask,
                    dp_even_first=dp_even_first)

    def e4m3_to_f32(self, b: np.ndarray) -> np.ndarray:
        self.lib.ktxo_e4m3_to_f32.restype = C.c_float
        lut = np.array([self.lib.ktxo_e4m3_to_f32(C.c_uint8(i)) for i in range(256)], np.float32)
        return lut[np.asarray(b, dtype=np.uint8)]

    def bucket(self, E: int, ids: np.ndarray, mask=None):
        T, k = ids.shape
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        num = np.zeros(E, np.int32)
        pos = np.zeros((T, k), np.int32)
        emap = np.zeros(E, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        n = self.lib.ktxo_bucket(C.c_int(E), _p(m), C.c_int(T), C.c_int(k), _p(ids), _p(num), _p(pos), _p(emap))
        return num, pos, emap[:n]


class Reference:
    """The reference's own kernels (oracle/_ref).  One worker pool per instance."""

    KIND = {FMT_AMXINT4: 0, FMT_AMXINT8: 1, FMT_RAWINT4: 2, FMT_FP8: 3, FMT_BF16: 4, FMT_FP8_PERCHANNEL: 5}

    def __init__(self, threads: int = 4, subpools: int = 1):
        if not reference_available():
            raise RuntimeError("oracle/_ref/libkt_ref.so not built or host lacks AVX512-VNNI/BF16")
        self.lib = C.CDLL(REF_SO)
        self.lib.ktref_pool_create.restype = C.c_void_p
        self.lib.ktref_moe_create.restype = C.c_void_p
        self.lib.ktref_last_error.restype = C.c_char_p
        self.threads = threads * subpools
        self.pool = C.c_void_p(self.lib.ktref_pool_create(C.c_int(subpools), C.c_int(threads)))
        if not self.pool:
            raise RuntimeError(self.lib.ktref_last_error().decode())

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.ktref_last_error().decode())

    def close(self):
        """Join and free the worker pool (its threads busy-wait between jobs: a pool that is no longer used must not stay)."""
        if self.pool:
            self.lib.ktref_pool_destroy.argtypes = [C.c_void_p]
            self.lib.ktref_pool_destroy(self.pool)
            self.pool = None

    def make_moe(self, fmt: int, gate, up, down, k: int, max_len: int = 64, group_size: int = 0,
                 gate_scale=None, up_scale=None, down_scale=None):
        E, I, H = gate.shape[0], (gate.shape[1] if fmt != FMT_RAWINT4 else gate.shape[1]), None
        if fmt in (FMT_AMXINT4, FMT_AMXINT8, FMT_BF16):
            E, I, H = gate.shape
        else:
            raise NotImplementedError("use make_moe_quant for pre-quantised formats")
        h = C.c_void_p(self.lib.ktref_moe_create(C.c_int(self.KIND[fmt]), C.c_int(E), C.c_int(k), C.c_int(H),
                                                 C.c_int(I), C.c_int(max_len), C.c_int(group_size), self.pool))
        if not h:
            raise RuntimeError(self.lib.ktref_last_error().decode())
        keep = [np.ascontiguousarray(a) for a in (gate, up, down)]
        self._chk(self.lib.ktref_moe_load(h, _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(gate_scale), _p(up_scale),
                                          _p(down_scale)))
        return dict(h=h, H=H, I=I, E=E, k=k)

    def make_moe_quant(self, fmt: int, E: int, H: int, I: int, k: int, gate, up, down, gate_scale, up_scale,
                       down_scale, max_len: int = 64, group_size: int = 0):
        h = C.c_void_p(self.lib.ktref_moe_create(C.c_int(self.KIND[fmt]), C.c_int(E), C.c_int(k), C.c_int(H),
                                                 C.c_int(I), C.c_int(max_len), C.c_int(group_size), self.pool))
        if not h:
            raise RuntimeError(self.lib.ktref_last_error().decode())
        keep = [np.ascontiguousarray(a) for a in (gate, up, down, gate_scale, up_scale, down_scale)]
        self._chk(self.lib.ktref_moe_load(h, *[_p(a) for a in keep]))
        return dict(h=h, H=H, I=I, E=E, k=k, keep=keep)

    def moe_forward(self, moe: dict, ids, weights, x_bf16, y_prev=None):
        T, k = ids.shape
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        weights = np.ascontiguousarray(weights, dtype=np.float32)
        x_bf16 = np.ascontiguousarray(x_bf16, dtype=np.uint16)
        y = np.zeros((T, moe["H"]), np.uint16) if y_prev is None else np.ascontiguousarray(y_prev, np.uint16).copy()
        self._chk(self.lib.ktref_moe_forward(moe["h"], C.c_int(T), C.c_int(k), _p(ids), _p(weights), _p(x_bf16),
                                             _p(y), C.c_int(0 if y_prev is None else 1)))
        return y

    def free_moe(self, moe: dict):
        self.lib.ktref_moe_destroy(moe["h"])

    def quant_roundtrip_int4(self, w_bf16: np.ndarray):
        N, K = w_bf16.shape
        out = np.empty((N, K), np.uint16)
        d = np.empty(N, np.float32)
        self._chk(self.lib.ktref_quant_roundtrip(C.c_int(0), C.c_int(N), C.c_int(K),
                                                 _p(np.ascontiguousarray(w_bf16)), _p(out), _p(d)))
        return out, d

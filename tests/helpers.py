"""Shared test helpers: seeded synthetic inputs in the reference tests' style
(kt-kernel/test/per_commit/test_moe_amx_accuracy_int4.py:104-130, test_moe_rawint4_accuracy.py:175-183)."""
import numpy as np

from oracle.oracle import bf16_to_f32, f32_to_bf16  # noqa: F401


def make_case(seed, E, k, H, I, T, wscale=0.1, xscale=0.01, invalid_ids=False):
    rng = np.random.default_rng(seed)
    gate = f32_to_bf16((rng.standard_normal((E, I, H)) * wscale).astype(np.float32))
    up = f32_to_bf16((rng.standard_normal((E, I, H)) * wscale).astype(np.float32))
    down = f32_to_bf16((rng.standard_normal((E, H, I)) * wscale).astype(np.float32))
    x = f32_to_bf16((rng.standard_normal((T, H)) * xscale).astype(np.float32))
    ids = np.stack([rng.permutation(E)[:k] for _ in range(T)]).astype(np.int64) if T else np.zeros((0, k), np.int64)
    w = rng.random((T, k)).astype(np.float32)
    if invalid_ids and T:
        ids[0, 0] = -1
        ids[-1, -1] = E + 3
    return dict(gate=gate, up=up, down=down, x=x, ids=ids, w=w)


def torch_bf16(a_u16, device):
    import torch
    return torch.from_numpy(a_u16.view(np.int16).copy()).view(torch.bfloat16).to(device)


def numpy_u16(t):
    import torch
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


_E4M3_LUT = None


def e4m3_lut():
    """float value of every e4m3 byte exactly as the reference's LUT decodes it (0x7F/0xFF -> +-480)."""
    global _E4M3_LUT
    if _E4M3_LUT is None:
        v = np.zeros(256, np.float32)
        for b in range(256):
            e, m = (b >> 3) & 15, b & 7
            f = m * 2.0 ** -9 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7)
            v[b] = -f if b & 0x80 else f
        _E4M3_LUT = v
    return _E4M3_LUT


def fp8_block_quant(w_f32):
    """[E,N,K] fp32 -> (e4m3 bytes [E,N,K], fp32 scale_inv [E,N/128,K/128]); scale = amax/448 per 128x128 block, nearest
    e4m3 (the scheme of kt-kernel/test/per_commit/test_moe_avx2_accuracy_fp8.py:28-63).  Codes 0x7F/0xFF (NaN in OCP
    e4m3fn, 480 in the reference's table) are never produced."""
    E, N, K = w_f32.shape
    grid = e4m3_lut()[:127].astype(np.float32)          # non-negative finite codes 0..126 (max 448)
    wb = w_f32.reshape(E, N // 128, 128, K // 128, 128)
    amax = np.abs(wb).max(axis=(2, 4), keepdims=True)
    scale = (amax / 448.0).astype(np.float32)
    scale[scale == 0] = 1
    v = (wb / scale).astype(np.float32)
    mid = (grid[1:] + grid[:-1]) / 2
    idx = np.searchsorted(mid, np.abs(v)).astype(np.uint8)
    q = idx | ((v < 0).astype(np.uint8) << 7)
    return q.reshape(E, N, K), scale.reshape(E, N // 128, K // 128).astype(np.float32)


def fp8_perchannel_quant(w_f32):
    """[E,N,K] fp32 -> (e4m3 bytes [E,N,K], fp32 scale [E,N]): one scale per output row, scale = row amax / 448, nearest e4m3
    (the per-channel scheme of GLM-4.7-FP8 style checkpoints the reference's FP8_PERCHANNEL method loads,
    kt-kernel/operators/amx/fp8-perchannel-moe.hpp:508-555)."""
    E, N, K = w_f32.shape
    grid = e4m3_lut()[:127].astype(np.float32)
    amax = np.abs(w_f32).max(axis=2, keepdims=True)
    scale = (amax / 448.0).astype(np.float32)
    scale[scale == 0] = 1
    v = (w_f32 / scale).astype(np.float32)
    mid = (grid[1:] + grid[:-1]) / 2
    idx = np.searchsorted(mid, np.abs(v)).astype(np.uint8)
    q = idx | ((v < 0).astype(np.uint8) << 7)
    return q, scale.reshape(E, N).astype(np.float32)


def rawint4_quantize(w_f32, group=32):
    """[E,N,K] fp32 -> (packed uint8 [E,N,K/2] with byte = ((q1+8)<<4)|(q0+8), bf16 scale bits uint16 [E,N,K/group]),
    vectorised form of rawint4_quantize in kt-kernel/test/per_commit/test_moe_rawint4_accuracy.py:69-94
    (scale = amax/7 or 1, stored as bf16; q = clamp(round(w/scale)+8, 0, 15) with Python's round-half-even on the
    fp32/fp64 quotient)."""
    E, N, K = w_f32.shape
    wg = w_f32.reshape(E, N, K // group, group).astype(np.float32)
    amax = np.abs(wg).max(-1, keepdims=True).astype(np.float64)
    scale = np.where(amax > 0, amax / 7.0, 1.0)
    sb = f32_to_bf16(scale.astype(np.float32))                     # the test stores the scale in a bf16 tensor...
    q = np.clip(np.rint(wg.astype(np.float64) / scale) + 8, 0, 15).astype(np.uint8)   # ...but divides by the fp64 one
    q = q.reshape(E, N, K)
    packed = (q[..., 1::2] << 4) | q[..., 0::2]
    return np.ascontiguousarray(packed), np.ascontiguousarray(sb.reshape(E, N, K // group))


def write_gguf(path, tensors, meta=None, alignment=32):
    """Minimal GGUF v3 writer for the loader tests.  tensors: {name: (ggml_type, shape_ggml_order, raw_bytes)};
    shape is in ggml order (fastest dim first), raw_bytes the tensor's blocks."""
    import struct

    def s(x):
        b = x.encode()
        return struct.pack("<Q", len(b)) + b

    meta = dict(meta or {})
    meta.setdefault("general.architecture", "deepseek2")
    meta.setdefault("general.alignment", alignment)
    kv = b""
    for k, v in meta.items():
        if isinstance(v, str):
            kv += s(k) + struct.pack("<I", 8) + s(v)
        else:
            kv += s(k) + struct.pack("<I", 4) + struct.pack("<I", int(v))
    infos, blob, off = b"", b"", 0
    for name, (ty, shape, raw) in tensors.items():
        raw = bytes(raw)
        off += (alignment - off % alignment) % alignment
        blob += b"\0" * (off - len(blob)) + raw
        infos += s(name) + struct.pack("<I", len(shape)) + b"".join(struct.pack("<Q", d) for d in shape) + struct.pack("<IQ", ty, off)
        off += len(raw)
    head = b"GGUF" + struct.pack("<IQQ", 3, len(tensors), len(meta)) + kv + infos
    pad = (alignment - len(head) % alignment) % alignment
    with open(path, "wb") as f:
        f.write(head + b"\0" * pad + blob)


def hash_bf16(shape, seed, scale, device="cpu", center=0.0):
    """Deterministic pseudo-random bf16 tensor whose BITS are identical on the CPU and on the GPU: an integer hash of the
    element index (int64 arithmetic, exact everywhere) -> the sum of two 16-bit uniforms (exact in fp32) -> one fp32 multiply
    and add -> torch's round-to-nearest-even bf16 cast.  Standard deviation `scale`, mean `center`, triangular distribution.
    Lets a GPU test rebuild on the device the gigabytes of weights a CPU golden run was made with, from a seed
    (tests/golden/make_v3_layer_golden.py)."""
    import torch
    n = int(np.prod(shape))
    out = torch.empty(n, dtype=torch.bfloat16, device=device)
    step = 1 << 24
    k = np.float32(scale / 26754.68)      # sqrt(2 * (65536^2 - 1) / 12): the std of the sum of two 16-bit uniforms
    for a in range(0, n, step):
        i = torch.arange(a, min(n, a + step), dtype=torch.int64, device=device)
        x = (i * 0x9E3779B1 + (seed + 1) * 0x85EBCA77) & 0xFFFFFFFF
        x ^= x >> 15
        x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
        x ^= x >> 12
        x = (x * 0x297A2D39) & 0xFFFFFFFF
        x ^= x >> 15
        v = ((x & 0xFFFF).to(torch.float32) + (x >> 16).to(torch.float32) - 65535.0) * float(k)
        if center:
            v = v + float(np.float32(center))
        out[a:a + i.numel()] = v.to(torch.bfloat16)
    return out.view(*shape)

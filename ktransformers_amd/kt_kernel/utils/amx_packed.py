"""Inverse of the reference's AMX tile packing, so that checkpoints written by its converter (`SafeTensorLoader` keys
`blk.L.ffn_{gate,up,down}_exps.E.numa.N.{weight,scale}`) can be ingested through `ktx_moe_load_quantized`.

Layout restated from BufferBInt4Impl::_pack_block / GemmKernel224Int8::BufferB::_pack_block
(kt-kernel/operators/amx/la/amx_buffers.hpp:542-627, amx_kernels.hpp:1103-1151) with the constants of
GemmKernel224Int4 / Int8 (amx_kernels.hpp:960-985, 1559-1585):

  * rows are grouped in N_BLOCKs (128 for int4, 64 for int8), K in K_BLOCKs of 3584; block (nb, kb) starts at element offset
    nb_begin*k + kb_begin*nb_size and holds [nb_size/32 row groups][kb_size/KS k-steps] tiles of 32 rows x 64 bytes
    (KS = 128 elements for int4 — low nibble = element j, high nibble = element 64+j of the step — and 64 for int8);
  * each 16-row half of a tile is stored transposed as a 16x16 matrix of 32-bit words (the VNNI layout `transpose_16x16_32bit`
    produces; it is its own inverse);
  * int4 nibbles are two's-complement and stand for nibble*16 (the multiplicand the int8 dot product sees), scale d = amax/112.

Pinned bit-for-bit against the reference's own packer: tests/test_amx_packed_cpu.py (oracle/_ref `ktref_pack_b`)."""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

N_STEP, K_BLOCK = 32, 3584
N_BLOCK = {4: 128, 8: 64}


def unpack_matrix(packed: np.ndarray, n: int, k: int, bits: int) -> np.ndarray:
    """packed bytes of one BufferB (n*k*bits/8) -> int8 [n, k] integer multiplicands (int4: nibble*16)."""
    if bits not in (4, 8):
        raise ValueError("bits must be 4 or 8")
    ks = 128 if bits == 4 else 64  # elements per 64-byte tile row
    if n % N_STEP or k % ks:
        raise ValueError(f"AMX packing needs n % {N_STEP} == 0 and k % {ks} == 0, got n={n}, k={k}")
    raw = np.ascontiguousarray(packed).view(np.uint8).reshape(-1)
    if raw.size != n * k * bits // 8:
        raise ValueError(f"packed matrix has {raw.size} bytes, expected {n * k * bits // 8} for [{n}, {k}] at {bits} bits")
    out = np.empty((n, k), np.int8)
    for nb in range(0, n, N_BLOCK[bits]):
        nbs = min(N_BLOCK[bits], n - nb)
        for kb in range(0, k, K_BLOCK):
            kbs = min(K_BLOCK, k - kb)
            start = (nb * k + kb * nbs) * bits // 8
            tiles = raw[start:start + nbs * kbs * bits // 8].reshape(nbs // N_STEP, kbs // ks, 2, 16, 16, 4)
            rows = tiles.transpose(0, 2, 4, 1, 3, 5).reshape(nbs, kbs // ks, 64)  # [row][k-step][byte]
            if bits == 8:
                out[nb:nb + nbs, kb:kb + kbs] = rows.reshape(nbs, kbs).view(np.int8)
            else:
                lo = ((rows & 0x0F) ^ 8).astype(np.int16) - 8
                hi = ((rows >> 4) ^ 8).astype(np.int16) - 8
                out[nb:nb + nbs, kb:kb + kbs] = (np.concatenate([lo, hi], axis=2) * 16).astype(np.int8).reshape(nbs, kbs)
    return out


def unpack_expert(parts: Sequence[np.ndarray], scales: Sequence[np.ndarray], n: int, k: int, bits: int,
                  split: str) -> Tuple[np.ndarray, np.ndarray]:
    """One expert matrix from its NUMA parts -> (int8 [n, k], fp32 [n]).  gate/up are sharded over rows (`split="n"`): parts
    are concatenated.  down is sharded over K (`split="k"`) with one scale per (row, shard) — the reference then sums fp32
    partials per shard (operators/amx/moe_base.hpp:749-791); a single per-row scale cannot express that: multi-part
    checkpoints are loaded part by part into one handle per part instead (AMXMoEWrapper._load_tp_parts), which calls this
    function with one part at a time."""
    tp = len(parts)
    if tp == 1:
        return unpack_matrix(parts[0], n, k, bits), np.ascontiguousarray(scales[0]).view(np.float32).reshape(n).copy()
    if split == "n":
        if n % tp:
            raise ValueError(f"{n} rows do not split over {tp} NUMA parts")
        q = np.concatenate([unpack_matrix(p, n // tp, k, bits) for p in parts], axis=0)
        s = np.concatenate([np.ascontiguousarray(x).view(np.float32).reshape(n // tp) for x in scales])
        return q, s
    raise NotImplementedError(f"a matrix sharded over K across {tp} NUMA parts has one scale per (row, part) and cannot be merged into "
                              "one per-row-scaled matrix; load it part by part (AMXMoEWrapper._load_tp_parts)")

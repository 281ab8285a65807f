"""CPU: the AMX tile un-packer (ktransformers_amd/kt_kernel/utils/amx_packed.py) against the reference's OWN packer
(BufferB::from_mat of GemmKernel224Int4 / Int8 through oracle/_ref `ktref_pack_b`): the unpacked multiplicands and scales
must equal an independent numpy evaluation of the quantiser, for shapes that exercise several N blocks, a partial N block
and two K blocks."""
import ctypes as C

import numpy as np
import pytest

from ktransformers_amd.kt_kernel.utils.amx_packed import unpack_expert, unpack_matrix
from oracle import oracle as O


def pack_with_reference(kind, w_bf16_bits, n, k):
    lib = C.CDLL(O.REF_SO)
    packed = np.zeros(n * k // (2 if kind == 0 else 1), np.uint8)
    scales = np.zeros(n, np.float32)
    rc = lib.ktref_pack_b(C.c_int(kind), C.c_int(n), C.c_int(k), w_bf16_bits.ctypes.data_as(C.c_void_p),
                          packed.ctypes.data_as(C.c_void_p), scales.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return packed, scales


def quantise_numpy(w, bits):
    """a7 / its int8 analogue: d = amax/112 (or /127); q = sat8(rne(w * (1/d))); int4: sign-magnitude round to a multiple of 16."""
    amax = np.abs(w).max(axis=1)
    d = (amax / np.float32(112.0 if bits == 4 else 127.0)).astype(np.float32)
    inv = np.divide(np.float32(1.0), d, out=np.zeros_like(d), where=d != 0)
    q = np.clip(np.rint(w * inv[:, None]), -128, 127).astype(np.int32)
    if bits == 4:
        q = np.sign(q) * ((np.abs(q) + 8) & 0xF0)
    return q.astype(np.int8), d


@pytest.mark.skipif(not O.reference_available(), reason="oracle/_ref not built or host lacks AVX512-VNNI")
@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("shape", [(32, 128), (160, 256), (256, 3584 + 256), (96, 1408)])
def test_unpack_matches_reference_packer(bits, shape):
    n, k = shape
    rng = np.random.default_rng(n + k + bits)
    w = O.bf16_to_f32(O.f32_to_bf16((rng.standard_normal((n, k)) * 0.05).astype(np.float32)))
    w[3] = 0  # an all-zero row: d = 0
    packed, scales = pack_with_reference(0 if bits == 4 else 1, O.f32_to_bf16(w), n, k)
    q = unpack_matrix(packed, n, k, bits)
    want_q, want_d = quantise_numpy(w, bits)
    assert np.array_equal(scales, want_d)
    assert np.array_equal(q, want_q), f"{int((q != want_q).sum())} of {q.size} multiplicands differ"


@pytest.mark.skipif(not O.reference_available(), reason="oracle/_ref not built or host lacks AVX512-VNNI")
def test_row_sharded_parts_concatenate_and_k_sharded_is_rejected():
    n, k = 128, 256
    rng = np.random.default_rng(5)
    w = O.bf16_to_f32(O.f32_to_bf16((rng.standard_normal((n, k)) * 0.05).astype(np.float32)))
    halves = [pack_with_reference(0, O.f32_to_bf16(np.ascontiguousarray(w[i * 64:(i + 1) * 64])), 64, k) for i in range(2)]
    q, s = unpack_expert([h[0] for h in halves], [h[1] for h in halves], n, k, 4, "n")
    want_q, want_d = quantise_numpy(w, 4)
    assert np.array_equal(q, want_q) and np.array_equal(s, want_d)
    with pytest.raises(NotImplementedError):
        unpack_expert([h[0] for h in halves], [h[1] for h in halves], n, k, 4, "k")


def test_shape_errors():
    with pytest.raises(ValueError):
        unpack_matrix(np.zeros(10, np.uint8), 32, 128, 4)
    with pytest.raises(ValueError):
        unpack_matrix(np.zeros(31 * 64, np.uint8), 31, 128, 4)
    with pytest.raises(ValueError):
        unpack_matrix(np.zeros(32 * 64, np.uint8), 32, 128, 5)

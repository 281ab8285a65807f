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

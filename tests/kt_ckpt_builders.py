"""Tiny synthetic expert checkpoints in every on-disk style the reference's kt_kernel loaders accept
(kt-kernel/python/utils/loader.py).  Deterministic: the golden maker (with the reference's loaders) and the test (with ours)
build byte-identical files from these recipes."""
import hashlib
import os

import numpy as np
import torch
from safetensors.torch import save_file

E, H, I = 3, 256, 128


def _rng(seed):
    return torch.Generator().manual_seed(seed)


def _fp8(shape, g):
    return (torch.randn(shape, generator=g) * 0.5).to(torch.float8_e4m3fn)


def _save(folder, shards):
    os.makedirs(folder, exist_ok=True)
    for name, tensors in shards.items():
        save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(folder, name))


def fp8_block(folder, prefix="model.layers.{L}.mlp.experts", names=("gate_proj", "up_proj", "down_proj"),
              scale="weight_scale_inv", per_channel=False, layers=(1, 2), seed=0, dims=None):
    E, H, I = dims or (globals()["E"], globals()["H"], globals()["I"])
    g = _rng(seed)
    shards = {}
    for L in layers:
        t = {}
        for e in range(E):
            for stem, (n, k) in zip(names, ((I, H), (I, H), (H, I))):
                base = f"{prefix.format(L=L)}.{e}.{stem}"
                t[f"{base}.weight"] = _fp8((n, k), g)
                t[f"{base}.{scale}"] = (torch.rand((n, 1) if per_channel else (n // 128, k // 128), generator=g) + 0.5).float()
        t[f"model.layers.{L}.self_attn.o_proj.weight"] = torch.randn(8, 8, generator=g).to(torch.bfloat16)
        shards[f"model-{L:05d}-of-00009.safetensors"] = t
    _save(folder, shards)


def bf16_per_expert(folder, prefix="model.layers.{L}.mlp.experts", names=("gate_proj", "up_proj", "down_proj"), seed=1, dims=None):
    E, H, I = dims or (globals()["E"], globals()["H"], globals()["I"])
    g = _rng(seed)
    t = {}
    for e in range(E):
        for stem, (n, k) in zip(names, ((I, H), (I, H), (H, I))):
            t[f"{prefix.format(L=3)}.{e}.{stem}.weight"] = torch.randn(n, k, generator=g).to(torch.bfloat16)
    _save(folder, {"model.safetensors": t})


def bf16_packed(folder, vl=False, seed=2):
    g = _rng(seed)
    base = "model.language_model.layers.4" if vl else "model.layers.4"
    _save(folder, {"model.safetensors": {
        f"{base}.mlp.experts.gate_up_proj": torch.randn(E, 2 * I, H, generator=g).to(torch.bfloat16),
        f"{base}.mlp.experts.down_proj": torch.randn(E, H, I, generator=g).to(torch.bfloat16)}})


def compressed_int4(folder, int32=True, with_shape=True, prefix="model.layers.5.mlp.experts", seed=3, dims=None):
    E, H, I = dims or (globals()["E"], globals()["H"], globals()["I"])
    g = _rng(seed)
    t = {}
    for e in range(E):
        for stem, (n, k) in zip(("gate", "up", "down"), ((I, H), (I, H), (H, I))):
            packed = torch.randint(0, 256, (n, k // 2), generator=g, dtype=torch.uint8)
            base = f"{prefix}.{e}.{stem}_proj"
            t[f"{base}.weight_packed"] = packed.view(torch.int32) if int32 else packed
            t[f"{base}.weight_scale"] = (torch.rand(n, k // 32, generator=g) * 0.01).to(torch.bfloat16)
            if with_shape:
                t[f"{base}.weight_shape"] = torch.tensor([n, k], dtype=torch.int32)
    _save(folder, {"model.safetensors": t})


def amx_packed(folder, numa=2, seed=4):
    g = _rng(seed)
    t = {}
    for e in range(E):
        for p, nbytes in (("up", I * H // 2 // numa), ("gate", I * H // 2 // numa), ("down", H * I // 2 // numa)):
            for n in range(numa):
                t[f"blk.6.ffn_{p}_exps.{e}.numa.{n}.weight"] = torch.randint(-128, 128, (nbytes,), generator=g, dtype=torch.int8)
                t[f"blk.6.ffn_{p}_exps.{e}.numa.{n}.scale"] = torch.rand((I if p != "down" else H) // (numa if p != "down" else 1),
                                                                         generator=g).float()
    _save(folder, {"a.safetensors": t})


# name -> (builder, builder kwargs, loader class name, loader kwargs, base_key)
CASES = {
    "fp8_deepseek_block": (fp8_block, {}, "FP8SafeTensorLoader", {}, "model.layers.2"),
    "fp8_mixtral_per_channel": (fp8_block, dict(prefix="model.layers.{L}.block_sparse_moe.experts", names=("w1", "w3", "w2"),
                                                scale="weight_scale", per_channel=True), "FP8SafeTensorLoader", {}, "model.layers.1"),
    "fp8_mistral_no_model_prefix_blockwise_weight_scale": (fp8_block, dict(prefix="layers.{L}.experts", names=("w1", "w3", "w2"),
                                                                           scale="weight_scale"), "FP8SafeTensorLoader", {}, "model.layers.2"),
    "fp8_vl_prefix": (fp8_block, dict(prefix="model.language_model.layers.{L}.mlp.experts"), "FP8SafeTensorLoader", {}, "model.layers.1"),
    "fp8_explicit_per_channel": (fp8_block, dict(scale="weight_scale", per_channel=True), "FP8SafeTensorLoader",
                                 dict(scale_suffix="weight_scale"), "model.layers.1"),
    "fp8_missing_layer": (fp8_block, {}, "FP8SafeTensorLoader", {}, "model.layers.7"),
    "bf16_deepseek": (bf16_per_expert, {}, "BF16SafeTensorLoader", {}, "model.layers.3"),
    "bf16_mixtral": (bf16_per_expert, dict(prefix="model.layers.{L}.block_sparse_moe.experts", names=("w1", "w3", "w2")),
                     "BF16SafeTensorLoader", {}, "model.layers.3"),
    "bf16_packed": (bf16_packed, {}, "BF16SafeTensorLoader", {}, "model.layers.4"),
    "bf16_packed_vl": (bf16_packed, dict(vl=True), "BF16SafeTensorLoader", {}, "model.layers.4"),
    "int4_int32_with_shape": (compressed_int4, {}, "CompressedSafeTensorLoader", {}, "model.layers.5"),
    "int4_uint8_no_shape": (compressed_int4, dict(int32=False, with_shape=False), "CompressedSafeTensorLoader", {}, "model.layers.5"),
    "int4_language_model_prefix": (compressed_int4, dict(prefix="language_model.model.layers.5.mlp.experts"),
                                   "CompressedSafeTensorLoader", {}, "model.layers.5"),
    "int4_missing": (compressed_int4, {}, "CompressedSafeTensorLoader", {}, "model.layers.9"),
    "amx_numa2": (amx_packed, {}, "SafeTensorLoader", {}, "blk.6"),
    "amx_missing": (amx_packed, {}, "SafeTensorLoader", {}, "blk.0"),
}

ATTRS = ("_detected_format", "_scale_suffix", "_is_per_channel", "_is_vl_model")


def digest(x):
    """(shape, dtype, sha1 of the bytes) of a torch tensor or numpy array."""
    if isinstance(x, torch.Tensor):
        raw = x.contiguous().view(torch.uint8).numpy().tobytes() if x.numel() else b""
        return [list(x.shape), str(x.dtype).replace("torch.", ""), hashlib.sha1(raw).hexdigest()]
    a = np.ascontiguousarray(x)
    return [list(a.shape), str(a.dtype), hashlib.sha1(a.tobytes()).hexdigest()]


def summarise(result):
    """Nested lists of tensors -> nested lists of digests."""
    def walk(v):
        return [walk(u) for u in v] if isinstance(v, (list, tuple)) else digest(v)
    return {k: walk(v) for k, v in sorted(result.items())}


def run_case(module, name, folder):
    """Build the checkpoint, load it through `module`'s loader -> JSON-able record (result digests or the exception type)."""
    builder, bkw, cls, lkw, base_key = CASES[name]
    builder(folder, **bkw)
    loader = getattr(module, cls)(folder, **lkw)
    rec = {"attrs": {a: getattr(loader, a) for a in ATTRS if hasattr(loader, a)}}
    try:
        rec["result"] = summarise(loader.load_experts(base_key))
    except (ValueError, KeyError) as e:
        rec["raises"] = type(e).__name__
    rec["has"] = [loader.has_tensor(k) for k in ("model.layers.2.self_attn.o_proj.weight", "nope")]
    loader.close_all_handles()
    return rec

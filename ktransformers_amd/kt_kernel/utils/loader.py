"""Checkpoint readers for the expert-weight formats the reference's live `kt_kernel` package ingests.

Same class names, constructor arguments, `load_experts(base_key, device)` return structures and error types as
kt-kernel/python/utils/loader.py, so `KTMoEWrapper` callers (SGLang) and checkpoints move over unchanged:

  SafeTensorLoader            AMX-packed, NUMA-sharded `blk.L.ffn_{up,gate,down}_exps.E.numa.N.{weight,scale}`   (loader.py:102-293)
  FP8SafeTensorLoader         DeepSeek / Mixtral / Mistral key styles, block-wise or per-channel scales            (loader.py:296-512)
  BF16SafeTensorLoader        the same three styles + the stacked `mlp.experts.gate_up_proj` layout               (loader.py:515-676)
  CompressedSafeTensorLoader  compressed-tensors int4 `weight_packed` / `weight_scale` / `weight_shape`           (loader.py:679-777)

Built differently from the reference: one key index (`_Index`) shared by every loader, one naming-scheme table, and a
generic expert-run probe.  GGUF is read by ktransformers_amd/util/gguf_loader.py.  Pinned against the reference's loaders
reading the same files: tests/golden/make_kt_loader_golden.py -> tests/test_kt_loader_cpu.py.
"""
from __future__ import annotations

import gc
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from safetensors import safe_open

# experts path below the layer key, (gate, up, down) tensor stems — kt-kernel/python/utils/loader.py:312-316
_SCHEMES: Dict[str, Tuple[str, Tuple[str, str, str]]] = {
    "deepseek": ("{base}.mlp.experts", ("gate_proj", "up_proj", "down_proj")),
    "mixtral": ("{base}.block_sparse_moe.experts", ("w1", "w3", "w2")),
    "mistral": ("{base}.experts", ("w1", "w3", "w2")),
}


class _Index:
    """key -> (file, lazily opened handle) over every *.safetensors below a directory (files visited in sorted order,
    later files win on duplicate keys, like the reference's walk)."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FileNotFoundError(f"Path not found: {path}")
        folder = os.path.dirname(path) if os.path.isfile(path) else path
        self.handles: Dict[str, object] = {}
        self.where: Dict[str, str] = {}
        seen = False
        for root, _, files in os.walk(folder):
            for name in sorted(files):
                if not name.endswith(".safetensors"):
                    continue
                seen = True
                if name not in self.handles:
                    try:
                        self.handles[name] = safe_open(os.path.join(root, name), framework="pt")
                    except Exception as e:  # unreadable shard: skipped, as the reference does
                        print(f"Error opening Safetensor file {os.path.join(root, name)}: {e}")
                        continue
                for key in self.handles[name].keys():
                    self.where[key] = name
        if not seen:
            raise FileNotFoundError(f"No Safetensor files found in {folder}")

    def get(self, key: str) -> torch.Tensor:
        if key not in self.where:
            raise KeyError(f"Key {key} not found in Safetensor files")
        handle = self.handles.get(self.where[key])
        if handle is None:
            raise FileNotFoundError(f"File {self.where[key]} not found in Safetensor files")
        return handle.get_tensor(key)


def _run_length(has, template: str) -> int:
    """Number of consecutive ids 0, 1, 2, ... for which `template.format(i)` exists."""
    n = 0
    while has(template.format(n)):
        n += 1
    return n


class SafeTensorLoader:
    """AMX-packed expert checkpoints (what `AMXMoEWrapper.load_weights` reads, utils/amx.py:403-430)."""

    def __init__(self, file_path: str):
        self._index = _Index(file_path)
        # attribute names other code pokes at (utils/amx.py, the reference's own scripts)
        self.tensor_file_map = self._index.where
        self.file_handle_map = self._index.handles
        self.tensor_type_map: dict = {}
        self.tensor_device_map: dict = {}

    def has_tensor(self, name: str) -> bool:
        return name in self._index.where

    def load_tensor(self, key: str, device: str = "cpu") -> torch.Tensor:
        return self._index.get(key).to(device)

    def close_all_handles(self) -> None:
        self._index.handles.clear()
        gc.collect()

    def load_experts(self, base_key: str, device: str = "cpu") -> dict:
        """-> {up, gate, down, up_scale, gate_scale, down_scale}: [numa_id][expert_id] numpy arrays
        (+ the `*_bwd` families when the checkpoint carries them)."""
        stems = {p: f"{base_key}.ffn_{p}_exps" for p in ("up", "gate", "down")}
        n_exp = _run_length(self.has_tensor, stems["up"] + ".{}.numa.0.weight")
        if n_exp == 0:
            raise ValueError(f"No experts found for key {base_key}")
        n_numa = _run_length(self.has_tensor, stems["up"] + ".0.numa.{}.weight")
        families = dict(stems)
        if self.has_tensor(f"{base_key}.ffn_gate_bwd_exps.0.numa.0.weight"):
            families.update({f"{p}_bwd": f"{base_key}.ffn_{p}_bwd_exps" for p in ("up", "gate", "down")})
        out = {}
        for fam, stem in families.items():
            for kind, suffix in (("", "weight"), ("_scale", "scale")):
                out[fam + kind] = [[self.load_tensor(f"{stem}.{e}.numa.{n}.{suffix}", device).numpy() for e in range(n_exp)]
                                   for n in range(n_numa)]
        return out


class _PerExpertLoader(SafeTensorLoader):
    """Shared by the FP8 and BF16 loaders: naming-scheme detection and the per-expert key walk."""

    TAG = "SafeTensorLoader"

    def load_tensor(self, key: str, device: str = "cpu") -> torch.Tensor:
        t = self._index.get(key)
        return t if device == "cpu" else t.to(device)

    def _sample_keys(self) -> List[str]:
        return list(self._index.where.keys())[:1000]

    def _detect_scheme(self, keys: Sequence[str]) -> str:
        for name, (_, (gate, _, _)) in _SCHEMES.items():
            for key in keys:
                if ".experts." not in key or f".{gate}.weight" not in key:
                    continue
                in_bsm, in_mlp = "block_sparse_moe.experts" in key, "mlp.experts" in key
                if (name == "mixtral" and in_bsm) or (name == "deepseek" and in_mlp and "block_sparse_moe" not in key) or (
                        name == "mistral" and ".mlp.experts" not in key and ".block_sparse_moe.experts" not in key):
                    print(f"[{self.TAG}] Detected format: {name}")
                    return name
        print(f"[{self.TAG}] No MoE format detected, defaulting to: deepseek")
        return "deepseek"

    def _get_proj_names(self) -> Tuple[str, str, str]:
        return _SCHEMES[self._detected_format][1]

    def _get_experts_prefix_candidates(self, base_key: str) -> List[str]:
        template = _SCHEMES[self._detected_format][0]
        if getattr(self, "_is_vl_model", False):
            base_key = base_key.replace("model.layers", "model.language_model.layers")
        cands = [template.format(base=base_key)]
        if base_key.startswith("model."):  # Mistral-native checkpoints drop the "model." prefix
            cands.append(template.format(base=base_key[len("model."):]))
        return list(dict.fromkeys(cands))

    def _locate(self, base_key: str) -> Tuple[str, int]:
        gate = self._get_proj_names()[0]
        cands = self._get_experts_prefix_candidates(base_key)
        for prefix in cands:
            n = _run_length(self.has_tensor, prefix + ".{}." + gate + ".weight")
            if n:
                return prefix, n
        raise ValueError(f"No experts found for keys: {cands}")


class FP8SafeTensorLoader(_PerExpertLoader):
    """e4m3 expert weights + `weight_scale_inv` (128x128 blocks) or `weight_scale` (per channel, or block-wise under that
    name — decided by the tensor's shape)."""

    TAG = "FP8SafeTensorLoader"
    MOE_FORMATS = {k: (v[0],) + v[1] for k, v in _SCHEMES.items()}

    def __init__(self, file_path: str, scale_suffix: Optional[str] = None):
        super().__init__(file_path)
        self._scale_suffix = scale_suffix
        self._is_per_channel = scale_suffix == "weight_scale"
        self._is_vl_model = False
        self._detected_format = None
        self._detect_format()

    def _detect_format(self) -> None:
        keys = self._sample_keys()
        self._detected_format = self._detect_scheme(keys)
        if self._scale_suffix is not None:
            kind = "per-channel" if self._is_per_channel else "block-wise"
            print(f"[{self.TAG}] Using explicit scale format: {kind} ({self._scale_suffix})")
            return
        gate = self._get_proj_names()[0]
        for key in keys:
            if f".{gate}.weight_scale_inv" in key:
                self._scale_suffix, self._is_per_channel = "weight_scale_inv", False
                print(f"[{self.TAG}] Detected scale format: block-wise (weight_scale_inv)")
                if key.startswith("model.language_model.") and self._detected_format == "deepseek":
                    self._is_vl_model = True  # Qwen3.5-VL style: model.language_model.layers.N
                    print(f"[{self.TAG}] Detected VL model")
                return
            if f".{gate}.weight_scale" in key:
                s = self.load_tensor(key)
                self._scale_suffix = "weight_scale"
                self._is_per_channel = s.dim() == 1 or (s.dim() == 2 and s.shape[1] == 1)
                kind = "per-channel" if self._is_per_channel else "block-wise"
                print(f"[{self.TAG}] Detected scale format: {kind} (weight_scale)")
                return
        self._scale_suffix, self._is_per_channel = "weight_scale_inv", False
        print(f"[{self.TAG}] No scale format detected, defaulting to: weight_scale_inv")

    def is_per_channel(self) -> bool:
        return self._is_per_channel

    def load_experts(self, base_key: str, device: str = "cpu") -> dict:
        prefix, n = self._locate(base_key)
        out = {k: [None] * n for k in ("gate", "up", "down", "gate_scale", "up_scale", "down_scale")}
        for fam, stem in zip(("gate", "up", "down"), self._get_proj_names()):
            for e in range(n):
                out[fam][e] = self.load_tensor(f"{prefix}.{e}.{stem}.weight", device).contiguous()
                s = self.load_tensor(f"{prefix}.{e}.{stem}.{self._scale_suffix}", device)
                if self._is_per_channel and s.dim() == 2 and s.shape[1] == 1:
                    s = s.squeeze(1)
                out[fam + "_scale"][e] = s.contiguous()
        return out


class BF16SafeTensorLoader(_PerExpertLoader):
    """Unquantised experts: per-expert tensors in any of the three key styles, or the stacked layout
    `mlp.experts.gate_up_proj [E, 2I, H]` + `mlp.experts.down_proj [E, H, I]`."""

    TAG = "BF16SafeTensorLoader"
    MOE_FORMATS = {k: (v[0],) + v[1] for k, v in _SCHEMES.items()}

    def __init__(self, file_path: str):
        super().__init__(file_path)
        self._detected_format = None
        self._detect_format()

    def _detect_format(self) -> None:
        keys = self._sample_keys()
        if any(k.endswith(".mlp.experts.gate_up_proj") for k in keys):
            self._detected_format = "packed"
            print(f"[{self.TAG}] Detected format: packed (Qwen3.5 MoE style)")
            return
        self._detected_format = self._detect_scheme(keys)

    def _resolve_packed_experts_prefix(self, base_key: str) -> str:
        head, _, tail = base_key.partition(".")
        for base in (base_key, f"{head}.language_model.{tail}" if tail else None):
            if base and self.has_tensor(f"{base}.mlp.experts.gate_up_proj"):
                return f"{base}.mlp.experts"
        raise ValueError(f"No packed experts found for base_key '{base_key}'.")

    def load_experts(self, base_key: str, device: str = "cpu") -> dict:
        if self._detected_format == "packed":
            prefix = self._resolve_packed_experts_prefix(base_key)
            gate_up = self.load_tensor(f"{prefix}.gate_up_proj", device)
            down = self.load_tensor(f"{prefix}.down_proj", device)
            half = gate_up.shape[1] // 2
            return {"gate": [gate_up[e, :half].contiguous() for e in range(gate_up.shape[0])],
                    "up": [gate_up[e, half:].contiguous() for e in range(gate_up.shape[0])],
                    "down": [down[e].contiguous() for e in range(down.shape[0])]}
        prefix, n = self._locate(base_key)
        return {fam: [self.load_tensor(f"{prefix}.{e}.{stem}.weight", device).contiguous() for e in range(n)]
                for fam, stem in zip(("gate", "up", "down"), self._get_proj_names())}


class CompressedSafeTensorLoader(SafeTensorLoader):
    """compressed-tensors pack-quantized int4 (Kimi-K2 native): `weight_packed` (int32, 8 nibbles each, or uint8),
    `weight_scale` bf16 [N, K/group], optional `weight_shape` [N, K].  Weights come back as uint8 [N, K/2]."""

    @staticmethod
    def _normalize_rawint4_weight(weight_tensor, scale_tensor, shape_tensor=None, key: str = "weight_packed"):
        if weight_tensor.dtype == torch.int32:  # same bytes, viewed two nibbles per byte
            rows, words = weight_tensor.shape
            weight_tensor = weight_tensor.contiguous().view(torch.uint8).view(rows, words * 4).contiguous()
        elif weight_tensor.dtype == torch.uint8:
            weight_tensor = weight_tensor.contiguous()
        else:
            raise TypeError(f"{key} must be torch.uint8 or torch.int32, got {weight_tensor.dtype}")
        if shape_tensor is None:
            return weight_tensor
        dims = shape_tensor.detach().cpu().tolist()
        if len(dims) != 2:
            raise ValueError(f"{key}.weight_shape must contain [out_features, in_features], got {dims}")
        n, k = int(dims[0]), int(dims[1])
        if n <= 0 or k <= 0 or k % 2 or tuple(weight_tensor.shape) != (n, k // 2):
            return weight_tensor  # shape record not usable for a consistency check
        if scale_tensor.dim() != 2 or scale_tensor.shape[0] != n or scale_tensor.shape[1] <= 0:
            raise ValueError(f"{key} scale shape {tuple(scale_tensor.shape)} is incompatible with weight_shape={dims}")
        if k % int(scale_tensor.shape[1]):
            raise ValueError(f"{key} in_features={k} is not divisible by scale columns={scale_tensor.shape[1]}")
        return weight_tensor

    def load_experts(self, base_key: str, device: str = "cpu") -> dict:
        prefix = f"{base_key}.mlp.experts"
        n = _run_length(self.has_tensor, prefix + ".{}.up_proj.weight_packed")
        if n == 0:
            prefix = f"language_model.{base_key}.mlp.experts"
            n = _run_length(self.has_tensor, prefix + ".{}.up_proj.weight_packed")
            if n == 0:
                raise ValueError(f"No experts found for key {prefix}")
        out = {}
        for fam in ("gate", "up", "down"):
            ws, ss = [], []
            for e in range(n):
                stem = f"{prefix}.{e}.{fam}_proj"
                for need in ("weight_packed", "weight_scale"):
                    if not self.has_tensor(f"{stem}.{need}"):
                        raise KeyError(f"Missing tensor: {stem}.{need}")
                w = self.load_tensor(f"{stem}.weight_packed", device).contiguous()
                s = self.load_tensor(f"{stem}.weight_scale", device).contiguous()
                shape = self.load_tensor(f"{stem}.weight_shape", "cpu") if self.has_tensor(f"{stem}.weight_shape") else None
                ws.append(self._normalize_rawint4_weight(w, s, shape, f"{stem}.weight_packed"))
                ss.append(s)
            out[fam], out[fam + "_scale"] = ws, ss
        return out

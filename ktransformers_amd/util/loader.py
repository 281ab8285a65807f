"""Weight sources with the loader protocol the operators consume (archive/ktransformers/util/custom_loader.py:278-526):
``has_tensor / load_tensor / load_experts / tensor_device_map``.  GGUF parsing and the AMX ``.kt`` layouts are §8(f)
"next" rows; this module carries the in-memory and safetensors sources the tests, the bench and bf16 checkpoints use."""
from __future__ import annotations

import os
import re
from typing import Dict

import torch


class DictLoader:
    """State-dict backed loader (HF names).  Experts are looked up as ``<key>.<e>.{gate,up,down}_proj.weight`` like
    SafeTensorLoader.load_experts (custom_loader.py:96-160) and returned stacked: gate/up [E,I,H], down [E,H,I]."""

    def __init__(self, state: Dict[str, torch.Tensor]):
        self.state = state
        self.tensor_device_map: dict = {}

    def has_tensor(self, name: str) -> bool:
        return name in self.state

    def load_tensor(self, name: str, device: str = "cpu") -> torch.Tensor:
        return self.state[name].to(device)

    def get_expert_count(self, key: str) -> int:
        pat = re.compile(re.escape(key) + r"\.(\d+)\.gate_proj\.weight(_packed)?$")
        ids = [int(m.group(1)) for k in self.state for m in [pat.match(k)] if m]
        return max(ids) + 1 if ids else 0

    def load_experts(self, key: str, device: str = "cpu") -> dict:
        n = self.get_expert_count(key)
        if n == 0:
            raise ValueError(f"Experts {key} not found in the weight source")
        out = {}
        for proj in ("gate", "up", "down"):
            out[proj] = torch.stack([self.state[f"{key}.{e}.{proj}_proj.weight"] for e in range(n)]).to(device)
            # DeepSeek block-fp8 checkpoints carry weight_scale_inv (kt-kernel/python/utils/loader.py:296-508); Kimi-K2
            # compressed-tensors int4 carries weight_packed + weight_scale (:683-777)
            for suffix in ("weight_scale_inv", "weight_scale"):
                k0 = f"{key}.0.{proj}_proj.{suffix}"
                if k0 in self.state:
                    out[proj + "_scale"] = torch.stack(
                        [self.state[f"{key}.{e}.{proj}_proj.{suffix}"] for e in range(n)]).to(device)
            kp = f"{key}.0.{proj}_proj.weight_packed"
            if kp in self.state:
                out[proj] = torch.stack([self.state[f"{key}.{e}.{proj}_proj.weight_packed"] for e in range(n)]).to(device)
        return out


class SafeTensorLoader(DictLoader):
    """Directory of *.safetensors shards, loaded lazily per tensor."""

    def __init__(self, path: str):
        from safetensors import safe_open

        self.tensor_device_map = {}
        self._files = {}
        self._index: Dict[str, str] = {}
        files = [path] if os.path.isfile(path) else sorted(
            os.path.join(path, f) for f in os.listdir(path) if f.endswith(".safetensors"))
        for f in files:
            h = safe_open(f, framework="pt", device="cpu")
            self._files[f] = h
            for k in h.keys():
                self._index[k] = f
        self.state = _LazyState(self)


class _LazyState(dict):
    def __init__(self, owner: SafeTensorLoader):
        super().__init__()
        self._o = owner

    def __contains__(self, k):
        return k in self._o._index

    def __getitem__(self, k):
        return self._o._files[self._o._index[k]].get_tensor(k)

    def __iter__(self):
        return iter(self._o._index)

    def keys(self):
        return self._o._index.keys()

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

    def _expert_stems(self, key: str):
        """Per-expert tensor name stems: HF `<key>.<e>.{gate,up,down}_proj` or the short `<key>.<e>.{gate,up,down}` some
        checkpoints use (custom_loader.py:149-173)."""
        short = f"{key}.0.up.weight" in self.state
        return {p: (p if short else p + "_proj") for p in ("gate", "up", "down")}

    def get_expert_count(self, key: str) -> int:
        stem = self._expert_stems(key)["up"]
        n = 0
        while any(f"{key}.{n}.{stem}.{suffix}" in self.state for suffix in ("weight", "weight_packed")):
            n += 1
        return n

    def load_experts(self, key: str, device: str = "cpu") -> dict:
        """Stacked per-expert tensors: gate/up [E, I, H(/2)], down [E, H, I(/2)] (+ `<proj>_scale` for block-fp8
        `weight_scale_inv`, kt-kernel/python/utils/loader.py:296-508, and compressed-tensors int4 `weight_packed` +
        `weight_scale`, :683-777).  The reference returns numpy views + ggml type ids for its CPU backend
        (custom_loader.py:113-223); the HIP operator takes the tensors as they are."""
        n = self.get_expert_count(key)
        if n == 0:
            raise ValueError(f"No experts found for key {key}")
        out = {}
        for proj, stem in self._expert_stems(key).items():
            packed = f"{key}.0.{stem}.weight_packed" in self.state
            name = "weight_packed" if packed else "weight"
            out[proj] = torch.stack([self.state[f"{key}.{e}.{stem}.{name}"] for e in range(n)]).to(device)
            for suffix in ("weight_scale_inv", "weight_scale"):
                if f"{key}.0.{stem}.{suffix}" in self.state:
                    out[proj + "_scale"] = torch.stack([self.state[f"{key}.{e}.{stem}.{suffix}"] for e in range(n)]).to(device)
        return out

    def load_gate(self, key: str, device: str = "cpu") -> dict:
        """{'weight', 'e_score_correction_bias'} of a router, None where absent (custom_loader.py:225-250)."""
        return {k: (self.state[f"{key}.{k}"].to(device) if f"{key}.{k}" in self.state else None)
                for k in ("weight", "e_score_correction_bias")}


class SafeTensorLoader(DictLoader):
    """Directory of *.safetensors shards, loaded lazily per tensor (archive/ktransformers/util/custom_loader.py:44-275).

    Also reads the reference's HYBRID checkpoints — the output of archive/merge_tensors/merge_safetensor_gguf.py (DeepSeek-V3 /
    R1 with fp8 linears and GGUF-quantised experts, the format its DeepSeek-V3-Chat-fp8-linear-ggml-experts.yaml rule file
    expects; BASELINE config C5).  That script stores EVERY tensor under its GGUF name (`translate_name_to_gguf`:
    model.layers.3.self_attn.q_a_proj.weight -> blk.3.attn_q_a.weight, ...), the linears as e4m3 `weight` +
    `weight_scale_inv`, and per MoE layer the experts as the raw ggml blocks of all experts in one uint8 tensor
    `blk.N.ffn_{gate,up,down}_exps.weight` next to an int `blk.N.ffn_*_exps.ggml_type`.  Like the reference's loader, every
    lookup tries the name as given and then its GGUF translation (custom_loader.py:96-111,272-275), and load_experts /
    load_gate take the "legacy hybrid" branch when the translated expert tensor exists (:122-146, :236-243)."""

    def __init__(self, path: str):
        from safetensors import safe_open

        self.tensor_device_map = {}
        self._files = {}
        self._index: Dict[str, str] = {}
        if not os.path.exists(path):
            raise FileNotFoundError(f"Path not found: {path}")
        files = [path] if os.path.isfile(path) else sorted(
            os.path.join(root, f) for root, _, fs in os.walk(path) for f in sorted(fs) if f.endswith(".safetensors"))
        for f in files:
            h = safe_open(f, framework="pt", device="cpu")
            self._files[f] = h
            for k in h.keys():
                self._index[k] = f
        self.state = _LazyState(self)

    @staticmethod
    def _gguf_name(name: str) -> str:
        from ktransformers_amd.util.gguf_loader import translate_name_to_gguf
        return translate_name_to_gguf(name)

    def _resolve(self, name: str):
        t = self._gguf_name(name)
        return t if t in self._index else (name if name in self._index else None)

    def has_tensor(self, name: str) -> bool:
        return self._resolve(name) is not None

    def load_tensor(self, name: str, device: str = "cpu") -> torch.Tensor:
        k = self._resolve(name)
        if k is None:
            raise KeyError(f"Key {name} not found in Safetensor files")
        return self.state[k].to(device)

    def is_hybrid_experts(self, key: str) -> bool:
        return self._gguf_name(key) + ".ffn_gate_exps.weight" in self._index

    def get_expert_count(self, key: str) -> int:
        if self.is_hybrid_experts(key):
            raise ValueError("a hybrid checkpoint stores all experts of a layer in one ggml tensor: take the count from the config")
        return DictLoader.get_expert_count(self, key)

    def load_experts(self, key: str, device: str = "cpu") -> dict:
        if not self.is_hybrid_experts(key):
            return DictLoader.load_experts(self, key, device)
        base = self._gguf_name(key)                     # "blk.N": raw ggml blocks + type ids, as the reference returns them
        out = {}
        for proj in ("gate", "up", "down"):
            out[proj] = self.state[f"{base}.ffn_{proj}_exps.weight"].to(device)
            out[proj + "_type"] = int(self.state[f"{base}.ffn_{proj}_exps.ggml_type"].item())
        return out

    def load_gate(self, key: str, device: str = "cpu") -> dict:
        res = {"weight": None, "e_score_correction_bias": None}
        for k in res:          # both branches of the reference's load_gate reduce to has_tensor / load_tensor with translation
            name = self._resolve(f"{key}.{k}")
            if name is not None:
                res[k] = self.state[name].to(device)
        return res


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

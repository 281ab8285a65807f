"""The three inference backends behind `KTMoEWrapper`, named after the reference's (kt-kernel/python/utils/{amx,llamafile}.py).
All of them end in one ktx_moe_t handle (include/ktx_moe.h); they differ in the checkpoint reader and the load call."""
from __future__ import annotations

import glob
import os
from typing import Optional

import torch

from .. import _native
from ..util.gguf_loader import GGUFLoader
from .experts_base import BaseMoEWrapper
from .utils.loader import BF16SafeTensorLoader, CompressedSafeTensorLoader, FP8SafeTensorLoader, SafeTensorLoader

_LAYER_PREFIXES = ("model.layers.{L}", "language_model.model.layers.{L}", "model.language_model.layers.{L}")  # utils/amx.py:730-734


class AMXMoEWrapper(BaseMoEWrapper):
    """AMXINT4 / AMXINT8 (utils/amx.py:248-546).  Weights: online quantisation from bf16 tensors — bit-identical to the
    reference's BufferB::from_mat — or an AMX-packed safetensors folder written by the reference's converter."""

    FORMAT = {"AMXINT4": "AMXINT4", "AMXINT8": "AMXINT8"}
    _safetensor_loader_instance: Optional[SafeTensorLoader] = None

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.load_merged_weight = bool(glob.glob(os.path.join(self.weight_path, "*.safetensors"))) if self.weight_path else False

    def load_weights_from_tensors(self, gate_proj, up_proj, down_proj, physical_to_logical_map_cpu=None):
        """gate/up [E, I, H], down [E, H, I] (bf16 or fp16, any device).  The reference's online branch indexes both source
        and destination by the LOGICAL id (operators/amx/moe.hpp:357-371), i.e. the map does not permute; neither do we."""
        ws = [t.to(device=self.device, dtype=torch.bfloat16).contiguous() for t in (gate_proj, up_proj, down_proj)]
        self.moe = self._new_handle()
        self.moe.load_bf16(*ws)

    def load_weights(self, physical_to_logical_map_cpu=None):
        if not self.load_merged_weight:
            raise FileNotFoundError(f"no *.safetensors under weight_path={self.weight_path!r}: convert the checkpoint with the "
                                    "reference's tool or call load_weights_from_tensors (online quantisation)")
        if AMXMoEWrapper._safetensor_loader_instance is None:
            AMXMoEWrapper._safetensor_loader_instance = SafeTensorLoader(self.weight_path)
        from .utils.amx_packed import unpack_expert
        w = AMXMoEWrapper._safetensor_loader_instance.load_experts(f"blk.{self.layer_idx}")
        order = self._logical_order(physical_to_logical_map_cpu, self.num_experts)
        bits = 4 if self.method == "AMXINT4" else 8
        n_parts = len(w["gate"])
        if n_parts > 1:
            self._load_tp_parts(w, order, bits, n_parts)
            return
        self.moe = self._new_handle()
        for which, fam, (n, k), split in ((_native.MAT_GATE, "gate", (self.moe_intermediate_size, self.hidden_size), "n"),
                                         (_native.MAT_UP, "up", (self.moe_intermediate_size, self.hidden_size), "n"),
                                         (_native.MAT_DOWN, "down", (self.hidden_size, self.moe_intermediate_size), "k")):
            for slot, logical in enumerate(order):
                q, s = unpack_expert([part[logical] for part in w[fam]], [part[logical] for part in w[fam + "_scale"]], n, k, bits, split)
                self.moe.load_quantized(slot, which, q, s)


    def _load_tp_parts(self, w: dict, order, bits: int, n_parts: int) -> None:
        """NUMA-sharded checkpoint (kt-kernel/python/utils/loader.py:179-290; operators/amx/moe.hpp:103-147): part p holds
        gate / up rows and down COLUMNS [p * I/P, (p + 1) * I/P) of every expert, quantised on its own — down has one scale per
        (row, part).  One handle of width I / P per part; the forward adds their fp32 outputs in part order (the reference's
        merge_results), which reproduces what the reference computes from this very checkpoint."""
        I, H = self.moe_intermediate_size, self.hidden_size
        if I % n_parts:
            raise ValueError(f"moe_intermediate_size {I} does not split over the checkpoint's {n_parts} NUMA parts")
        Ip = I // n_parts
        from .utils.amx_packed import unpack_expert
        parts = [self._new_handle(intermediate_size=Ip) for _ in range(n_parts)]
        for p, h in enumerate(parts):
            for which, fam, (n, k) in ((_native.MAT_GATE, "gate", (Ip, H)), (_native.MAT_UP, "up", (Ip, H)), (_native.MAT_DOWN, "down", (H, Ip))):
                for slot, logical in enumerate(order):
                    q, s = unpack_expert([w[fam][p][logical]], [w[fam + "_scale"][p][logical]], n, k, bits, "n")
                    h.load_quantized(slot, which, q, s)
        self.tp_parts, self.moe = parts, parts[0]
        self._tp_scratch = torch.empty((n_parts, max(1, int(self.chunked_prefill_size)), H), dtype=torch.float32, device=self.device)


class NativeMoEWrapper(BaseMoEWrapper):
    """RAWINT4 / FP8 / FP8_PERCHANNEL / BF16 straight from the model's own safetensors (utils/amx.py:549-959)."""

    FORMAT = {"RAWINT4": "RAWINT4", "FP8": "FP8", "FP8_PERCHANNEL": "FP8_PERCHANNEL", "BF16": "BF16"}
    _native_loader_instance = None

    def __init__(self, *args, **kw):
        method = kw.get("method", args[12] if len(args) > 12 else "RAWINT4")
        weight_path = kw.get("weight_path", args[8] if len(args) > 8 else None)
        if NativeMoEWrapper._native_loader_instance is None:  # one loader for all layers, opened with the first wrapper
            NativeMoEWrapper._native_loader_instance = self._create_loader(method, weight_path)
        super().__init__(*args, **kw)
        self.loader = NativeMoEWrapper._native_loader_instance

    @staticmethod
    def _create_loader(method: str, weight_path: str):
        if method == "RAWINT4":
            return CompressedSafeTensorLoader(weight_path)
        if method == "FP8":
            return FP8SafeTensorLoader(weight_path)
        if method == "FP8_PERCHANNEL":     # utils/amx.py:674-675: the same loader reading `weight_scale` (one scale per row)
            return FP8SafeTensorLoader(weight_path, scale_suffix="weight_scale")
        if method == "BF16":
            return BF16SafeTensorLoader(weight_path)
        raise NotImplementedError(f"Unsupported method for NativeMoEWrapper: {method}")

    @staticmethod
    def _release_loader(layer_idx: int = -1) -> None:
        if NativeMoEWrapper._native_loader_instance is not None:
            NativeMoEWrapper._native_loader_instance.close_all_handles()
            NativeMoEWrapper._native_loader_instance = None

    force_release_loader = _release_loader

    def load_weights_from_tensors(self, gate_proj, up_proj, down_proj, physical_to_logical_map_cpu=None):
        raise NotImplementedError("RAWINT4 wrapper expects pre-quantized safetensor weights.")  # utils/amx.py:704-711

    def load_weights(self, physical_to_logical_map_cpu=None):
        if NativeMoEWrapper._native_loader_instance is None:
            NativeMoEWrapper._native_loader_instance = self._create_loader(self.method, self.weight_path)
        self.loader = NativeMoEWrapper._native_loader_instance
        weights = None
        for tpl in _LAYER_PREFIXES:
            try:
                weights = self.loader.load_experts(tpl.format(L=self.layer_idx))
                break
            except (ValueError, KeyError):
                continue
        if weights is None:
            raise ValueError(f"No experts found for layer {self.layer_idx} under any prefix: "
                             f"{[t.format(L=self.layer_idx) for t in _LAYER_PREFIXES]}")
        if len(weights["gate"]) != self.num_experts:
            raise ValueError(f"checkpoint holds {len(weights['gate'])} experts for layer {self.layer_idx}, expected {self.num_experts}")
        order = self._logical_order(physical_to_logical_map_cpu, self.num_experts)

        def stack(name, dtype=None):
            ts = [weights[name][i] for i in order]
            ts = [t if dtype is None or t.dtype == dtype else t.to(dtype) for t in ts]
            return torch.stack(ts).to(self.device).contiguous()

        if self.method == "BF16":
            self.moe = self._new_handle()
            self.moe.load_bf16(stack("gate", torch.bfloat16), stack("up", torch.bfloat16), stack("down", torch.bfloat16))
        elif self.method == "FP8_PERCHANNEL":   # utils/amx.py:774-779, 895-897: fp32 scale per output row, per_channel = True
            self.moe = self._new_handle()
            self.moe.load_fp8_perchannel(stack("gate").view(torch.uint8), stack("up").view(torch.uint8), stack("down").view(torch.uint8),
                                         stack("gate_scale", torch.float32), stack("up_scale", torch.float32),
                                         stack("down_scale", torch.float32))
        elif self.method == "FP8":
            if getattr(self.loader, "is_per_channel", lambda: False)():
                raise ValueError("this checkpoint carries per-channel FP8 scales: load it with method='FP8_PERCHANNEL'")
            self.moe = self._new_handle(group_size=128)
            self.moe.load_fp8(stack("gate").view(torch.uint8), stack("up").view(torch.uint8), stack("down").view(torch.uint8),
                              stack("gate_scale", torch.float32), stack("up_scale", torch.float32), stack("down_scale", torch.float32))
        else:  # RAWINT4
            if weights["gate_scale"][0].dtype != torch.bfloat16:
                raise AssertionError("Expected bf16 scales for RAWINT4")
            group = self.hidden_size // weights["gate_scale"][0].shape[1]  # utils/amx.py:843-847
            self.moe = self._new_handle(group_size=group)
            self.moe.load_rawint4(stack("gate"), stack("up"), stack("down"), stack("gate_scale"), stack("up_scale"), stack("down_scale"))
        NativeMoEWrapper._release_loader(self.layer_idx)


class LlamafileMoEWrapper(BaseMoEWrapper):
    """GGUF experts (utils/llamafile.py:21-227): `blk.L.ffn_{gate,up,down}_exps.weight` raw blocks, Q4_K / Q6_K / IQ1_S."""

    FORMAT = {"LLAMAFILE": "GGUF"}
    _gguf_loader_instance: Optional[GGUFLoader] = None

    def __init__(self, *args, **kw):
        weight_path = kw.get("weight_path", args[8] if len(args) > 8 else None)
        inter = kw.get("moe_intermediate_size", args[4] if len(args) > 4 else 0)
        if not weight_path or not os.path.exists(weight_path):
            raise FileNotFoundError(f"GGUF weight path not found: {weight_path}")
        if inter % 256:
            raise ValueError(f"intermediate_size ({inter}) must be divisible by QK_K (256) for Llamafile backend")
        super().__init__(*args, **kw)
        if LlamafileMoEWrapper._gguf_loader_instance is None:
            LlamafileMoEWrapper._gguf_loader_instance = GGUFLoader(weight_path)
        self.gguf_loader = LlamafileMoEWrapper._gguf_loader_instance

    def load_weights_from_tensors(self, gate_proj, up_proj, down_proj, physical_to_logical_map_cpu=None):
        raise NotImplementedError("Llamafile backend does not support online quantization (load_weights_from_tensors).\n"
                                  "Please use pre-quantized GGUF weights and call load_weights() instead.")

    def load_weights(self, physical_to_logical_map_cpu: Optional[torch.Tensor] = None):
        order = self._logical_order(physical_to_logical_map_cpu, self.num_experts)
        mats, types = [], []
        for fam, (n, k) in (("gate", (self.moe_intermediate_size, self.hidden_size)), ("up", (self.moe_intermediate_size, self.hidden_size)),
                            ("down", (self.hidden_size, self.moe_intermediate_size))):
            name = f"blk.{self.layer_idx}.ffn_{fam}_exps.weight"
            ty = self.gguf_loader.get_ggml_type(name)
            if ty not in _native.GGML_BLOCK_BYTES:
                raise NotImplementedError(f"{name}: ggml type {ty} is not supported (Q2_K=10, Q3_K=11, Q4_K=12, Q5_K=13, Q6_K=14, IQ1_S=19, IQ4_XS=23; "
                                          "Q4_0=2, Q5_0=6, Q8_0=8)")
            raw = torch.from_numpy(self.gguf_loader.get_mmap_tensor(name).copy()).view(torch.uint8)
            raw = raw.reshape(self.num_experts, n, k // _native.ggml_block_elems(ty) * _native.GGML_BLOCK_BYTES[ty])
            if order != list(range(self.num_experts)):
                raw = raw[torch.tensor(order)]
            mats.append(raw.to(self.device).contiguous())
            types.append(ty)
        self.moe = self._new_handle()
        self.moe.load_gguf(*mats, *types)

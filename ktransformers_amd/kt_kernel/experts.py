"""`KTMoEWrapper` — the factory external code (SGLang) instantiates once per MoE layer, with the reference's signature
(kt-kernel/python/experts.py:72-313).  Methods whose arithmetic this library implements map to a backend; the rest of the
reference's method list is rejected with the reason instead of being silently emulated."""
from __future__ import annotations

from typing import List, Optional

import torch

from .backends import AMXMoEWrapper, LlamafileMoEWrapper, NativeMoEWrapper
from .experts_base import BaseMoEWrapper

# the reference's full lists (experts.py:33-68): used only to tell "unknown" from "known but not built here"
INFERENCE_METHODS = frozenset(["AMXINT4", "AMXINT8", "RAWINT4", "FP8", "BF16", "FP8_PERCHANNEL", "GPTQ_INT4", "SYCL_GPTQ_INT4",
                               "MXFP4", "MXFP8", "LLAMAFILE", "MOE_INT4", "MOE_INT8"])
SFT_METHODS = frozenset(["AMXBF16_SFT", "AMXFP8_SFT", "INT8_SFT", "AMXINT8_SFT", "AMXINT4_SFT", "AMXINT4_1_SFT", "AMXINT4_KGroup_SFT",
                         "AMXINT4_1KGroup_SFT", "AMXBF16_SFT_SkipLoRA", "AMXINT8_SFT_SkipLoRA", "AMXINT4_SFT_SkipLoRA",
                         "AMXINT4_1_SFT_SkipLoRA", "AMXINT4_KGroup_SFT_SkipLoRA", "AMXINT4_1KGroup_SFT_SkipLoRA"])
_BACKENDS = {"AMXINT4": AMXMoEWrapper, "AMXINT8": AMXMoEWrapper, "RAWINT4": NativeMoEWrapper, "FP8": NativeMoEWrapper,
             "FP8_PERCHANNEL": NativeMoEWrapper, "BF16": NativeMoEWrapper, "LLAMAFILE": LlamafileMoEWrapper}
SUPPORTED_METHODS = frozenset(_BACKENDS)


class KTMoEWrapper:
    """wrapper = KTMoEWrapper(layer_idx, num_experts, num_experts_per_tok, hidden_size, moe_intermediate_size,
    gpu_experts_mask, cpuinfer_threads, threadpool_count, weight_path, chunked_prefill_size, method="AMXINT4", ...)
    -> a BaseMoEWrapper with load_weights / load_weights_from_tensors / submit_forward / sync_forward / forward."""

    def __new__(cls, layer_idx: int, num_experts: int, num_experts_per_tok: int, hidden_size: int, moe_intermediate_size: int,
                gpu_experts_mask: Optional[torch.Tensor], cpuinfer_threads: int, threadpool_count: int, weight_path: str,
                chunked_prefill_size: int, cpu_save: bool = False, max_deferred_experts_per_token: Optional[int] = None,
                method: str = "AMXINT4", numa_nodes: Optional[List[int]] = None, mode: str = "inference", num_gpu_experts: int = 0,
                lora_rank: int = 16, lora_alpha: float = 32.0, lora_dropout: float = 0.0, max_cache_depth: int = 1,
                group_size: int = 128, zero_point: bool = True, full_weight_grad: bool = False, swiglu_limit: float = 0.0,
                swiglu_alpha: float = 0.0, device: Optional[torch.device] = None) -> BaseMoEWrapper:
        if mode not in ("inference", "sft"):
            raise ValueError(f"Unknown mode: '{mode}'. Supported modes: 'inference', 'sft'")
        if mode == "sft":
            if method not in SFT_METHODS:
                raise ValueError(f"Method '{method}' not supported for SFT mode. Supported methods: {sorted(SFT_METHODS)}")
            raise NotImplementedError("mode='sft' (LoRA fine-tuning of the experts) is outside this library's scope: inference only")
        if method not in INFERENCE_METHODS:
            raise ValueError(f"Method '{method}' not supported for inference mode. Supported methods: {sorted(INFERENCE_METHODS)}")
        if swiglu_limit != 0.0 or swiglu_alpha != 0.0:
            raise ValueError(f"swiglu_limit={swiglu_limit} / swiglu_alpha={swiglu_alpha} are only supported on method='MXFP4'/'MXFP8', "
                             f"got method={method!r}")
        if method not in _BACKENDS:
            raise NotImplementedError(f"method {method!r} is part of the reference's list but has no HIP implementation here; "
                                      f"available: {sorted(SUPPORTED_METHODS)}")
        return _BACKENDS[method](layer_idx=layer_idx, num_experts=num_experts, num_experts_per_tok=num_experts_per_tok,
                                 hidden_size=hidden_size, moe_intermediate_size=moe_intermediate_size, gpu_experts_mask=gpu_experts_mask,
                                 cpuinfer_threads=cpuinfer_threads, threadpool_count=threadpool_count, weight_path=weight_path,
                                 chunked_prefill_size=chunked_prefill_size, cpu_save=cpu_save,
                                 max_deferred_experts_per_token=max_deferred_experts_per_token, method=method, numa_nodes=numa_nodes,
                                 device=device)

    set_capture_batch_sizes = staticmethod(BaseMoEWrapper.set_capture_batch_sizes)
    get_capture_batch_sizes = staticmethod(BaseMoEWrapper.get_capture_batch_sizes)
    clear_buffer_cache = staticmethod(BaseMoEWrapper.clear_buffer_cache)

"""HIP mirror of the reference's live expert-wrapper base (kt-kernel/python/experts_base.py): same class, method and
argument names, same deferred-expert protocol — but every buffer lives in HBM and "submit" is a kernel enqueue.

  reference (CPU experts behind a GPU model)                     here (experts resident on the MI355X)
  -----------------------------------------------------------    ----------------------------------------------------------
  KExpertsCPUBuffer: pinned input/ids/weights/output staging,    KExpertsDeviceBuffer: double-buffered bf16 OUTPUT tensors per
    double-buffered, cached per captured batch size (:75-142)      batch size (inputs are read in place: stream order protects them)
  submit_forward: D2H copies + cpu_infer.submit_with_cuda_stream  submit_forward: ktx_moe_forward enqueued on `cuda_stream`
    (moe.forward_task(bsz*, k, ids*, w*, x*, y*, incremental))      (same argument list, include/ktx_moe.h) (:377-455)
  sync_forward: cudaLaunchHostFunc barrier + H2D copy (:457-483)  sync_forward: returns the slot's output tensor; later work on
                                                                   `cuda_stream` is ordered after the kernels — no host sync
  gpu_experts_mask (pinned uint8 shared with C++, :281-291)       ktx_moe_set_expert_mask: masked experts contribute nothing

Deferred experts (max_deferred_experts_per_token > 0, :347-375, 412-455): the `protected_k` highest-scored experts of a token
run now into output[slot]; the rest run into output[next slot] and are folded in by the NEXT layer's forward through
`incremental` — reproduced exactly, including the layer_idx % 2 slot rotation and the class-level pending table."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, List, Optional, Tuple

import torch

from .. import _native


def generate_gpu_experts_masks(activation_freq: torch.Tensor, num_gpu_experts: int) -> torch.Tensor:
    """Bool [num_layers, num_experts] on CPU marking the `num_gpu_experts` most frequently activated (layer, expert) pairs
    (kt-kernel/python/experts_base.py:21-72).  "GPU" keeps the reference's meaning: experts the caller runs itself and this
    wrapper must skip."""
    layers, per_layer = activation_freq.shape
    total = layers * per_layer
    n = max(0, min(int(num_gpu_experts), total))
    mask = torch.zeros(total, dtype=torch.bool, device="cpu")
    if n:
        mask[torch.topk(activation_freq.reshape(-1).to("cpu"), k=n, largest=True, sorted=False).indices] = True
    return mask.view(layers, per_layer)


class KExpertsDeviceBuffer:
    """Output double-buffer per batch size; buffers of sizes listed in `capture_bs` are kept for HIP-graph replay, one other
    size is cached as the 'temp' entry (same policy as KExpertsCPUBuffer.get_buffer)."""

    capture_bs: List[int] = []
    capture_buffers: Dict[tuple, tuple] = {}
    temp_key: Optional[tuple] = None
    temp_buffer: tuple = ()
    buffer_depth: int = 2

    @classmethod
    def get_buffer(cls, hidden_states: torch.Tensor, num_experts_per_tok: int) -> tuple:
        bsz, hidden = hidden_states.shape
        key = (bsz, hidden, hidden_states.device)
        if key in cls.capture_buffers:
            return cls.capture_buffers[key]
        if key == cls.temp_key:
            return cls.temp_buffer
        buf = tuple(torch.zeros((bsz, hidden), device=hidden_states.device, dtype=torch.bfloat16) for _ in range(cls.buffer_depth))
        if bsz in cls.capture_bs:
            cls.capture_buffers[key] = buf
        cls.temp_key, cls.temp_buffer = key, buf
        return buf

    @classmethod
    def clear(cls) -> None:
        cls.capture_buffers.clear()
        cls.temp_key, cls.temp_buffer = None, ()


class BaseMoEWrapper(ABC):
    """One MoE layer's routed experts.  Subclasses differ in where the weights come from (`load_weights*`)."""

    _layer_has_pending_deferred: Dict[int, bool] = {}
    FORMAT: Dict[str, str] = {}  # reference method name -> ktx_moe_format name (include/ktx_moe.h)

    def __init__(self, layer_idx: int, num_experts: int, num_experts_per_tok: int, hidden_size: int, moe_intermediate_size: int,
                 gpu_experts_mask: Optional[torch.Tensor], cpuinfer_threads: int, threadpool_count: int, weight_path: str,
                 chunked_prefill_size: int, cpu_save: bool = False, max_deferred_experts_per_token: Optional[int] = None,
                 method: str = "AMXINT4", numa_nodes: Optional[List[int]] = None, swiglu_limit: float = 0.0,
                 device: Optional[torch.device] = None):
        if swiglu_limit != 0.0:
            raise ValueError(f"swiglu_limit={swiglu_limit} is only defined for the MXFP4/MXFP8 methods, which this build lacks")
        self.layer_idx = layer_idx
        self.num_experts = num_experts
        self.num_experts_per_tok = num_experts_per_tok
        self.hidden_size = hidden_size
        self.moe_intermediate_size = moe_intermediate_size
        self.gpu_experts_mask = torch.zeros(num_experts, dtype=torch.bool, device="cpu")
        if gpu_experts_mask is not None:
            self.gpu_experts_mask.copy_(gpu_experts_mask)
        self.num_gpu_experts = int(self.gpu_experts_mask.sum().item())
        self.weight_path = weight_path
        self.chunked_prefill_size = chunked_prefill_size
        self.cpu_save = cpu_save
        self.max_deferred_experts_per_token = int(max_deferred_experts_per_token) if max_deferred_experts_per_token is not None else 0
        self.method = method
        self.swiglu_limit = 0.0
        # cpuinfer_threads / threadpool_count / numa_nodes size the reference's CPU worker pool; there is none here
        self.cpuinfer_threads, self.threadpool_count, self.numa_nodes = cpuinfer_threads, threadpool_count, numa_nodes
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("KTMoEWrapper needs a HIP device: the experts are resident on the GPU, there is no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        BaseMoEWrapper._layer_has_pending_deferred[self.layer_idx] = False
        self.moe: Optional[_native.MoEHandle] = None
        # a checkpoint converted with threadpool_count = P holds every expert as P NUMA parts of width I / P with their own
        # scales; TP_MOE then runs P complete MoEs and adds their fp32 outputs (operators/moe-tp.hpp:201-216).  Same here: one
        # handle per part, fp32 partials, ktx_moe_merge_partials.  Empty for single-part weights.
        self.tp_parts: List[_native.MoEHandle] = []
        self._tp_scratch: Optional[torch.Tensor] = None

    # ---- weights -----------------------------------------------------------------------------------------------
    def _new_handle(self, group_size: int = 0, intermediate_size: Optional[int] = None) -> _native.MoEHandle:
        h = _native.MoEHandle(self.num_experts, self.num_experts_per_tok, self.hidden_size,
                              intermediate_size or self.moe_intermediate_size,
                              max_len=max(1, int(self.chunked_prefill_size)), method=self.FORMAT[self.method], device=self.device,
                              group_size=group_size)
        h.set_expert_mask(self.gpu_experts_mask.numpy().astype("uint8"))
        return h

    @staticmethod
    def _logical_order(physical_to_logical_map_cpu: Optional[torch.Tensor], n: int) -> List[int]:
        """Physical slot i is filled from logical expert map[i] (operators/amx/fp8-moe.hpp:194-208, common.hpp:48)."""
        if physical_to_logical_map_cpu is None:
            return list(range(n))
        order = [int(v) for v in physical_to_logical_map_cpu.reshape(-1).tolist()]
        if len(order) != n or any(v < 0 or v >= n for v in order):
            raise ValueError(f"physical_to_logical_map must hold {n} expert ids in [0, {n})")
        return order

    @abstractmethod
    def load_weights_from_tensors(self, gate_proj: torch.Tensor, up_proj: torch.Tensor, down_proj: torch.Tensor,
                                  physical_to_logical_map_cpu: torch.Tensor):
        ...

    @abstractmethod
    def load_weights(self, physical_to_logical_map_cpu: torch.Tensor):
        ...

    # ---- forward -----------------------------------------------------------------------------------------------
    def select_deferred_experts(self, expert_ids: torch.Tensor, expert_scores: torch.Tensor,
                                protected_k: int) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """(immediate_ids, deferred_ids): an expert is immediate for EVERY token of the batch as soon as it is among the
        `protected_k` best-scored experts of ANY token; the other slots carry -1 (experts_base.py:347-375)."""
        topk = expert_ids.shape[1]
        protected_k = max(0, min(int(protected_k), topk))
        if protected_k == 0:
            return torch.full_like(expert_ids, -1), expert_ids.clone()
        best = torch.topk(expert_scores, k=protected_k, dim=-1, largest=True, sorted=False).indices
        flag = torch.zeros(self.num_experts, dtype=torch.int32, device=expert_ids.device)
        flag.scatter_(0, torch.gather(expert_ids, -1, best).reshape(-1), 1)
        protected = torch.gather(flag, 0, expert_ids.reshape(-1)).ne(0).view_as(expert_ids)
        return expert_ids.masked_fill(~protected, -1), expert_ids.masked_fill(protected, -1)

    def _enqueue(self, ids: torch.Tensor, weights: torch.Tensor, x: torch.Tensor, out: torch.Tensor, incremental: bool, stream) -> None:
        if self.tp_parts:
            st, T, buf = _stream_handle(stream, x.device), x.shape[0], self._tp_scratch
            for i, part in enumerate(self.tp_parts):
                _native.check(_native.lib.ktx_moe_forward_ex(part._h, None, T, ids.shape[1], ids.data_ptr(), weights.data_ptr(),
                                                             x.data_ptr(), buf[i].data_ptr(), 2, st))       # KTX_FWD_PARTIAL_F32
            _native.check(_native.lib.ktx_moe_merge_partials(len(self.tp_parts), T, self.hidden_size, buf.data_ptr(), buf.stride(0),
                                                             out.data_ptr(), 1 if incremental else 0, None, st))
            return
        _native.check(_native.lib.ktx_moe_forward(self.moe._h, None, x.shape[0], ids.shape[1], ids.data_ptr(), weights.data_ptr(),
                                                  x.data_ptr(), out.data_ptr(), 1 if incremental else 0, _stream_handle(stream, x.device)))

    def submit_forward(self, hidden_states: torch.Tensor, topk_ids: torch.Tensor, topk_weights: torch.Tensor, cuda_stream) -> None:
        """Enqueue this layer's routed experts on `cuda_stream` (a raw stream handle, a torch stream, or None = current)."""
        if self.moe is None:
            raise RuntimeError("submit_forward before load_weights")
        x = hidden_states.reshape(-1, hidden_states.shape[-1])
        if x.dtype != torch.bfloat16 or x.device != self.device:
            raise ValueError(f"hidden_states must be bf16 on {self.device} (the reference's hidden_type is bf16, experts.py:200)")
        x = x.contiguous()
        out = KExpertsDeviceBuffer.get_buffer(x, self.num_experts_per_tok)
        slot = self.layer_idx % KExpertsDeviceBuffer.buffer_depth
        ids = topk_ids.to(torch.long).contiguous()
        w = topk_weights.to(torch.float32).contiguous()
        deferred = None
        if self.max_deferred_experts_per_token > 0:
            ids, deferred = self.select_deferred_experts(ids, w, self.num_experts_per_tok - self.max_deferred_experts_per_token)
        self._keep = (x, ids, w, deferred)  # inputs must outlive the enqueued kernels when the caller drops them
        incremental = BaseMoEWrapper._layer_has_pending_deferred.get(self.layer_idx - 1, False)
        self._enqueue(ids, w, x, out[slot], incremental, cuda_stream)
        BaseMoEWrapper._layer_has_pending_deferred[self.layer_idx] = False
        if deferred is not None:
            self._enqueue(deferred, w, x, out[(slot + 1) % KExpertsDeviceBuffer.buffer_depth], False, cuda_stream)
            BaseMoEWrapper._layer_has_pending_deferred[self.layer_idx] = True

    def sync_forward(self, hidden_states: torch.Tensor, cuda_stream) -> torch.Tensor:
        """The slot's output tensor, valid for work enqueued on `cuda_stream` after this call (no host synchronisation)."""
        x = hidden_states.reshape(-1, hidden_states.shape[-1])
        return KExpertsDeviceBuffer.get_buffer(x, self.num_experts_per_tok)[self.layer_idx % KExpertsDeviceBuffer.buffer_depth]

    def forward(self, hidden_states: torch.Tensor, topk_ids: torch.Tensor, topk_weights: torch.Tensor, cuda_stream) -> torch.Tensor:
        self.submit_forward(hidden_states, topk_ids, topk_weights, cuda_stream)
        return self.sync_forward(hidden_states, cuda_stream)

    # ---- buffer policy (static, shared by every layer) ---------------------------------------------------------------
    @staticmethod
    def set_capture_batch_sizes(capture_bs: List[int]) -> None:
        KExpertsDeviceBuffer.capture_bs = list(capture_bs)

    @staticmethod
    def get_capture_batch_sizes() -> List[int]:
        return KExpertsDeviceBuffer.capture_bs

    @staticmethod
    def clear_buffer_cache() -> None:
        KExpertsDeviceBuffer.clear()


def _stream_handle(stream, device) -> int:
    if stream is None:
        return torch.cuda.current_stream(device).cuda_stream
    return int(getattr(stream, "cuda_stream", stream))

"""Routed-expert operators — mirror of archive/ktransformers/operators/experts.py for HBM-resident experts.

Reference surface kept (file:line in archive/ktransformers/operators/experts.py):
  KExpertsBase                :68-141   forward(input_tensor, expert_ids, weights) / load / unload / load_weights
  KExpertsCPU                 :143-365  the op the YAML rules name for decode; here its role is taken by KExpertsHIP
      .submit_for_one_decode / .sync_for_one_decode  :293-318  (fast path used by K*MoE.forward when capturing)
  EXPERTS_MAP                 :680-684
  KTransformersExperts        :686-757  prefill_op / generate_op switch, load / unload / set_inference_mode
  KDeepseekV2MoE / KDeepseekV3MoE   :874-971 / :972-1012   gate -> routed experts (+ shared experts) orchestration
  KMistralSparseMoEBlock      :1074-1170  Mixtral (BASELINE config C1): nn.Linear router, softmax -> top-k -> renormalise
  KDeepseekV3MoEV2 / KTransformersExpertsV2  :1172-1350  batched-serving variants: forward(..., bsz_tensor, cuda_graph_idx)

What changed underneath: the reference copies hidden states to pinned host memory, runs the experts on the CPU
(cpuinfer_ext MOE / AMX*_MOE) and copies the result back, ordered by cudaLaunchHostFunc; here the quantized experts
live in HBM and the forward is a few HIP launches on the caller's stream (ktransformers_amd/_native.py ->
include/ktx_moe.h), so submit/sync degenerate to "enqueue" and "nothing".  There is no CPU fallback.
"""
from __future__ import annotations

import os
from abc import ABC, abstractmethod

import torch
from torch import nn

from ktransformers_amd.operators.base_operator import BaseInjectedModule
from ktransformers_amd.util.utils import InferenceState

# reference backend names (KExpertsCPU(backend=...), experts.py:166,199-262; kt-kernel method table experts.py:316-360)
_BACKEND_TO_METHOD = {
    "AMXInt4": "AMXINT4", "AMXINT4": "AMXINT4", "int4": "AMXINT4",
    "AMXInt8": "AMXINT8", "AMXINT8": "AMXINT8", "int8": "AMXINT8",
    "AMXBF16": "BF16", "BF16": "BF16",
    "FP8": "FP8",                       # DeepSeek block-fp8 (weight + weight_scale_inv)
    "RAWINT4": "RAWINT4",               # Kimi-K2 native int4 (weight_packed + weight_scale, group 32)
    "llamafile": "GGUF", "GGUF": "GGUF",  # GGUF k-quant blocks + ggml types (the reference's default backend, experts.py:199-224)
}
_GROUP_SIZE = {"FP8": 128, "RAWINT4": 32}


class KExpertsBase(ABC):
    def __init__(self, key: str, gguf_loader, config, orig_module: nn.Module, device: str = "cuda", **kwargs):
        self.key = key
        self.gguf_loader = gguf_loader
        self.config = config
        self.device = device

    @abstractmethod
    def forward(self, input_tensor, expert_ids, weights):
        ...

    @abstractmethod
    def load(self, w: dict | None = None, device: str | None = None, warmup: bool = False):
        ...

    @abstractmethod
    def unload(self):
        ...

    def load_weights(self, override_key: list | None = None, device: str = "cpu") -> dict:
        """{key: {"gate","up","down"}} from the loader; bf16 stacked per expert (reference :89-135 returns the GGUF
        blobs + ggml types; the online-quant AMX backends require BF16 there — :226-228,246-248)."""
        res = {}
        for key in (override_key or [self.key]):
            if hasattr(self.gguf_loader, "load_experts"):
                res[key] = self.gguf_loader.load_experts(key, device=device)
            else:
                raise ValueError(f"Experts {key} not found in gguf_loader")
        return res


class KExpertsHIP(KExpertsBase):
    """All routed experts of one layer, quantized and resident in HBM; forward = ktx_moe_forward on the current stream.

    Registered in EXPERTS_MAP under its own name *and* under "KExpertsCPU"/"KExpertsTorch"/"KExpertsMarlin", so the
    reference's unmodified optimize_rules (generate_op: "KExpertsCPU", prefill_op: "KExpertsTorch") select it."""

    def __init__(self, key: str, gguf_loader, config, n_routed_experts: int, orig_module: nn.Module = None,
                 device: str = "cuda", out_device: str = "cuda", **kwargs):
        super().__init__(key, gguf_loader, config, orig_module, device, **kwargs)
        self.n_routed_experts = n_routed_experts
        self.out_device = out_device
        backend = kwargs.get("backend", "llamafile")      # the reference's default (experts.py:167): GGUF blocks, llamafile arithmetic
        if backend not in _BACKEND_TO_METHOD:
            raise ValueError(f"KExpertsHIP: unsupported backend {backend!r} (have {sorted(_BACKEND_TO_METHOD)})")
        self.method = _BACKEND_TO_METHOD[backend]
        self.max_len = int(kwargs.get("max_len", kwargs.get("chunk_size", 8192)))
        self.exact = bool(kwargs.get("exact", False))      # include/ktx_moe.h ktx_moe_set_exact (RAWINT4 prompt chunks: exact vs fast)
        self.expert_begin = int(kwargs.get("expert_begin", 0))
        self.expert_count = int(kwargs.get("expert_count", n_routed_experts))
        self.handle = None
        self._decode_out = None
        self._ep = None          # ExpertParallelMoE when ktransformers_amd.parallel.enable_expert_parallel() is on

    # the device the kernels run on: "cpu" in a reference YAML means "where KExpertsCPU ran" -> use out_device
    def _hip_device(self) -> torch.device:
        name = self.out_device if str(self.device).lower() == "cpu" else self.device
        dev = torch.device(name)
        if dev.type != "cuda":
            raise RuntimeError(f"KExpertsHIP needs a HIP device, got {name!r}: there is no CPU path")
        return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())

    def load(self, w: dict | None = None, device: str | None = None, warmup: bool = False):
        from ktransformers_amd._native import MoEHandle

        if self.handle is not None:
            return
        dev = self._hip_device()
        if w is None:
            w = self.load_weights(device=str(dev))[self.key]
        from ktransformers_amd import parallel
        ep_on = parallel.EP_STATE["enabled"] and torch.distributed.is_available() and torch.distributed.is_initialized()
        handle_len = self.max_len                            # (self.max_len itself is never changed: load() after unload())
        if ep_on:   # shard the experts over the ranks; routing ids stay global
            world = torch.distributed.get_world_size(parallel.EP_STATE["group"])
            rank = torch.distributed.get_rank(parallel.EP_STATE["group"])
            self.expert_begin, self.expert_count = parallel.expert_range(self.n_routed_experts, world, rank)
            handle_len = self.max_len * min(world, 8)        # decode gathers every rank's tokens; prefill receives rows
        cfg = self.config
        inter = getattr(cfg, "moe_intermediate_size", None) or cfg.intermediate_size
        method = self.method                 # THIS load's format; the configured self.method is never changed (unload() -> load() again)
        if method == "GGUF" and "gate_type" not in w:
            # a "llamafile" rule over a bf16 (safetensors) weight source: the reference would reject it; quantise online
            # to the int4 format instead of failing, like its AMX backends do for bf16 sources.
            method = "AMXINT4"
        elif method != "GGUF" and "gate_type" in w and int(w["gate_type"]) not in (0, 1, 30):
            # raw quantised ggml blocks under an online-quantising backend: the reference's AMX backends insist on a BF16
            # source too (experts.py:226-228, 246-248)
            raise ValueError(f"{self.key}: backend {self.method} quantises bf16 weights online, but the loader holds ggml type "
                             f"{int(w['gate_type'])} blocks; use backend 'llamafile' for GGUF k-quant experts")
        elif method == "GGUF":
            from ktransformers_amd._native import GGML_BLOCK_BYTES, GGML_LEGACY_TYPES
            types = {n: int(w[f"{n}_type"]) for n in ("gate", "up", "down")}
            native = all(t in GGML_BLOCK_BYTES for t in types.values()) and len({t in GGML_LEGACY_TYPES for t in types.values()}) == 1
            if os.environ.get("KTX_GGUF_BF16_FALLBACK") == "1" and any(t in GGML_LEGACY_TYPES for t in types.values()):
                native = False      # opt-in (rounds 2-4 served Q4_0 / Q5_0 / Q8_0 experts this way): exact weights, un-quantised activations
            if not native:
                # a ggml type (or a mix of the two families) the expert kernels do not read natively — they read Q2_K..Q6_K, IQ1_S, IQ4_XS
                # with Q8_K activations and Q4_0 / Q5_0 / Q8_0 with Q8_0 activations, llamafile's arithmetic both: the blocks are
                # de-quantised on the host with the loader's (reference-pinned) codecs and served as BF16 experts — exact
                # weights, un-quantised activations; NOT the llamafile arithmetic, and 2 bytes per weight.
                import warnings

                from ktransformers_amd.util.gguf_loader import GGML_NAMES, GGML_QUANT_SIZES, dequantize_expert_blocks
                warnings.warn(f"{self.key}: GGUF expert types {[GGML_NAMES.get(t, t) for t in types.values()]} have no native expert kernel "
                              "(as this combination / by request); de-quantising to BF16 experts", RuntimeWarning, stacklevel=2)
                E_all, e0, en = self.n_routed_experts, self.expert_begin, self.expert_count

                def local(name, rows, cols):   # only THIS rank's experts are de-quantised; the others stay zero rows the slice below drops
                    raw = w[name]
                    flat = (raw.detach().cpu() if isinstance(raw, torch.Tensor) else torch.as_tensor(raw)).reshape(-1).view(torch.uint8)
                    n_el, n_by = GGML_QUANT_SIZES[types[name]]
                    per = rows * cols // n_el * n_by
                    out = torch.zeros((E_all, rows, cols), dtype=torch.bfloat16)
                    out[e0:e0 + en] = dequantize_expert_blocks(flat[e0 * per:(e0 + en) * per], types[name], en, rows, cols)
                    return out
                w = {"gate": local("gate", inter, cfg.hidden_size), "up": local("up", inter, cfg.hidden_size),
                     "down": local("down", cfg.hidden_size, inter)}
                method = "BF16"
        h = MoEHandle(self.expert_count, cfg.num_experts_per_tok, cfg.hidden_size, inter, max_len=handle_len,
                      method=method, device=dev, expert_begin=self.expert_begin,
                      global_expert_num=self.n_routed_experts, group_size=_GROUP_SIZE.get(method, 0))
        sl = slice(self.expert_begin, self.expert_begin + self.expert_count)

        def prep(t, dtype=None):
            t = t if isinstance(t, torch.Tensor) else torch.as_tensor(t)
            t = t[sl].to(device=dev)
            return (t.to(dtype) if dtype is not None else t).contiguous()

        if method in ("AMXINT4", "AMXINT8", "BF16"):
            # bf16 source weights; the integer formats are quantised online exactly like the reference's AMX backends
            h.load_bf16(prep(w["gate"], torch.bfloat16), prep(w["up"], torch.bfloat16), prep(w["down"], torch.bfloat16))
        elif method == "GGUF":
            # raw ggml blocks [E, N, K/256 * block_bytes] + type ids, as KExpertsBase.load_weights returns them
            # (reference experts.py:89-135: gate/up/down mmap blobs and gate_type/up_type/down_type)
            def blocks(t, n):
                t = t if isinstance(t, torch.Tensor) else torch.as_tensor(t)
                return t.view(torch.uint8).reshape(self.n_routed_experts, n, -1)[sl].to(device=dev).contiguous()
            h.load_gguf(blocks(w["gate"], inter), blocks(w["up"], inter), blocks(w["down"], cfg.hidden_size),
                        int(w["gate_type"]), int(w["up_type"]), int(w["down_type"]))
        elif method == "FP8":
            h.load_fp8(prep(w["gate"]).view(torch.uint8), prep(w["up"]).view(torch.uint8), prep(w["down"]).view(torch.uint8),
                       prep(w["gate_scale"], torch.float32), prep(w["up_scale"], torch.float32),
                       prep(w["down_scale"], torch.float32))
        else:  # RAWINT4
            h.load_rawint4(prep(w["gate"]).view(torch.uint8), prep(w["up"]).view(torch.uint8), prep(w["down"]).view(torch.uint8),
                           prep(w["gate_scale"], torch.bfloat16), prep(w["up_scale"], torch.bfloat16),
                           prep(w["down_scale"], torch.bfloat16))
        # exact / fast switch (include/ktx_moe.h ktx_moe_set_exact): rule-file kwarg `exact: true` or KTX_MOE_EXACT=1
        if bool(self.exact) or os.environ.get("KTX_MOE_EXACT") == "1":
            h.set_exact(True)
        self.handle = h
        self.loaded_method = method          # the format this load actually built (the BF16 fallback of GGUF types without a kernel)
        if ep_on:
            self._ep = parallel.ExpertParallelMoE(h, parallel.EP_STATE["group"])
        if warmup and not ep_on:
            x = torch.zeros((1, cfg.hidden_size), dtype=torch.bfloat16, device=dev)
            ids = torch.zeros((1, cfg.num_experts_per_tok), dtype=torch.int64, device=dev)
            self.handle.forward(x, ids, torch.zeros((1, cfg.num_experts_per_tok), dtype=torch.float32, device=dev))

    def unload(self):
        if self.handle is not None:
            self.handle.close()
            self.handle = None

    def forward(self, input_tensor, expert_ids, weights, bsz_tensor=None, cuda_graph_idx=0, side=None):
        """side = (LinearHandle, side_x [T, K], residual [T, H] | None): an extension of the reference signature — the call
        returns residual + (experts(x) + side_linear(side_x)), the tail of the whole MoE block (MoEHandle.forward_side)."""
        if self.handle is None:
            raise RuntimeError("KExpertsHIP.forward before load()")
        dev = self.handle.device
        x = input_tensor.to(device=dev, dtype=torch.bfloat16).contiguous()
        ids = expert_ids.to(device=dev, dtype=torch.int64).contiguous()
        w = weights.to(device=dev, dtype=torch.float32).contiguous()
        if x.dim() == 1:
            x, ids, w = x.unsqueeze(0), ids.unsqueeze(0), w.unsqueeze(0)
        if side is not None:
            if self._ep is not None:
                raise RuntimeError("KExpertsHIP.forward(side=...) is a single-GPU path")
            lin, sx, res = side
            out = self.handle.forward_side(x, ids, w, lin, sx.reshape(x.shape[0], -1).contiguous(),
                                           None if res is None else res.reshape(x.shape[0], -1).contiguous(), bsz_tensor=bsz_tensor)
            return out.to(device=self.out_device if str(self.out_device) != "cuda" else dev)
        if self._ep is not None:
            # decode-sized batches: all-gather + local partial + reduce-scatter; prompts: all-to-all-v dispatch / return.
            # Contract (as for any collective): every rank calls with the same number of tokens when it is <= 16 — the
            # choice of path and the gather's row count are taken from the local T, there is no extra collective to agree on it.
            out = self._ep.forward(x, ids, w) if x.shape[0] <= 16 else self._ep.forward_prefill(x, ids, w)
        else:
            out = self.handle.forward(x, ids, w, bsz_tensor=bsz_tensor)
        return out.to(device=self.out_device if str(self.out_device) != "cuda" else dev)

    # reference fast path (experts.py:293-318): enqueue on the capturing stream, result later.  On the GPU both halves
    # are stream-ordered launches, so submit runs the forward and sync returns its output buffer.
    def submit_for_one_decode(self, input_tensor, expert_ids, weights, bsz_tensor=None, cuda_graph_idx=0, side=None):
        self._decode_out = self.forward(input_tensor.view(1, -1), expert_ids.view(1, -1), weights.view(1, -1),
                                        bsz_tensor=bsz_tensor, side=side)

    def sync_for_one_decode(self, cuda_graph_idx=0):
        out, self._decode_out = self._decode_out, None
        return out


EXPERTS_MAP = {
    "KExpertsHIP": KExpertsHIP,
    # names used by the reference's rule files; all of them resolve to the HBM-resident implementation
    "KExpertsCPU": KExpertsHIP,
    "KExpertsTorch": KExpertsHIP,
    "KExpertsMarlin": KExpertsHIP,
}


class KTransformersExperts(BaseInjectedModule, KExpertsBase):
    """archive/ktransformers/operators/experts.py:686-757.  With one HBM-resident implementation serving both phases,
    prefill_op and generate_op share a single KExpertsHIP instance when they name the same device."""

    def __init__(self, key: str, gguf_loader, config, orig_module: nn.Module, prefill_device: str = "cuda",
                 prefill_op: str | None = "KExpertsTorch", generate_device: str = "cpu",
                 generate_op: str | None = "KExpertsCPU", **kwargs):
        BaseInjectedModule.__init__(self, key, gguf_loader, config, orig_module, prefill_device, generate_device, **kwargs)
        KExpertsBase.__init__(self, key, gguf_loader, config, orig_module, generate_device, **kwargs)
        n = len(orig_module) if orig_module is not None and hasattr(orig_module, "__len__") else config.n_routed_experts
        gen = EXPERTS_MAP[generate_op](key, gguf_loader, config, n, device=generate_device, **kwargs) \
            if generate_op is not None else None
        if prefill_op is not None and gen is not None and EXPERTS_MAP[prefill_op] is EXPERTS_MAP[generate_op]:
            pre = gen
        else:
            pre = EXPERTS_MAP[prefill_op](key, gguf_loader, config, n, device=prefill_device, **kwargs) \
                if prefill_op is not None else None
        object.__setattr__(self, "generate_experts", gen)
        object.__setattr__(self, "prefill_experts", pre)
        object.__setattr__(self, "gpu_mlp_type", prefill_op)
        object.__setattr__(self, "cpu_mlp_type", generate_op)
        object.__setattr__(self, "mode", InferenceState.UNLOAD)

    def load(self, w: dict = None, mode: InferenceState = None, warmup: bool = True):
        if not mode:
            mode = InferenceState.GENERATE
        if mode == InferenceState.GENERATE:
            if self.prefill_experts is not None and self.prefill_experts is not self.generate_experts:
                self.prefill_experts.unload()
            self.generate_experts.load(w, warmup=warmup)
            self.device = self.generate_experts.device
        elif mode == InferenceState.PREFILL:
            if self.generate_experts is not None and self.prefill_experts is not self.generate_experts:
                self.generate_experts.unload()
            self.prefill_experts.load(w, warmup=warmup)
            self.device = self.prefill_experts.device
        elif mode == InferenceState.UNLOAD:
            self.unload()
        else:
            raise ValueError("mode must be either InferenceState.GENERATE, InferenceState.PREFILL or InferenceState.UNLOAD")
        self.mode = mode

    def unload(self):
        for e in (self.generate_experts, self.prefill_experts):
            if e is not None:
                e.unload()
        if self.generate_experts is not None:
            self.device = self.generate_experts.device

    def forward(self, input_tensor, expert_ids, weights, bsz_tensor=None, cuda_graph_idx=0):
        """`bsz_tensor` / `cuda_graph_idx`: the batched-serving contract of KTransformersExpertsV2 (experts.py:1331-1339) — an
        int32 device scalar holding the number of valid rows, read by the kernels at execution time so one captured graph
        serves every batch size up to the captured one; rows beyond it are left untouched."""
        if self.mode == InferenceState.GENERATE:
            assert self.generate_experts is not None, "generate_experts is None"
            return self.generate_experts.forward(input_tensor, expert_ids, weights, bsz_tensor, cuda_graph_idx)
        elif self.mode == InferenceState.PREFILL:
            assert self.prefill_experts is not None, "prefill_experts is None"
            return self.prefill_experts.forward(input_tensor, expert_ids, weights, bsz_tensor, cuda_graph_idx)
        raise ValueError("load or set_inference_mode before forward")

    def set_inference_mode(self, mode: InferenceState):
        if mode in (InferenceState.GENERATE, InferenceState.PREFILL):
            self.load(mode=mode, warmup=False)
        elif mode == InferenceState.UNLOAD:
            self.unload()
        else:
            raise ValueError("mode must be either InferenceState.GENERATE, InferenceState.PREFILL or InferenceState.UNLOAD")


class _KMoEBlock(BaseInjectedModule):
    """Shared orchestration of the model-specific MoE blocks: gate -> routed experts (+ shared experts)."""
    SUPPORTS_FUSION, RESIDUAL_KW, PRE_NORM_KW = True, "residual", "pre_norm"

    def moe_kexperts(self, x: torch.Tensor, topk_ids: torch.Tensor, topk_weight: torch.Tensor) -> torch.Tensor:
        return self.experts(x, topk_ids, topk_weight)

    def forward(self, hidden_states, residual=None, pre_norm=None):
        """Fusion hooks of the decoder-layer glue: `residual` -> returns residual + mlp(hidden_states), the two adds riding
        in the shared experts' down_proj epilogue; `pre_norm` (the layer's post_attention_layernorm module) ->
        hidden_states is the un-normalised residual stream and the norm runs inside the router launch."""
        return self._forward(hidden_states, residual, pre_norm)

    def _finish(self, y, identity, residual, orig_shape, shared_act=None, shared_raw=False):
        shared = getattr(self.config, "n_shared_experts", None) is not None
        se = self.shared_experts if shared else None
        if shared_act is not None:                                         # gate|up of the shared experts already done (rode with the router)
            # (shared_raw: the un-activated [gate | up] rows of a block-fp8 MLP — SiLU * up runs in down_proj's prologue)
            return se.down(shared_act, orig_shape, add1=y.view(*orig_shape), add2=residual, glu_in=shared_raw).view(*orig_shape)
        # (running the shared experts' first GEMV on a side stream, forked from and joined to the captured stream so it
        # overlaps the routed launches like the reference overlaps its CPU experts, was measured: 325 vs 465 tok/s —
        # a cross-stream join inside the HIP graph costs more than the launch it hides.)
        if se is not None and hasattr(se, "_gate_up"):                     # KDeepseekV3MLP: adds fused into down_proj
            return se(identity, add1=y.view(*orig_shape), add2=residual).view(*orig_shape)
        if se is not None:
            y = y.view(*orig_shape) + se(identity).view(*orig_shape)
        y = y.view(*orig_shape)
        return y if residual is None else residual + y

    def _forward(self, hidden_states, residual=None, pre_norm=None):
        orig_shape = hidden_states.shape
        sequence_length = orig_shape[1]
        shared_act = None
        side = self._router_side_linear(hidden_states, pre_norm, allow_cat=True)
        shared_raw = False
        if side is not None:
            # decode: the router rides in the launch of the shared experts' gate|up GEMV (same input row, independent results)
            glu = side.fmt == "W4"                           # block-FP8 shared experts: [gate | up] rows, SiLU * up in down_proj's prologue
            topk_idx, topk_weight, xn, shared_act = self.gate.forward_with_linear(
                hidden_states, (pre_norm.weight, pre_norm.variance_epsilon), side, glu=glu)
            shared_raw = not glu
            hidden_states = xn.view(*orig_shape)
        elif pre_norm is not None and hasattr(self.gate, "_handle"):
            topk_idx, topk_weight, xn = self.gate(hidden_states, norm=(pre_norm.weight, pre_norm.variance_epsilon))
            hidden_states = xn.view(*orig_shape)
        else:
            if pre_norm is not None:
                hidden_states = pre_norm(hidden_states)
            topk_idx, topk_weight = self.gate(hidden_states)
        identity = hidden_states
        hidden_states = hidden_states.view(-1, hidden_states.shape[-1])
        shared = getattr(self.config, "n_shared_experts", None) is not None

        gen = getattr(self.experts, "generate_experts", None)
        if (sequence_length == 1 and gen is not None and hasattr(gen, "submit_for_one_decode")
                and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            tail = self._tail_side(None if shared_raw else shared_act, residual, gen)
            gen.submit_for_one_decode(hidden_states[0], topk_idx[0], topk_weight[0], **({"side": tail} if tail else {}))
            y = gen.sync_for_one_decode().unsqueeze(0)
            return y.view(*orig_shape) if tail else self._finish(y, identity, residual, orig_shape, shared_act, shared_raw)

        ex = self.experts
        op = getattr(ex, "generate_experts", None) if getattr(ex, "mode", None) == InferenceState.GENERATE else \
            getattr(ex, "prefill_experts", None)
        tail = self._tail_side(None if shared_raw else shared_act, residual, op)
        if tail:
            return op.forward(hidden_states, topk_idx, topk_weight, side=tail).view(*orig_shape).to(device=hidden_states.device)
        y = self.moe_kexperts(hidden_states, topk_idx, topk_weight).view(*orig_shape).to(device=hidden_states.device)
        return self._finish(y, identity, residual, orig_shape, shared_act, shared_raw)

    def _tail_side(self, shared_act, residual, experts_op):
        """(down_proj handle, shared_act, residual) when the routed experts' call can carry the shared experts' down projection and
        the block's closing adds (KExpertsHIP.forward(side=...)): decode steps whose shared gate|up half already ran beside the
        router, a single-GPU HBM experts operator, a W4 down_proj onto hidden_size."""
        if shared_act is None or not isinstance(experts_op, KExpertsHIP) or experts_op._ep is not None \
                or os.environ.get("KTX_MOE_SEPARATE_SHARED_DOWN"):
            return None
        down = getattr(getattr(self.shared_experts, "orig_module", None), "down_proj", None)
        lin = getattr(down, "generate_linear", None) if getattr(down, "mode", None) == InferenceState.GENERATE else \
            getattr(down, "prefill_linear", None)
        h = getattr(lin, "_h", None)
        if h is None or getattr(h, "fmt", None) != "W4" or h.N != self.config.hidden_size or getattr(lin, "has_bias", False):
            return None
        res = None if residual is None else residual.reshape(-1, residual.shape[-1])
        return (h, shared_act, res)

    def _router_side_linear(self, hidden_states, pre_norm, allow_cat: bool = False):
        """The LinearHandle of the shared experts' merged gate|up operator when this call can use the combined launch: a decode
        step (<= 4 rows) with the layer's post-attention norm handed in, a HIP router and merged shared experts (W4 GLU-interleaved
        rows; with allow_cat also the block-FP8 [gate | up] rows, whose SiLU * up the caller runs)."""
        if pre_norm is None or not hasattr(self.gate, "forward_with_linear") or os.environ.get("KTX_MOE_SEPARATE_ROUTER"):
            return None
        if getattr(self.config, "n_shared_experts", None) is None or hidden_states.numel() // hidden_states.shape[-1] > 4:
            return None
        if hidden_states.dtype != torch.bfloat16 or not hidden_states.is_cuda:
            return None
        w = getattr(getattr(self.gate, "orig_module", None), "weight", None)
        if w is None or w.dtype != torch.bfloat16:
            return None
        gu = getattr(self.shared_experts, "_gate_up", None)
        h = getattr(gu, "_h", None)
        if h is None and allow_cat:
            h = getattr(getattr(self.shared_experts, "_gate_up_cat", None), "_h", None)
            if h is not None and getattr(h, "fmt", None) != "FP8":
                h = None
        elif getattr(h, "fmt", None) != "W4":
            h = None
        if h is None or not hasattr(self.shared_experts, "down"):
            return None
        return h


class KDeepseekV2MoE(_KMoEBlock):
    """archive/ktransformers/operators/experts.py:874-971."""


class KDeepseekV3MoE(_KMoEBlock):
    """archive/ktransformers/operators/experts.py:972-1012 ("V3 MoE" block; sigmoid noaux_tc gate in front)."""


class KTransformersExpertsV2(KTransformersExperts):
    """archive/ktransformers/operators/experts.py:1273-1350: the balance_serve engine's expert switch; its forward always
    carries (bsz_tensor, cuda_graph_idx) — KTransformersExperts.forward accepts both spellings."""


class KDeepseekV3MoEV2(_KMoEBlock):
    """archive/ktransformers/operators/experts.py:1172-1213: forward(hidden_states, bsz_tensor, cuda_graph_idx=0) of the
    batched-serving engine.  hidden_states [1, T_max, H] with `bsz_tensor[0]` valid rows; the router and the shared experts
    run over all T_max rows (the reference's shared_experts(identity, bsz_tensor) skips the padding rows, whose values
    nobody reads), the routed experts honour bsz_tensor on the device."""

    def forward(self, hidden_states, bsz_tensor=None, cuda_graph_idx=0):
        orig_shape = hidden_states.shape
        identity = hidden_states
        topk_idx, topk_weight = self.gate(hidden_states)
        x = hidden_states.view(-1, hidden_states.shape[-1])
        y = self.experts(x, topk_idx, topk_weight, bsz_tensor, cuda_graph_idx).view(*orig_shape).to(device=hidden_states.device)
        return self._finish(y, identity, None, orig_shape)


class KMistralSparseMoEBlock(BaseInjectedModule):
    """archive/ktransformers/operators/experts.py:1074-1128 (Mixtral-8x7B, BASELINE config C1): router logits from the
    original nn.Linear gate, fp32 softmax over all experts, top-k, renormalise, weights cast to the activation dtype (the
    reference does, :1090, so they reach the experts bf16-rounded) -> routed experts.  Returns (y, router_logits)."""

    def moe_kexperts(self, x: torch.Tensor, topk_ids: torch.Tensor, topk_weight: torch.Tensor) -> torch.Tensor:
        return self.experts(x, topk_ids, topk_weight)

    def forward(self, hidden_states: torch.Tensor):
        orig_shape = hidden_states.shape
        sequence_length = orig_shape[1]
        x = hidden_states.view(-1, orig_shape[-1])
        router_logits = self.gate(x)
        routing_weights = torch.nn.functional.softmax(router_logits, dim=1, dtype=torch.float)
        routing_weights, selected_experts = torch.topk(routing_weights, self.top_k, dim=-1)
        routing_weights = routing_weights / routing_weights.sum(dim=-1, keepdim=True)
        routing_weights = routing_weights.to(x.dtype)
        gen = getattr(self.experts, "generate_experts", None)
        if (sequence_length == 1 and x.shape[0] == 1 and gen is not None and hasattr(gen, "submit_for_one_decode")
                and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            gen.submit_for_one_decode(x[0], selected_experts[0], routing_weights[0])
            y = gen.sync_for_one_decode()
        else:
            y = self.moe_kexperts(x, selected_experts, routing_weights)
        return y.view(*orig_shape).to(device=hidden_states.device), router_logits

"""MoE router operator — mirror of archive/ktransformers/operators/gate.py:91-127 (KMoEGate).

The reference keeps the HF MoEGate math (a chain of ~10 torch kernels: fp32 linear, sigmoid/softmax, bias add, group
top-2 sums, two topk calls, scatter, masked_fill, gather, normalise — models/modeling_deepseek_v3.py:430-481) and only
re-homes the parameters.  Here forward() runs two HIP launches (ktransformers_amd/csrc/ktx_gate.hip)."""
from __future__ import annotations

import torch
from torch import nn

from ktransformers_amd.operators.base_operator import BaseInjectedModule


class KMoEGate(BaseInjectedModule):
    def __init__(self, key: str, gguf_loader, config, orig_module: nn.Module = None, generate_device: str = "cuda",
                 prefill_device: str = "cuda", **kwargs):
        BaseInjectedModule.__init__(self, key, gguf_loader, config, orig_module, prefill_device, generate_device, **kwargs)
        object.__setattr__(self, "_gate", None)

    def _handle(self):
        if self._gate is None:
            from ktransformers_amd._native import GateHandle

            c = self.config
            object.__setattr__(self, "_gate", GateHandle(
                c.n_routed_experts, c.hidden_size, c.num_experts_per_tok, getattr(c, "n_group", 1) or 1,
                getattr(c, "topk_group", 1) or 1, getattr(c, "scoring_func", "softmax"),
                getattr(c, "topk_method", "greedy"), bool(getattr(c, "norm_topk_prob", False)),
                float(getattr(c, "routed_scaling_factor", 1.0))))
        return self._gate

    def forward(self, hidden_states, norm=None):
        """(topk_idx, topk_weight) like MoEGate.forward; with norm=(weight, eps) the input is the un-normalised hidden
        state, RMSNorm runs inside the router launch and the normalised rows come back as a third value."""
        h = hidden_states.shape[-1]
        x = hidden_states.reshape(-1, h)
        w = self.orig_module.weight
        bias = getattr(self.orig_module, "e_score_correction_bias", None)
        return self._handle().forward(x.to(torch.bfloat16).contiguous(), w, bias, norm=norm)

    def forward_with_linear(self, hidden_states, norm, linear_handle, glu: bool = True):
        """Decode step: forward(hidden_states, norm=norm) AND linear_handle.forward(hidden_states, norm=norm, glu=glu) — the
        shared experts' merged gate|up GEMV, independent work on the same row — in one launch
        (ktx_linear_forward_fused_gate).  Returns (topk_idx, topk_weight, xn, y_linear)."""
        from ktransformers_amd._native import gate_with_linear

        x = hidden_states.reshape(-1, hidden_states.shape[-1]).to(torch.bfloat16).contiguous()
        bias = getattr(self.orig_module, "e_score_correction_bias", None)
        return gate_with_linear(self._handle(), linear_handle, x, self.orig_module.weight, bias, norm, glu=glu)

    def load(self, w: dict | None = None, device: str | None = None):
        if device is None:
            device = self.device
        if w is None:
            w = {"weight": self.gguf_loader.load_tensor(self.key + ".weight", device=device)}
            if self.gguf_loader.has_tensor(self.key + ".e_score_correction_bias"):
                w["e_score_correction_bias"] = self.gguf_loader.load_tensor(self.key + ".e_score_correction_bias", device=device)
        if not isinstance(w, dict):
            raise ValueError("Invalid weight type")
        self.orig_module.weight = nn.Parameter(w["weight"].to(device), requires_grad=False)
        if w.get("e_score_correction_bias") is not None:
            self.orig_module.e_score_correction_bias = nn.Parameter(w["e_score_correction_bias"].to(device), requires_grad=False)

    def unload(self):
        self.orig_module.weight = None
        if hasattr(self.orig_module, "e_score_correction_bias"):
            self.orig_module.e_score_correction_bias = None

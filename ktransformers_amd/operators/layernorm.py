"""RMSNorm operator — mirror of archive/ktransformers/operators/layernorm.py:47-87 (class RMSNorm): `forward(x)` is
forward_native's math, `forward(x, batch_size_tensor)` flashinfer's rmsnorm, `forward(x, batch_size_tensor, residual)` the
in-place fused add + norm returning (x, residual).  All three run csrc/ktx_ops.hip."""
from __future__ import annotations

import torch
from torch import nn

from ktransformers_amd.operators.base_operator import BaseInjectedModule


class RMSNorm(BaseInjectedModule):
    def __init__(self, key, gguf_loader, config, orig_module: nn.Module, prefill_device: str = "cuda",
                 generate_device: str = "cuda", **kwargs):
        BaseInjectedModule.__init__(self, key, gguf_loader, config, orig_module, prefill_device, generate_device, **kwargs)

    def load(self):
        w = self.gguf_loader.load_tensor(self.key + ".weight", device=self.device)
        self.orig_module.weight = nn.Parameter(w.to(torch.bfloat16), requires_grad=False)

    def forward(self, x: torch.Tensor, batch_size_tensor: torch.Tensor = None, residual: torch.Tensor = None):
        from ktransformers_amd._native import fused_add_rmsnorm, rmsnorm

        w, eps = self.orig_module.weight, self.orig_module.variance_epsilon
        if batch_size_tensor is None:
            return rmsnorm(x, w, eps, native_rounding=True)
        if residual is not None:
            fused_add_rmsnorm(x, residual, w, eps, batch_size_tensor)
            return x, residual
        return rmsnorm(x, w, eps, native_rounding=False, bsz_tensor=batch_size_tensor)


KQwen2MoeRMSNorm = KQwen3MoeRMSNorm = DeepseekV3RMSNorm = RMSNorm

"""Dense / shared-expert MLP operator — counterpart of archive/ktransformers/operators/mlp.py (kDeepseekV3MLP):
down_proj(act_fn(gate_proj(x)) * up_proj(x)) (models/modeling_deepseek_v3.py:396-398).

gate_proj and up_proj share their input, so they are loaded as ONE row-concatenated quantised linear (one GEMV launch,
exactly the rows the two separate operators give), followed by ktx_silu_mul and down_proj; the residual-style adds that
follow an MLP in the decoder layer can ride in down_proj's epilogue (`add1`, `add2`)."""
from __future__ import annotations

import torch
from torch import nn

from ktransformers_amd.operators.base_operator import BaseInjectedModule
from ktransformers_amd.operators.linear import KTransformersLinear, build_merged_linear
from ktransformers_amd.util.utils import InferenceState


class KDeepseekV3MLP(BaseInjectedModule):
    SUPPORTS_FUSION, RESIDUAL_KW, PRE_NORM_KW = True, "add1", "pre_norm"

    def __init__(self, key: str, gguf_loader, config, orig_module: nn.Module, prefill_device: str = "cuda",
                 generate_device: str = "cuda", **kwargs):
        BaseInjectedModule.__init__(self, key, gguf_loader, config, orig_module, prefill_device, generate_device, **kwargs)
        object.__setattr__(self, "_gate_up", None)
        object.__setattr__(self, "_gate_up_cat", None)

    def load(self):
        down, gate, up = self.orig_module.down_proj, self.orig_module.gate_proj, self.orig_module.up_proj
        from ktransformers_amd.util.utils import load_weights
        load_weights(down, self.gguf_loader, self.key + ".down_proj.")
        merged = cat = None
        if all(isinstance(m, KTransformersLinear) for m in (down, gate, up)) and down.generate_linear is not None:
            from ktransformers_amd.operators.linear import KLinearFP8
            keys = [self.key + ".gate_proj", self.key + ".up_proj"]
            # decided BEFORE anything is built (each build loads and quantises both matrices): block-fp8 sources take the plain
            # concatenation [gate ; up] — the strip-interleaved GLU layout would put gate and up rows into one 128-row scale block, the
            # concatenation keeps every block whole (intermediate sizes are multiples of 128): one GEMV (+ the fused input norm) and one
            # SiLU * up launch — every other format takes the interleaved layout with the GLU epilogue
            fp8 = isinstance(down.generate_linear, KLinearFP8) and all(self.gguf_loader.has_tensor(k + ".weight_scale_inv") for k in keys)
            if fp8:
                cat = build_merged_linear(down.generate_linear, keys, self.gguf_loader, down.generate_linear.device, interleave8=False)
                if cat is not None and (len(cat[1]) != 2 or cat[1][0] != cat[1][1] or getattr(cat[0], "FMT", None) != "FP8"):
                    cat = None
            else:
                merged = build_merged_linear(down.generate_linear, keys, self.gguf_loader, down.generate_linear.device, interleave8=True)
        if merged is not None or cat is not None:
            object.__setattr__(self, "_gate_up" if merged is not None else "_gate_up_cat", (merged or cat)[0])
            for m in (gate, up):                       # their rows live in the merged operator
                for op in {id(m.generate_linear): m.generate_linear, id(m.prefill_linear): m.prefill_linear}.values():
                    if op is not None:
                        op.unload()
                        op.loaded = True
        else:
            load_weights(gate, self.gguf_loader, self.key + ".gate_proj.")
            load_weights(up, self.gguf_loader, self.key + ".up_proj.")

    def gate_up(self, x: torch.Tensor, norm: tuple | None = None) -> torch.Tensor:
        """First half: act_fn(gate_proj(x)) * up_proj(x) -> [T, intermediate] (one launch when merged)."""
        from ktransformers_amd._native import rmsnorm, silu_mul

        x2 = x.reshape(-1, x.shape[-1])
        if self._gate_up is not None:       # merged GEMV with the SiLU * up epilogue
            return self._gate_up.forward(x2, norm=norm, glu=True)
        if self._gate_up_cat is not None:   # [gate | up] rows of one GEMV, then SiLU * up
            return silu_mul(self._gate_up_cat.forward(x2, norm=norm))
        if norm is not None:
            x2 = rmsnorm(x2, norm[0], norm[1], native_rounding=True)
        return silu_mul(torch.cat([self.orig_module.gate_proj(x2), self.orig_module.up_proj(x2)], dim=-1))

    def down(self, a: torch.Tensor, shape, add1: torch.Tensor | None = None, add2: torch.Tensor | None = None,
             glu_in: bool = False) -> torch.Tensor:
        """Second half: down_proj(a) (+ add1, + add2 in its epilogue).  glu_in: `a` is the un-activated [gate | up] block of the
        first half (block-fp8: the concatenated GEMV has no GLU epilogue) and SiLU * up runs in down_proj's prologue
        (LinearHandle.forward(glu_in=True): inside the decode kernel, a ktx_silu_mul launch in front of the prompt kernels)."""
        down = self.orig_module.down_proj
        fusion = {k: v.reshape(-1, shape[-1]) for k, v in (("add1", add1), ("add2", add2)) if v is not None}
        if glu_in:
            op = down.prefill_linear if isinstance(down, KTransformersLinear) and down.mode == InferenceState.PREFILL else \
                getattr(down, "generate_linear", None)
            if getattr(op, "_h", None) is not None and hasattr(op._h, "_forward_glu_in"):
                fusion["glu_in"] = True
            else:
                from ktransformers_amd._native import silu_mul
                a = silu_mul(a)
        if isinstance(down, KTransformersLinear):
            y = down(a, **fusion)
        else:
            y = down(a)
            for v in fusion.values():
                y = v + y
        return y.reshape(*shape[:-1], y.shape[-1])

    def forward(self, x: torch.Tensor, add1: torch.Tensor | None = None, add2: torch.Tensor | None = None,
                norm: tuple | None = None, pre_norm=None) -> torch.Tensor:
        if pre_norm is not None:
            norm = (pre_norm.weight, pre_norm.variance_epsilon)
        if self._gate_up_cat is not None:   # block-fp8: [gate | up] rows of one GEMV; SiLU * up in down_proj's prologue
            return self.down(self._gate_up_cat.forward(x.reshape(-1, x.shape[-1]), norm=norm), x.shape, add1, add2, glu_in=True)
        return self.down(self.gate_up(x, norm), x.shape, add1, add2)


kDeepseekV3MLP = KDeepseekV3MLP

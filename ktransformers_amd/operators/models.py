"""Model-level operator named by the reference's rule files (`- match: {name: "^model$"}  replace: {class:
ktransformers.operators.models.KDeepseekV2Model, kwargs: {per_layer_prefill_intput_threshold: 0}}`,
archive/ktransformers/operators/models.py:547-742).

In the reference this wrapper owns two things: moving the hidden state between the devices the layers are placed on
(`transfer_map`) and "per-layer prefill" for very long prompts (load one layer's CPU-resident experts to the GPU, run the whole
prompt through it, unload; threshold in tokens).  With every expert resident in HBM neither exists here — a long prompt is
chunked by `util.generate.prefill_and_generate` instead — so the operator keeps the constructor contract (the reference's
unmodified YAML files must load) and runs the skeleton model's own forward."""
from __future__ import annotations

from torch import nn

from ktransformers_amd.operators.base_operator import BaseInjectedModule


class KDeepseekV2Model(BaseInjectedModule):
    def __init__(self, key: str, gguf_loader, config, orig_module: nn.Module, device: str = "cuda",
                 per_layer_prefill_intput_threshold: int | None = 30000, transfer_map: dict | None = None, **kwargs):
        kwargs.pop("prefill_device", None)
        kwargs.pop("generate_device", None)
        BaseInjectedModule.__init__(self, key, gguf_loader, config, orig_module, device, device, **kwargs)
        if transfer_map:
            raise NotImplementedError("transfer_map (layers spread over several devices inside one process) is not supported: "
                                      "this library scales as one process per GPU with expert parallelism (parallel.py)")
        object.__setattr__(self, "per_layer_prefill_intput_threshold", per_layer_prefill_intput_threshold)
        object.__setattr__(self, "transfer_map", transfer_map)

    def forward(self, *args, per_layer_prefill_intput_threshold=None, is_prefill=None, **kwargs):
        return self.orig_module(*args, **kwargs)


KDeepseekV3Model = KDeepseekV2Model

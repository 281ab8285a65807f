"""Host utilities mirrored from archive/ktransformers/util/utils.py (only what the hot path's operators need)."""
from __future__ import annotations

import enum

import torch
from torch import nn


class InferenceState(enum.Enum):
    """archive/ktransformers/util/utils.py:317-322."""
    UNLOAD = 0
    PREFILL = 1
    GENERATE = 2
    RESTORE = 3


def set_module(model: nn.Module, submodule_key: str, module: nn.Module) -> None:
    """archive/ktransformers/util/utils.py:252-262: replace ``model.<dotted key>`` by ``module``."""
    tokens = submodule_key.split(".")
    cur = model
    for name in tokens[:-1]:
        cur = getattr(cur, name)
    setattr(cur, tokens[-1], module)


def load_cur_state_dict(module: nn.Module, loader, prefix: str = "", device: str = "cuda") -> None:
    """Materialise the (meta) parameters/buffers of a non-injected module from the loader
    (archive/ktransformers/util/utils.py:264-333, minus the GGUF-dequant branches which live in the loader)."""
    persistent = {k: v for k, v in module._buffers.items() if k not in module._non_persistent_buffers_set}
    for name, param in list(module._parameters.items()) + list(persistent.items()):
        if param is None:
            continue
        key = prefix + name
        if not loader.has_tensor(key):
            raise KeyError(f"can't find {key} in the weight source")
        target = loader.tensor_device_map.get(prefix[:-1], {}).get("generate_device", device) \
            if hasattr(loader, "tensor_device_map") else device
        w = loader.load_tensor(key, device=target)
        if isinstance(param, nn.Parameter):
            module._parameters[name] = nn.Parameter(w.to(param.dtype if param.dtype.is_floating_point else w.dtype),
                                                    requires_grad=False)
        else:
            module._buffers[name] = w


def load_weights(module: nn.Module, loader, prefix: str = "", device: str = "cuda") -> None:
    """archive/ktransformers/util/utils.py:335-342: injected modules load themselves, the rest is filled from the loader."""
    from ktransformers_amd.operators.base_operator import BaseInjectedModule

    if isinstance(module, BaseInjectedModule):
        module.load()
        return
    load_cur_state_dict(module, loader, prefix, device=device)
    for name, child in module._modules.items():
        if child is not None:
            load_weights(child, loader, prefix + name + ".", device=device)

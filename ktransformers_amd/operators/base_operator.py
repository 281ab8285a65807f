"""BaseInjectedModule — the contract every YAML-injected replacement module honours.

Mirrors archive/ktransformers/operators/base_operator.py:12-63: constructed as
``cls(key, gguf_loader, config, orig_module, prefill_device=, generate_device=, **kwargs)`` (optimize.py:45), keeps the
replaced module reachable as ``orig_module`` and lets every attribute it does not define itself fall through to it (the
"load-bearing quirk" SURVEY.md §7 notes), and exposes ``load()`` for utils.load_weights.
"""
from __future__ import annotations

from typing import Any

from torch import nn


class BaseInjectedModule(nn.Module):
    def __init__(self, key: str, gguf_loader, config, orig_module: nn.Module, prefill_device: str = "cuda",
                 generate_device: str = "cuda", **kwargs):
        nn.Module.__init__(self)
        nn.Module.__setattr__(self, "orig_module", orig_module)
        for name, value in (("key", key), ("gguf_loader", gguf_loader), ("config", config),
                            ("prefill_device", prefill_device), ("generate_device", generate_device),
                            ("device", generate_device)):
            object.__setattr__(self, name, value)

    # -- attribute fall-through -----------------------------------------------------------------------------------
    def __getattr__(self, name: str) -> Any:
        try:
            return object.__getattribute__(self, name)
        except AttributeError:
            pass
        if name == "orig_module":
            return nn.Module.__getattr__(self, "orig_module")
        orig = nn.Module.__getattr__(self, "orig_module")
        try:
            return nn.Module.__getattr__(orig, name)       # parameters / buffers / submodules of the original
        except AttributeError:
            return object.__getattribute__(orig, name)      # plain attributes (e.g. Linear.out_features)

    def __setattr__(self, name: str, value) -> None:
        if name == "orig_module":
            return nn.Module.__setattr__(self, "orig_module", value)
        try:
            object.__getattribute__(self, name)
            own = True
        except AttributeError:
            own = name in self.__dict__ or name in self._modules or name in self._parameters or name in self._buffers
        if own:
            if isinstance(value, nn.Module) or name in self._modules:
                return nn.Module.__setattr__(self, name, value)
            return object.__setattr__(self, name, value)
        return nn.Module.__getattr__(self, "orig_module").__setattr__(name, value)

    def forward(self, *args, **kwargs):
        return self.orig_module.forward(*args, **kwargs)

    def load(self):
        from ktransformers_amd.util.utils import load_weights
        for name, child in self._modules.items():
            load_weights(child, self.gguf_loader, self.key + ".")

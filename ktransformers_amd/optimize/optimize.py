"""YAML rule matching + module injection — mirror of archive/ktransformers/optimize/optimize.py:28-163.

Rule format (unchanged):  - match: {name: <regex on dotted module name>, class: <dotted class path>}
                            replace: {class: <dotted class path> | "default", kwargs: {...}}
                            recursive: bool
First matching rule wins; every replacement is constructed as
``cls(key=, gguf_loader=, config=, orig_module=, **kwargs)`` and swapped in with set_module.

Class paths written for the reference package (``ktransformers.operators.experts.KTransformersExperts`` ...) resolve to
this package's mirrors when the reference package itself is not importable, so the reference's rule files run as-is.
"""
from __future__ import annotations

import copy
import importlib
import re
from typing import List, Mapping

import torch
import yaml
from torch import nn

from ktransformers_amd.util.utils import load_weights, set_module

_ALIAS_PREFIXES = (("ktransformers.operators.", "ktransformers_amd.operators."),
                   ("ktransformers.optimize.", "ktransformers_amd.optimize."),
                   ("ktransformers.util.", "ktransformers_amd.util."),
                   # the host model tree: both reference modeling files map onto the one skeleton
                   ("ktransformers.models.modeling_deepseek_v3.", "ktransformers_amd.models.modeling_deepseek."),
                   ("ktransformers.models.modeling_deepseek.", "ktransformers_amd.models.modeling_deepseek."),
                   ("ktransformers.models.", "ktransformers_amd.models."))


def resolve_class(path: str):
    """Import ``a.b.C``; paths into the reference package fall back to the ktransformers_amd mirror."""
    mod_name, _, cls_name = path.rpartition(".")
    candidates = [mod_name]
    for old, new in _ALIAS_PREFIXES:
        if (mod_name + ".").startswith(old):
            candidates.insert(0, new + mod_name[len(old):] if len(mod_name) >= len(old) else new.rstrip("."))
    last = None
    for m in candidates:
        try:
            return getattr(importlib.import_module(m), cls_name)
        except (ImportError, AttributeError) as e:
            last = e
    raise ImportError(f"cannot resolve class {path!r}: {last}")


def inject(module: nn.Module, local_optimization_dict: Mapping, model_config, gguf_loader, prefix: str = "") -> None:
    """optimize.py:28-54."""
    for name, child in list(module._modules.items()):
        if child is None:
            continue
        child_prefix = prefix + name
        if child_prefix not in local_optimization_dict:
            continue
        meta = local_optimization_dict[child_prefix]
        kwargs = meta.get("kwargs", {}) or {}
        gguf_loader.tensor_device_map[meta["key"]] = kwargs
        if meta["class"] != "default":
            cls = resolve_class(meta["class"])
            print(f"Injecting {child_prefix} as {cls.__module__} . {cls.__name__}")
            set_module(module, name, cls(key=meta["key"], gguf_loader=gguf_loader, config=model_config,
                                         orig_module=child, **kwargs))
        child_prefix += "."
        sub = {k: v for k, v in local_optimization_dict.items() if k.startswith(child_prefix)}
        inject(child, sub, model_config, gguf_loader, child_prefix)


def gen_optimize_config(module: nn.Module, out_data: dict, rule_list: List, prefix: str = "",
                        default_device: str = "cuda:0") -> None:
    """optimize.py:67-118: first matching rule wins; `recursive: False` stops the descent."""
    module_name = prefix[:-1]
    recursive = True
    for rule in rule_list:
        match_meta = rule["match"]
        if "class" not in match_meta and "name" not in match_meta:
            raise Exception("match must have at least one of \"class\" and \"name\"")
        if "class" in match_meta:
            try:
                cls = resolve_class(match_meta["class"])
            except ImportError:
                continue  # a rule written for a model family that is not installed can never match
            if not isinstance(module, cls):
                continue
        if "name" in match_meta and re.search(match_meta["name"], module_name) is None:
            continue
        if "replace" not in rule:
            raise Exception("replace must be in rule")
        rep = rule["replace"]
        if module_name not in out_data:
            out_data[module_name] = {"key": module_name, "class": rep.get("class", "default"),
                                     "kwargs": copy.deepcopy(rep.get("kwargs", {}) or {})}
        else:
            if out_data[module_name]["class"] == "default":
                out_data[module_name]["class"] = rep.get("class", "default")
            out_data[module_name]["kwargs"].update(copy.deepcopy(rep.get("kwargs", {}) or {}))
        if "recursive" in rule:
            recursive = bool(rule["recursive"])
        break
    if module_name not in out_data:
        out_data[module_name] = {"class": "default", "key": module_name,
                                 "kwargs": {"generate_device": default_device, "prefill_device": default_device}}
    if recursive:
        for name, child in module._modules.items():
            if child is not None:
                gen_optimize_config(child, out_data, rule_list, prefix + name + ".", default_device=default_device)


def load_rules(rule_file: str) -> list:
    with open(rule_file, "r", encoding="utf-8") as f:
        return yaml.load(f.read(), Loader=yaml.FullLoader)


def optimize_and_load(module: nn.Module, rule_file: str, loader, model_config, default_device: str = "cuda:0",
                      load: bool = True) -> dict:
    """optimize_and_load_gguf (optimize.py:129-163) with the weight source passed in as a loader object."""
    rule_list = load_rules(rule_file)
    optimize_config: dict = {}
    gen_optimize_config(module, optimize_config, rule_list, default_device=default_device)
    if not hasattr(loader, "tensor_device_map"):
        loader.tensor_device_map = {}
    inject(module, optimize_config, model_config, loader)
    if load:
        load_weights(module, loader, device=default_device)
    module.gguf_loader = loader
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return optimize_config


optimize_and_load_gguf = optimize_and_load  # reference name

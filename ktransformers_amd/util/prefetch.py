"""Decode-step read-ahead (opt-in experiment: KTX_PREFETCH=1).

What the reference overlaps: its batch-1 decode runs the CPU experts beside the GPU attention of the same step
(archive/ktransformers/operators/experts.py:974-1012 submit_for_one_decode / sync_for_one_decode).  Here everything is one
launch chain on one GPU, and five of a MoE layer's eight launches are latency chains that leave HBM idle.
This module forks a side stream off the decode stream and lets small read-only launches (ktx_prefetch) pull the weights of
the launches that come NEXT into the die-level Infinity Cache (256 MiB, memory side) while those chains run:

  set 1 — forked at the layer's start (beside q_a|kv_a, q_b + absorb, the MLA kernel, merge + un-absorb):
          this layer's W_UV, o_proj, router weight, shared experts' gate|up and down;
  set 2 — forked when the attention operator has enqueued its last launch (beside the router | shared gate|up launch):
          the NEXT layer's q_a|kv_a, q_b and W_UK.

The routed experts cannot be read ahead (their ids do not exist yet) and are not: the two expert launches already stream
at 0.5-0.6 of the HBM roof.  The side stream is joined back once, at the end of the model's forward, so a captured graph
keeps its single chain of kernels plus one parallel chain of read-only nodes.  Results are not affected in any way.

KTX_PREFETCH        unset / 0 = off (default), 1 = both sets, "s1" / "s2" = one of them
KTX_PREFETCH_WGS    workgroups per read-ahead launch (default 128)
"""
from __future__ import annotations

import os

import torch


def enabled() -> str:
    v = os.environ.get("KTX_PREFETCH", "0").strip().lower()
    return "" if v in ("", "0", "off", "false") else v


def _wgs() -> int:
    return max(1, int(os.environ.get("KTX_PREFETCH_WGS", "128")))


def _handles(*objs) -> list:
    """LinearHandles behind operator objects: a LinearHandle, a KLinear* operator (`_h`), a KTransformersLinear
    (`generate_linear`), an injected module wrapping one (`orig_module`), or a (merged operator, split) tuple."""
    from ktransformers_amd._native import LinearHandle

    out = []
    for o in objs:
        seen = 0
        while o is not None and seen < 4:
            seen += 1
            if isinstance(o, (tuple, list)):
                o = o[0] if o else None
                continue
            if isinstance(o, LinearHandle):
                if getattr(o, "_h", None):
                    out.append(o)
                break
            nxt = getattr(o, "_h", None)
            if nxt is None:
                nxt = getattr(o, "generate_linear", None)
            o = nxt
    return out


class LayerPlan:
    """What one decoder layer reads ahead: `early` (set 1, its own later launches) and `head` (what the PREVIOUS layer's set 2
    fetches for it: the first launches of its attention)."""

    def __init__(self, layer):
        attn, mlp = layer.self_attn, layer.mlp
        qa, oa = (None, None)
        if hasattr(attn, "get_absorbed"):
            try:
                qa, oa = attn.get_absorbed()
            except Exception:
                qa, oa = None, None
        first = getattr(attn, "_qkv", None)
        orig = getattr(attn, "orig_module", attn)
        head = _handles(first) if first is not None else _handles(getattr(orig, "q_a_proj", None) or getattr(orig, "q_proj", None),
                                                                     getattr(orig, "kv_a_proj_with_mqa", None))
        self.head = head + _handles(getattr(orig, "q_b_proj", None), qa)
        self.early = _handles(oa, getattr(orig, "o_proj", None))
        self.tensors = []
        shared = getattr(mlp, "shared_experts", None)
        if shared is not None:                      # MoE block: router weight + the shared experts (not the routed ones)
            gate = getattr(mlp, "gate", None)
            w = getattr(getattr(gate, "orig_module", gate), "weight", None)
            if isinstance(w, torch.Tensor) and w.is_cuda and w.is_contiguous():
                self.tensors.append(w)
            so = getattr(shared, "orig_module", shared)
            self.early += _handles(getattr(shared, "_gate_up", None), getattr(so, "down_proj", None))

    def run(self, which: str, stream, wgs: int) -> None:
        from ktransformers_amd._native import prefetch_tensor

        for h in (self.early if which == "early" else self.head):
            h.prefetch(stream, wgs)
        if which == "early":
            for t in self.tensors:
                prefetch_tensor(t, stream, wgs)


class DecodePrefetcher:
    """One per model and device: the side stream, the per-layer plans (built on first use, after the operators have loaded),
    fork / join around the layers of ONE decode step."""

    def __init__(self, layers):
        self.layers = list(layers)
        self.plans: dict = {}
        self.side = None
        self.forked = False

    def _plan(self, i: int) -> LayerPlan:
        p = self.plans.get(i)
        if p is None:
            p = self.plans[i] = LayerPlan(self.layers[i])
        return p

    def _fork(self, dev):
        if self.side is None or self.side.device != dev:
            self.side = torch.cuda.Stream(device=dev)
        self.side.wait_stream(torch.cuda.current_stream(dev))
        self.forked = True
        return self.side

    def layer_start(self, i: int, dev, mode: str) -> None:
        if mode in ("1", "s1", "both"):
            self._plan(i).run("early", self._fork(dev), _wgs())

    def attention_done(self, i: int, dev, mode: str) -> None:
        if mode in ("1", "s2", "both") and i + 1 < len(self.layers):
            self._plan(i + 1).run("head", self._fork(dev), _wgs())

    def join(self, dev) -> None:
        if self.forked and self.side is not None:
            torch.cuda.current_stream(dev).wait_stream(self.side)
        self.forked = False

"""Dev probe for PMC passes: V2-Lite-shaped experts, T=2048 prompt, uniform routing; runs the chunk-pipelined (knob 1) and the
streaming (knob 2) grouped GEMMs back to back so one rocprofv3 pass sees both kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native

dev = torch.device("cuda", 0)
E, H, I, k, T = 64, 2048, 1408, 6, 2048
g = torch.Generator(device="cpu").manual_seed(1)
mk = lambda *s: (torch.randn(*s, generator=g) * 0.03).to(torch.bfloat16).to(dev)
h = _native.MoEHandle(E, k, H, I, max_len=T, method="AMXINT4", device=dev)
h.load_bf16(mk(E, I, H), mk(E, I, H), mk(E, H, I))
x = torch.randn(T, H, generator=g).to(torch.bfloat16).to(dev)
ids = torch.multinomial(torch.ones(T, E), k, generator=g).to(torch.int64).to(dev)
w = torch.rand(T, k, generator=g).to(dev)
y = torch.empty(T, H, dtype=torch.bfloat16, device=dev)
for knob in (1, 2):
    _native.lib.ktx_debug_set(4, knob)
    for _ in range(5):
        h.forward(x, ids, w, out=y)
    torch.cuda.synchronize()
_native.lib.ktx_debug_set(4, 0)

"""Dev probe: chunk-pipelined vs streaming vs register-tile grouped GEMM (ktx_debug_set(4, 1|2|3)) — bit-exact comparison +
per-kernel HIP-event timing."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def run_case(name, E, H, I, k, T, fmt, reps=6, uniform=False):
    g = torch.Generator(device="cpu").manual_seed(1)
    gate = (torch.randn(E, I, H, generator=g) * 0.03).to(torch.bfloat16).to(dev)
    up = (torch.randn(E, I, H, generator=g) * 0.03).to(torch.bfloat16).to(dev)
    down = (torch.randn(E, H, I, generator=g) * 0.03).to(torch.bfloat16).to(dev)
    h = _native.MoEHandle(E, k, H, I, max_len=T, method=fmt, device=dev)
    h.load_bf16(gate, up, down)
    del gate, up, down
    x = torch.randn(T, H, generator=g).to(torch.bfloat16).to(dev)
    # skewed routing: some experts get few rows, some many (ragged tiles)
    p = torch.rand(E, generator=g) ** 2 + 0.02
    if uniform:
        p = torch.ones(E)
    ids = torch.multinomial(p.expand(T, E), k, generator=g).to(torch.int64).to(dev)
    w = torch.rand(T, k, generator=g).to(dev)
    outs = {}
    for knob in (1, 2, 3):   # chunk-pipelined, streaming, register-tile
        _native.lib.ktx_debug_set(4, knob)
        y = torch.empty(T, H, dtype=torch.bfloat16, device=dev)
        for _ in range(2):
            h.forward(x, ids, w, out=y)
        torch.cuda.synchronize()
        _native.profile_enable(True)
        _native.profile_collect()
        for _ in range(reps):
            h.forward(x, ids, w, out=y)
        prof = _native.profile_collect()
        _native.profile_enable(False)
        outs[knob] = y.clone()
        s = " ".join(f"{n}={ms / max(c, 1) * 1e3:.1f}us" for n, (ms, c) in prof.items())
        tot = sum(ms / max(c, 1) for ms, c in prof.values())
        print(f"{name} knob={knob}: {s} total={tot * 1e3:.1f}us  ({2 * 3 * H * I * k * T / tot / 1e9:.0f} TOP/s)", flush=True)
    same = all(torch.equal(outs[1].view(torch.int16), outs[kk].view(torch.int16)) for kk in (2, 3))
    print(f"{name}: bit-exact={same} finite={bool(torch.isfinite(outs[2].float()).all())}", flush=True)
    _native.lib.ktx_debug_set(4, 0)
    return same


ok = True
ok &= run_case("v2lite-int4 T=2048", 64, 2048, 1408, 6, 2048, "AMXINT4")
ok &= run_case("v2lite-int4 T=2048 uniform", 64, 2048, 1408, 6, 2048, "AMXINT4", uniform=True)
ok &= run_case("v2lite-int4 T=600", 64, 2048, 1408, 6, 600, "AMXINT4")
ok &= run_case("v3shape-int4 E=16 T=1024", 16, 7168, 2048, 8, 1024, "AMXINT4")
print("ALL_OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

"""Dev tool: the grouped (prompt) expert kernels of one non-GGUF format at the V3 / K2 expert shape with the library's five stage
timers.    python scripts/fmt_prompt_bench.py --fmt FP8|RAWINT4|AMXINT4|BF16 [--experts 256] [--T 2048] [--layers 1] [--knob IDX=VAL]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native as n

ap = argparse.ArgumentParser()
ap.add_argument("--fmt", default="FP8")
ap.add_argument("--experts", type=int, default=256)
ap.add_argument("--T", type=int, default=2048)
ap.add_argument("--layers", type=int, default=1)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--knob", action="append", default=[])
a = ap.parse_args()
dev = torch.device("cuda", 0)
for kv in a.knob:
    n.lib.ktx_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
E_, k, H, I, T = a.experts, 8, 7168, 2048, a.T
g = torch.Generator(device=dev); g.manual_seed(0)
rb = lambda *s: torch.randint(0, 256, s, generator=g, device=dev, dtype=torch.uint8)
hs = []
for _ in range(a.layers):
    if a.fmt in ("AMXINT4", "BF16"):
        h = n.MoEHandle(E_, k, H, I, max_len=T, method=a.fmt, device=0)
        mk = lambda *s: (torch.randn(s, generator=g, device=dev, dtype=torch.bfloat16) * 0.1)
        h.load_bf16(mk(E_, I, H), mk(E_, I, H), mk(E_, H, I))
    elif a.fmt == "FP8":
        h = n.MoEHandle(E_, k, H, I, max_len=T, method="FP8", device=0, group_size=128)
        fp8 = lambda *s: (torch.randn(s, generator=g, device=dev) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
        sc = lambda r, c: torch.rand((E_, r // 128, c // 128), generator=g, device=dev) * 0.01 + 0.001
        h.load_fp8(fp8(E_, I, H), fp8(E_, I, H), fp8(E_, H, I), sc(I, H), sc(I, H), sc(H, I))
    else:
        h = n.MoEHandle(E_, k, H, I, max_len=T, method="RAWINT4", device=0, group_size=32)
        sc = lambda r, c: (torch.rand((E_, r, c // 32), generator=g, device=dev) * 0.01 + 0.001).to(torch.bfloat16)
        h.load_rawint4(rb(E_, I, H // 2), rb(E_, I, H // 2), rb(E_, H, I // 2), sc(I, H), sc(I, H), sc(H, I))
    hs.append(h)
x = (torch.randn((T, H), generator=g, device=dev) * 0.5).to(torch.bfloat16)
ids = torch.multinomial(torch.ones(T, E_), k, generator=torch.Generator().manual_seed(1)).to(torch.int64).to(dev)   # (device randperm dies under rocprofv3 --pmc)
w = torch.rand((T, k), generator=g, device=dev)
for h in hs:
    h.forward(x, ids, w)
torch.cuda.synchronize()
n.profile_enable(True); n.profile_collect()
for _ in range(a.iters):
    for h in hs:
        h.forward(x, ids, w)
torch.cuda.synchronize()
prof = n.profile_collect(); n.profile_enable(False)
print(f"knobs {a.knob} {a.fmt} E={E_} T={T} ({T * k / E_:.0f} rows/expert):", "  ".join(f"{nm} {ms / max(c, 1):.3f} ms" for nm, (ms, c) in prof.items()))

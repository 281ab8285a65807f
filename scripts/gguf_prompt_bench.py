"""Dev tool / profile artefact: the grouped (prompt) GGUF expert kernels alone — T tokens through L layers of routed experts at a given
shape, with the library's five stage timers (prep+Q8_K, gate|up, Q8_K(a), down, combine) and the int8-MFMA fraction of each GEMM.
Random valid weight bytes (timing only).
    python scripts/gguf_prompt_bench.py [--shape mixtral|v3] [--types 12,12,14] [--T 2048] [--layers 2] [--iters 5]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native as n

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="mixtral", choices=("mixtral", "v3"))
ap.add_argument("--types", default="12,12,14")
ap.add_argument("--T", type=int, default=2048)
ap.add_argument("--layers", type=int, default=2)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--experts", type=int, default=0)
ap.add_argument("--knob", action="append", default=[], help="idx=val dev knobs (ktx_debug_set), e.g. 21=1: unfolded kernels, 22=1: 64-row tiles")
args = ap.parse_args()
dev = torch.device("cuda", 0)
for kv in args.knob:
    n.lib.ktx_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
E, k, H, I = (8, 2, 4096, 14336) if args.shape == "mixtral" else (args.experts or 64, 8, 7168, 2048)
if args.experts:
    E = args.experts
T, L = args.T, args.layers
types = tuple(int(t) for t in args.types.split(","))
g = torch.Generator(device=dev); g.manual_seed(0)
bb = n.GGML_BLOCK_BYTES
F16_OFF = {10: (80, 82), 11: (108,), 12: (0, 2), 13: (0, 2), 14: (208,), 19: (0,), 23: (0,)}


def blocks(N, K, ty):
    t = torch.randint(0, 256, (E, N, K // 256, bb[ty]), generator=g, device=dev, dtype=torch.uint8)
    d = (torch.rand((E, N, K // 256), generator=g, device=dev) * 0.004 + 0.001).to(torch.float16).view(torch.uint8)
    for off in F16_OFF[ty]:
        t[..., off:off + 2] = d.view(E, N, K // 256, 2)
    return t.reshape(E, N, -1).contiguous()


hs = []
for _ in range(L):
    h = n.MoEHandle(E, k, H, I, max_len=T, method="GGUF", device=0)
    h.load_gguf(blocks(I, H, types[0]), blocks(I, H, types[1]), blocks(H, I, types[2]), *types)
    hs.append(h)
x = (torch.randn((T, H), generator=g, device=dev) * 0.5).to(torch.bfloat16)
ids = torch.stack([torch.randperm(E, generator=g, device=dev)[:k] for _ in range(T)]).to(torch.int64)
w = torch.rand((T, k), generator=g, device=dev)
for h in hs:
    h.forward(x, ids, w)
torch.cuda.synchronize()
n.profile_enable(True)
n.profile_collect()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.iters):
    for h in hs:
        h.forward(x, ids, w)
e1.record()
torch.cuda.synchronize()
prof = n.profile_collect()
n.profile_enable(False)
calls = args.iters * L
ms_layer = e0.elapsed_time(e1) / calls
print(f"knobs {args.knob} " + f"shape {args.shape} E={E} k={k} H={H} I={I} T={T} types={types}: {ms_layer:.3f} ms/layer "
      f"({2 * 3 * H * I * k * T / ms_layer / 1e9:.0f} TOP/s whole layer, {T * k / E:.0f} rows/expert)")
flops = {"gate_up": 2 * 2 * H * I * k * T, "down": 2 * H * I * k * T}
for name, (ms, cnt) in prof.items():
    per = ms / max(cnt, 1)
    key = "gate_up" if "gate" in name or "up" in name else ("down" if "down" in name else None)
    extra = f"  {flops[key] / per / 1e9:8.0f} TOP/s = {flops[key] / per / 1e9 / 5000:.4f} of int8 peak" if key and per > 0 else ""
    print(f"  {name:12s} {per:8.3f} ms x {cnt}{extra}")

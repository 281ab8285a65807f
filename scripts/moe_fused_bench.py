"""A/B of the MoE half of a DeepSeek-V3 decode step: ONE persistent launch (csrc/ktx_moe_layer.inc) against the three-launch chain
(router || shared gate|up, routed gate/up, routed down + shared down + adds), L distinct layers chained in one captured graph.
python scripts/moe_fused_bench.py [layers=6]"""
import subprocess
import sys
import time

import torch

sys.path.insert(0, ".")
from ktransformers_amd import _native as N  # noqa: E402

E, K, H, I = 256, 8, 7168, 2048
L = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)


def u(shape, scale):
    return ((torch.rand(shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1) * scale).to(torch.bfloat16)


print(subprocess.run("/opt/rocm/bin/rocm-smi --showuniqueid | grep Unique", shell=True, capture_output=True, text=True).stdout.strip())
layers = []
base = [u((32, I, H), 0.02), u((32, I, H), 0.02), u((32, H, I), 0.04)]
for l in range(L):
    o = {}
    ex = N.MoEHandle(E, K, H, I, max_len=8, method="AMXINT4", device=0)
    gw = torch.cat([torch.roll(base[0], l * 8 + i, 0) for i in range(8)])
    uw = torch.cat([torch.roll(base[1], l * 8 + i + 3, 1) for i in range(8)])
    dw = torch.cat([torch.roll(base[2], l * 8 + i + 5, 1) for i in range(8)])
    ex.load_bf16(gw, uw, dw)
    del gw, uw, dw
    o["experts"] = ex
    o["sgu"] = N.LinearHandle(H, 2 * I, "W4", 64, 8, dev); o["sgu"].load_bf16(u((2 * I, H), 0.02))
    o["sdown"] = N.LinearHandle(I, H, "W4", 64, 8, dev); o["sdown"].load_bf16(u((H, I), 0.04))
    o["gate"] = N.GateHandle(E, H, K, 8, 4, "sigmoid", "noaux_tc", True, 2.5)
    o["gate_w"] = u((E, H), 0.05)
    o["gate_b"] = ((torch.rand(E, generator=g, device=dev) - 0.5) * 0.2).float().contiguous()
    o["norm_w"] = (1 + u((H,), 0.2).float()).to(torch.bfloat16)
    layers.append(o)
torch.cuda.empty_cache()
xs = [u((1, H), 1.0)] + [torch.zeros((1, H), dtype=torch.bfloat16, device=dev) for _ in range(L)]


def three(l):
    o = layers[l]
    idx, wt, xn, act = N.gate_with_linear(o["gate"], o["sgu"], xs[l], o["gate_w"], o["gate_b"], (o["norm_w"], 1e-6), glu=True)
    o["experts"].forward_side(xn, idx, wt, o["sdown"], act, residual=xs[l], out=xs[l + 1])


ARGS = [N.moe_layer_args(o["experts"], o["sgu"], o["sdown"], o["gate"], o["gate_w"], o["gate_b"], xs[l].reshape(-1), xs[l + 1].reshape(-1),
                         (o["norm_w"], 1e-6)) for l, o in enumerate(layers)]


def fused(l, chain=(7,)):
    for i, ph in enumerate(chain):
        N.moe_layer_decode(ARGS[l], dev, phases=ph, last=(i == len(chain) - 1))


def capture(fn):
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for l in range(L):
            fn(l)
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for l in range(L):
            fn(l)
    return gr


def timeit(gr, reps=30):
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps / L * 1e6)
    return best


cfgs = [("three launches", three), ("one launch", lambda l: fused(l)), ("phases as 3 launches", lambda l: fused(l, (1, 2, 4))),
        ("1 + 6", lambda l: fused(l, (1, 6))), ("3 + 4", lambda l: fused(l, (3, 4)))]
graphs = [(name, capture(fn)) for name, fn in cfgs]
ref_y = None
for rnd in range(2):
    for name, gr in graphs:
        us = timeit(gr)
        y = xs[L].clone()
        if ref_y is None:
            ref_y = y
        print(f"{name:24s} {us:8.2f} us per layer   output == three launches: {bool(torch.equal(y.view(torch.int16), ref_y.view(torch.int16)))}"
              f"   status {N.moe_layer_status(dev):#x}", flush=True)
st = torch.zeros(64, dtype=torch.int64, device=dev)
N.lib.ktx_moe_layer_debug_stamps(st.data_ptr())
for l in range(L):
    fused(l)
torch.cuda.synchronize()
N.lib.ktx_moe_layer_debug_stamps(None)
t = st.tolist()
names = ["entry", "row normalised", "F: logit / slice done", "F: shared strip published", "selection done", "G done", "H: activations polled", "H done"]
print("stamps of workgroup 0, last layer (us from entry):")
for i, nme in enumerate(names):
    if t[i]:
        print(f"  {nme:28s} {(t[i] - t[0]) * 0.01:7.2f}")

"""Dev tool: time one MoE layer forward (HIP graph of L distinct layers, cycled) for a given shape and batch."""
import argparse, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS, build_layers, DecodeRunner

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="v2lite-int4")
ap.add_argument("--layers", type=int, default=0)
ap.add_argument("--T", type=int, default=1)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--dbg", default="", help="idx=val,idx=val")
args = ap.parse_args()
wl = dict(WORKLOADS[args.workload])
if args.layers: wl["L"] = args.layers
dev = torch.device("cuda", 0)
from ktransformers_amd import _native
for kv in filter(None, args.dbg.split(",")):
    i, val = kv.split("="); _native.lib.ktx_debug_set(int(i), int(val))
layers = build_layers(wl, dev, max_len=max(args.T, 1))
r = DecodeRunner(wl, layers, T=args.T, dev=dev)
if not args.no_graph: r.capture()
for i in range(10): r.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.steps): r.step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
H, I, E, k, L = wl["H"], wl["I"], wl["E"], wl["k"], wl["L"]
import math
U = k if args.T == 1 else E * (1 - (1 - k / E) ** args.T)
bytes_layer = U * 3 * H * I * 0.5
print(f"{args.workload} T={args.T} L={L} dbg[{args.dbg}]: {dt*1e6/L:.2f} us/layer, {bytes_layer/(dt/L)/1e9:.0f} GB/s algorithmic, "
      f"{2*3*H*I*k*args.T/(dt/L)/1e12:.1f} TFLOP/s, step {dt*1e3:.3f} ms", flush=True)

"""Dev probe: routed-expert prefill chunk (T tokens through L V2-Lite layers); used for the PMC pass in profiles/r01_pmc_prefill.json."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ktransformers_amd import _native

dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS["v2lite-int4"])
wl["L"] = int(os.environ.get("L", "4"))
T = int(os.environ.get("T", "2048"))
layers = bench.build_layers(wl, dev, max_len=T)
for knob in (0,):
    rp = bench.DecodeRunner(wl, layers, T=T, dev=dev, nsets=2, seed=5)
    for i in range(3):
        rp.step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for i in range(n):
        rp.step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    H, I, k = wl["H"], wl["I"], wl["k"]
    print(f"knob {knob}: T={T} L={wl['L']}: {dt*1e3/wl['L']:.3f} ms/layer, {2*3*H*I*k*T*wl['L']/dt/1e12:.0f} TOP/s")

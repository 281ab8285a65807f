"""Dev probe: is the whole-model decode graph host-bound?  Per replay: host time to return from graph.replay() vs GPU time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
mr = bench.ModelDecodeRunner(dev, 4096, 600)
g = mr.runner.graph
for i in range(50):
    mr.step(i)
torch.cuda.synchronize()
host, total = [], []
for i in range(30):
    t0 = time.perf_counter()
    g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    total.append(t2 - t0)
host.sort(); total.sort()
print(f"graph.replay() host return: median {host[15]*1e6:.0f} us, min {host[0]*1e6:.0f}; replay+sync median {total[15]*1e6:.0f} us")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(100):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"100 back-to-back replays: {e0.elapsed_time(e1)*10:.0f} us per replay (GPU timeline)")
t0 = time.perf_counter()
for i in range(100):
    mr.step(i)
torch.cuda.synchronize()
print(f"100 bench steps: {(time.perf_counter()-t0)*1e4:.0f} us per step")
print("cpu count", os.cpu_count(), "loadavg", os.getloadavg())

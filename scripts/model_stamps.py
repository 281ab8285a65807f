"""Dev probe: the decode GEMVs' phase stamps INSIDE the whole-model decode graph of bench.py (same model, same graph).
Prints one line per (in, out) shape: averages over the layers of the last replays."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ktransformers_amd import _native as n

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
wl = bench.WORKLOADS["v3-int4"]
layers = int(os.environ.get("LAYERS", "16"))
n.lib.ktx_debug_set_ptr.argtypes = [C.c_int, C.c_void_p]
mr = bench.ModelDecodeRunner(wl, layers, dev, 4096, 256, use_graph=False)
NREC = 4096
stamps = torch.zeros(NREC * 16, dtype=torch.int64, device=dev)
n.lib.ktx_debug_set_ptr(0, C.c_void_p(stamps.data_ptr()))
mr.capture(True)
n.lib.ktx_debug_set_ptr(0, None)
assert mr.graph_ok, mr.graph_error
acc, reps = None, 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for r in range(reps + 20):
    if r >= 20:
        stamps.view(NREC, 16)[:, 0] = torch.iinfo(torch.int64).max
        stamps.view(NREC, 16)[:, 1] = 0
        torch.cuda.synchronize()
        e0.record()
    mr.step()
    if r >= 20:
        e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
        s = stamps.view(NREC, 16).cpu()
        acc = s.double() if acc is None else acc + s.double()
s = acc / reps
shape = stamps.view(NREC, 16)[:, 8].cpu()
print(f"{layers} layers, graph step {tot / reps:.3f} ms")
# the capture warm-up ran the step eagerly first: the graph's launches are the LAST records that were written
used = [i for i in range(NREC) if int(shape[i]) != 0 and s[i, 1] > 0]
by = {}
for i in used:
    K, N = int(shape[i]) >> 32, int(shape[i]) & 0xffffffff
    by.setdefault((K, N), []).append(i)
for (K, N), idx in by.items():
    t = s[idx] / 100.0
    span = (t[:, 1] - t[:, 0]).mean().item()
    ph = [(t[:, b] - t[:, a]).mean().item() for a, b in ((0, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 1))]
    mhz = ((s[idx, 12] - s[idx, 11]) / ((s[idx, 7] - s[idx, 2]) / 100.0)).mean().item() if (s[idx, 12] > 0).all() else float("nan")
    print(f"W4 {K}->{N} x{len(idx)}: shader clock {mhz:5.0f} MHz | span {span:5.2f} | wg0: entry+{ph[0]:.2f} stage {ph[1]:.2f} sync {ph[2]:.2f} stream {ph[3]:.2f} "
          f"reduce {ph[4]:.2f} out {ph[5]:.2f} | tail {ph[6]:.2f}")

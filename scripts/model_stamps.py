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
cu_map = torch.zeros(NREC * 1024, dtype=torch.int32, device=dev)
n.lib.ktx_debug_set_ptr(1, C.c_void_p(cu_map.data_ptr()))
n.lib.ktx_debug_set_ptr(0, C.c_void_p(stamps.data_ptr()))
mr.capture(True)
n.lib.ktx_debug_set_ptr(0, None)
n.lib.ktx_debug_set_ptr(1, None)
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
    cm = cu_map.view(NREC, 1024)[idx].cpu()
    occ = []
    for row in cm:
        ids = row[row != 0] & 0x7fffffff
        if len(ids):
            _, counts = torch.unique(ids, return_counts=True)
            occ.append((len(ids), len(counts), int(counts.max())))
    if occ:
        print(f"      placement: {occ[0][0]} workgroups on {sum(o[1] for o in occ) / len(occ):.0f} CUs, at most {max(o[2] for o in occ)} per CU")
    print(f"W4 {K}->{N} x{len(idx)}: shader clock {mhz:5.0f} MHz | span {span:5.2f} | wg0: entry+{ph[0]:.2f} stage {ph[1]:.2f} sync {ph[2]:.2f} stream {ph[3]:.2f} "
          f"reduce {ph[4]:.2f} out {ph[5]:.2f} | tail {ph[6]:.2f}")

# ---- the isolated q_a|kv_a chain of scripts/lin_stamps.py, but INSIDE this process (152 GiB resident): does the footprint
# of the process change a kernel that is fast in a small process?
if os.environ.get("CHAIN_IN_PROCESS", "1") == "1":
    K, N, L = 7168, 2112, 32
    hs = []
    for i in range(L):
        h = n.LinearHandle(K, N, "W4", 64, 4, dev)
        h.load_bf16((torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16))
        hs.append(h)
    xs = [torch.randn(1, K, device=dev).to(torch.bfloat16) for _ in range(2)]
    y = torch.empty(1, N, device=dev, dtype=torch.bfloat16)
    nw = torch.ones(K, device=dev, dtype=torch.bfloat16)
    st2 = torch.zeros(L * 16, dtype=torch.int64, device=dev)

    def run():
        for i, h in enumerate(hs):
            h.forward(xs[i & 1], out=y, norm=(nw, 1e-6))
            xs[(i + 1) & 1][:, :N].copy_(y)

    run()
    torch.cuda.synchronize()
    n.lib.ktx_debug_set_ptr(0, C.c_void_p(st2.data_ptr()))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run()
    n.lib.ktx_debug_set_ptr(0, None)
    acc2 = None
    for r in range(6):
        st2.view(L, 16)[:, 0] = torch.iinfo(torch.int64).max
        st2.view(L, 16)[:, 1] = 0
        mr.step()                      # a whole model step first: caches / TLBs in the model's state
        gr.replay()
        torch.cuda.synchronize()
        if r:
            v = st2.view(L, 16).cpu().double() / 100.0
            acc2 = v if acc2 is None else acc2 + v
    v = acc2 / 5
    print(f"in-process isolated chain W4 {K}->{N} +norm: span {(v[:, 1] - v[:, 0])[1:].mean().item():.2f} us "
          f"(stage {(v[:, 3] - v[:, 2])[1:].mean().item():.2f}, stream {(v[:, 5] - v[:, 4])[1:].mean().item():.2f}, out {(v[:, 7] - v[:, 6])[1:].mean().item():.2f})")

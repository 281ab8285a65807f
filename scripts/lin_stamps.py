"""Dev probe: phase stamps of the decode GEMV (lin_dec_kernel) inside a graph chain that looks like the model's: L distinct
matrices per shape, each launch reading the previous one's output row, optional fused RMSNorm, the L3 flushed before a
replay.  Prints, per shape, averages over the chain (µs; wall_clock64 at 100 MHz):
    span      first workgroup entry -> last workgroup exit (all workgroups)
    gap       previous launch's last exit -> this launch's first entry  (the boundary)
    wg0: stage / sync / stream / reduce / out      phases of workgroup 0
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native as n

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="7168x2112n,7168x2112,16384x7168,7168x4096n,1536x24576n,7168x36864n,18432x7168")
ap.add_argument("--sw", type=int, default=0)
ap.add_argument("--direct", action="store_true", help="square shapes: feed y straight back as the next x (scattered 32-byte writes from many workgroups), no copy kernel")
ap.add_argument("--knobs", default="", help="ktx_debug_set pairs, e.g. 17=1,18=1")
ap.add_argument("--thrash", action="store_true", help="evict the instruction caches between launches (two 48 KB kernels)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
n.lib.ktx_debug_set_ptr.argtypes = [C.c_int, C.c_void_p]
n.lib.ktx_debug_icache_thrash.argtypes = [C.c_void_p, C.c_void_p]
scratch = torch.zeros(4096, device=dev)
for kv in filter(None, args.knobs.split(",")):
    k, v = kv.split("=")
    n.lib.ktx_debug_set(int(k), int(v))
print(f"knobs: {args.knobs or '(defaults)'}")
flush = torch.zeros(512 << 20, dtype=torch.int8, device=dev)
for sh in args.shapes.split(","):
    norm = sh.endswith("n")
    K, N = (int(v) for v in sh.rstrip("n").split("x"))
    mb = K * N * 0.5625 / 1e6
    L = max(4, min(32, int(400 / mb) + 1))
    hs = []
    for i in range(L):
        h = n.LinearHandle(K, N, "W4", 64, 4, dev)
        h.load_bf16((torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16))
        hs.append(h)
    xs = [torch.randn(1, K, device=dev).to(torch.bfloat16) for _ in range(2)]
    y = torch.empty(1, N, device=dev, dtype=torch.bfloat16)
    nw = torch.ones(K, device=dev, dtype=torch.bfloat16)
    stamps = torch.zeros(L * 16, dtype=torch.int64, device=dev)
    n.lib.ktx_debug_set(8, args.sw)

    ys = [torch.empty(1, N, device=dev, dtype=torch.bfloat16) for _ in range(2)]

    def run_direct():
        cur = xs[0]
        for i, h in enumerate(hs):
            h.forward(cur, out=ys[i & 1], norm=(nw, 1e-6) if norm else None)
            cur = ys[i & 1]

    def run():
        if args.direct and K == N:
            return run_direct()
        for i, h in enumerate(hs):
            h.forward(xs[i & 1], out=y, norm=(nw, 1e-6) if norm else None)
            xs[(i + 1) & 1][:, :min(K, N)].copy_(y[:, :min(K, N)])     # dependent chain (a tiny copy kernel in between)
            if args.thrash:
                n.lib.ktx_debug_icache_thrash(C.c_void_p(scratch.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))

    run()
    torch.cuda.synchronize()
    n.lib.ktx_debug_set_ptr(0, C.c_void_p(stamps.data_ptr()))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run()
    n.lib.ktx_debug_set_ptr(0, None)
    acc = None
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for r in range(reps + 1):
        stamps.view(L, 16)[:, 0] = torch.iinfo(torch.int64).max
        stamps.view(L, 16)[:, 1] = 0
        flush.add_(1)
        e0.record(); gr.replay(); e1.record(); e1.synchronize()
        if r:
            tot += e0.elapsed_time(e1)
            s = stamps.view(L, 16).cpu().double() / 100.0
            acc = s if acc is None else acc + s
    s = acc / reps
    span = (s[:, 1] - s[:, 0])[1:].mean().item()
    gap = (s[1:, 0] - s[:-1, 1]).mean().item()
    ph = [(s[:, b] - s[:, a])[1:].mean().item() for a, b in ((0, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 1))]
    if norm:
        print(f"      x arrived + sumsq (wave 0): {(s[:, 9] - s[:, 2])[1:].mean().item():.2f}   norm barrier: {(s[:, 10] - s[:, 9])[1:].mean().item():.2f}")
    print(f"{'[icache thrashed] ' if args.thrash else ''}{'[y -> x direct] ' if args.direct and K == N else ''}W4 {K}->{N}{' +norm' if norm else ''} ({mb:.1f} MB, L={L}): chain {tot / reps / L * 1e3:6.2f} us/launch | span {span:5.2f} "
          f"gap(+copy kernel) {gap:5.2f} | wg0: entry+{ph[0]:.2f} stage {ph[1]:.2f} sync {ph[2]:.2f} stream {ph[3]:.2f} "
          f"reduce {ph[4]:.2f} out {ph[5]:.2f} | tail (wg0 end -> last exit) {ph[6]:.2f}", flush=True)
    n.lib.ktx_debug_set(8, 0)
    del hs

"""Dev tool: time the decode GEMV (lin_dec_kernel) for the DeepSeek-V3 linear shapes, sweeping the strips-per-workgroup split
(knob 8).  L distinct matrices per shape (> the 256 MB L3 in total), one HIP graph, L3 flushed before every replay."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native as n

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="7168x2112,1536x24576,16384x7168,7168x4096,2048x7168,7168x36864,18432x7168")
args = ap.parse_args()
dev = torch.device("cuda", 0)
flush = torch.zeros(512 << 20, dtype=torch.int8, device=dev)
for sh in args.shapes.split(","):
    K, N = (int(v) for v in sh.split("x"))
    mb = K * N * 0.5625 / 1e6
    L = max(4, min(48, int(400 / mb) + 1))
    hs = []
    for i in range(L):
        h = n.LinearHandle(K, N, "W4", 64, 4, dev)
        h.load_bf16((torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16))
        hs.append(h)
    x = torch.randn(1, K, device=dev).to(torch.bfloat16)
    y = torch.empty(1, N, device=dev, dtype=torch.bfloat16)
    for sw in (0, 1, 2, 4, 8):
        n.lib.ktx_debug_set(8, sw)
        for h in hs: h.forward(x, out=y)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for h in hs: h.forward(x, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for r in range(6):
            flush.add_(1); e0.record(); gr.replay(); e1.record(); e1.synchronize()
            if r: tot += e0.elapsed_time(e1)
        us = tot / 5 / L * 1e3
        print(f"W4 {K}->{N} ({mb:.1f} MB) SW={sw or 'auto'}: {us:7.2f} us  {mb / us:5.2f} TB/s", flush=True)
    n.lib.ktx_debug_set(8, 0)
    del hs

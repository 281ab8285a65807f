"""Dev tool: the W4 prompt GEMM (T = 2048) on the DeepSeek-V3 linear shapes, one strip per wavefront (lin_gemm_kernel) against
2 / 4 strips (lin_gemm_w4n_kernel, knob 12), in TFLOP/s of the bf16 MFMA work.

    python scripts/lin_prompt_sweep.py [--T 2048]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native as n

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=2048)
args = ap.parse_args()
dev = torch.device("cuda", 0)
SHAPES = [(7168, 2112), (1536, 24576), (16384, 7168), (7168, 4096), (2048, 7168), (7168, 36864), (18432, 7168)]
for K, N in SHAPES:
    h = n.LinearHandle(K, N, "W4", 64, args.T, dev)
    h.load_bf16((torch.randn(N, K, device=dev) / 10).to(torch.bfloat16))
    x = (torch.randn(args.T, K, device=dev) / 10).to(torch.bfloat16)
    line = f"W4 {K}->{N} T={args.T}:"
    for knob in (1, 2, 4, 0):
        n.lib.ktx_debug_set(12, knob)
        for _ in range(3):
            h.forward(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            h.forward(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"  {'auto' if knob == 0 else 'NSW=%d' % knob} {ms * 1e3:7.1f} us {2.0 * args.T * K * N / ms / 1e9:6.1f} TF"
    n.lib.ktx_debug_set(12, 0)
    print(line, flush=True)
    h.close() if hasattr(h, "close") else None

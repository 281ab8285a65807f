"""Dev tool: the block-fp8 linear prompt GEMM (lin_fp8_gemm_kernel) with one K = 128 MFMA per block (default) against the four K = 32
MFMAs of rounds 4-5 (dev knob 27 = 1) on the DeepSeek-V3 linear shapes: time, TFLOP/s, and how far the two are apart.
    python scripts/fp8_gemm_ab.py [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native as n

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for N, K in [(2112, 7168), (24576, 1536), (7168, 16384), (4096, 7168), (7168, 2048), (36864, 7168)]:
    w = (torch.randn(N, K, device=dev) / 4).to(torch.float8_e4m3fn)
    sc = (torch.rand((N + 127) // 128, K // 128, device=dev) + 0.5) / 32
    x = (torch.randn(T, K, device=dev) / 10).to(torch.bfloat16)
    h = n.LinearHandle(K, N, "FP8", 128, T)
    h.load_fp8(w, sc)
    res = {}
    for knob in (0, 1, 0, 1):
        n.check(n.lib.ktx_debug_set(27, knob))
        for _ in range(3):
            y = h.forward(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = h.forward(x)
        e1.record(); torch.cuda.synchronize()
        res[knob] = (min(res.get(knob, (1e9,))[0], e0.elapsed_time(e1) / 10), y.float())
    n.check(n.lib.ktx_debug_set(27, 0))
    a, b = res[0][1], res[1][1]
    fl = 2.0 * T * N * K
    print(f"T={T} N={N} K={K}: K=128 MFMA {res[0][0]:.3f} ms ({fl / res[0][0] / 1e9:.0f} TF/s)   4 x K=32 {res[1][0]:.3f} ms ({fl / res[1][0] / 1e9:.0f} TF/s)   "
          f"outputs differing {float((a != b).float().mean()):.4f}, max |diff| / max |y| {float((a - b).abs().max() / b.abs().max()):.2e}", flush=True)

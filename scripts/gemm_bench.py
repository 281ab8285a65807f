"""Prompt-sized GEMM shapes of one DeepSeek-V3 chunk (T = 2048): the library's own kernel (both LDS-stage variants) beside the
vendor GEMM it replaces (torch F.linear -> hipBLASLt).  TFLOP/s = 2*M*N*K / time, random operands (the guide's rule: never quote
zero-filled operands), median of 5 windows of 20 launches on the current stream.

    python scripts/gemm_bench.py [T]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ktransformers_amd._native import gemm_bf16_nt

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
SHAPES = [("q_a|kv_a 7168->2112", 2112, 7168), ("q_b 1536->24576", 24576, 1536), ("o_proj 16384->7168", 7168, 16384),
          ("shared gate|up 7168->4096", 4096, 7168), ("shared down 2048->7168", 7168, 2048), ("dense gate|up 7168->36864", 36864, 7168),
          ("dense down 18432->7168", 7168, 18432), ("router planes 7168->768", 768, 7168)]
dev = torch.device("cuda", 0)


def timed(fn, reps=20, windows=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


print(f"T = {T}; TFLOP/s (ms)")
VARIANTS = (1, 2, 3, 4, 5)   # 128x128x64 one stage, 128x128x64 two stages, 256x128x32 two stages, 256x128x64 one stage, 256x256x64 ping-pong (round 6)
print(f"{'shape':34s} " + " ".join(f"{'ktx variant %d' % v:>16s}" for v in VARIANTS) + f" {'torch F.linear':>16s}")
tot = [0.0] * (len(VARIANTS) + 1)
for name, N, K in SHAPES:
    x = torch.randn((T, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
    out = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
    fl = 2.0 * T * N * K
    ms = [timed(lambda v=v: gemm_bf16_nt(x, w, out=out, variant=v)) for v in VARIANTS] + [timed(lambda: torch.nn.functional.linear(x, w))]
    for i in range(len(ms)):
        tot[i] += ms[i]
    print(f"{name:34s} " + " ".join(f"{fl / m / 1e9:8.0f} ({m:5.3f})" for m in ms), flush=True)
print(f"{'sum of the rows, ms':34s} " + " ".join(f"{m:16.3f}" for m in tot))
# kv_b expansion of the non-absorbed prompt kernel: 128 heads, latent rows shared
H, kv, lora, d = 128, T, 512, 128
lat = torch.randn((kv, lora), device=dev).to(torch.bfloat16)
wk = (torch.randn((H, d, lora), device=dev) / 22).to(torch.bfloat16)
fl = 2.0 * H * kv * lora * d
for name, f_own, f_lib in (("K_nope = latent @ W_UK[h]^T", lambda v: gemm_bf16_nt(lat, wk, variant=v), lambda: torch.matmul(lat.unsqueeze(0), wk.transpose(1, 2))),
                           ("V^T = W_UV[h] @ latent^T", lambda v: gemm_bf16_nt(wk, lat, variant=v), lambda: torch.matmul(wk, lat.t().unsqueeze(0)))):
    ms = [timed(lambda v=v: f_own(v)) for v in VARIANTS] + [timed(f_lib)]
    print(f"{name:34s} " + " ".join(f"{fl / m / 1e9:8.0f} ({m:5.3f})" for m in ms), flush=True)

"""Non-absorbed prompt attention (ktx_mla_prefill) at the chunk shape of bench.py's prefill leg: T queries = the last T of kv_len
keys, 128 heads, qk 192 / v 128, causal.  Times the default kernel (two workgroups per CU) against dev knob 22 = 1 (the
unconstrained register allocation, one workgroup per CU) and checks both against each other.

    python scripts/mla_prefill_bench.py [T] [kv_len]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ktransformers_amd import _native as n  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
kv_len = int(sys.argv[2]) if len(sys.argv) > 2 else T
H, dev = 128, torch.device("cuda", 0)
kv_pad = (kv_len + 63) // 64 * 64
q = torch.randn((T, H, 192), device=dev).to(torch.bfloat16)
q_pe = q[:, :, 128:].contiguous()
k_nope = torch.randn((H, kv_pad, 128), device=dev).to(torch.bfloat16)
cache = torch.randn((kv_pad, 576), device=dev).to(torch.bfloat16)
v_t = torch.randn((H, 128, kv_pad), device=dev).to(torch.bfloat16)
scale = 192 ** -0.5
# causal: query t (position kv_len - T + t) sees kv_len - T + t + 1 keys
pairs = sum(kv_len - T + t + 1 for t in range(T))
flop = 2.0 * pairs * H * (192 + 128)


def run():
    return n.mla_prefill(q[:, :, :128], q_pe, k_nope, cache[:, 512:], v_t, kv_len, scale)


def timed(reps=10, windows=5):
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


outs = {}
for knob, name in ((0, "two workgroups per CU (<= 256 registers)"), (1, "one workgroup per CU (340 registers)")):
    n.check(n.lib.ktx_debug_set(22, knob))
    ms = timed()
    outs[knob] = run().clone()
    print(f"T={T} kv={kv_len} H={H}  {name:44s} {ms:7.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s  ({flop / ms / 1e9 / 2500:.3f} of the bf16 MFMA peak)", flush=True)
n.check(n.lib.ktx_debug_set(22, 0))
print("outputs identical:", bool(torch.equal(outs[0], outs[1])))
# round 5: mask / rescale skipped where they are the identity (default) against round 3's unconditional ones (knob 23 = 1)
n.check(n.lib.ktx_debug_set(23, 1))
ms_old = timed()
old = run().clone()
n.check(n.lib.ktx_debug_set(23, 0))
ms_new = timed()
print(f"mask + rescale unconditional (round 3) {ms_old:7.3f} ms   skipped where identity (default) {ms_new:7.3f} ms = "
      f"{flop / ms_new / 1e9:7.1f} TFLOP/s ({flop / ms_new / 1e9 / 2500:.3f} of peak)   outputs identical: {bool(torch.equal(old, run()))}")

# round 5: all query blocks of a head on one XCD (default) against the 2-D grid of rounds 2-4 (knob 24 = 1)
n.check(n.lib.ktx_debug_set(24, 1))
ms_old = timed()
old = run().clone()
n.check(n.lib.ktx_debug_set(24, 0))
ms_new = timed()
print(f"2-D grid (rounds 2-4) {ms_old:7.3f} ms   one head per XCD (default) {ms_new:7.3f} ms = {flop / ms_new / 1e9:7.1f} TFLOP/s "
      f"({flop / ms_new / 1e9 / 2500:.3f} of peak)   outputs identical: {bool(torch.equal(old, run()))}")

# dev knob 31 = 1: one query tile per wavefront (64 queries per workgroup), three wavefronts per SIMD
n.check(n.lib.ktx_debug_set(31, 1))
ms_1 = timed()
one = run().clone()
n.check(n.lib.ktx_debug_set(31, 0))
ref = run()
print(f"one query tile per wavefront, 3 waves per SIMD (knob 31) {ms_1:7.3f} ms = {flop / ms_1 / 1e9:7.1f} TFLOP/s ({flop / ms_1 / 1e9 / 2500:.3f} of peak)"
      f"   outputs identical to the default: {bool(torch.equal(one, ref))}")

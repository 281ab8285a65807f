"""A/B of the attention half of a DeepSeek-V3 decode step: ONE persistent launch (csrc/ktx_attn.hip) against the five-launch
chain, L distinct layers chained in one captured graph (layer l's output row is layer l+1's input), with the phase stamps of
workgroup 0.  python scripts/attn_fused_bench.py [layers=16] [ctx=4096]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from ktransformers_amd import _native as N  # noqa: E402

H, NOPE, ROPE, LORA, VDIM, QLORA, HIDDEN, PAGE = 128, 128, 64, 512, 128, 1536, 7168, 64
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CTX = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
if os.environ.get("ATTN_BOUND_K"):   # dev knob 1: context bound (K tokens) up to which the one launch is offered (csrc/ktx_mla.hip)
    N.lib.ktx_debug_set(1, int(os.environ["ATTN_BOUND_K"]))
g = torch.Generator(device=dev)
g.manual_seed(1)


def u(shape, scale):
    return ((torch.rand(shape, generator=g, device=dev, dtype=torch.float32) * 2 - 1) * scale).to(torch.bfloat16)


import os  # noqa: E402
FMT = os.environ.get("KTX_ATTN_BENCH_FMT", "W4")      # "FP8": block-fp8 projections (DeepSeek fp8 checkpoints)
if os.environ.get("KTX_ATTN_BENCH_NSPLIT"):           # dev knob 7: force the KV split count (both paths follow it)
    N.lib.ktx_debug_set(7, int(os.environ["KTX_ATTN_BENCH_NSPLIT"]))


def proj(K, Nout, scale):
    w = u((Nout, K), scale)
    if FMT == "FP8":
        import bench
        h = N.LinearHandle(K, Nout, "FP8", 128, 8, dev)
        q, sc = bench.fp8_block_quant(w)
        h.load_fp8(q, sc)
    else:
        h = N.LinearHandle(K, Nout, "W4", 64, 8, dev)
        h.load_bf16(w)
    return h


layers = []
for _ in range(L):
    o = {}
    o["qkv_a"] = proj(HIDDEN, QLORA + LORA + ROPE, 0.03)
    o["q_b"] = proj(QLORA, H * (NOPE + ROPE), 0.05)
    o["qabs"] = N.LinearHandle(NOPE, LORA, "BF16", 0, 8, dev, batch=H); o["qabs"].load_bf16(u((H, LORA, NOPE), 0.08))
    o["oabs"] = N.LinearHandle(LORA, VDIM, "BF16", 0, 8, dev, batch=H); o["oabs"].load_bf16(u((H, VDIM, LORA), 0.05))
    o["o_proj"] = proj(H * VDIM, HIDDEN, 0.02)
    o["in_norm"] = (1 + u((HIDDEN,), 0.2).float()).to(torch.bfloat16)
    o["qa_norm"] = (1 + u((QLORA,), 0.2).float()).to(torch.bfloat16)
    o["kv_norm"] = (1 + u((LORA,), 0.2).float()).to(torch.bfloat16)
    pages = (CTX + 512 + PAGE - 1) // PAGE
    o["cache"] = u((pages, PAGE, 1, LORA + ROPE), 1.0)
    layers.append(o)
inv_freq = (1.0 / (10000.0 ** (torch.arange(0, ROPE, 2, device=dev, dtype=torch.float32) / ROPE))).contiguous()
pages = layers[0]["cache"].shape[0]
position = torch.tensor([CTX - 1], dtype=torch.int64, device=dev)
kv_len = torch.tensor([CTX], dtype=torch.int32, device=dev)
kv_indptr = torch.tensor([0, pages], dtype=torch.int32, device=dev)
hint = min(CTX - 1 + 512, pages * PAGE)
xs = [u((1, HIDDEN), 1.0)] + [torch.zeros((1, HIDDEN), dtype=torch.bfloat16, device=dev) for _ in range(L)]
eps = 1e-6
wrapper = N.MLAWrapper(1, pages, use_cuda_graph=True, device=dev, max_q_tokens=1)
wrapper.plan(None, kv_indptr, None, kv_len, None, H, LORA, ROPE, PAGE, 0.1147, torch.bfloat16, torch.bfloat16, max_kv_len=hint, identity_pages=True)


def five(l):
    o = layers[l]
    qkv = o["qkv_a"].forward(xs[l], norm=(o["in_norm"], eps))
    q_nope, q_pe, ckv_new, kpe_new = N.qb_absorb_and_prep(o["q_b"], o["qabs"], qkv[:, :QLORA], (o["qa_norm"], eps), qkv[:, QLORA:], o["kv_norm"], eps,
                                                          position, inv_freq, 1.0, H, NOPE, ROPE, LORA)
    parts = wrapper.run_partials(q_nope, q_pe, o["cache"][:, :, 0, :LORA], o["cache"][:, :, 0, LORA:], new_ckv=ckv_new, new_kpe=kpe_new)
    out = N.merge_and_unabsorb(o["oabs"], parts, 1, H)
    o["o_proj"].forward(out.reshape(1, H * VDIM), add1=xs[l], out=xs[l + 1])


ARGS = []
for l in range(L):
    o = layers[l]
    ARGS.append(N.attn_decode_args(o["qkv_a"], o["q_b"], o["qabs"], o["oabs"], o["o_proj"], xs[l].reshape(-1), xs[l + 1].reshape(-1),
                                   (o["in_norm"], eps), (o["qa_norm"], eps), (o["kv_norm"], eps), position, inv_freq, 1.0, H, NOPE, ROPE, LORA,
                                   VDIM, o["cache"][:, :, 0, :LORA], o["cache"][:, :, 0, LORA:], PAGE, kv_indptr, None, kv_len, hint, 0.1147))


def fused(l, chain=(31,)):
    for i, ph in enumerate(chain):
        N.attn_decode(ARGS[l], dev, phases=ph, last=(i == len(chain) - 1))


def capture(fn):
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for l in range(L):
            fn(l)
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for l in range(L):
            fn(l)
    return gr


def timeit(gr, reps=30):
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps / L * 1e6)
    return best


cfgs = [("five launches", five), ("one launch", lambda l: fused(l)), ("phases as 5 launches", lambda l: fused(l, (1, 2, 4, 8, 16))),
        ("3 + 4 + 24", lambda l: fused(l, (3, 4, 24))), ("3 + 28", lambda l: fused(l, (3, 28)))]
import subprocess
print(subprocess.run("/opt/rocm/bin/rocm-smi --showuniqueid | grep Unique", shell=True, capture_output=True, text=True).stdout.strip())
if len(sys.argv) > 3:
    cfgs = [c for c in cfgs if c[0] in ("five launches", "one launch")]
if FMT == "FP8":      # (the five-launch chain of this script uses the W4-only q_b + absorb launch: compare with the phases as launches)
    cfgs = [c for c in cfgs if c[0] in ("one launch", "phases as 5 launches")]   # (the phase masks instantiated for FP8)
graphs = [(name, capture(fn)) for name, fn in cfgs]
ref_y = None
for rnd in range(2):
    for name, gr in graphs:
        us = timeit(gr)
        y = xs[L].clone()
        if ref_y is None:
            ref_y = y
        same = bool(torch.equal(y.view(torch.int16), ref_y.view(torch.int16)))
        print(f"{name:24s} {us:8.2f} us per layer   output == five launches: {same}   status {N.attn_status(dev):#x}", flush=True)
# phase stamps of workgroup 0 (one launch), last layer
st = torch.zeros(64, dtype=torch.int64, device=dev)
N.lib.ktx_attn_debug_stamps(st.data_ptr())
for l in range(L):
    fused(l)
torch.cuda.synchronize()
N.lib.ktx_attn_debug_stamps(None)
t = st.tolist()
names = ["entry", "A staged", "A done", "B: A's row polled", "B: q_a staged", "B: q_nope exchanged", "B done", "C: q polled", "C tiles done",
         "C done", "D: partials polled", "D: merged exchanged", "D done", "E: attn rows polled", "E streamed", "E done"]
print("stamps of workgroup 0, last layer (us from entry):")
for i, nme in enumerate(names):
    if t[i]:
        print(f"  {nme:24s} {(t[i] - t[0]) * 0.01:7.2f}")
extra = {18: "wave 0: every request of the prologue issued", 16: "wave 0: x landed, sum of squares done", 17: "wave 0: past the norm barrier",
         19: "wave 0: its 7 k-steps of phase A done", 23: "wave 7: entry", 25: "wave 7: requests issued", 24: "wave 7: x landed, sum of squares done",
         26: "wave 7: its 7 k-steps of phase A done", 20: "B: wave 0 q_b k-steps done", 21: "B: past the q_b barrier",
         22: "B: own q_nope piece published", 27: "B: partner's piece polled", 28: "E: attention row staged",
         29: "E: ring group 0 done (wave 0)", 30: "E: ring group 1 done", 31: "E: ring group 2 done"}
for i, nme in extra.items():
    if t[i]:
        print(f"  [{nme}] {(t[i] - t[0]) * 0.01:7.2f}")

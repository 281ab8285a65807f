"""Dev tool / profile artefact: one-token decode of the routed experts in EVERY weight format at the DeepSeek-V3 / R1 / Kimi-K2
expert shape (H = 7168, I = 2048, top-8), with the library's per-launch log: which kernels a decode token launches per format,
their time and the algorithmic HBM rate.  Random valid weight bytes (timing only), L distinct layers (> the L3), L3 flushed
before every step.        python scripts/moe_formats_bench.py [--experts 32] [--layers 6]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native as n

ap = argparse.ArgumentParser()
ap.add_argument("--experts", type=int, default=32)
ap.add_argument("--layers", type=int, default=6)
ap.add_argument("--T", type=int, default=1)
ap.add_argument("--json", default="")
ap.add_argument("--formats", default="", help="comma-separated substrings of format names (default: all)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
E, k, H, I, L, T = args.experts, 8, 7168, 2048, args.layers, args.T
g = torch.Generator(device=dev); g.manual_seed(0)
rb = lambda *shape: torch.randint(0, 256, shape, generator=g, device=dev, dtype=torch.uint8)
flush = torch.zeros(512 << 20, dtype=torch.int8, device=dev)
BPW = {"AMXINT4": 0.5, "AMXINT8": 1.0, "RAWINT4": 0.5 + 2 / 32, "FP8": 1.0 + 4 / 16384, "BF16": 2.0,
       "GGUF q4_k_m": (2 * 144 + 210) / 3 / 256, "GGUF IQ1_S": 50 / 256}


def build(fmt):
    hs = []
    for _ in range(L):
        if fmt in ("AMXINT4", "AMXINT8", "BF16"):
            h = n.MoEHandle(E, k, H, I, max_len=max(T, 8), method=fmt, device=0)
            mk = lambda *s: (torch.randn(s, generator=g, device=dev, dtype=torch.bfloat16) * 0.1)
            h.load_bf16(mk(E, I, H), mk(E, I, H), mk(E, H, I))
        elif fmt == "FP8":
            h = n.MoEHandle(E, k, H, I, max_len=max(T, 8), method="FP8", device=0, group_size=128)
            fp8 = lambda *s: (torch.randn(s, generator=g, device=dev) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
            sc = lambda a, b: torch.rand((E, a // 128, b // 128), generator=g, device=dev) * 0.01 + 0.001
            h.load_fp8(fp8(E, I, H), fp8(E, I, H), fp8(E, H, I), sc(I, H), sc(I, H), sc(H, I))
        elif fmt == "RAWINT4":
            h = n.MoEHandle(E, k, H, I, max_len=max(T, 8), method="RAWINT4", device=0, group_size=32)
            sc = lambda a, b: (torch.rand((E, a, b // 32), generator=g, device=dev) * 0.01 + 0.001).to(torch.bfloat16)
            h.load_rawint4(rb(E, I, H // 2), rb(E, I, H // 2), rb(E, H, I // 2), sc(I, H), sc(I, H), sc(H, I))
        else:
            types = (12, 12, 14) if fmt == "GGUF q4_k_m" else (19, 19, 19)
            bb = n.GGML_BLOCK_BYTES

            def blocks(N, K, ty):   # random bytes with a small fp16 super-block scale so nothing overflows
                t = rb(E, N, K // 256, bb[ty])
                d = (torch.rand((E, N, K // 256), generator=g, device=dev) * 0.004 + 0.001).to(torch.float16).view(torch.uint8)
                off = {12: 0, 14: 208, 19: 0}[ty]
                t[..., off:off + 2] = d.view(E, N, K // 256, 2)
                if ty == 12:
                    t[..., 2:4] = d.view(E, N, K // 256, 2)
                return t.reshape(E, N, -1).contiguous()
            h = n.MoEHandle(E, k, H, I, max_len=max(T, 16), method="GGUF", device=0)
            h.load_gguf(blocks(I, H, types[0]), blocks(I, H, types[1]), blocks(H, I, types[2]), *types)
        hs.append(h)
    return hs


out = {}
for fmt in ("AMXINT4", "AMXINT8", "RAWINT4", "FP8", "BF16", "GGUF q4_k_m", "GGUF IQ1_S"):
    if args.formats and not any(f in fmt for f in args.formats.split(",")):
        continue
    try:
        hs = build(fmt)
    except Exception as e:
        print(f"{fmt}: build failed: {e}", flush=True)
        continue
    x = (torch.randn((T, H), generator=g, device=dev) * 0.01).to(torch.bfloat16)
    ids = torch.stack([torch.randperm(E, generator=g, device=dev)[:k] for _ in range(T)]).to(torch.int64)
    w = torch.rand((T, k), generator=g, device=dev)
    y = torch.empty_like(x)
    for h in hs:
        h.forward(x, ids, w, out=y)
    torch.cuda.synchronize()
    n.timing_enable(1)
    logs = []
    for r in range(4):
        for _ in range(4):
            flush.add_(1)
        for h in hs:
            h.forward(x, ids, w, out=y)
        torch.cuda.synchronize()
        log = n.timing_collect()
        if r:
            logs.append(log)
    n.timing_enable(0)
    agg = {}
    for log in logs:
        for label, nb, us in log:
            a = agg.setdefault(label, [0, 0.0])
            a[0] += 1; a[1] += us or 0.0
    per_layer = sum(v[1] for v in agg.values()) / (len(logs) * L)
    U = k if T == 1 else E * (1 - (1 - k / E) ** T)
    nbytes = U * 3 * H * I * BPW[fmt]
    out[fmt] = {"us_per_layer": round(per_layer, 2), "launches_per_layer": round(sum(v[0] for v in agg.values()) / (len(logs) * L), 2),
                "algorithmic_MB": round(nbytes / 1e6, 1), "GBs": round(nbytes / per_layer / 1e3, 1),
                "frac_of_hbm_peak": round(nbytes / per_layer / 1e3 / 8000, 4),
                "kernels": {lab: round(v[1] / v[0], 2) for lab, v in agg.items()}}
    print(f"{fmt:12s} T={T}: {per_layer:8.2f} us/layer  {out[fmt]['launches_per_layer']:.0f} launches  {nbytes / 1e6:7.1f} MB  "
          f"{out[fmt]['GBs']:7.1f} GB/s ({out[fmt]['frac_of_hbm_peak'] * 100:.1f} % of 8 TB/s)", flush=True)
    for lab, v in agg.items():
        print(f"      {v[1] / v[0]:8.2f} us  {lab}", flush=True)
    del hs
    torch.cuda.empty_cache()
if args.json:
    json.dump({"shape": {"E": E, "k": k, "H": H, "I": I, "T": T, "layers": L}, "formats": out}, open(args.json, "w"), indent=1)

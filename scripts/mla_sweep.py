"""Dev tool: time the MLA split-KV kernel and the merge kernel separately for a head count / context, sweeping the workgroup
shape (knob 6) and the split count (knob 7).  Distinct caches per launch (> the 256 MB L3 in total)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native as n

ap = argparse.ArgumentParser()
ap.add_argument("--heads", type=int, default=128)
ap.add_argument("--ctx", type=int, default=4096)
ap.add_argument("--layers", type=int, default=64)
ap.add_argument("--shapes", default="1,2,4", help="knob-6 codes: 1 = 1x4 (head blocks x dim slices), 2 = 2x4, 4 = 4x2")
ap.add_argument("--splits", default="8,16,32,64,128")
ap.add_argument("--no-stamps", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
Hq, ctx, L = args.heads, args.ctx, args.layers
pages = (ctx + 1 + 63) // 64
g = torch.Generator(device=dev); g.manual_seed(0)
kv = [torch.randn((pages, 64, 576), generator=g, device=dev).to(torch.bfloat16) for _ in range(L)]
qn = torch.randn((1, Hq, 512), generator=g, device=dev).to(torch.bfloat16)
qp = torch.randn((1, Hq, 64), generator=g, device=dev).to(torch.bfloat16)
nc = torch.randn((1, 512), generator=g, device=dev).to(torch.bfloat16)
nk = torch.randn((1, 64), generator=g, device=dev).to(torch.bfloat16)
kvlen = torch.tensor([ctx + 1], dtype=torch.int32, device=dev)
w = n.MLAWrapper(1, pages, device=dev, max_q_tokens=1)
flush = torch.zeros(512 << 20, dtype=torch.int8, device=dev)


def run_all():
    for a in range(L):
        ckv, kpe = torch.split(kv[a], [512, 64], dim=-1)
        w.run(qn, qp, ckv, kpe, new_ckv=nc, new_kpe=nk)


def timed(only):
    n.lib.ktx_debug_set(5, only)
    w.plan(None, None, None, kvlen, None, Hq, 512, 64, 64, 192 ** -0.5, max_kv_len=ctx + 1)
    run_all(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run_all()
    n.lib.ktx_debug_set(5, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for r in range(6):
        flush.add_(1); e0.record(); gr.replay(); e1.record(); e1.synchronize()
        if r: tot += e0.elapsed_time(e1)
    return tot / 5 / L * 1e3


for shape in (int(v) for v in args.shapes.split(",")):
    if (shape == 2 and Hq % 32) or (shape == 4 and Hq % 64):
        continue
    for ns in (int(v) for v in args.splits.split(",")):
        n.lib.ktx_debug_set(6, shape); n.lib.ktx_debug_set(7, ns)
        try:
            both, dec, mer = timed(0), timed(1), timed(2)
            print(f"Hq={Hq} ctx={ctx} shape={shape} nsplit={ns}: decode {dec:6.2f} us  merge {mer:6.2f} us  both {both:6.2f} us", flush=True)
        except Exception as e:
            print(f"shape={shape} nsplit={ns}: {e}")
n.lib.ktx_debug_set(6, 0); n.lib.ktx_debug_set(7, 0)
if args.no_stamps:
    sys.exit(0)

# phase breakdown of the split-KV kernel (wall-clock stamps, 10 ns ticks): one eager launch per configuration
names = ["start->request resolved", "->first tile issued", "->first tile landed (barrier)", "->S = QK^T done", "->first tile done",
         "->all tiles done", "->partials stored"]
for shape, ns in ((2, 64), (4, 128), (2, 32)):
    n.lib.ktx_debug_set(6, shape); n.lib.ktx_debug_set(7, ns); n.lib.ktx_debug_set(5, 1)
    buf = torch.zeros(16 * 4096, dtype=torch.int64, device=dev)
    w.plan(None, None, None, kvlen, None, Hq, 512, 64, 64, 192 ** -0.5, max_kv_len=ctx + 1)
    run_all(); torch.cuda.synchronize()
    flush.add_(1)
    n.lib.ktx_mla_debug_stamps(buf.data_ptr())
    ckv, kpe = torch.split(kv[0], [512, 64], dim=-1)
    w.run(qn, qp, ckv, kpe, new_ckv=nc, new_kpe=nk)
    torch.cuda.synchronize()
    n.lib.ktx_mla_debug_stamps(None)
    t = buf.view(-1, 16).cpu()
    t = t[t[:, 0] > 0][:, :8].double()
    for k in (3, 4, 5):   # the stamps inside the tile loop exist only in a -DKTX_MLA_LOOP_STAMPS build of csrc/ktx_mla.hip: without them
        t[:, k] = torch.where(t[:, k] > 0, t[:, k], t[:, k - 1])   # their intervals read 0 and "all tiles done" holds the whole loop
    d = (t[:, 1:] - t[:, :-1]) / 100.0
    span = (t[:, 7].max() - t[:, 0].min()) / 100.0
    print(f"shape={shape} nsplit={ns}: {t.shape[0]} workgroups, first start -> last end {span:.2f} us; start skew {(t[:, 0].max() - t[:, 0].min()) / 100.0:.2f} us")
    for i, nm in enumerate(names):
        print(f"    {nm:34s} mean {d[:, i].mean():6.2f} us  max {d[:, i].max():6.2f} us")
n.lib.ktx_debug_set(5, 0); n.lib.ktx_debug_set(6, 0); n.lib.ktx_debug_set(7, 0)

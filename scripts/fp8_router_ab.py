"""A/B on one resident model: decode step of a block-fp8 workload with the router riding in the shared experts' gate|up launch
(lin_dec_gate_kernel<FP8>, default) vs the two separate launches (KTX_MOE_SEPARATE_ROUTER=1).  The choice is baked into the
captured graph, so the step is re-captured between the legs.  usage: python scripts/fp8_router_ab.py [workload] [layers] [env switch of the second leg]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "r1-iq1s"
    wl = bench.WORKLOADS[name]
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else wl["layers"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    mr = bench.ModelDecodeRunner(wl, layers, dev, 4096, 1024, seed=0, use_graph=True)
    assert mr.graph_ok, mr.graph_error
    res = {}
    knob = sys.argv[3] if len(sys.argv) > 3 else "KTX_MOE_SEPARATE_ROUTER"   # the environment switch of the "separate" leg
    for rnd in range(2):
        for leg in ("combined", "separate"):
            if leg == "separate" and knob == "none":      # (one leg only: e.g. to compare two builds of the library)
                continue
            if leg == "separate":
                os.environ[knob] = "1"
            else:
                os.environ.pop(knob, None)
            mr.capture(True)
            assert mr.graph_ok, mr.graph_error
            for i in range(30):
                mr.step(i)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(200):
                mr.step(i)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / 200 * 1e3
            res.setdefault(leg, []).append(ms)
            print(f"{name} L={layers} {leg:9s} round {rnd}: {ms:.3f} ms/step  {1e3 / ms:.1f} tok/s", flush=True)
    os.environ.pop(knob, None)
    # one eager step with HIP events around every library launch: what a decode step of this workload consists of
    from ktransformers_amd import _native as n
    mr.step_eager()
    torch.cuda.synchronize(dev)
    n.timing_collect()
    n.timing_enable(1)
    mr.step_eager()
    torch.cuda.synchronize(dev)
    rows = n.timing_collect()
    n.timing_enable(0)
    agg = {}
    for lab, nbytes, us in rows:
        a = agg.setdefault(lab, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += us or 0.0
        a[2] += nbytes
    tot = sum(v[1] for v in agg.values())
    print(f"eager step: {len(rows)} library launches, {tot / 1e3:.3f} ms of kernel time")
    for lab, (cnt, us, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"  {cnt:4d} x {us / cnt:8.2f} us  {nb / max(us, 1e-9) / 1e6:7.2f} TB/s  {100 * us / tot:5.1f} %  {lab}")
    if "separate" not in res:
        print(f"best: {min(res['combined']):.3f} ms ({1e3 / min(res['combined']):.1f} tok/s)")
        return
    a, b = min(res["combined"]), min(res["separate"])
    print(f"best: combined {a:.3f} ms ({1e3 / a:.1f} tok/s)  separate {b:.3f} ms ({1e3 / b:.1f} tok/s)  gain {100 * (b / a - 1):.2f} %")


if __name__ == "__main__":
    main()

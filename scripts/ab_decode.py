"""Dev tool: A/B timing of launch-time choices on the SAME box and the SAME resident model — boxes of the pool differ by
10-30 % on latency-bound kernels, so two bench runs cannot be compared.  The whole-model decode graph is re-captured per
configuration (knobs are read at launch time) and the configurations are timed interleaved, several rounds.

    python scripts/ab_decode.py [--layers 32] [--rounds 3] [--steps 60]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ktransformers_amd import _native as n

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="v3-int4")
ap.add_argument("--layers", type=int, default=0)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--ctx", type=int, default=4096)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
wl = bench.WORKLOADS[args.workload]
mr = bench.ModelDecodeRunner(wl, args.layers or wl["layers"], dev, args.ctx, 4096)

ONLY = os.environ.get("KTX_AB_ONLY")   # comma-separated substrings of configuration names
CONFIGS = {   # name: (knobs {idx: val}, env {k: v})
    "default": ({}, {}),
    "mla: 32 splits": ({7: 32}, {}),
    "mla: 48 splits": ({7: 48}, {}),
    "mla: 63 splits": ({7: 63}, {}),
    "mla: 4x2 shape, 128 splits": ({6: 4, 7: 128}, {}),
    "mla: separate prep launch": ({}, {"KTX_MLA_SEPARATE_PREP": "1"}),
    "moe: router in its own launch": ({}, {"KTX_MOE_SEPARATE_ROUTER": "1"}),
    "moe: shared down in its own launch": ({}, {"KTX_MOE_SEPARATE_SHARED_DOWN": "1"}),
    "sampling: torch argmax on fp32 logits": ({}, {"KTX_TORCH_ARGMAX": "1"}),
    "mla: q_b and q-absorb as two launches": ({}, {"KTX_MLA_SEPARATE_QB": "1"}),
    "mla: merge and un-absorb as two launches": ({}, {"KTX_MLA_SEPARATE_MERGE": "1"}),
    "lin: round-2 decode GEMVs (no all-CU kernel)": ({16: 1}, {}),
    "lin: router as wavefront 7 of the all-CU gate|up kernel + selector workgroup": ({19: 2}, {}),
    "lin: round-2 router workgroups in front of the all-CU gate|up kernel": ({19: 3}, {}),
    "lin: two workgroups per CU": ({18: 2}, {}),
    "gate: store-ack hand-off (no granules)": ({21: 1}, {}),
    "gate: 8 experts per router workgroup (32 workgroups at E = 256)": ({25: 1}, {}),
    "gate: 2 experts per router workgroup (128 workgroups at E = 256)": ({25: 2}, {}),
    "gate: selection by one wavefront (rounds 1-3)": ({28: 1}, {}),
    "attn: five launches (rounds 2-3)": ({}, {"KTX_ATTN_SEPARATE": "1"}),
}
if ONLY:
    CONFIGS = {k: v for k, v in CONFIGS.items() if k == "default" or any(o in k for o in ONLY.split(","))}
res = {k: [] for k in CONFIGS}
toks = {}


def token_trace(n_steps=8):
    """The tokens of n_steps greedy steps from a fixed state (position ctx, token 1): every configuration must produce the same."""
    mr.set_position(args.ctx)
    mr.cur.fill_(1)
    out = []
    for _ in range(n_steps):
        mr.step()
        out.append(int(mr.cur.item()))
    return out


for r in range(args.rounds):
    for name, (knobs, env) in CONFIGS.items():
        for i, v in knobs.items():
            n.lib.ktx_debug_set(i, v)
        for k, v in env.items():
            os.environ[k] = v
        try:
            mr.capture(True)
            if not mr.graph_ok:
                print(f"  !! {name}: graph capture failed ({mr.graph_error}); timed eagerly", flush=True)
            if r == 0:
                toks[name] = token_trace()
                if toks[name] != toks["default"]:
                    print(f"  !! {name}: tokens differ from default: {toks[name]} vs {toks['default']}", flush=True)
            for _ in range(40):
                mr.step()
            mr.set_position(args.ctx)
            dt = bench.timed(mr.step, args.steps, 10, dev, False)
            res[name].append(dt / args.steps * 1e3)
            print(f"round {r} {name:32s} {res[name][-1]:.3f} ms/step", flush=True)
        except Exception as e:      # one configuration failing (a capture error, say) must not lose the others
            if name == "default":
                raise
            print(f"round {r} {name:32s} FAILED: {type(e).__name__}: {e}", flush=True)
            torch.cuda.synchronize(dev)
        finally:
            for i in knobs:
                n.lib.ktx_debug_set(i, 0)
            for k in env:
                os.environ.pop(k, None)
base = min(res["default"])
for name, v in res.items():
    if not v:
        print(f"{name:32s} no result", flush=True)
        continue
    print(f"{name:32s} best {min(v):.3f} ms  median {sorted(v)[len(v) // 2]:.3f} ms  vs default {min(v) / base:.3f}x  "
          f"tokens {'same' if toks.get(name) == toks.get('default') else 'DIFFER'}", flush=True)

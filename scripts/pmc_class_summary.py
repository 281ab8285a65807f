"""Average of every counter of a rocprofv3 --pmc run per kernel class (the identifier in front of '<' / '('), library kernels
only.  Usage: pmc_class_summary.py <rocprof dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not files:
    print("no counter_collection.csv under", d)
    sys.exit(0)
rows = list(csv.DictReader(open(files[0])))
cut = 0   # every dispatch of a library kernel (the model build's kernels have other names)
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    if int(r["Dispatch_Id"]) < cut:
        continue
    name = r["Kernel_Name"]
    cls = name.split("<")[0].split("(")[0].split(" ")[-1]
    if not any(k in cls for k in ("lin_", "moe_", "mla_", "gate_", "argmax", "rmsnorm", "attn_", "layer_")):
        continue
    grid = r.get("Grid_Size", "")
    acc[(cls, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for v in acc.values() for c in v})
print(f"{'kernel class':40s} {'grid':>8s} {'n':>4s} " + " ".join(f"{c:>22s}" for c in counters))
for (cls, grid), v in sorted(acc.items()):
    n = max(len(x) for x in v.values())
    print(f"{cls:40s} {grid:>8s} {n:4d} " + " ".join(f"{(sum(v[c]) / len(v[c]) if v.get(c) else float('nan')):22.1f}" for c in counters))

"""Per-kernel averages of every counter found in rocprofv3 --pmc output directories: pmc_any_summary.py <dir>... <out.json>"""
import collections
import csv
import glob
import json
import sys

*srcs, dst = sys.argv[1:]
out = collections.defaultdict(dict)
for src in srcs:
    for f in glob.glob(f"{src}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "moe_gemm" in name:
                acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (name, c), v in acc.items():
            out[name][c] = round(sum(v) / len(v), 1)
            out[name]["dispatches"] = len(v)
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))

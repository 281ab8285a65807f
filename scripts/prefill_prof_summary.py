"""Summarise the rocprofv3 --kernel-trace CSV of scripts/prefill_prof.py: the dispatches between the two marker scans, grouped by
kernel, and the share of the chunk's kernel time that runs in this library's kernels vs torch glue vs vendor GEMMs.

    python scripts/prefill_prof_summary.py <kernel_trace.csv> <reps> > profiles/rNN_prefill_chunk_kernels.txt
"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "Scan" in r["Kernel_Name"] or "scan" in r["Kernel_Name"]]
assert len(marks) >= 2, "marker scans not found"
lo = max(i for i in marks if i < marks[-1] - 10) if any(i < marks[-1] - 10 for i in marks) else marks[0]
body = rows[lo + 1:marks[-1]]
body = [r for r in body if "Scan" not in r["Kernel_Name"] and "scan" not in r["Kernel_Name"]]


def klass(name):
    if name.startswith(("void at::", "at::", "void at_cuda", "at_cuda")) or "rocclr" in name or "elementwise" in name:
        return "torch glue (elementwise / copies / index ops)"
    if name.startswith("Cijk_") or "rocblas" in name.lower() or "hipblas" in name.lower() or "Tensile" in name:
        return "vendor GEMM (hipBLASLt / rocBLAS)"
    return "libktx_hip.so"


agg, cls = defaultdict(lambda: [0, 0.0]), defaultdict(float)
for r in body:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = re.sub(r"\s+", " ", r["Kernel_Name"])
    agg[n][0] += 1
    agg[n][1] += d
    cls[klass(n)] += d
tot = sum(cls.values())
span = (int(body[-1]["End_Timestamp"]) - int(body[0]["Start_Timestamp"])) / 1e3
print(f"{len(body)} dispatches between the markers = {reps} prompt chunks; kernel time {tot / reps / 1e3:.3f} ms per chunk, "
      f"span {span / reps / 1e3:.3f} ms per chunk (kernel time / span = {tot / span:.3f})")
print()
for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
    print(f"{v / tot * 100:6.2f} %  {v / reps / 1e3:8.3f} ms/chunk  {k}")
print()
print(f"{'share':>7s} {'ms/chunk':>9s} {'calls/chunk':>11s} {'avg us':>9s}  kernel")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{d / tot * 100:6.2f}% {d / reps / 1e3:9.3f} {c / reps:11.1f} {d / c:9.1f}  {n[:150]}")

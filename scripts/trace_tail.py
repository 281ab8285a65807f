"""Keep the tail of a rocprofv3 --kernel-trace CSV (sorted by start time, trimmed columns, gzip) so it fits gpurun_out/."""
import csv
import gzip
import sys

src, dst = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cols = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X", "LDS_Block_Size",
        "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size"]
with gzip.open(dst, "wt") as f:
    w = csv.writer(f)
    w.writerow(cols)
    for r in rows[-n:]:
        w.writerow([r.get(c, "")[:100] for c in cols])
print(f"{len(rows)} dispatches, kept {min(n, len(rows))}")

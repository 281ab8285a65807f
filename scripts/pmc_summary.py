"""Summarise rocprofv3 --pmc passes (one counter per pass, as MI355X_MICROARCH.md §HBM prescribes) into the small JSON that
bench.py reports as roofline.traffic.

    cd /tmp && for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d <out>/pmc_$c -- \
        python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prefill --no-model --no-graph --hot-path moe; done
    python scripts/pmc_summary.py <out> profiles/r01_pmc_decode.json

Units and corrections (guide): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of a
wide coalesced streaming read (16 B per lane), so reads are doubled; WRITE_SIZE is uncalibrated and reported as is."""
import collections
import csv
import glob
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per-dispatch averages",
       "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read undercount, MI355X_MICROARCH.md #HBM)", "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{src}/pmc_{c}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and ("moe_dec_" in r["Kernel_Name"] or "mla_" in r["Kernel_Name"] or "gate_" in r["Kernel_Name"]):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        e = out["kernels"].setdefault(k, {})
        e["dispatches"] = len(v)
        avg = sum(v) / len(v)
        if c == "FETCH_SIZE":
            e["fetch_kib_raw"] = round(avg, 2)
            e["hbm_read_bytes"] = int(avg * 1024 * 2)
        else:
            e["write_kib_raw"] = round(avg, 2)
            e["hbm_write_bytes"] = int(avg * 1024)
for e in out["kernels"].values():
    e["hbm_bytes"] = e.get("hbm_read_bytes", 0) + e.get("hbm_write_bytes", 0)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))

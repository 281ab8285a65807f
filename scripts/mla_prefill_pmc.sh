#!/bin/bash
# SQ counters of mla_prefill_kernel at the prompt chunk's shape (T = kv = 2048, 128 heads): what bounds it after round 5?
#   mla_prefill_pmc.sh <outdir>      two passes of 8 SQ counters; per-kernel averages into gpurun_out/<outdir>/pmc_mla_prefill.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmc_pf_$i
  timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc_pf_$i -- python $R/scripts/mla_prefill_bench.py 2048 2048 > $O/pmc_run_$i.log 2>&1
done
python - "$O/pmc_mla_prefill.txt" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    for f in glob.glob(f"/tmp/pmc_pf_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "mla_prefill" in name:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1], "w") as out:
    for name, cs in sorted(acc.items()):
        line = f"{name}  (dispatches {len(next(iter(cs.values())))}; every variant the bench script times is in the average)"
        print(line); out.write(line + "\n")
        for c, v in sorted(cs.items()):
            line = f"    {c:28s} {sum(v) / len(v):16.1f}   (min {min(v):.1f}, max {max(v):.1f})"
            print(line); out.write(line + "\n")
PY

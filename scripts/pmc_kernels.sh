#!/bin/bash
# SQ counters (two passes of <= 8) of every library kernel whose name contains <substr>, for an arbitrary python command:
#   pmc_kernels.sh <outdir> <tag> <kernel-name substring> -- python scripts/...py args
# Per-kernel averages go to gpurun_out/<outdir>/pmc_<tag>.txt.  --pmc passes run with --kernel-trace only (gpurun's rule).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; TAG=$2; SUB=$3; shift 4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmc_k_$i
  (cd $R && rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc_k_$i -- "$@") > $O/pmc_${TAG}_run_$i.log 2>&1
done
python - "$O/pmc_$TAG.txt" "$SUB" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    for f in glob.glob(f"/tmp/pmc_k_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            if sys.argv[2] in name:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1], "w") as out:
    for name, cs in sorted(acc.items()):
        line = f"{name}  (dispatches {len(next(iter(cs.values())))})"
        print(line); out.write(line + "\n")
        for c, v in sorted(cs.items()):
            line = f"    {c:28s} {sum(v) / len(v):16.1f}"
            print(line); out.write(line + "\n")
PY

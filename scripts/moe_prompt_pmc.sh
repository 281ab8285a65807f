#!/bin/bash
# SQ counters of the grouped prompt GEMMs of one expert format (T tokens over E experts at the V3 / K2 expert shape):
#   moe_prompt_pmc.sh <outdir> <format substring> [T] [experts]
# Two passes of <= 8 SQ counters each; per-kernel averages into gpurun_out/<outdir>/pmc_<format>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; F=$2; T=${3:-256}; E=${4:-48}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmc_mp_$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc_mp_$i -- python $R/scripts/moe_formats_bench.py --T $T --experts $E --layers 2 --formats "$F" > $O/pmc_run_$i.log 2>&1
done
python - "$O/pmc_$(echo $F | tr ' ' '_').txt" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    for f in glob.glob(f"/tmp/pmc_mp_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if name.startswith(("moe_", "gg_")) and "pack" not in name and "quant_rows" not in name:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1], "w") as out:
    for name, cs in sorted(acc.items()):
        line = f"{name}  (dispatches {len(next(iter(cs.values())))})"
        print(line); out.write(line + "\n")
        for c, v in sorted(cs.items()):
            line = f"    {c:28s} {sum(v) / len(v):16.1f}"
            print(line); out.write(line + "\n")
PY

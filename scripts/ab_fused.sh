#!/bin/bash
# A/B of the one-launch attention / MoE halves inside the whole-model decode bench, same box, same process order:
#   bash scripts/ab_fused.sh <tag>   -> gpurun_out/<tag>/ab_fused.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-ab}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/rocm-smi --showuniqueid --showdriverversion | grep "Unique\|Driver" | tee $O/ab_fused.txt
for cfg in "" "KTX_MOE_SEPARATE=1" "KTX_ATTN_SEPARATE=1" "KTX_MOE_SEPARATE=1 KTX_ATTN_SEPARATE=1" ""; do
  env $cfg python $R/bench.py --steps 100 --warmup 10 --windows 2 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null > $O/b.json
  python - <<PY | tee -a $O/ab_fused.txt
import json
d = json.load(open("$O/b.json"))
print(f"{'$cfg' or 'default (both halves one launch each)':60s} {d['ms_per_step']:.4f} ms/step  {d['value']:.2f} tok/s  median {d.get('median_ms_per_step')}")
PY
done

#!/bin/bash
# A/B of two builds of the library on one box: ktransformers_amd/lib/libktx_hip.so (new) against ktransformers_amd/lib/base/libktx_hip.so
# (a copy of the previous build), the attention half alone (scripts/attn_fused_bench.py) and the whole-model decode step (bench.py).
#   bash scripts/ab_attn_libs.sh <tag> [pytest-file ...]   -> gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-ab}; mkdir -p $O
shift
LIB=$R/ktransformers_amd/lib
cd $R
/opt/rocm/bin/rocm-smi --showuniqueid | grep Unique | tee $O/box.txt
if [ $# -gt 0 ]; then timeout 1500 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt; fi
cp $LIB/libktx_hip.so /tmp/new.so
for rnd in 1 2; do
  for which in new base; do
    if [ $which = base ]; then cp $LIB/base/libktx_hip.so $LIB/libktx_hip.so; else cp /tmp/new.so $LIB/libktx_hip.so; fi
    echo "== $which (round $rnd)" | tee -a $O/attn.txt $O/step.txt
    timeout 600 python scripts/attn_fused_bench.py 16 4096 x 2>&1 | grep -v "^$" | tee -a $O/attn.txt
    timeout 900 python bench.py --steps 100 --warmup 10 --windows 2 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null > $O/b_$which$rnd.json
    python - <<PY | tee -a $O/step.txt
import json
d = json.load(open("$O/b_$which$rnd.json"))
print(f"{d['ms_per_step']:.4f} ms/step  {d['value']:.2f} tok/s  median {d.get('median_tok_s')}  attention launch {d.get('roofline', {}).get('avg_launch_us')} us")
PY
  done
done
cp /tmp/new.so $LIB/libktx_hip.so

#!/bin/bash
# every bench workload once, short (dev check of the workload plumbing): scripts/bench_all.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-ball}; mkdir -p $O
for w in ${WL:-r1-iq1s v3-fp8 k2-rawint4 mixtral-q4km v2lite-int4}; do
  timeout 600 python $R/bench.py --workload $w --steps 30 --warmup 5 --windows 0 --no-kernels --no-prefill --no-cpu-baseline --no-secondary > $O/$w.json 2> $O/$w.err
  echo "== $w rc=$?"; python -c "
import json,sys
try:
    d=json.load(open('$O/$w.json')); print(d['value'], d['ms_per_step'], d.get('whole_step'), d.get('config',{}).get('hip_graph'))
except Exception as e:
    print('no json:', e)
"; grep -v "amdgpu.ids" $O/$w.err | tail -4
done

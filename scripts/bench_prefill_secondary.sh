#!/bin/bash
# Prompt-chunk tables of the secondary workloads (one bench.py call each, decode kept short): gpurun_out/<name>/<workload>.json
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-prefill_sec}; mkdir -p $O; shift
for w in ${@:-k2-rawint4 v3-fp8 r1-iq1s}; do
  python $R/bench.py --workload $w --steps 20 --warmup 3 --windows 0 --no-secondary --no-cpu-baseline --no-pmc --no-kernels > $O/$w.json 2> $O/$w.err
  python -c "
import json; d=json.load(open('$O/$w.json')); p=d['prefill']; print('$w', 'decode', d['value'], 'prefill', p['value'], 'tok/s', p['ms_per_chunk'], 'ms', 'frac', p['roofline']['frac'])
for r in p['per_kernel'][:6]: print('    ', r)"
done

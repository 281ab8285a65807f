#!/bin/bash
# The measurement artefacts of a round, into gpurun_out/<name>/ (copied to profiles/ by hand): full GPU suite, the bench line,
# rocprofv3 kernel stats of the same bench command, kernel stats of every secondary workload.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r03}; mkdir -p $O
/opt/rocm/bin/rocm-smi --showuniqueid | grep Unique > $O/box.txt
if [ "${2:-all}" = "all" ] || [ "$2" = "tests" ]; then
  # (the reference's worker pools print to the C stdout, which is flushed after pytest's own summary: keep every line that matters)
  (cd $R && timeout 1500 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu_full.log 2>&1); grep -E "^[.sFEx]+ *\[|passed|failed|error|SKIPPED|FAILED|ERROR" $O/pytest_gpu_full.log > $O/pytest_gpu.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
fi
if [ "${2:-all}" = "all" ] || [ "$2" = "bench" ]; then
  (cd $R && python bench.py > $O/bench.json 2> $O/bench.err); python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d.get('median_tok_s'), d['roofline']['kernel'], d['roofline']['frac'])"
  bash $R/scripts/prof_bench.sh $(basename $O)/prof_v3 --steps 60 --warmup 10 --windows 0 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels > /dev/null
fi
if [ "${2:-all}" = "all" ] || [ "$2" = "secondary" ]; then
  for w in r1-iq1s v3-fp8 k2-rawint4 mixtral-q4km v2lite-int4; do
    bash $R/scripts/prof_bench.sh $(basename $O)/prof_$w --workload $w --steps 40 --warmup 5 --windows 0 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels > /dev/null
    python -c "
import json; d=json.load(open('$O/prof_$w/bench_prof.json')); print('$w', d['value'], d['ms_per_step'])"
  done
fi

#!/bin/bash
# rocprofv3 --kernel-trace over bench.py (graph replays): stats CSV + the tail of the per-dispatch trace into gpurun_out/$1/
# usage: scripts/prof_bench.sh <outdir-name> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py "$@" > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
s=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp $s $O/kernel_stats.csv
[ -n "$f" ] && python $R/scripts/trace_tail.py $f $O/ktrace_tail.csv.gz 20000
tail -2 $O/bench_prof.err

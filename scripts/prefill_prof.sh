#!/bin/bash
# rocprofv3 --kernel-trace over scripts/prefill_prof.py + the per-class summary into gpurun_out/$1/   (usage: prefill_prof.sh <outdir> [layers] [tokens])
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pf
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_pf -- python $R/scripts/prefill_prof.py "$@" > $O/prefill_prof.log 2> $O/prefill_prof.err
f=$(find /tmp/prof_pf -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $R/scripts/prefill_prof_summary.py $f 4 > $O/prefill_chunk_kernels.txt
tail -1 $O/prefill_prof.log; head -8 $O/prefill_chunk_kernels.txt

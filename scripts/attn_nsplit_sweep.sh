#!/bin/bash
# The one-launch attention against the KV split count (dev knob 7; the default rule picks 49 at a 4.6 K bound): attention half alone.
#   bash scripts/attn_nsplit_sweep.sh <tag>   -> gpurun_out/<tag>/nsplit.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-ns}; mkdir -p $O; cd $R
/opt/rocm/bin/rocm-smi --showuniqueid | grep Unique | tee $O/nsplit.txt
for ns in 0 43 49 56 64 0; do
  echo "== nsplit $ns" | tee -a $O/nsplit.txt
  if [ $ns = 0 ]; then unset KTX_ATTN_BENCH_NSPLIT; else export KTX_ATTN_BENCH_NSPLIT=$ns; fi
  timeout 600 python scripts/attn_fused_bench.py 16 4096 x 2>&1 | grep "one launch\|five launches\|q polled\|tiles done\|C done\|partials polled\|merged" | tee -a $O/nsplit.txt
done

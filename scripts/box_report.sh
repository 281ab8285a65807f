#!/bin/bash
# Fingerprint of the box a gpurun call landed on (run first in every measurement call of round 3): clocks / partition modes,
# dependent-load latency, boundary cost, streaming rate, and the fixed cost of dependent GEMV-like kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-box}; mkdir -p $O
( bash $R/scripts/box_info.sh; $R/scripts/box_probe; $R/scripts/dep_chain ) > $O/box_report.txt 2>&1
grep -h "chase across\|Unique ID\|sclk\|fclk\|Partition\|chase  2048\|graph of 200 empty kernels, grid  256\|stream 4\|gemv" $O/box_report.txt | grep -v "^HIP_" 

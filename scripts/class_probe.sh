#!/bin/bash
# box class (short decode bench with the five / three launch path and with the one-launch halves) + the translation probe
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-cls}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/rocm-smi --showuniqueid --showdriverversion | grep "Unique\|Driver" | tee $O/class_probe.txt
$R/scripts/tlb_probe 48 2>&1 | tee -a $O/class_probe.txt
KTX_MOE_SEPARATE=1 KTX_ATTN_SEPARATE=1 python $R/bench.py --steps 60 --warmup 5 --windows 1 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('decode step, eight launches per layer:', d['ms_per_step'], 'ms')" | tee -a $O/class_probe.txt

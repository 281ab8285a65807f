#!/bin/bash
# On a box of the SLOW class: does anything a user can set about clocks / power management change the class?
# (perf level auto -> high, power profile -> COMPUTE, fclk / mclk / sclk under load, deep-sleep feature mask.)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pwr}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
S=/opt/rocm/bin/rocm-smi
bench() { env $2 python $R/bench.py --steps 60 --warmup 5 --windows 1 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1:', d['ms_per_step'], 'ms/step')"; }
{
$S --showuniqueid --showdriverversion | grep "Unique\|Driver"
bench "eight launches, as found" "KTX_MOE_SEPARATE=1 KTX_ATTN_SEPARATE=1" > $O/cls.txt; cat $O/cls.txt
ms=$(sed 's/.*: \([0-9.]*\) ms.*/\1/' $O/cls.txt)
if python -c "import sys; sys.exit(0 if float('$ms') > ${SLOW_MS:-4.6} else 1)"; then
  echo "SLOW CLASS"
  for c in /sys/class/drm/card*/device; do [ -f $c/power_dpm_force_performance_level ] && echo "$c: $(cat $c/power_dpm_force_performance_level) ; pp_features:" && cat $c/pp_features 2>/dev/null | head -40; done
  $S --showprofile 2>&1 | grep -v "^=\|^$" | head -20
  # clocks under load, as found
  (KTX_MOE_SEPARATE=1 KTX_ATTN_SEPARATE=1 python $R/bench.py --steps 3000 --warmup 5 --windows 0 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels > /dev/null 2>&1 &) ; sleep 13
  for i in 1 2 3; do $S --showclocks --showpower --showuse 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk\|Power (W)\|GPU use" | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 1; done
  sleep 8
  echo "--- perf level high"
  $S --setperflevel high 2>&1 | grep -v "^=\|^$" | head -5
  cat /sys/class/drm/card*/device/power_dpm_force_performance_level 2>/dev/null | tr '\n' ' '; echo
  bench "eight launches, perf level high" "KTX_MOE_SEPARATE=1 KTX_ATTN_SEPARATE=1"
  bench "one-launch attention, perf level high" "KTX_MOE_SEPARATE=1"
  $S --setperflevel auto 2>&1 | grep -v "^=\|^$" | head -3
  echo "--- power profile COMPUTE"
  $S --setprofile COMPUTE 2>&1 | grep -v "^=\|^$" | head -5
  bench "eight launches, profile COMPUTE" "KTX_MOE_SEPARATE=1 KTX_ATTN_SEPARATE=1"
  $S --resetprofile 2>&1 | grep -v "^=\|^$" | head -3
  echo "--- perf determinism / sclk pinned"
  $S --setperfdeterminism 2100 2>&1 | grep -v "^=\|^$" | head -5
  bench "eight launches, perf determinism 2100" "KTX_MOE_SEPARATE=1 KTX_ATTN_SEPARATE=1"
  $S --resetperfdeterminism 2>&1 | grep -v "^=\|^$" | head -3
  bench "eight launches, back to as found" "KTX_MOE_SEPARATE=1 KTX_ATTN_SEPARATE=1"
fi
} 2>&1 | tee $O/power.txt

#!/bin/bash
# Where does the reference's CPU leg run best on this box?  (lift the thread cap, pin, say what the host is.)
# Prints the container's CPU allowance, then the bs=1 / prompt legs of bench.py's cpu_baseline for several thread counts, pinned and not.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-cpu_leg}; mkdir -p $O
{
echo "nproc $(nproc)   online $(cat /sys/devices/system/cpu/online)"
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)   cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
grep -i "Cpus_allowed_list\|Mems_allowed_list" /proc/self/status
for n in /sys/devices/system/node/node*/cpulist; do echo "$n $(cat $n)"; done
uptime
grep "nr_throttled\|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null
for cfg in "128 0" "64 0" "64 1" "32 0" "128 1"; do
  set -- $cfg
  echo "== threads $1 nobind $2"
  KTX_HWLOC_NOBIND=$2 python $R/bench.py --cpu-baseline-only --cpu-threads $1 --cpu-budget 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d.get('us_per_layer'), 'us/layer', d.get('GBs'), 'GB/s   prefill', d.get('prefill',{}).get('value'), d.get('prefill',{}).get('int8_TOPs'))"
  grep "nr_throttled\|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
done
} 2>&1 | tee $O/cpu_leg_probe.txt

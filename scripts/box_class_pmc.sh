#!/bin/bash
# what differs between the two classes of boxes of the pool (host build 6.18.50 = slow, 6.18.51 = fast).
# Run first in a measurement call:  bash scripts/box_class_pmc.sh <tag>   -> gpurun_out/<tag>/{box_params.txt,bench_short.json,pmc_*.txt}
#   1. everything the box says about itself that rocm-smi does not: amdgpu module parameters (mtype_local, noretry, ...), host
#      kernel, CPU count / sockets / governor;
#   2. the short decode bench (ms per step names the class);
#   3. L2 (TCC) hit / miss / HBM-read-request counters per kernel class of the eager decode step (separate --pmc passes, counters
#      only with --kernel-trace: the combination gpurun allows).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-boxclass}; mkdir -p $O
{
  echo "== uname"; uname -a
  echo "== cpus"; nproc; lscpu | grep -i "model name\|socket(s)\|core(s) per socket\|thread(s) per core\|numa node(s)\|^CPU(s)"
  echo "== governor"; cat /sys/devices/system/cpu/cpu0/cpufreq/scaling_governor 2>/dev/null || echo "(no cpufreq)"
  echo "== amdgpu module parameters"
  for f in /sys/module/amdgpu/parameters/*; do printf "%s=%s\n" "$(basename $f)" "$(cat $f 2>/dev/null | tr '\n' ' ')"; done
  echo "== amdgpu version"; cat /sys/module/amdgpu/version 2>/dev/null; cat /sys/module/amdgpu/srcversion 2>/dev/null
  echo "== rocm-smi"; /opt/rocm/bin/rocm-smi --showuniqueid --showvbios --showdriverversion --showperflevel --showclocks --showmemorypartition --showcomputepartition 2>&1 | grep -v "^=\|^$" | head -40
  echo "== kfd topology (node 1+ = GPU)"; for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "-- $n"; grep -h "simd_count\|cu_count\|array_count\|num_xcc\|max_waves_per_simd\|lds_size\|gfx_target\|sdma\|caches_count" $n/properties 2>/dev/null; done | head -60
  echo "== mem banks / caches of the GPU node"; for c in /sys/class/kfd/kfd/topology/nodes/1/caches/*; do grep -h "level\|size\|type" $c/properties 2>/dev/null | tr '\n' ' '; echo; done | sort | uniq -c | head -12
} > $O/box_params.txt 2>&1
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 50 --warmup 5 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null > $O/bench_short.json
python - <<EOF
import json
d = json.load(open("$O/bench_short.json"))
print("bench:", d["ms_per_step"], "ms/step", d["value"], "tok/s", d.get("box", {}).get("driver"))
EOF
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $set | tr ' ' '+')
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python $R/bench.py --pmc-child --workload v3-int4 --ctx 4096 > $O/pmc_child_$tag.log 2>&1
  python $R/scripts/pmc_class_summary.py /tmp/pmc_$tag > $O/pmc_$tag.txt 2>&1
done
cat $O/pmc_*.txt | head -80

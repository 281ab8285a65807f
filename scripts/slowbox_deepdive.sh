#!/bin/bash
# Run at the start of a measurement call: classify the box with the short eight-launch decode bench; on the SLOW class (>= 4.6 ms per
# step) collect what is still missing about it: the cold-page probe, in-graph per-kernel tables (rocprofv3 child pass of bench.py)
# for three launch structures, and the in-model phase stamps of the one-launch kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-deep}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/rocm-smi --showuniqueid --showdriverversion | grep "Unique\|Driver" | tee $O/deepdive.txt
KTX_MOE_SEPARATE=1 KTX_ATTN_SEPARATE=1 python $R/bench.py --steps 60 --warmup 5 --windows 1 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null > $O/cls.json
ms=$(python -c "import json; print(json.load(open('$O/cls.json'))['ms_per_step'])")
echo "decode step, eight launches per layer: $ms ms" | tee -a $O/deepdive.txt
if python -c "import sys; sys.exit(0 if float('$ms') > ${SLOW_MS:-4.6} else 1)"; then
  echo "SLOW CLASS" | tee -a $O/deepdive.txt
  $R/scripts/tlb_probe 48 2>&1 | tee -a $O/deepdive.txt
  for cfg in "KTX_MOE_SEPARATE=1 KTX_ATTN_SEPARATE=1" "KTX_MOE_SEPARATE=1"; do
    tag=$(echo "$cfg" | tr ' =' '__')
    env $cfg python $R/bench.py --steps 60 --warmup 5 --windows 1 --no-prefill --no-secondary --no-cpu-baseline --no-pmc 2>/dev/null > $O/bench_$tag.json
    python - <<PY | tee -a $O/deepdive.txt
import json
d = json.load(open("$O/bench_$tag.json"))
print("== $cfg :", d["ms_per_step"], "ms/step")
for r in d.get("per_kernel", [])[:14]:
    print("   ", {k: r[k] for k in ("kernel", "avg_launch_us", "launches_per_step", "us_per_step", "GBs") if k in r})
PY
  done
  python $R/scripts/model_fused_stamps.py 32 2>&1 | grep -v amdgpu.ids | tee -a $O/deepdive.txt
fi

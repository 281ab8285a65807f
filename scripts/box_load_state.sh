#!/bin/bash
# Clocks / power of the GPU WHILE the decode graph replays (rocm-smi at idle says nothing about the DPM states under load).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-boxload}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 3000 --warmup 5 --windows 0 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null > $O/bench_long.json &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp --showuse 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk\|Power (W)\|junction\|GPU use\|HBM" | tr -s ' \t' ' ' | tr '\n' ';'; echo
  sleep 1.5
done > $O/load_state.txt
wait $BP
python - <<PY
import json
d = json.load(open("$O/bench_long.json")); print("bench_long:", d["ms_per_step"], "ms/step", d.get("box", {}).get("driver"), d.get("box", {}).get("unique_id"))
PY
cat $O/load_state.txt

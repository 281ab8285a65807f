#!/bin/bash
# A/B of several builds of the library on one box: every ktransformers_amd/lib/ab/<name>/libktx_hip.so (dev builds, git-ignored) takes
# the place of ktransformers_amd/lib/libktx_hip.so in turn; per build the attention half alone (scripts/attn_fused_bench.py, with the
# phase stamps of workgroup 0) and the whole-model decode step (bench.py).  The shipped build is restored at the end.
#   [AB_ROUNDS=2] [AB_NO_STEP=1] bash scripts/ab_attn_libs.sh <tag> [pytest-file ...]   -> gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-ab}; mkdir -p $O
shift
LIB=$R/ktransformers_amd/lib
cd $R
/opt/rocm/bin/rocm-smi --showuniqueid | grep Unique | tee $O/box.txt
if [ $# -gt 0 ]; then timeout 1500 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt; fi
cp $LIB/libktx_hip.so /tmp/shipped.so
for rnd in $(seq 1 ${AB_ROUNDS:-2}); do
  for d in $LIB/ab/*/; do
    which=$(basename $d)
    cp $d/libktx_hip.so $LIB/libktx_hip.so
    echo "== $which (round $rnd)" | tee -a $O/attn.txt $O/step.txt
    timeout 600 python scripts/attn_fused_bench.py 16 4096 x 2>&1 | grep -v "^$\|amdgpu.ids\|=====" | tee -a $O/attn.txt
    [ -n "$AB_NO_STEP" ] && continue
    timeout 900 python bench.py --steps 100 --warmup 10 --windows 2 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null > $O/b_${which}_$rnd.json
    python - <<PY | tee -a $O/step.txt
import json
d = json.load(open("$O/b_${which}_$rnd.json"))
print(f"{d['ms_per_step']:.4f} ms/step  {d['value']:.2f} tok/s  median {d.get('median_tok_s')}  bs8 {(d.get('bs8') or {}).get('value')}")
PY
  done
done
cp /tmp/shipped.so $LIB/libktx_hip.so

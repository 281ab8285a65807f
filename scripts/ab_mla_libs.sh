#!/bin/bash
# A/B of two builds of the library on one box for the stand-alone MLA decode kernel: ktransformers_amd/lib/ab/<name>/libktx_hip.so in turn;
# scripts/mla_sweep.py at several context lengths (the split-KV kernel and the merge timed separately).
#   bash scripts/ab_mla_libs.sh <tag> [pytest-file ...]   -> gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-abmla}; mkdir -p $O
shift
LIB=$R/ktransformers_amd/lib
cd $R
/opt/rocm/bin/rocm-smi --showuniqueid | grep Unique | tee $O/box.txt
if [ $# -gt 0 ]; then timeout 1500 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt; fi
cp $LIB/libktx_hip.so /tmp/shipped.so
for d in $LIB/ab/*/; do
  which=$(basename $d)
  cp $d/libktx_hip.so $LIB/libktx_hip.so
  for ctx in 4096 16384 32768 131072; do
    layers=64; [ $ctx -ge 32768 ] && layers=16
    echo "== $which ctx $ctx" | tee -a $O/mla.txt
    timeout 600 python scripts/mla_sweep.py --heads 128 --ctx $ctx --layers $layers 2>&1 | grep -v "^$\|amdgpu.ids" | grep "^Hq=" | tee -a $O/mla.txt
  done
done
cp /tmp/shipped.so $LIB/libktx_hip.so

#!/bin/bash
# Every measurement call of round 3 starts here: box identity + a short bench; on a SLOW box (the decode step of the same
# binary takes >= 15 % longer than on most boxes) collect the evidence that is missing so far: in-model phase stamps,
# the isolated chain, and a kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-diag}; mkdir -p $O
/opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" | tee $O/id.txt
python $R/bench.py --steps 50 --warmup 5 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null > $O/bench_short.json
ms=$(python -c "import json; print(json.load(open('$O/bench_short.json'))['ms_per_step'])")
echo "bench: $ms ms/step"
slow=$(python -c "print(1 if $ms > ${SLOW_MS:-4.6} else 0)")
if [ "$slow" = "1" ]; then
  echo "SLOW BOX: collecting stamps"
  bash $R/scripts/box_report.sh $(basename $O) > /dev/null
  python $R/scripts/lin_stamps.py --direct --shapes 7168x7168n,7168x7168 2>&1 | grep -v amdgpu.ids | tee $O/lin_stamps_direct.txt
  python $R/scripts/lin_stamps.py --shapes 7168x7168n,7168x7168 2>&1 | grep -v amdgpu.ids | tee -a $O/lin_stamps_direct.txt
  LAYERS=32 python $R/scripts/model_stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/model_stamps32.txt
  python $R/scripts/lin_stamps.py --shapes 7168x2112n,16384x7168,7168x4096n 2>&1 | grep -v amdgpu.ids | tee $O/lin_stamps.txt
  KTX_ARENA=0 python $R/bench.py --steps 50 --warmup 5 --no-prefill --no-secondary --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('second bench run:', d['value'], d['ms_per_step'])"
fi

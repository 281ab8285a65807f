# Dev A/B: the whole-model decode step (bench.py, v3-int4) at 16 K / 24 K context with the one-launch attention bound at 12 K (dev knob 1 = 12) and at its default.
for K in 12 0; do for C in 16384 24576; do
python - <<PY
import sys, json, io, contextlib
from ktransformers_amd import _native as n
n.lib.ktx_debug_set(1, $K)
import bench
sys.argv = ["bench.py", "--ctx", "$C", "--steps", "60", "--warmup", "10", "--windows", "1", "--no-prefill", "--no-secondary", "--no-cpu-baseline", "--no-pmc", "--no-kernels"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("bound", "$K" if $K else "32 (default)", "K  ctx $C:", d["ms_per_step"], "ms/step", d["value"], "tok/s")
PY
done; done

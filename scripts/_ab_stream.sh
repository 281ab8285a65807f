mkdir -p gpurun_out/r6z
for T in 1024 1536 2048 2560 3072 4096; do for kn in "25=0" "25=2"; do
python scripts/fmt_prompt_bench.py --fmt AMXINT4 --T $T --knob $kn 2>&1 | tail -1
done; done | tee gpurun_out/r6z/stream_tile80_sweep.txt

mkdir -p gpurun_out/r6A
timeout 600 python -m pytest tests/test_mla_gpu.py -x -q -k prefill 2>&1 | tail -2
python scripts/mla_prefill_bench.py 2048 2>&1 | tail -4 | tee gpurun_out/r6A/mla_prefill_bench.txt
python scripts/mla_prefill_bench.py 8192 2>&1 | tail -4 | tee -a gpurun_out/r6A/mla_prefill_bench.txt
python bench.py --steps 20 --warmup 5 --windows 0 --no-cpu-baseline --no-batched --no-secondary --no-pmc > gpurun_out/r6A/bench_prefill.json 2> gpurun_out/r6A/bench_prefill.err; tail -c 1500 gpurun_out/r6A/bench_prefill.json; cp gpurun_out/bench_detail.json gpurun_out/r6A/bench_detail.json

timeout 900 python -m pytest tests/test_attn_fused_gpu.py tests/test_mla_gpu.py -x -q 2>&1 | tail -3

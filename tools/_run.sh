timeout 900 python -m pytest tests/test_linear_gpu.py tests/test_modules_gpu.py -x -q -m gpu 2>&1 | tail -2
RLIPV2_FUSED_FFN=0 timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{' | cut -c1-330
RLIPV2_FUSED_FFN=1 timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{' | cut -c1-330

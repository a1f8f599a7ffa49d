timeout 900 python -m pytest tests/test_linear_gpu.py -x -q -m gpu 2>&1 | tail -1
timeout 300 python tools/wgrad_small.py 2>&1 | grep "T=" | head -8

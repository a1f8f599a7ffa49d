timeout 900 python -m pytest tests/test_msda_gpu.py -x -q -m gpu 2>&1 | tail -1
timeout 300 python tools/msda_microbench.py --quick --variants quad --out gpurun_out/mb.json 2>&1 | grep -E "bfloat16 +quad +fwd"

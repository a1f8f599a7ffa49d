timeout 600 python -m pytest tests/test_linear_gpu.py -x -q -m gpu -k "expand or fused" 2>&1 | tail -1
ONLY=2048 timeout 300 python tools/expand_bench.py 2>&1 | grep "N="
RLIPV2_LIB_PATH=$PWD/rlipv2_amd/_timeline.so timeout 300 python tools/expand_timeline.py 2>&1 | grep -v -i warn | grep -A2 "dgrad"

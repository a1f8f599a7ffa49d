#!/bin/bash
# GPU visit for the model stack: module/model parity tests, then the train-step bench.
TAG=${1:-m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest modules gpu"
timeout 900 python -m pytest tests/test_modules_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_modules.txt
echo "== bench train_step"
timeout 1200 python bench.py --steps ${2:-5} --warmup 2 ${3:-} 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/bench_train.json

#!/bin/bash
# 1-rank RCCL checks of the data-parallel step (run on the GPU box through gpurun)
set -x
mkdir -p gpurun_out
python -m pytest tests/test_modules_gpu.py -x -q -k "overlapped or one_rank" > gpurun_out/dp_tests_full.log 2>&1; grep -E "passed|failed|Error|error:" gpurun_out/dp_tests_full.log | tail -8 > gpurun_out/dp_tests.log
STEP_DP=1 python tools/step_timeline.py 8 > gpurun_out/dp_timeline_overlap.log 2>&1
STEP_DP=1 RLIPV2_DP_OVERLAP=0 python tools/step_timeline.py 8 > gpurun_out/dp_timeline_flat.log 2>&1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
# (the three schedules of bench.py on a 1-rank RCCL group: auto = train.choose_dp_schedule's verdict, reported in config.parallelism)
RLIPV2_FORCE_DP=1 python bench.py --no-cpu-baseline --dp-schedule overlapped > gpurun_out/dp_bench_overlap.log 2>&1
RLIPV2_FORCE_DP=1 python bench.py --no-cpu-baseline --dp-schedule flat > gpurun_out/dp_bench_flat.log 2>&1
RLIPV2_FORCE_DP=1 python bench.py --no-cpu-baseline --dp-schedule auto > gpurun_out/dp_bench_auto.log 2>&1

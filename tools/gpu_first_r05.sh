#!/bin/bash
# First GPU call of round 5: the GPU suite in evidence order (no -x: every failure is wanted), smoke, one bench line.
#   gpurun --timeout 1800 -- 'bash tools/gpu_first_r05.sh'
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocm-smi --showproductname > $OUT/box.txt 2>&1
( timeout 900 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt )
tail -30 $OUT/pytest_gpu.txt
( timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt ); tail -3 $OUT/smoke.txt
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
tail -c 3000 $OUT/bench_line.json; tail -5 $OUT/bench_stderr.txt
python tools/promote_r05.py $OUT/bench_line.json > $OUT/promote.txt 2>&1; cat $OUT/promote.txt

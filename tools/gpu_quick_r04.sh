#!/bin/bash
# The short form of tools/gpu_reopen_r04.sh for a GPU that appears late in a session: GPU suite + smoke + one bench line
# (~25 minutes).    gpurun --timeout 1800 -- 'bash tools/gpu_quick_r04.sh'
OUT=$GRAFT_REPO_ROOT/gpurun_out/quick_r04
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt )
tail -8 $OUT/pytest_gpu.txt
( timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt ); tail -3 $OUT/smoke.txt
timeout 420 python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
tail -c 1500 $OUT/bench_line.json

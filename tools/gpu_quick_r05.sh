#!/bin/bash
# When only a few GPU minutes are left: the op-level parity tests of the product path, the tests of the records route, its A/B at the
# bench shape and one bench line without the experiments leg (~10 minutes).  The full first call is tools/gpu_first_r05.sh.
#   gpurun --timeout 780 -- 'bash tools/gpu_quick_r05.sh'
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 240 python -m pytest tests/test_msda_gpu.py -m gpu -q -x > $OUT/pytest_msda.txt 2>&1; echo "rc=$?" >> $OUT/pytest_msda.txt ); tail -3 $OUT/pytest_msda.txt
( timeout 180 python -m pytest tests/test_zzz_records_gpu.py -m gpu -q > $OUT/pytest_records.txt 2>&1; echo "rc=$?" >> $OUT/pytest_records.txt ); tail -12 $OUT/pytest_records.txt
( timeout 90 python tools/experiments_r05.py --records > $OUT/records_ab.txt 2>&1 ); tail -c 2500 $OUT/records_ab.txt
timeout 240 python bench.py --no-experiments --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/bench_stderr.txt; tail -c 1500 $OUT/bench_line.json

#!/bin/bash
# One GPU-box visit: parity tests, smoke, kernel microbench, bench line, rocprof kernel stats.
# usage (from the repo root, via gpurun): bash tools/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo" ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== microbench"
timeout 900 python tools/msda_microbench.py --quick --out $OUT/microbench.json 2>&1 | tail -60 | tee $OUT/microbench.txt
echo "== bench"
timeout 900 python bench.py --steps 10 --warmup 2 2>$OUT/bench.stderr | tail -1 | tee $OUT/bench.json
echo "== rocprof"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1 )
find $OUT/prof -name "*kernel_stats*" | head -3
for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -12 $f | cut -c1-200; cp $f $OUT/kernel_stats.csv; done
# keep the merge-back small
find $OUT/prof -name "*kernel_trace*" -size +20M -delete
du -sh $OUT

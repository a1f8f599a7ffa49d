#!/bin/bash
# rocprofv3 kernel trace of the graphed train-step bench + gap analysis on the box
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-gaps}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 tools/gap_analysis.py "$f" ${2:-4} | tee $OUT/gaps.txt
rm -f "$f"

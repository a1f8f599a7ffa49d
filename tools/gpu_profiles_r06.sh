#!/bin/bash
# Profiles of a GPU session (run on the GPU box through gpurun; last step of tools/gpu_triage_r06.sh; was gpu_final_r03.sh):
# kernel-trace stats of the bench command, HBM-traffic
# counter passes (separate --pmc runs, --kernel-trace only) of the MSDA kernels the train step runs (fused route) and of the
# B0-signature kernels, the bench line itself.  The program goes directly after `--` (no env / bash -c hop).
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"      # (gpurun exports it; a local run falls back to the tree the script is in)
TAG=${1:-r06/final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 )
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/msda_$C -o p -- python3 $GRAFT_REPO_ROOT/tools/fused_once.py bwd 5 > $OUT/log_fusedbwd_$C.txt 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/fwd_$C -o p -- python3 $GRAFT_REPO_ROOT/tools/fused_once.py fwd 5 > $OUT/log_fusedfwd_$C.txt 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/b0_$C -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd_once.py auto bf16 init 5 > $OUT/log_b0bwd_$C.txt 2>&1 )
done
cd $GRAFT_REPO_ROOT
# (the table is also copied to gpurun_out/<tag>_final_traffic.json: commit it as profiles/r0N_final_traffic.json -- the newest
#  such file is what bench.py's roofline.traffic quotes)
python3 tools/pmc_final_summary.py $OUT $GRAFT_REPO_ROOT/gpurun_out/$(basename $TAG)_final_traffic.json > $OUT/traffic_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
cat $OUT/traffic_summary.txt
f=$(find $OUT/bench_stats -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-160

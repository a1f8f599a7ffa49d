#!/bin/bash
# A/B of dest_kernel's item queues (one global queue vs one per XCD), ablation build: kernel durations + HBM fetch
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/destxcd; mkdir -p $O
export RLIPV2_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/librlipv2_msda_ablation.so
for x in 0 1 0 1; do
    export RLIPV2_DEST_XCD=$x
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$x -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd_once.py dest bf16 model 20 > $O/log.txt 2>&1 )
    echo "xcd=$x $(grep dest_kernel $O/t_$x/p_kernel_stats.csv | awk -F'","' '{print "calls", $2, "avg_ns", $4}')" | tee -a $O/summary.txt
    find $O -name "*kernel_trace.csv" -delete
done
for x in 0 1; do
    export RLIPV2_DEST_XCD=$x
    ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f_$x -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd_once.py dest bf16 model 5 > $O/log.txt 2>&1 )
    python3 - $O/f_$x <<'P' | tee -a $O/summary.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "dest_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    print(sys.argv[1][-3:], "FETCH_SIZE (raw, x64 B... as reported) per launch:", sum(v) / max(len(v), 1))
P
done

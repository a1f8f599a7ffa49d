#!/bin/bash
# per-kernel durations of the weight-gradient ablation arms (rocprofv3 --kernel-trace --stats, ablation build)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/wgprof; mkdir -p $O
export RLIPV2_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/librlipv2_msda_ablation.so
for arm in "0 512 0" "0 512 1" "0 512 3" "0 512 5" "0 512 8"; do
    set -- $arm
    export RLIPV2_WGRAD_WIDE=$1 RLIPV2_WGRAD_BLOCKS=$2 RLIPV2_WGRAD_DBG=$3 WGRAD_SHAPES=2048x256
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/a_$1_$3 -o p -- python3 $GRAFT_REPO_ROOT/tools/wgrad_big.py child > $O/log_$1_$3.txt 2>&1 )
    echo "wide $1 dbg $3: $(grep -E 'wgrad|reduce_partials' $O/a_$1_$3/p_kernel_stats.csv | awk -F'","' '{printf "%s calls %s avg_ns %s | ", substr($1,1,60), $2, $4}')" | tee -a $O/summary.txt
    find $O -name "*kernel_trace.csv" -delete
done

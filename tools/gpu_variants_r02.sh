#!/bin/bash
# Non-best-case and other-config bench lines, one MI355X.
OUT=$GRAFT_REPO_ROOT/gpurun_out/variants_r02
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; python bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err; }
run default
run padded --padded
run var_targets --var-targets
run eager --no-graph
run batch8 --batch 8
run swin_large_b2 --backbone swin_large --batch 2
run msda_step --workload msda_step

#!/bin/bash
# SECOND call when a GPU is available again (rounds 3-5 ended without one; the first is tools/gpu_first_r05.sh: suite + smoke +
# one bench line with the experiments table).  Order = value of the evidence per GPU minute:
#   1. the GPU suite + smoke on the product library (HEAD has never run on hardware)            -> profiles/r0N_gputest_*.txt
#   2. one bench line (the driver's metric)                                                     -> bench_line.json
#   3. A/B of round 4's host-side restructures (tools/r04_host_ab.py)                           -> host_ab.txt
#   4. every arm of tools/experiments_r05.py (--all: no time budget) + tools/promote_r05.py's reading -> experiments.json, promote.txt
#   5. cell_forward_kernel: opt-in tests + timing + a bench line with it                        -> pytest_cell_forward.txt, bench_line_fwd_cell.json
#      and a bench line with the records route (msda.records_route)                             -> bench_line_records.json
#   6. kernel stats + HBM-traffic counter passes of the kernels that run (tools/gpu_final_r03.sh, separate --pmc passes)
#   7. Swin-L (config 4) bench line with its self-checked routes, and with them forced off     -> bench_line_swin*.json
#   gpurun --timeout 3300 -- 'bash tools/gpu_reopen_r05.sh'
OUT=$GRAFT_REPO_ROOT/gpurun_out/reopen_r05
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt )
tail -8 $OUT/pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt ); tail -3 $OUT/smoke.txt
timeout 900 python bench.py --no-experiments > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
( timeout 1000 python tools/experiments_r05.py --all > $OUT/experiments.json 2> $OUT/experiments_table.txt ); cat $OUT/experiments_table.txt
python tools/promote_r05.py $OUT/experiments.json > $OUT/promote.txt 2>&1; cat $OUT/promote.txt
tail -c 1500 $OUT/bench_line.json
( timeout 900 python tools/r04_host_ab.py 20 > $OUT/host_ab.txt 2>&1 ); cat $OUT/host_ab.txt | tail -12
( RLIPV2_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_msda_cell_forward_gpu.py -q -m gpu > $OUT/pytest_cell_forward.txt 2>&1; timeout 300 python tools/cell_forward_check.py >> $OUT/pytest_cell_forward.txt 2>&1 )
tail -25 $OUT/pytest_cell_forward.txt
timeout 600 python bench.py --no-cpu-baseline --no-experiments --msda-fwd-cell > $OUT/bench_line_fwd_cell.json 2> $OUT/bench_fwd_cell_stderr.txt
tail -c 600 $OUT/bench_line_fwd_cell.json
# the records route of the encoder forward / backward pair in the whole step (only worth reading if experiments.json above shows
# its gradients bit-equal): product line above against this one
timeout 600 python bench.py --no-cpu-baseline --no-experiments --set msda.records_route=1 > $OUT/bench_line_records.json 2> $OUT/bench_records_stderr.txt
tail -c 600 $OUT/bench_line_records.json
# brief item 6 decided by one measurement: today's chunk plan against one chunk + direct stores (small-token Linears)
( export RLIPV2_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/librlipv2_msda_ablation.so; for m in 8 100000; do RLIPV2_WGRAD_MINSTEPS=$m timeout 300 python tools/wgrad_plan_ab.py; done > $OUT/wgrad_plan_ab.txt 2>&1 ); cat $OUT/wgrad_plan_ab.txt
timeout 700 python bench.py --no-cpu-baseline --backbone swin_large --batch 2 > $OUT/bench_line_swin.json 2> $OUT/bench_swin_stderr.txt; tail -c 700 $OUT/bench_line_swin.json
timeout 700 python bench.py --no-cpu-baseline --backbone swin_large --batch 2 --host-routes off > $OUT/bench_line_swin_routes_off.json 2>> $OUT/bench_swin_stderr.txt; tail -c 300 $OUT/bench_line_swin_routes_off.json
bash tools/gpu_final_r03.sh reopen_r05/final

#!/bin/bash
# THE script of a GPU session (round 6; replaces gpu_first_r05.sh, gpu_quick_r05.sh, gpu_reopen_r05.sh).  Order = value of the
# evidence per GPU minute; every step under its own timeout, nothing after a step depends on it having succeeded.
#   gpurun --timeout 3300 -- 'bash tools/gpu_triage_r06.sh'            # everything below (~45-55 min: tight for one call, or split:)
#   gpurun --timeout 900  -- 'bash tools/gpu_triage_r06.sh quick'      # steps 1-3 only (~10 min): parity, smoke, the bench line
#   gpurun --timeout 1500 -- 'bash tools/gpu_triage_r06.sh suite'      # step 4 only (~15-20 min)
#   gpurun --timeout 1800 -- 'bash tools/gpu_triage_r06.sh triage'     # step 5 only (<= 20 min + process starts)
#   gpurun --timeout 1800 -- 'bash tools/gpu_triage_r06.sh profiles'   # step 6 only (~15-25 min)
#   gpurun --timeout 3300 -- 'bash tools/gpu_triage_r06.sh extras'     # the A/B bench lines and host A/Bs of a second session
# 1. op-level parity of the hot path against the oracle / goldens (tests/test_msda_gpu.py)                -> pytest_msda.txt
# 2. smoke()                                                                                             -> smoke.txt
# 3. ONE default bench line (product kernels only; the host-route self-check runs in a child process)    -> bench_line.json
# 4. the whole GPU suite, no -x (every failure is wanted)                                                -> pytest_gpu.txt
# 5. triage of the never-run kernels, family by family (tools/gpu_triage_r06.py, <= 20 min)               -> triage.json / .txt
# 6. kernel stats + HBM-traffic counter passes of the kernels that run (tools/gpu_profiles_r06.sh)       -> final/
# Copy what is to be judged from gpurun_out/r06/ into profiles/r06_*.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"      # (gpurun exports it; a local run falls back to the tree the script is in)
MODE=${1:-all}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocm-smi --showproductname > $OUT/box.txt 2>&1
if [ "$MODE" = "extras" ]; then
  ( timeout 900 python tools/r04_host_ab.py 20 > $OUT/host_ab.txt 2>&1 ); tail -12 $OUT/host_ab.txt
  timeout 600 python bench.py --no-cpu-baseline --msda-fwd-cell > $OUT/bench_line_fwd_cell.json 2> $OUT/bench_fwd_cell_stderr.txt; tail -c 600 $OUT/bench_line_fwd_cell.json
  timeout 600 python bench.py --no-cpu-baseline --set msda.records_route=1 > $OUT/bench_line_records.json 2> $OUT/bench_records_stderr.txt; tail -c 600 $OUT/bench_line_records.json
  ( [ -f tools/_build/librlipv2_msda_ablation.so ] || make -s -C rlipv2_amd/csrc -j8 ablation; export RLIPV2_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/librlipv2_msda_ablation.so; for m in 8 100000; do RLIPV2_WGRAD_MINSTEPS=$m timeout 300 python tools/wgrad_plan_ab.py; done > $OUT/wgrad_plan_ab.txt 2>&1 ); cat $OUT/wgrad_plan_ab.txt
  timeout 700 python bench.py --no-cpu-baseline --backbone swin_large --batch 2 > $OUT/bench_line_swin.json 2> $OUT/bench_swin_stderr.txt; tail -c 700 $OUT/bench_line_swin.json
  timeout 700 python bench.py --no-cpu-baseline --backbone swin_large --batch 2 --host-routes off > $OUT/bench_line_swin_routes_off.json 2>> $OUT/bench_swin_stderr.txt; tail -c 300 $OUT/bench_line_swin_routes_off.json
  ( timeout 900 bash tools/gpu_dp_check.sh > $OUT/dp_check.txt 2>&1 ); grep -h "parallelism" gpurun_out/dp_bench_auto.log | cut -c1-400
  ( timeout 1000 python tools/experiments_r05.py --all > $OUT/experiments.json 2> $OUT/experiments_table.txt ); python tools/promote_r05.py $OUT/experiments.json > $OUT/promote.txt 2>&1; cat $OUT/promote.txt
  exit 0
fi
want() { [ "$MODE" = "all" ] || [ "$MODE" = "$1" ] || { [ "$MODE" = "quick" ] && [ "$1" = "first" ]; }; }
if want first; then
( timeout 300 python -m pytest tests/test_msda_gpu.py -m gpu -q > $OUT/pytest_msda.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_msda.txt ); tail -4 $OUT/pytest_msda.txt
( timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt ); tail -3 $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
tail -c 3000 $OUT/bench_line.json; tail -5 $OUT/bench_stderr.txt
fi
if want suite; then
( timeout 1200 python -m pytest tests -m gpu -q -rxX --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt ); tail -45 $OUT/pytest_gpu.txt
fi
if want triage; then
timeout 1500 python tools/gpu_triage_r06.py --out $OUT | tee $OUT/triage_lines.txt
cat $OUT/triage.txt
fi
if want profiles; then
bash tools/gpu_profiles_r06.sh r06/final
fi

#!/bin/bash
# First call after the GPU pool reopens (round 3): the GPU test suite on the product library, the A/B table of the
# default-off experiment arms (ablation build; every arm must reproduce the default's gradients bit for bit), one bench line.
#   gpurun --timeout 2400 -- 'bash tools/gpu_reopen_r03.sh'
OUT=$GRAFT_REPO_ROOT/gpurun_out/reopen_r03
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt )
tail -5 $OUT/pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt ); tail -3 $OUT/smoke.txt
( export RLIPV2_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/librlipv2_msda_ablation.so; timeout 900 python tools/r03_experiments.py > $OUT/experiments.txt 2>&1 )
cat $OUT/experiments.txt
# where the cycles of cell_backward_kernel go, product kernel vs arm 3 (cycle stamps of thread 0, summed over the workgroups)
( export RLIPV2_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/librlipv2_msda_ablation.so; for m in 0 3; do echo "== cell timeline, RLIPV2_CELL_SHARED=$m"; RLIPV2_CELL_SHARED=$m timeout 300 python tools/cell_timeline.py init; done > $OUT/cell_timeline.txt 2>&1 )
cat $OUT/cell_timeline.txt
# LDS bank conflicts of the patch pass, product layout vs the experimental instantiation (operand images with exchanged halves):
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel (counter pass: --kernel-trace only, the program directly after --)
for arm in 0 1; do
  ( export RLIPV2_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/librlipv2_msda_ablation.so RLIPV2_PATCH_MULTI=$arm; cd /tmp && export TMPDIR=/tmp && \
    timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/lds_multi$arm -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd_once.py auto bf16 init 3 > $OUT/log_lds_multi$arm.txt 2>&1 )
done
python3 - <<'PY' > $OUT/lds_conflicts.txt 2>&1
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/reopen_r03"
for arm in (0, 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{out}/lds_multi{arm}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "patch_dest" in k or "cell_backward" in k:
            c, a = v.get("SQ_LDS_BANK_CONFLICT", 0.0), v.get("SQ_LDS_IDX_ACTIVE", 0.0)
            print(f"RLIPV2_PATCH_MULTI={arm}  {k:60s} conflict cycles {c:.3e}  LDS-array cycles {a:.3e}  ratio {c / a if a else 0:.3f}")
PY
cat $OUT/lds_conflicts.txt
( RLIPV2_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_msda_cell_forward_gpu.py -q -m gpu > $OUT/pytest_cell_forward.txt 2>&1; timeout 300 python tools/cell_forward_check.py >> $OUT/pytest_cell_forward.txt 2>&1 )
tail -25 $OUT/pytest_cell_forward.txt
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
tail -c 1500 $OUT/bench_line.json
# the train step with the encoder forward through the geometry kernel + cell_forward_kernel (only meaningful once the tests above pass)
timeout 600 python bench.py --no-cpu-baseline --msda-fwd-cell > $OUT/bench_line_fwd_cell.json 2> $OUT/bench_fwd_cell_stderr.txt
tail -c 600 $OUT/bench_line_fwd_cell.json

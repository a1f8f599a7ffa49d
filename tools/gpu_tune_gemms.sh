#!/bin/bash
# Full TunableOp pass over one eager train step, then an A/B of the resulting table against the shipped one (same box).
mkdir -p gpurun_out/tg
TUNE_MS=${TUNE_MS:-60} RLIPV2_TUNED_GEMMS=0 python tools/tune_gemms.py > gpurun_out/tg/tune.log 2>&1
tail -2 gpurun_out/tg/tune.log
cp gpurun_out/tunableop_step.csv gpurun_out/tg/full.csv; wc -l gpurun_out/tg/full.csv
for t in shipped full shipped full; do
    if [ $t = full ]; then export RLIPV2_TUNED_GEMM_TABLE=$PWD/gpurun_out/tg/full.csv; else unset RLIPV2_TUNED_GEMM_TABLE; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/tg/bench_$t.json 2> gpurun_out/tg/bench_$t.err
    echo "$t: $(python -c "import json; d=json.loads(open('gpurun_out/tg/bench_$t.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a gpurun_out/tg/summary.txt
done

#!/bin/bash
# A/B of the package's switches on ONE box: tools/gpu_ab.sh NAME "--set decoder.fused_glue=0" "--set alif.fused_attention=0 --no-graph" ...
# (each quoted argument = the extra bench.py flags of one run; "" = the product configuration)
out=gpurun_out/$1; shift
mkdir -p $out
i=0
for flags in "$@"; do
    i=$((i + 1))
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-experiments $flags > $out/run$i.json 2> $out/run$i.err
    echo "[$flags] $(python -c "import json,sys; d=json.loads(open('$out/run$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1)" | tee -a $out/summary.txt
done

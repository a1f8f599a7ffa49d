#!/bin/bash
# A/B of environment switches on ONE box: tools/gpu_ab.sh NAME "ENV=.. ENV=.." "ENV=.." ...   (each quoted set = one bench run)
out=gpurun_out/$1; shift
mkdir -p $out
i=0
for envs in "$@"; do
    i=$((i + 1))
    env $envs python bench.py --steps 30 --warmup 8 > $out/run$i.json 2> $out/run$i.err
    echo "[$envs] $(python -c "import json,sys; d=json.loads(open('$out/run$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1)" | tee -a $out/summary.txt
done

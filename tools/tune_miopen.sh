#!/bin/bash
# Records MIOpen's Find results for the convolution shapes of the train step into a user find-db under
# gpurun_out/miopen_db (copy the files to rlipv2_amd/tuned/miopen/ to ship them; rlipv2_amd/__init__.py points
# MIOPEN_USER_DB_PATH there).  Runs on the GPU box:  gpurun -- bash tools/tune_miopen.sh
set -u
db=$PWD/gpurun_out/miopen_db
rm -rf $db; mkdir -p $db
export MIOPEN_USER_DB_PATH=$db
export RLIPV2_TUNED_MIOPEN=0
for flags in "" "--padded" "--batch 2" "--batch 8" "--precision autocast" "--backbone swin_tiny"; do
    t0=$(date +%s)
    RLIPV2_MIOPEN_FIND=1 python bench.py --steps 3 --warmup 3 --no-cpu-baseline $flags > $db/../tune_$(echo $flags | tr -d ' -').json 2> $db/../tune_err.txt
    echo "find [$flags]: $(( $(date +%s) - t0 )) s" | tee -a $db/../tune_log.txt
done
ls -la $db | tee -a $db/../tune_log.txt
for mode in tuned untuned; do
    if [ $mode = untuned ]; then export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_empty; mkdir -p $MIOPEN_USER_DB_PATH; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $db/../tune_check_$mode.json 2>> $db/../tune_err.txt
    echo "$mode: $(python -c "import json; d=json.loads(open('$db/../tune_check_$mode.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")" | tee -a $db/../tune_log.txt
done

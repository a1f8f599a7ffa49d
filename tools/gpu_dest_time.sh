#!/bin/bash
# dest_kernel duration (rocprofv3 --kernel-trace --stats, encoder shape, bf16, model-like locations), shipped library
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/desttime; mkdir -p $O
for r in 1 2; do
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$r -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd_once.py dest bf16 model 20 > $O/log.txt 2>&1 )
python3 - $O/t_$r <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + "/p_kernel_stats.csv")):
    if "msda" in r["Name"]:
        print(f"{r['Name'].split('(')[0][-40:]:42s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
P
find $O -name "*kernel_trace.csv" -delete
done

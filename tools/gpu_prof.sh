#!/bin/bash
# rocprofv3 kernel stats of a python script: bash tools/gpu_prof.sh <tag> <script> [args...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/"$@" > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} tot_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
find $OUT -name "*kernel_trace.csv" -size +10M -delete

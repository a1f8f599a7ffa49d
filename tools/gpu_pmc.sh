#!/bin/bash
# HBM traffic counters of the MSDA kernels (separate passes, as MI355X_MICROARCH.md prescribes):
# bash tools/gpu_pmc.sh <tag> <script> [args...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o p -- python3 $GRAFT_REPO_ROOT/"$@" > $OUT/log_$C.txt 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "msda" not in name: continue
            import re
            mm = re.search(r"(quad_forward_kernel|quad_forward_shared_kernel|quad_backward_shared_kernel|quad_backward_kernel|tile_forward_kernel|scatter_kernel|generic_\w+_kernel|prep_\w+_kernel)<([^>]*)>", name)
            short = f"{mm.group(1)}<{mm.group(2)}>" if mm else name[:60]
            res[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(res.items()):
    line = [k[:70]]
    for c, v in sorted(d.items()):
        line.append(f"{c} mean {sum(v)/len(v):.0f} KB x{len(v)}")
    print("  ".join(line))
PY
find $OUT -name "*kernel_trace.csv" -size +5M -delete

#!/bin/bash
# Shader-side counters of the MSDA backward kernels (one rocprofv3 --pmc pass per counter group):
# bash tools/gpu_pmc_sq.sh <tag>
TAG=${1:-pmcsq}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for GROUP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_INSTS_GDS"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/g$i -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd_chunk.py > $OUT/log_g$i.txt 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        mm = re.search(r"(quad_backward_shared_kernel|quad_backward_kernel|scatter_kernel)", name)
        if not mm: continue
        res[mm.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(res.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.0f}  (x{len(v)})")
PY
find $OUT -name "*kernel_trace.csv" -delete

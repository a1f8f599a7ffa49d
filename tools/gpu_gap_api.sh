#!/bin/bash
# rocprofv3 kernel + HIP-API trace of a short bench run: input of tools/gap_api.py (which HIP call the host sits in while the GPU idles)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-gapapi}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --hip-trace --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
kf=$(find $OUT -name "*kernel_trace.csv" | head -1); af=$(find $OUT -name "*hip_api_trace.csv" | head -1)
python3 tools/gap_api.py "$kf" "$af" | tee $OUT/gap_api.txt
rm -f "$kf" "$af"

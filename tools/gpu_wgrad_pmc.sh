#!/bin/bash
# SQ counters of the weight-gradient main kernel (2048x256, T = 88 892): LDS bank conflicts, LDS-array activity, MFMA busy
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/wgpmc; mkdir -p $O
export RLIPV2_LIB_PATH=$GRAFT_REPO_ROOT/tools/_build/librlipv2_msda_ablation.so
for arm in "0 512 0" "0 512 3" "0 512 5"; do
    set -- $arm
    export RLIPV2_WGRAD_WIDE=$1 RLIPV2_WGRAD_BLOCKS=$2 RLIPV2_WGRAD_DBG=$3 WGRAD_SHAPES=2048x256
    for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS"; do
        tag=$(echo $C | tr ' ' '+')
        ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/a_$1_$3_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/wgrad_big.py child > $O/log.txt 2>&1 )
    done
done
python3 - <<'P' > $O/summary.txt
import csv, glob, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "wgpmc")
for d in sorted(glob.glob(root + "/a_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if "wgrad" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        print(os.path.basename(d), {k: round(v[1] / v[0]) for k, v in agg.items()})
P
find $O -name "*.csv" -size +1M -delete
cat $O/summary.txt

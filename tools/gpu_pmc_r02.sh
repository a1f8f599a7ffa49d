#!/bin/bash
# Round-2 counter passes (each rocprofv3 --pmc run separately, with --kernel-trace only, as MI355X_MICROARCH.md prescribes):
#   1. HBM traffic (FETCH_SIZE / WRITE_SIZE) of the MSDA forward + destination-stationary backward kernels
#   2. MFMA-busy / busy cycles of the dense kernels of one eager train step (own wgrad / expand kernels, hipBLASLt)
# bash tools/gpu_pmc_r02.sh <tag>
TAG=${1:-pmc_r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/msda_$C -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd_once.py dest bf16 model 5 > $OUT/log_msda_$C.txt 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/fwd_$C -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd_once.py fwd bf16 model 5 > $OUT/log_fwd_$C.txt 2>&1 )
done
i=0
for GROUP in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_BF16 SQ_WAVE_CYCLES"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/mfma_g$i -o p -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py 3 > $OUT/log_mfma_g$i.txt 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_r02_summary.py $OUT | tee $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete

#!/bin/bash
# Fast GPU iteration: selected parity tests + kernel microbench.
# usage: bash tools/gpu_quick.sh <tag> "<pytest -k expr>" "<variants>"
TAG=${1:-q}
KEXPR=${2:-"golden or random or ragged or encoder_backward"}
VARIANTS=${3:-"generic,quad,window"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest -m gpu -k '$KEXPR'"
timeout 1200 python -m pytest tests -x -q -m gpu -k "$KEXPR" 2>&1 | tail -30 | tee $OUT/pytest_gpu.txt
echo "== microbench"
timeout 900 python tools/msda_microbench.py --quick --variants "$VARIANTS" --out $OUT/microbench.json 2>&1 | grep -v amdgpu.ids | tail -60 | tee $OUT/microbench.txt

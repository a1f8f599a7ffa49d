#!/bin/bash
# Ablation builds of csrc/expand_gemm.hip (XDBG bits: 1 no MFMA, 2 no fragment reads, 4 no output stores,
# 16 no B-tile DMA, 32 no A loads) timed with tools/expand_bench.py.  Build part runs where hipcc is; `run` on the GPU.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p gpurun_out/ablate
  for d in 0 1 2 3 4 16 32 63; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DXDBG=$d -c rlipv2_amd/csrc/expand_gemm.hip -o /tmp/xg_$d.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rlipv2_amd/_ablate_$d.so $(ls rlipv2_amd/csrc/_obj/*.o | grep -v expand_gemm) /tmp/xg_$d.o
  done
else
  for d in 0 1 2 3 4 16 32 63; do
    echo "== XDBG $d"; RLIPV2_LIB_PATH=$PWD/rlipv2_amd/_ablate_$d.so ONLY=2048 timeout 300 python tools/expand_bench.py 2>&1 | grep "N="
  done
fi

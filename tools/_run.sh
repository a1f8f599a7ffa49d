timeout 900 python -m pytest tests/test_msda_gpu.py -x -q -m gpu 2>&1 | tail -1
for lib in librlipv2_msda.so _k2nolb.so; do echo "$lib"; RLIPV2_LIB_PATH=$PWD/rlipv2_amd/$lib timeout 600 python tools/msda_microbench.py --quick --variants window --out gpurun_out/mb.json 2>&1 | grep -E "^enc .*(model|uniform) .*(bfloat16|float32) +window +bwd|^dec300.*bfloat16.*window +bwd"; done
echo "debug 2 (no atomics) / grid 256"
RLIPV2_MSDA_DEBUG=2 timeout 600 python tools/msda_microbench.py --quick --variants window --out gpurun_out/mb.json 2>&1 | grep -E "^enc .*(model) .*(bfloat16) +window +bwd"
RLIPV2_MSDA_GRID=256 timeout 600 python tools/msda_microbench.py --quick --variants window --out gpurun_out/mb.json 2>&1 | grep -E "^enc .*(model) .*(bfloat16) +window +bwd"

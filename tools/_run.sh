timeout 900 python -m pytest tests/test_msda_gpu.py -x -q -m gpu 2>&1 | tail -1
timeout 600 python tools/msda_microbench.py --quick --variants window --out gpurun_out/mb.json 2>&1 | grep -E "^enc .*(model|uniform) .*bfloat16 +window +bwd|^dec.*bfloat16.*window +bwd"
RLIPV2_LIB_PATH=$PWD/rlipv2_amd/_k2timeline.so timeout 300 python tools/k2_timeline.py model 2>&1 | grep "workgroup 0"

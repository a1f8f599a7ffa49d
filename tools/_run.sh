timeout 900 python -m pytest tests/test_msda_gpu.py -x -q -m gpu 2>&1 | tail -1
timeout 600 python tools/msda_microbench.py --quick --variants window --out gpurun_out/mb.json 2>&1 | grep -E "^enc .*(model|uniform) .*(bfloat16|float32) +window +bwd|^dec.*bfloat16.*window +bwd"

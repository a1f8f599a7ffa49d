s=$(date +%s); RLIPV2_MIOPEN_FIND=1 timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/find1.err | grep '^{' | cut -c1-330; echo "wall $(( $(date +%s) - s )) s"
s=$(date +%s); RLIPV2_MIOPEN_FIND=0 timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/find0.err | grep '^{' | cut -c1-330; echo "wall $(( $(date +%s) - s )) s"

for i in 1 2; do for v in 0 1; do echo -n "ROWVEC=$v "; RLIPV2_ROWVEC=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | cut -c190-215; done; done

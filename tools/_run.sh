RLIPV2_TUNED_GEMMS=1 timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep -v -i "warn" | tail -5 | cut -c1-400
RLIPV2_TUNED_GEMMS=0 timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep -v -i "warn" | tail -1 | cut -c1-400

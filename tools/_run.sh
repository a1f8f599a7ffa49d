timeout 1200 python tools/train_sanity.py 2>&1 | grep -v -i "warn\|run_backward" | tail -30

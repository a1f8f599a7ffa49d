echo new; timeout 300 python tools/prep_bench.py 2>&1 | grep -v -i "warn\|run_backward"
echo old; RLIPV2_LIB_PATH=$PWD/rlipv2_amd/_prep_old.so timeout 300 python tools/prep_bench.py 2>&1 | grep -v -i "warn\|run_backward"

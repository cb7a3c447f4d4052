#!/bin/bash
step() { echo "[$(date +%T)] $1"; }
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
step "tests"
timeout -k 10 900 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_widening.py tests/test_gpu_leftovers.py tests/test_gpu_large.py tests/test_gpu_fullsize.py tests/test_gpu_psd_shortcut.py tests/test_gpu_round2.py -m gpu -q -x > $OUT/r3e_tests.log 2>&1; echo "rc=$?" >> $OUT/r3e_tests.log; tail -5 $OUT/r3e_tests.log
step "large n timing"
{ timeout -k 10 300 python3 scripts/large_n_timing.py 5 256; timeout -k 10 300 python3 scripts/large_n_timing.py 4 1024; } > $OUT/r3e_large_n_timing.txt 2>&1; cat $OUT/r3e_large_n_timing.txt
step "process timing"
timeout -k 10 300 python3 scripts/process_timing.py > $OUT/r3e_process_timing.txt 2>&1; cat $OUT/r3e_process_timing.txt
step done

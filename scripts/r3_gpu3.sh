#!/bin/bash
# round 3, GPU session 3: sign-clip rewrite -- tests + timings of the kernels that use it
step() { echo "[$(date +%T)] $1"; }
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
step "GPU suite"
timeout -k 10 1500 python3 -m pytest tests -m gpu -q -x > $OUT/r3c_gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/r3c_gpu_tests.log; tail -6 $OUT/r3c_gpu_tests.log
step "large n timing"
{ timeout -k 10 300 python3 scripts/large_n_timing.py 5 256; timeout -k 10 300 python3 scripts/large_n_timing.py 5 2048; timeout -k 10 300 python3 scripts/large_n_timing.py 4 1024; } > $OUT/r3c_large_n_timing.txt 2>&1; cat $OUT/r3c_large_n_timing.txt
step "process timing"
timeout -k 10 300 python3 scripts/process_timing.py > $OUT/r3c_process_timing.txt 2>&1; cat $OUT/r3c_process_timing.txt
step done

#!/bin/bash
step() { echo "[$(date +%T)] $1"; }
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
step "tests (n = 4, 5 and everything that clips)"
timeout -k 10 900 python3 -m pytest tests/test_gpu_large.py tests/test_gpu_fullsize.py tests/test_gpu_leftovers.py tests/test_gpu_round2.py tests/test_gpu_selection.py -m gpu -q -x > $OUT/r3j_tests.log 2>&1; echo "rc=$?" >> $OUT/r3j_tests.log; tail -4 $OUT/r3j_tests.log
step "large n timing"
{ timeout -k 10 300 python3 scripts/large_n_timing.py 5 256; timeout -k 10 300 python3 scripts/large_n_timing.py 5 2048; timeout -k 10 300 python3 scripts/large_n_timing.py 4 1024; } 2>&1 | grep -v amdgpu
step "phase stamps n = 5"
QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so timeout -k 10 300 python3 scripts/phase_timing_large.py 5 256 2>&1 | grep -v amdgpu | head -36
step done

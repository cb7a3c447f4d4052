#!/bin/bash
# round 3, first GPU session: new tests, the whole GPU suite, the driver's bench command, idle ramp, warm ceilings
step() { echo "[$(date +%T)] $1"; }
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
step "new tests"
timeout -k 10 900 python3 -m pytest tests/test_gpu_selection.py -x -q -m gpu > $OUT/r3_sel.log 2>&1; echo "rc=$?" >> $OUT/r3_sel.log; tail -4 $OUT/r3_sel.log
step "whole GPU suite"
timeout -k 10 1200 python3 -m pytest tests -m gpu -q > $OUT/r3_gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/r3_gpu_tests.log; tail -4 $OUT/r3_gpu_tests.log
step "bench (driver command)"
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r3_bench_driver_cmd.json 2> $OUT/r3_bench_driver_cmd.err; echo "rc=$?"; tail -3 $OUT/r3_bench_driver_cmd.err
step "idle ramp"
timeout -k 10 300 python3 scripts/idle_ramp.py > $OUT/r3_idle_ramp.txt 2>&1; tail -20 $OUT/r3_idle_ramp.txt
step "warm ceilings"
timeout -k 10 300 ./scripts/ubench/valu_f64_warm > $OUT/r3_ubench_valu_f64_warm.txt 2>&1; cat $OUT/r3_ubench_valu_f64_warm.txt
timeout -k 10 300 ./scripts/ubench/mfma_f64_warm > $OUT/r3_ubench_mfma_f64_warm.txt 2>&1; cat $OUT/r3_ubench_mfma_f64_warm.txt
step done

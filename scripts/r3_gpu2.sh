#!/bin/bash
# round 3, GPU session 2: whole GPU suite, the driver's bench command, bench --profile
step() { echo "[$(date +%T)] $1"; }
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
step "whole GPU suite"
timeout -k 10 1500 python3 -m pytest tests -m gpu -q > $OUT/r3b_gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/r3b_gpu_tests.log; tail -6 $OUT/r3b_gpu_tests.log
step "bench (driver command)"
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r3b_bench_driver_cmd.json 2> $OUT/r3b_bench_driver_cmd.err; echo "rc=$?"; tail -3 $OUT/r3b_bench_driver_cmd.err
step "bench --profile"
timeout -k 10 900 python3 bench.py --profile > $OUT/r3b_profile.json 2> $OUT/r3b_profile.err; echo "rc=$?"; tail -5 $OUT/r3b_profile.err; cat $OUT/r3b_profile.json | head -c 1500
cp profiles/round3_pmc_traffic.json profiles/round3_bench_kernel_stats.csv $OUT/ 2>/dev/null
step done

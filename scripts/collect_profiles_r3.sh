#!/bin/bash
# Everything profiles/round3_* is copied from, in one GPU-box session:   bash scripts/collect_profiles_r3.sh [tag]
# rocprofv3: kernel-trace/stats and every PMC counter are SEPARATE passes, the program itself follows "--".
step() { echo "[$(date +%T)] $1"; }
TAG=${1:-r3final}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
step "pytest -m gpu"
timeout -k 10 1500 python3 -m pytest tests -m gpu -q > $OUT/${TAG}_gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_gpu_tests.log; tail -3 $OUT/${TAG}_gpu_tests.log
step "coverage tables (verbose)"
timeout -k 10 600 python3 -m pytest tests/test_gpu_moments.py -m gpu -q -s -k coverage > $OUT/${TAG}_coverage_tables.txt 2>&1; tail -3 $OUT/${TAG}_coverage_tables.txt
step "bench, driver command"
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_cmd.json 2> $OUT/${TAG}_bench_driver_cmd.err; echo "rc=$?"
step "bench --profile (PMC traffic + stats of the short command)"
timeout -k 10 900 python3 bench.py --profile > $OUT/${TAG}_profile.json 2> $OUT/${TAG}_profile.err; echo "rc=$?"
cp profiles/round3_pmc_traffic.json $OUT/${TAG}_pmc_traffic.json; cp profiles/round3_bench_kernel_stats.csv $OUT/${TAG}_short_cmd_kernel_stats.csv
step "bench again, driver command (now with the traffic of THIS build)"
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_cmd_with_traffic.json 2> $OUT/${TAG}_bench_driver_cmd2.err; echo "rc=$?"
step "rocprofv3 --kernel-trace --stats of the driver command"
( cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.err ); echo "rc=$?"
cp $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null
step "bench, default flags"
timeout -k 10 900 python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; echo "rc=$?"
step "bench, two gloo ranks sharing cuda:0 / one-rank RCCL"
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --share-gpu0 --no-cpu-baseline > $OUT/${TAG}_bench_2rank_gloo_shared_gpu.json 2> $OUT/${TAG}_bench_2rank.err; echo "rc=$?"
QT_BENCH_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench_1rank_rccl.json 2> $OUT/${TAG}_bench_1rank.err; echo "rc=$?"
step "timing scripts"
timeout -k 10 300 python3 scripts/headline_timing.py 500 > $OUT/${TAG}_headline_timing.txt 2>&1
timeout -k 10 300 python3 scripts/iterating_timing.py > $OUT/${TAG}_iterating_timing.txt 2>&1
{ timeout -k 10 300 python3 scripts/process_timing.py 1024; timeout -k 10 300 python3 scripts/process_timing.py 16384; timeout -k 10 300 python3 scripts/process_timing.py 64; } > $OUT/${TAG}_process_timing.txt 2>&1
{ timeout -k 10 300 python3 scripts/process_timing.py 64 3; timeout -k 10 300 python3 scripts/process_timing.py 256 3; timeout -k 10 300 python3 scripts/process_timing.py 1024 3; } > $OUT/${TAG}_process3_timing.txt 2>&1
timeout -k 10 600 python3 scripts/pgdb3_timing.py > $OUT/${TAG}_pgdb3_timing.txt 2>&1
timeout -k 10 300 python3 scripts/lifp16_stream_timing.py > $OUT/${TAG}_lifp16_stream_timing.txt 2>&1
{ timeout -k 10 300 python3 scripts/cp_hard_spectra.py; timeout -k 10 300 python3 scripts/cp_accuracy_probe.py; } > $OUT/${TAG}_cp_accuracy_n3.txt 2>&1
timeout -k 10 600 python3 scripts/clip_accuracy_n5.py > $OUT/${TAG}_clip_accuracy_n5.txt 2>&1
timeout -k 10 600 python3 scripts/cptp_sweep.py 96 > $OUT/${TAG}_cptp_sweep.txt 2>&1
{ timeout -k 10 300 python3 scripts/large_n_timing.py 5 256; timeout -k 10 300 python3 scripts/large_n_timing.py 5 2048; timeout -k 10 300 python3 scripts/large_n_timing.py 4 1024; } > $OUT/${TAG}_large_n_timing.txt 2>&1
timeout -k 10 300 python3 scripts/moment_coverage_timing.py > $OUT/${TAG}_moment_coverage_timing.txt 2>&1
timeout -k 10 300 python3 scripts/bootstrap_timing.py > $OUT/${TAG}_bootstrap_end_to_end.txt 2>&1
timeout -k 10 300 python3 scripts/idle_ramp.py > $OUT/${TAG}_idle_ramp_device_draw.txt 2>&1
timeout -k 10 300 ./scripts/ubench/valu_f64_warm > $OUT/${TAG}_ubench_valu_f64_warm.txt 2>&1
timeout -k 10 300 ./scripts/ubench/mfma_f64_warm > $OUT/${TAG}_ubench_mfma_f64_warm.txt 2>&1
step "phase stamps (profile build)"
QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so timeout -k 10 300 python3 scripts/phase_timing_large.py 5 256 > $OUT/${TAG}_phase_timing_n5.txt 2>&1
{ QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so timeout -k 10 300 python3 scripts/phase_timing_large_bfgs.py 5 256; QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so timeout -k 10 300 python3 scripts/phase_timing_large_bfgs.py 4 1024; } > $OUT/${TAG}_phase_timing_large_bfgs.txt 2>&1
{ QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so timeout -k 10 300 python3 scripts/phase_timing_cptp.py 1024; QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so timeout -k 10 300 python3 scripts/phase_timing_cptp.py 256; } > $OUT/${TAG}_phase_timing_cptp.txt 2>&1
QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so timeout -k 10 300 python3 scripts/phase_timing.py > $OUT/${TAG}_phase_timing_B1000.txt 2>&1
QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so timeout -k 10 300 python3 scripts/gemm_diag.py > $OUT/${TAG}_gemm_phase_switches.txt 2>&1
step "parity sweeps"
timeout -k 10 900 python3 scripts/parity_sweep.py 24 > $OUT/${TAG}_parity_sweep.txt 2>&1; tail -2 $OUT/${TAG}_parity_sweep.txt
timeout -k 10 900 python3 scripts/parity_sweep.py 8 split > $OUT/${TAG}_parity_sweep_split.txt 2>&1; tail -2 $OUT/${TAG}_parity_sweep_split.txt
step done

#!/bin/bash
# Everything profiles/round2_* is copied from, in one GPU-box session (see profiles/README.md):
#   bash scripts/final_round2_profiles.sh <tag>
set -e
step() { echo "[$(date +%T)] $1"; }
TAG=${1:-r2final}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
step "python3 -m pytest tests -m gpu -x -q"
python3 -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.log 2>&1 || true
tail -3 $OUT/${TAG}_gpu_tests.log
step "bash scripts/collect_profiles_r2.sh $TAG"
bash scripts/collect_profiles_r2.sh $TAG > $OUT/${TAG}_collect.log 2>&1
step "python3 bench.py --gpus 1 --steps 20 --warmup 5"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_cmd.json 2> $OUT/${TAG}_bench_driver_cmd.err
step "python3 bench.py"
python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
step "python3 bench.py --gpus 2 --backend gloo --share-gpu0 --no-cpu-baselin"
python3 bench.py --gpus 2 --backend gloo --share-gpu0 --no-cpu-baseline > $OUT/${TAG}_bench_2rank_gloo_shared_gpu.json 2> $OUT/${TAG}_bench_2rank.err
step "QT_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline"
QT_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench_1rank_rccl.json 2> $OUT/${TAG}_bench_1rank.err
step "python3 scripts/headline_timing.py 500"
python3 scripts/headline_timing.py 500 > $OUT/${TAG}_headline_timing.txt 2>&1
step "python3 scripts/iterating_timing.py"
python3 scripts/iterating_timing.py > $OUT/${TAG}_iterating_timing.txt 2>&1
step "python3 scripts/process_timing.py"
python3 scripts/process_timing.py > $OUT/${TAG}_process_timing.txt 2>&1
step "{ python3 scripts/process_timing.py 64 3; python3 scripts/process_timi"
{ python3 scripts/process_timing.py 64 3; python3 scripts/process_timing.py 256 3; } > $OUT/${TAG}_process3_timing.txt 2>&1
step "python3 scripts/experiment_latency.py"
python3 scripts/experiment_latency.py > $OUT/${TAG}_experiment_latency.txt 2>&1
step "{ python3 scripts/large_n_timing.py 5 256; python3 scripts/large_n_tim"
{ python3 scripts/large_n_timing.py 5 256; python3 scripts/large_n_timing.py 4 1024 100000; python3 scripts/large_n_timing.py 4 1024; python3 scripts/large_n_timing.py 5 2048; } > $OUT/${TAG}_large_n_timing.txt 2>&1
step "QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so python3 scripts/phase_timin"
QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so python3 scripts/phase_timing_large.py 5 256 > $OUT/${TAG}_phase_timing_n5.txt 2>&1
step "QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so python3 scripts/phase_timin"
QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so python3 scripts/phase_timing.py > $OUT/${TAG}_phase_timing_B1000.txt 2>&1
step "python3 scripts/parity_sweep.py 24"
python3 scripts/parity_sweep.py 24 > $OUT/${TAG}_parity_sweep.txt 2>&1
step "python3 scripts/parity_sweep.py 8 split"
python3 scripts/parity_sweep.py 8 split > $OUT/${TAG}_parity_sweep_split.txt 2>&1
step "python3 scripts/state_api_latency.py"
python3 scripts/state_api_latency.py > $OUT/${TAG}_state_api_latency.txt 2>&1 || true
step "python3 scripts/process_api_latency.py"
python3 scripts/process_api_latency.py > $OUT/${TAG}_process_api_latency.txt 2>&1 || true
step "python3 scripts/bootstrap_timing.py"
python3 scripts/bootstrap_timing.py > $OUT/${TAG}_bootstrap_end_to_end.txt 2>&1 || true
echo done

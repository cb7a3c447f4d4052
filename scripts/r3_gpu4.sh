#!/bin/bash
step() { echo "[$(date +%T)] $1"; }
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
step "process tests"
timeout -k 10 900 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_widening.py tests/test_gpu_leftovers.py tests/test_gpu_large.py tests/test_gpu_fullsize.py tests/test_gpu_moments.py -m gpu -q -x > $OUT/r3d_tests.log 2>&1; echo "rc=$?" >> $OUT/r3d_tests.log; tail -5 $OUT/r3d_tests.log
step "process timing"
timeout -k 10 300 python3 scripts/process_timing.py > $OUT/r3d_process_timing.txt 2>&1; cat $OUT/r3d_process_timing.txt
step "kernel stats of the process path"
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/r3d_proc_stats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/process_timing.py > $OUT/r3d_proc_stats.log 2>&1; cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3d_proc_stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}")
PY
step done

#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's dominant kernel on the GPU box.
#   bash scripts/collect_profiles.sh <tag>       -> gpurun_out/<tag>_{stats,fetch,write}/ + summaries
# Kernel-trace/stats and each PMC counter are separate passes (the pool forbids mixing them with
# sys/hip traces); the program itself follows "--" (no env/bash hop after the profiler preload).
set -e
TAG=${1:-prof}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
ARGS="bench.py --steps 20 --warmup 3 --no-cpu-baseline --bootstrap-points 0 --saturation-batch 0 --no-other-configs"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o run --output-format csv -- python3 bench.py > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o run --output-format csv -- python3 $ARGS > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o run --output-format csv -- python3 $ARGS > $OUT/${TAG}_write.log 2>&1
python3 scripts/summarise_profiles.py $TAG > $OUT/${TAG}_summary.json
cat $OUT/${TAG}_summary.json

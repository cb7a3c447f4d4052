#!/bin/bash
# Round-2 rocprofv3 evidence, collected on the GPU box into gpurun_out/<tag>_*; the summaries are then copied to
# profiles/round2_*.  Kernel-trace/stats and every PMC counter are SEPARATE passes (the pool forbids mixing PMC with
# sys/hip traces); the program itself follows "--" (no env / bash hop behind the profiler's preload).
#   bash scripts/collect_profiles_r2.sh <tag>
set -e
TAG=${1:-r2}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
SHORT="bench.py --steps 20 --warmup 3 --no-cpu-baseline --bootstrap-points 0 --saturation-batch 0 --no-other-configs"
# 1. per-kernel durations of the DEFAULT bench command (the line the driver records)
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o run --output-format csv -- python3 bench.py > $OUT/${TAG}_stats.log 2>&1
echo "[$(date +%T)] stats done" >> $OUT/${TAG}_progress.txt
# 2. HBM traffic of the dominant kernel (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o run --output-format csv -- python3 $SHORT > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o run --output-format csv -- python3 $SHORT > $OUT/${TAG}_write.log 2>&1
# 3. executed instructions / busy cycles of the n = 3 kernels at a saturating batch, one counter per pass
for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY; do
  echo "[$(date +%T)] pmc $C" >> $OUT/${TAG}_progress.txt
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_pmc_$C -o run --output-format csv -- python3 scripts/kernel_breakdown.py 65536 > $OUT/${TAG}_pmc_$C.log 2>&1 || echo "counter $C not collected" >> $OUT/${TAG}_pmc_missing.txt
done
python3 scripts/summarise_profiles.py $TAG > $OUT/${TAG}_summary.json
python3 - "$TAG" <<'PY' > $OUT/${TAG}_pmc_instruction_counts.json
import csv, glob, json, statistics, sys
tag = sys.argv[1]
res = {}
for f in glob.glob(f"gpurun_out/{tag}_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "qt::" in k:
            res.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(json.dumps({k: {c: statistics.median(v) for c, v in d.items()} for k, d in res.items()}, indent=1))
PY
cat $OUT/${TAG}_summary.json

#!/bin/bash
# VALU / LDS instruction counts of the n = 3 kernels at a saturating batch (PMC passes only; no traces mixed in).
set -e
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
for C in SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o run --output-format csv -- python3 scripts/kernel_breakdown.py 65536 > $OUT/pmc_$C.log 2>&1
done
python3 - <<'PY'
import csv, glob, statistics, json
res = {}
for f in glob.glob("gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "qt::" not in k:
            continue
        res.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {k: {c: statistics.median(v) for c, v in d.items()} for k, d in res.items()}
print(json.dumps(out, indent=1))
PY

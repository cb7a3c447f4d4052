#!/bin/bash
# one counter per pass vs five counters in one pass: SQ_INSTS_VALU of k_mle_start<3> at B = 65536
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
SHORT="$GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --bootstrap-points 0 --no-other-configs --saturation-batch 65536"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d $OUT/pmcchk_single -o run --output-format csv -- python3 $SHORT > $OUT/pmcchk_single.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $OUT/pmcchk_pair -o run --output-format csv -- python3 $SHORT > $OUT/pmcchk_pair.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES -d $OUT/pmcchk_five -o run --output-format csv -- python3 $SHORT > $OUT/pmcchk_five.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d $OUT/pmcchk_kb -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/kernel_breakdown.py 65536 > $OUT/pmcchk_kb.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, statistics
for tag in ("single", "pair", "five", "kb"):
    per = {}
    for f in glob.glob(f"gpurun_out/pmcchk_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mle_start<3" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(tag, {k: (statistics.median(v), len(v), min(v), max(v)) for k, v in per.items()})
PY

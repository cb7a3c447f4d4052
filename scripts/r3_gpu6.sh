#!/bin/bash
step() { echo "[$(date +%T)] $1"; }
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
step "2-rank gloo bench sharing GPU 0"
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --share-gpu0 --no-cpu-baseline --steps 50 --warmup 5 > $OUT/r3h_bench_2rank_gloo.json 2> $OUT/r3h_bench_2rank_gloo.err; echo "rc=$?"; tail -3 $OUT/r3h_bench_2rank_gloo.err
step "1-rank RCCL bench"
QT_BENCH_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 > $OUT/r3h_bench_1rank_rccl.json 2> $OUT/r3h_bench_1rank_rccl.err; echo "rc=$?"; tail -3 $OUT/r3h_bench_1rank_rccl.err
step "1-rank plain (same flags) for comparison"
timeout -k 10 600 python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 > $OUT/r3h_bench_1rank_plain.json 2> $OUT/r3h_bench_1rank_plain.err; echo "rc=$?"
python3 - <<'PY'
import json
for f in ("r3h_bench_2rank_gloo", "r3h_bench_1rank_rccl", "r3h_bench_1rank_plain"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["n_gpus"], d["rccl_ranks"], round(d["value"]/1e6, 2), "M/s")
    for k in ("bootstrap_ci", "bootstrap_ci_device_sampler", "bootstrap_ci_large", "bootstrap_ci_large_device_draw", "bootstrap_ci_n5"):
        b = d.get(k)
        if b: print("   ", k, b["wall_ms"], b["quantiles_hs"], b.get("quantile_path"), "serial", b.get("serial_ms"))
PY
step done

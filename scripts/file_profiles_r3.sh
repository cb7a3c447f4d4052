#!/bin/bash
# Copy what scripts/collect_profiles_r3.sh <tag> left under gpurun_out/ into profiles/round3_* (run in the dev container).
TAG=${1:-r3final}
cd "$(dirname "$0")/../gpurun_out" || exit 1
extract() { python3 - "$1" "$2" <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
open(sys.argv[2], "w").write(json.dumps(json.loads(lines[-1]), indent=1) + "\n")
PY
}
extract ${TAG}_bench_driver_cmd_with_traffic.json ../profiles/round3_final_bench_driver_cmd.json
extract ${TAG}_bench_default.json ../profiles/round3_final_bench_default.json
extract ${TAG}_bench_under_rocprof.json ../profiles/round3_final_bench_under_rocprof.json
extract ${TAG}_bench_2rank_gloo_shared_gpu.json ../profiles/round3_bench_2rank_gloo_shared_gpu.json
extract ${TAG}_bench_1rank_rccl.json ../profiles/round3_bench_1rank_rccl.json
cp ${TAG}_bench_kernel_stats.csv ../profiles/round3_final_bench_kernel_stats.csv
cp ${TAG}_short_cmd_kernel_stats.csv ../profiles/round3_bench_kernel_stats.csv
cp ${TAG}_pmc_traffic.json ../profiles/round3_pmc_traffic.json
for t in gpu_tests.log headline_timing.txt iterating_timing.txt process_timing.txt process3_timing.txt large_n_timing.txt moment_coverage_timing.txt bootstrap_end_to_end.txt phase_timing_B1000.txt parity_sweep.txt parity_sweep_split.txt; do grep -v "amdgpu.ids" ${TAG}_$t > ../profiles/round3_final_$t; done
for t in cptp_sweep.txt pgdb3_timing.txt lifp16_stream_timing.txt cp_accuracy_n3.txt clip_accuracy_n5.txt phase_timing_large_bfgs.txt coverage_tables.txt idle_ramp_device_draw.txt ubench_valu_f64_warm.txt ubench_mfma_f64_warm.txt phase_timing_n5.txt phase_timing_cptp.txt gemm_phase_switches.txt; do grep -v "amdgpu.ids" ${TAG}_$t > ../profiles/round3_$t; done

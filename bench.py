#!/usr/bin/env python3
"""Headline benchmark: state reconstructions/s, 3-qubit MLE (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: qt_mle_batch on B = 1000 independent
3-qubit trials per GPU ('proj-set' POVM, 1e5 shots per setting, Ginibre state of
np.random.default_rng(1234); counts drawn on the host from np.random.seed(7) in the
reference's call order -- SURVEY.md section 8d).  Counts are resident in HBM before the timed
region.

N > 1: one rank per GPU over RCCL.  Either the driver starts the ranks (torchrun: WORLD_SIZE is
set) or `python bench.py --gpus N` starts them itself as a CHILD `python -m torch.distributed.run`
before anything in this process has touched the GPU.  Each rank reconstructs its own 1000 trials
(weak scaling, no data-path collective); the group size is asserted equal to --gpus and reported as
"rccl_ranks".  After the headline the sharded legs run on every rank:
  "configs4_weak"     5-qubit MLE (configs[4]), 256 trials per rank, weak-scaled;
and five STRONG-scaled bootstrap CIs (interval.py:598-612), each rank doing its shard of: [draw +] reconstruct with the
Hilbert-Schmidt distance in the same pass (qt_mle_dist_batch) + sort of its shard + the order statistics across ranks
(quantpy_amd.distributed.ShardedSample: two all-gathers of a few hundred KB, never the sample):
  "bootstrap_ci"      the 2000-resample CI of configs[3] on the reference's stream (end_to_end_ms includes the serial host draw);
  "bootstrap_ci_device_sampler"  the same with the resamples drawn in HBM (qt_device_multinomial, opt-in), draw inside the region;
  "bootstrap_ci_n5"   2000 resamples at 5 qubits;
  "bootstrap_ci_large" 2 097 152 resamples at n = 3 (a size where sharding matters), with "serial_ms" = what does not shard;
  "bootstrap_ci_large_device_draw"  the same size, every resample distinct and drawn by its rank, the draw inside the region.

The contract's W warm-up + K timed steps run twice: straight after the host drew the counts ("no_preroll") and again after
--preroll-ms (40 ms, untimed, reported as "preroll_ms") of the same step, which brings the GPU back to its running clocks;
`value` is the second.  `python bench.py --profile` regenerates profiles/round3_pmc_traffic.json (rocprofv3 PMC passes as
child processes); `roofline.traffic` is reported only when that file was measured on this build of the library.

Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "roofline":     the dominant kernel against the HBM roofline, from HIP-event timing of
                  back-to-back launches on the stream the kernel runs on;
  "cpu_baseline": the CPU oracle (oracle/quantpy_oracle.py: scipy BFGS + forward differences,
                  i.e. the reference's algorithm) timed on one host core on a bounded sample;
  "iterating":    the regime in which BFGS really iterates (configs[1] counts from the fully mixed
                  start; a rank-1 state from the 'lin' start), with parity against the oracle.
"""
import argparse
import json
import os
import sys
import time


def _cpu_share():
    """CPUs this process may actually use: the cgroup quota (cpu.max) when there is one, else the affinity mask."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


# BLAS / OpenMP pools sized by the VISIBLE CPUs (256 on the GPU box) against a cgroup quota of 16 get the whole process
# throttled while their idle workers spin: the single-threaded host sampler then reads 80-100 ms instead of 16
# (seen in 1 run of 3; scripts/host_sampler_timing.py: 15.7 ms when nothing else spins).  Size them by the share.
for _var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_var, str(min(_cpu_share(), 8)))

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # vector FP64 and FP64 matrix, SURVEY.md section 8d
# what bare loops reach on this chip, measured warm in round 3 (scripts/ubench/*.hip -DQT_UBENCH_WARM; profiles/round3_ubench_*)
FP64_MFMA_MEASURED_TFLOPS = 46.4
VALU_WARM_NS = 2.42       # ns per FP64 wave-instruction per SIMD at four or more waves per SIMD (54.2 TFLOP/s)
VALU_DATASHEET_NS = 4 / 2.4  # 4 cycles at 2.4 GHz


def ginibre(rng, d):
    g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    rho = g @ g.conj().T
    return rho / np.trace(rho)


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import quantpy_oracle as qo

    return qo


def _cpu_worker(job):
    """One host core of the all-cores CPU baseline: the oracle (the reference's algorithm) on its slice."""
    counts, n_trials = job
    qo = _oracle()
    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=1)
    except Exception:
        pass
    povm = qo.measurement_matrix("proj-set", 3)
    t0 = time.perf_counter()
    for i in range(n_trials):
        qo.mle_estimate(counts[i % len(counts)], povm)
    return time.perf_counter() - t0


def cpu_all_cores(shots, per_core):
    """The CPU baseline on every host core (SURVEY 8d ii).  Runs BEFORE anything touches the GPU: the workers
    are fresh interpreters (spawn), and a process that has initialised HIP must not fork + exec."""
    import concurrent.futures as cf
    import multiprocessing as mp

    qo = _oracle()
    # the CPU share that goes with one GPU of the node is 16 cores, whatever the affinity mask shows
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    povm = qo.measurement_matrix("proj-set", 3)
    bloch = qo.bloch_from_matrix(ginibre(np.random.default_rng(1234), 8))
    np.random.seed(7)
    counts = np.stack([qo.sample_counts(povm, bloch, np.ones(povm.shape[0]) * shots) for _ in range(64)])
    tw = time.perf_counter()
    with cf.ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as pool:
        busy = list(pool.map(_cpu_worker, [(counts, per_core)] * cores))
    wall = time.perf_counter() - tw
    total = cores * per_core
    return {"value": round(total / max(busy), 1), "unit": "reconstructions/s", "cores": cores,
            "sample": f"{per_core} reconstructions on each of {cores} worker processes (one per core of the box's CPU share, capped at 16), slowest worker {max(busy):.1f} s, {wall:.1f} s with process start-up"}


def launch_ranks(args):
    """Run `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a child process, pass its
    output through (rank 0 prints the one JSON line) and return its exit code."""
    import socket
    import subprocess

    port = args.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def vector_resamples(povm, bloch, shots, n, seed):
    """`n` multinomial count tensors (n, S, K) for one state, drawn setting by setting in vectorised calls of a
    seeded Generator.  Input generation for the LARGE bootstrap legs only: valid resamples, but not the reference's
    stream order (that order -- one legacy-RNG call per setting per resample -- is kept wherever counts are
    compared with the reference: the headline batch, configs[3]'s 2000 resamples, the tests)."""
    d = int(round(np.sqrt(povm.shape[-1])))
    p = np.clip(np.einsum("ijk,k->ij", np.asarray(povm), bloch) * d, 0, 1)
    p = p / p.sum(-1, keepdims=True)
    rng = np.random.default_rng(seed)
    out = np.empty((n,) + p.shape, dtype=np.int64)
    for s in range(p.shape[0]):
        out[:, s, :] = rng.multinomial(int(shots[s]), p[s], size=n)
    return out


PMC_FILE = "round3_pmc_traffic.json"
SHORT_FLAGS = ["--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--bootstrap-points", "0", "--no-other-configs"]


def library_hash():
    """The source hash libqtomo.so was built from (quantpy_amd/build.py): ties a committed profile to a library."""
    try:
        with open(os.path.join(ROOT, "quantpy_amd", "lib", "libqtomo.so.srchash")) as fh:
            return fh.read().strip()
    except OSError:
        return None


def run_profile():
    """`python bench.py --profile`: regenerate profiles/round3_pmc_traffic.json (HBM bytes per launch of the dominant
    kernel) and profiles/round3_bench_kernel_stats.csv from three rocprofv3 passes of the short bench command -- kernel
    trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE, each in its own pass with nothing but --kernel-trace beside it
    (MI355X_MICROARCH.md, HBM section) -- started as CHILD processes before this one touches the GPU.  FETCH_SIZE is
    calibrated in the same pass on a known byte count in the same access pattern (the guide: widths other than 16 B per
    lane are uncalibrated): the cold 65 536-trial launch of k_mle_start reads 65 536 x 8 M bytes of counts exactly once."""
    import csv
    import glob
    import shutil
    import statistics
    import subprocess

    out_root = os.path.join(ROOT, "gpurun_out", "profile_r3")
    shutil.rmtree(out_root, ignore_errors=True)
    os.makedirs(out_root, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    me = os.path.abspath(__file__)
    passes = {"stats": ["--kernel-trace", "--stats"], "fetch": ["--kernel-trace", "--pmc", "FETCH_SIZE"],
              "write": ["--kernel-trace", "--pmc", "WRITE_SIZE"],
              # executed wave-instructions of the n = 3 kernels (SQ block: eight counter slots per pass, five used)
              "sq": ["--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VALU_MFMA_MOPS_F64",
                     "SQ_WAVES"]}
    for name, flags in passes.items():
        cmd = ["rocprofv3", *flags, "-d", os.path.join(out_root, name), "-o", "run", "--output-format", "csv", "--",
               sys.executable, me, *SHORT_FLAGS, "--saturation-batch", "65536"]
        with open(os.path.join(out_root, name + ".log"), "w") as log:
            rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT).returncode
        print(f"[profile] {name}: rc {rc}", file=sys.stderr, flush=True)
        if rc != 0:
            return rc

    def counter(sub, name):
        per = {}
        for f in glob.glob(os.path.join(out_root, sub, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    if r["Counter_Name"] == name and "qt::" in r["Kernel_Name"]:
                        per.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(float(r["Counter_Value"]))
        return per

    fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
    sq = {name: counter("sq", name) for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VALU_MFMA_MOPS_F64",
                                                  "SQ_WAVES")}
    fused = "qt::k_mle_fused<3, false>"
    start = "qt::k_mle_start<3, false>"
    m_rows, d_el = 216, 64
    known_counts = 65536 * 8 * m_rows
    # the 65 536-trial launches: the first is cold (the buffer was just uploaded), every one streams 113 MB > L2
    cal = None
    if start in fetch and fetch[start]:
        cal = known_counts / (statistics.median(fetch[start]) * 1024.0)
    stats_rows = []
    for f in glob.glob(os.path.join(out_root, "stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(ROOT, "profiles", "round3_bench_kernel_stats.csv"))
        with open(f, newline="") as fh:
            stats_rows = [r for r in csv.DictReader(fh) if "qt::" in r["Name"]]
    avg_ns = {r["Name"].split("(")[0].replace("void ", ""): (float(r["AverageNs"]), int(r["Calls"])) for r in stats_rows}
    f_kib, w_kib = statistics.median(fetch[fused]), statistics.median(write[fused])
    f_bytes = f_kib * 1024.0 * (cal if cal else 1.0)
    executed = None
    if start in sq["SQ_WAVES"] and sq["SQ_WAVES"][start]:
        # the 65 536-trial launches of k_mle_start<3>: one trial per wavefront, the whole nit = 0 path of a reconstruction
        waves = statistics.median(sq["SQ_WAVES"][start])
        per = {k: statistics.median(v[start]) / waves for k, v in sq.items() if start in v and k != "SQ_WAVES"}
        executed = {
            "source": "python bench.py --profile: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS "
                      "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES (one pass) over the saturated launches of " + start +
                      " = the whole nit = 0 path of one reconstruction per wavefront",
            "waves_per_launch": waves,
            "valu_per_reconstruction": round(per.get("SQ_INSTS_VALU", 0.0), 1),
            "salu_per_reconstruction": round(per.get("SQ_INSTS_SALU", 0.0), 1),
            "lds_per_reconstruction": round(per.get("SQ_INSTS_LDS", 0.0), 1),
            "mfma_f64_ops": round(per.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0), 1),
            "note": "the n <= 3 estimators execute no MFMA instruction: the contractions are factorised (8x fewer flops than "
                    "the dense GEMM form) and account for ~13 % of the VALU instructions (DESIGN.md 4.3)"}
    res = {
        "executed": executed,
        "source": "python bench.py --profile: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) and "
                  "--kernel-trace --stats over `bench.py " + " ".join(SHORT_FLAGS) + " --saturation-batch 65536`",
        "library_srchash": library_hash(),
        "kernel": fused, "launches": len(fetch[fused]),
        "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
        "fetch_calibration": {
            "factor": cal, "kernel": start, "known_bytes": known_counts,
            "FETCH_SIZE_KiB_median": statistics.median(fetch[start]) if start in fetch else None,
            "note": "known bytes / reported bytes for this kernel family's 8 B-per-lane count reads (the guide's factor 2 is "
                    "for 16 B per lane); WRITE_SIZE is exact for these stores (checked in round 2 on k_povm_setup / k_povm_kron)"},
        "traffic_bytes_per_launch": int(round(f_bytes + w_kib * 1024.0)),
        "algorithmic_bytes_per_launch": int(1000 * (8 * m_rows + 16 * d_el) + 2 * 8 * m_rows * d_el),
        "kernel_avg_us_rocprof": (avg_ns[fused][0] / 1e3 if fused in avg_ns else None),
        "kernel_calls_rocprof": (avg_ns[fused][1] if fused in avg_ns else None),
    }
    ratio = res["traffic_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
    res["traffic_over_algorithmic"] = round(ratio, 3)
    res["note"] = ("algorithmic bytes count the operand tables once per launch; each of the 250 workgroups stages its own copy "
                   "of the product-POVM tables and row weights (~2 KB) and the timed step replays ONE counts buffer (1.73 MB, "
                   "partly L2-resident from the previous step): a ratio near 1 means no wasted re-reads of the counts")
    with open(os.path.join(ROOT, "profiles", PMC_FILE), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1000, help="trials per GPU per step (configs[1]: 1000)")
    ap.add_argument("--shots", type=int, default=100000)
    ap.add_argument("--preroll-ms", type=float, default=40.0,
                    help="run the step for this long BEFORE the W warm-up steps: the counts were just drawn on the host "
                         "(GPU idle, clocks down); 0 = off.  Untimed, reported as \"preroll_ms\"")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2000, help="trials timed on the CPU oracle")
    ap.add_argument("--cpu-per-core", type=int, default=1500,
                    help="reconstructions per worker in the all-cores CPU baseline; 0 = skip it")
    ap.add_argument("--bootstrap-points", type=int, default=2000)
    ap.add_argument("--bootstrap-large", type=int, default=2097152,
                    help="resamples of the large n = 3 bootstrap leg (a size at which 8 GPUs still have ~2 ms of work each); 0 = off")
    ap.add_argument("--saturation-batch", type=int, default=65536, help="extra (untimed-contract) measurement; 0 = off")
    ap.add_argument("--pipelined-steps", type=int, default=0,
                    help="extra: this many steps alternated over two streams (off by default so that a rocprofv3 pass over the "
                         "default command sees single-stream launches only)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[2] / configs[4] / iterating side measurements")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N > 1 (nccl = RCCL; gloo for rehearsal)")
    ap.add_argument("--share-gpu0", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself")
    ap.add_argument("--launch-check", action="store_true", help="form the process group (gloo, CPU) and exit: launcher test")
    ap.add_argument("--profile", action="store_true",
                    help="regenerate profiles/" + PMC_FILE + " and the kernel stats from rocprofv3 passes of the short command (child processes)")
    args = ap.parse_args()
    if args.profile:
        sys.exit(run_profile())

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves.  This parent has imported neither torch nor
        # libqtomo (nothing here has touched the GPU); the ranks are CHILD processes, never an exec.
        sys.exit(launch_ranks(args))
    force_dist = os.environ.get("QT_BENCH_FORCE_DIST") == "1"  # 1-rank group: rehearsal of the RCCL path
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a mislabelled number",
              file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    cpu_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_per_core > 0:
        try:
            cpu_all = cpu_all_cores(args.shots, args.cpu_per_core)
        except Exception as exc:  # the single-core figure below is the contract; this one is a supplement
            cpu_all = {"error": f"{type(exc).__name__}: {exc}"}

    import torch
    import torch.distributed as dist

    if args.launch_check:
        # CPU-only check of the launcher path (tests/test_host_logic.py): the ranks this process started (or the
        # driver's torchrun) form a group of --gpus members; nothing here touches a GPU.
        dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        probe = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(probe)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": dist.get_world_size(), "rank_sum": int(probe.item())}), flush=True)
        dist.destroy_process_group()
        return

    dev_index = 0 if args.share_gpu0 else local_rank
    if not args.share_gpu0 and world > torch.cuda.device_count():
        print(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s) (use --share-gpu0 for a rehearsal)",
              file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(dev_index)
    use_dist = world > 1 or force_dist
    rccl_ranks = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        # prove that the collective library sees every rank: sum of (rank + 1) over the group
        probe = torch.tensor([float(rank + 1)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(probe)
        assert int(probe.item()) == world * (world + 1) // 2, probe
        rccl_ranks = dist.get_world_size() if args.backend == "nccl" else 0
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    import quantpy_amd as qp
    from quantpy_amd import distributed as qd
    from quantpy_amd.tomography.state import simulate_counts

    def barrier():
        if use_dist:
            dist.barrier()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    n, d, B = 3, 8, args.batch
    rho_true = ginibre(np.random.default_rng(1234), d)
    state = qp.Qobj(rho_true)
    povm = qp.generate_measurement_matrix("proj-set", n)  # (27, 8, 64), assembled on the GPU
    S, K, D = povm.shape
    M = S * K
    shots = np.ones(S) * args.shots

    # ---- synthetic counts: one global legacy stream, rank r owns trials [r*B, (r+1)*B) --------
    np.random.seed(7)
    bloch = state.bloch
    all_counts = simulate_counts(povm, bloch, shots, repeats=B * world)  # the reference's stream order, one C call
    counts = all_counts[rank * B:(rank + 1) * B]

    eng = qp.get_engine(n)  # this process's GPU (torch.cuda.current_device())
    assert eng.device == dev_index, (eng.device, dev_index)
    eng.set_povm(povm, shots)
    counts_d = torch.from_numpy(np.ascontiguousarray(counts)).cuda()
    rho_d = torch.empty((B, d, d), dtype=torch.complex128, device="cuda")
    nit_d = torch.zeros(B, dtype=torch.int32, device="cuda")
    nfev_d = torch.zeros(B, dtype=torch.int32, device="cuda")
    st_d = torch.zeros(B, dtype=torch.int32, device="cuda")

    def step():
        eng.mle_dev(counts_d, rho_d, init="lin", max_iter=100, tol=1e-3, nit=nit_d, nfev=nfev_d, status=st_d)

    def timed_steps():
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronise pairs; max over ranks."""
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()  # (the engine runs on torch's current stream: this covers its work)
        barrier()
        if use_dist:
            torch.cuda.synchronize()  # an RCCL barrier is itself GPU work
        t0 = time.perf_counter()
        eng.timer_begin()
        for _ in range(args.steps):
            step()
        eng.timer_stop()  # HIP events on the stream the kernel runs on; the interval is read after the region
        torch.cuda.synchronize()
        barrier()
        if use_dist:
            torch.cuda.synchronize()
        return max_over_ranks(time.perf_counter() - t0), eng.timer_elapsed() / args.steps

    # First the contract's W + K exactly as written: the GPU sat idle while the host drew the counts, and W x 15 us of
    # warm-up end before its power state has moved -- reported as "no_preroll" (both figures).
    cold_elapsed, cold_kernel_ms = timed_steps() if args.preroll_ms > 0 else (None, None)
    # pre-roll: bring the GPU back to its running clocks with the same step.  Outside the timed region.
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
        for _ in range(64):
            step()
        torch.cuda.synchronize()
    preroll_ms = (time.perf_counter() - t_pre) * 1e3 if args.preroll_ms > 0 else 0.0
    elapsed, kernel_ms = timed_steps()
    assert torch.cuda.current_device() == dev_index  # engine calls leave the thread's device alone

    nit = nit_d.cpu().numpy()
    nfev = nfev_d.cpu().numpy()
    status = st_d.cpu().numpy()
    rho_h = rho_d.cpu().numpy()
    assert np.all(status == 0), "non-zero trial status in the benchmark batch"
    value = world * B * args.steps / elapsed

    # ---- roofline of the dominant kernel (algorithmic bytes per reconstruction) -----
    bytes_per_recon = 8 * M + 16 * d * d + 2 * (8 * M * D) / B  # counts in, rho out, A' and A'^T amortised
    launch_bytes = bytes_per_recon * B
    achieved_gbs = launch_bytes / (kernel_ms * 1e-3) / 1e9
    # FP64 work of the DENSE analytic-gradient algorithm for the same reconstructions: per evaluation
    # 2*(2 M D) for the two POVM contractions + small d^3 terms; per trial one linear inversion 2 M D and a
    # Jacobi eigensolve.  (A dense-equivalent rate: the factorised POVM path executes ~8x fewer flops for the
    # contractions, and a trial whose start matrix is positive definite skips the eigensolve.)
    flops_eval = 4 * M * D + 16 * D * d + 2 * 8 * d**3
    flops_trial = 2 * M * D + 8 * D * d + 60 * 8 * d**3 + float(nfev.mean()) * flops_eval
    fp64_tflops = flops_trial * B / (kernel_ms * 1e-3) / 1e12
    # HBM bytes per launch from the PMC passes of `bench.py --profile` (committed under profiles/); trusted only while the
    # library is the one that was profiled (source hash), otherwise reported as null with the reason
    traffic, traffic_src, executed = None, None, None
    pmc_file = os.path.join(ROOT, "profiles", PMC_FILE)
    if B == 1000 and os.path.exists(pmc_file):
        with open(pmc_file) as fh:
            pmc = json.load(fh)
        if pmc.get("library_srchash") and pmc.get("library_srchash") == library_hash():
            traffic = pmc["traffic_bytes_per_launch"]
            traffic_src = (f"profiles/{PMC_FILE} (python bench.py --profile: rocprofv3 --pmc FETCH_SIZE x calibration "
                           f"{pmc['fetch_calibration']['factor']:.3f} + WRITE_SIZE; same library source hash)")
            executed = pmc.get("executed")  # instruction counts of the same build (SQ pass of --profile)
        else:
            traffic_src = f"profiles/{PMC_FILE} was measured on another build of the library (source hash differs): not reported"
    if executed is None:
        exec_file = os.path.join(ROOT, "profiles", "round2_pmc_traffic.json")
        if os.path.exists(exec_file):
            with open(exec_file) as fh:
                executed = json.load(fh).get("executed")
            if isinstance(executed, dict):
                executed = dict(executed, measured_on="round-2 build of k_mle_start<3> (instruction counts per reconstruction): "
                                                      "no SQ pass of this build under profiles/")
    # qt_mle_batch runs the single-launch kernel while the batch fits one wave per SIMD (<= 1024 waves)
    dominant_kernel = ("qt::k_mle_fused<3,false>" if B <= 1024 else
                       "qt::k_mle_start<3,false> (+ qt::k_mle_bfgs<3,false> for the trials that iterate)")
    roofline = {
        "bound": "hbm", "kernel": dominant_kernel, "achieved": round(achieved_gbs, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 6), "traffic": traffic,
        "traffic_source": traffic_src,
        "bytes_per_launch": int(launch_bytes), "kernel_ms": round(kernel_ms, 5),
        "note": "n=3 working set is LDS/register resident: the step is bound by single-wave instruction issue "
                "(one wave per SIMD at B=1000), not by HBM (SURVEY 8d, DESIGN.md 4.1); dense-equivalent FP64 rate beside it",
        "executed_instruction_utilisation": executed,
        "fp64_dense_equivalent": {"achieved": round(fp64_tflops, 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(fp64_tflops / FP64_PEAK_TFLOPS, 6),
                                  "note": "flops of the dense (unfactorised) algorithm for the same work / FP64 vector peak "
                                          "-- NOT utilisation: the factorised kernel executes ~8x fewer"},
    }

    def timed(fn, reps, engine=None):
        e = engine or eng
        fn()
        e.sync()
        e.timer_begin()
        for _ in range(reps):
            fn()
        return e.timer_end() / reps

    # ---- the same step at a batch that fills the chip (occupancy hides the single-wave latency) ----
    sat = None
    if rank == 0 and args.saturation_batch > 0:
        Bs = args.saturation_batch
        reps = (Bs + B - 1) // B
        big = torch.from_numpy(np.ascontiguousarray(np.concatenate([counts] * reps)[:Bs])).cuda()
        rho_s = torch.empty((Bs, d, d), dtype=torch.complex128, device="cuda")
        ms = timed(lambda: eng.mle_dev(big, rho_s), 10)
        sat = {"batch": Bs, "ms_per_step": round(ms, 4), "value": round(Bs / ms * 1e3, 1), "unit": "reconstructions/s",
               "hbm_GBps": round(bytes_per_recon * Bs / (ms * 1e-3) / 1e9, 2),
               "fp64_TFLOPs_dense_equivalent": round(flops_trial * Bs / (ms * 1e-3) / 1e12, 3)}
        del big, rho_s
        if isinstance(executed, dict) and "valu_per_reconstruction" in executed and Bs == 65536:
            # executed VALU wave-instructions (PMC, committed under profiles/) against the chip's measured issue rate
            # against BOTH ceilings: the data sheet's FP64 issue rate (one wave-instruction per 4 cycles
            # at 2.4 GHz = 78.6 TFLOP/s) and the rate a bare v_fma_f64 loop reaches on this chip WARM (>= 40 ms pre-roll,
            # >= 20 ms kernels: 54.2 TFLOP/s = 2.42 ns; profiles/round3_ubench_valu_f64_warm.txt -- 2.57 ns cold in round 1)
            per_simd = executed["valu_per_reconstruction"] * Bs / 1024
            executed = dict(executed,
                            valu_issue_ceiling_ns_per_wave_instruction_per_simd=VALU_WARM_NS,
                            valu_issue_utilisation_at_B65536=round(per_simd * VALU_WARM_NS * 1e-6 / ms, 4),
                            valu_issue_utilisation_vs_datasheet_at_B65536=round(per_simd * VALU_DATASHEET_NS * 1e-6 / ms, 4),
                            mfma_utilisation=0.0)
            roofline["executed_instruction_utilisation"] = executed

    # ---- the same steps issued alternately on two handles (two HIP streams) ----------------------
    piped = None
    if rank == 0 and args.pipelined_steps > 0:
        from quantpy_amd.engine import Engine

        engs = (Engine(n, dev_index, stream="own"), Engine(n, dev_index, stream="own"))
        for e in engs:
            e.set_povm(povm, shots)
        outs = (rho_d, torch.empty_like(rho_d))
        torch.cuda.synchronize()
        for k in range(20):
            engs[k & 1].mle_dev(counts_d, outs[k & 1])
        for e in engs:
            e.sync()
        tp = time.perf_counter()
        for k in range(args.pipelined_steps):
            engs[k & 1].mle_dev(counts_d, outs[k & 1])
        for e in engs:
            e.sync()
        dt = time.perf_counter() - tp
        same = bool(torch.equal(outs[0], outs[1]))
        piped = {"steps": args.pipelined_steps, "streams": 2, "ms_per_step": round(dt / args.pipelined_steps * 1e3, 5),
                 "value": round(B * args.pipelined_steps / dt, 1), "unit": "reconstructions/s",
                 "outputs_identical_across_streams": same}
        for e in engs:
            e.close()

    # ---- the regime in which BFGS iterates (state.py:204-215 is THE hot loop of SURVEY 3.3) -------
    iterating = None
    if rank == 0 and not args.no_other_configs:
        qo = _oracle()
        iterating = {}
        povm_np = np.asarray(povm)
        prng = np.random.default_rng(77)
        psi = prng.standard_normal(d) + 1j * prng.standard_normal(d)
        psi /= np.linalg.norm(psi)
        pure = qp.Qobj(np.outer(psi, psi.conj()))
        np.random.seed(9)
        pure_counts = simulate_counts(povm, pure.bloch, shots, repeats=B)
        # (The 1000-trial figure of the 'lin'-start case is not taken in the default run: its launches would carry the
        #  name of the timed kernel, k_mle_fused<3,false>, and skew a profiler's per-kernel average of the headline step;
        #  the fully mixed start runs as k_mle_fused_mixed.  scripts/iterating_timing.py measures both.)
        for name, cts, init, small in (("configs[1] counts, init='mixed'", counts, "mixed", True),
                                       ("rank-1 state, 1e5 shots, init='lin'", pure_counts, "lin", False)):
            c_d = torch.from_numpy(np.ascontiguousarray(cts)).cuda()
            nb = len(cts) if small else 64
            r_d = torch.empty((nb, d, d), dtype=torch.complex128, device="cuda")
            ni_d = torch.zeros(nb, dtype=torch.int32, device="cuda")
            nf_d = torch.zeros(nb, dtype=torch.int32, device="cuda")
            s_d = torch.zeros(nb, dtype=torch.int32, device="cuda")
            run_small = lambda: eng.mle_dev(c_d[:nb], r_d, init=init, nit=ni_d, nfev=nf_d, status=s_d)  # noqa: E731
            if small:
                ms = timed(run_small, 20)
            else:
                run_small()
                eng.sync()
            r_h, ni_h = r_d.cpu().numpy(), ni_d.cpu().numpy()
            same_nit, worst = 0, 0.0
            for i in range(64):
                ref, ri = qo.mle_estimate(cts[i], povm_np, init=init, return_info=True, solver="port")
                same_nit += int(ri["nit"] == ni_h[i])
                worst = max(worst, abs(qo.infidelity(ref, r_h[i])))
            Bs = 65536
            big = c_d.repeat((Bs + len(cts) - 1) // len(cts), 1, 1)[:Bs].contiguous()
            rb = torch.empty((Bs, d, d), dtype=torch.complex128, device="cuda")
            nib = torch.zeros(Bs, dtype=torch.int32, device="cuda")
            ms_big = timed(lambda: eng.mle_dev(big, rb, init=init, nit=nib), 3)
            entry = {"unit": "reconstructions/s", "mean_nit": float(nib.float().mean().item()), "max_nit": int(nib.max().item()),
                     "nonzero_status": int((s_d != 0).sum().item()),
                     "parity_vs_oracle_64_trials": {"identical_nit": same_nit, "max_infidelity": float(f"{worst:.3e}")},
                     "saturated": {"batch": Bs, "ms_per_step": round(ms_big, 3), "value": round(Bs / ms_big * 1e3, 1)}}
            if small:
                entry.update(batch=nb, ms_per_step=round(ms, 4), value=round(nb / ms * 1e3, 1))
            iterating[name] = entry
            del big, rb

    # ---- the other single-GPU configurations of BASELINE.json, one short measurement each ---------
    others = None
    if rank == 0 and not args.no_other_configs:
        others = {}
        # configs[2]: 2-qubit process tomography of a depolarizing channel, Choi linear inversion + CPTP
        np.random.seed(11)
        ptm = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 2))
        ptm.experiment(10000, "proj-set")
        peng = ptm._engine()
        pb = 1024
        pc = torch.from_numpy(np.ascontiguousarray(np.stack([ptm.results] * pb))).cuda()
        pout = torch.empty((pb, 16, 16), dtype=torch.complex128, device="cuda")
        choi_bytes = 16 * 36 * 8 + 16 * 16 * 16  # per process: int64 counts in, complex128 Choi matrix out
        for cptp in (False, True):
            ms = timed(lambda: peng.lifp_dev(pc, pout, cptp=cptp), 5, peng)
            gbs = pb * choi_bytes / (ms * 1e-3) / 1e9
            others["configs[2] lifp" + (" + CPTP projection" if cptp else "")] = {
                "batch": pb, "ms_per_launch": round(ms, 4), "value": round(pb / ms * 1e3, 1), "unit": "Choi reconstructions/s",
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_process": choi_bytes,
                             "note": ("k_lifp16: one wavefront per process through the Kronecker factors of the left inverse, 34 "
                                      "v_mfma_f64_16x16x4_f64 per process (the dense operator costs 288)" +
                                      (" + k_cptp_wave16 (Dykstra with one wavefront per process; bound by the matrix instructions of its sign iteration, "
                                       "not by HBM)" if cptp else ""))}}
        # the same streaming kernel at a batch that fills the chip, and the dense-operator path (frequencies + FP64 GEMM
        # against the 576 x 256 complex left inverse: what POVMs with M % 4 != 0 take) beside it
        pbig = 65536  # 302 MB of counts in, 268 MB of Choi matrices out per launch: beyond L2 (32 MB) and the 256 MB Infinity Cache
        pcb = pc.repeat(pbig // pb, 1, 1, 1).contiguous()
        poutb = torch.empty((pbig, 16, 16), dtype=torch.complex128, device="cuda")
        ms = timed(lambda: peng.lifp_dev(pcb, poutb, cptp=False), 5, peng)
        gbs = pbig * choi_bytes / (ms * 1e-3) / 1e9
        others["configs[2] lifp, saturated"] = {
            "batch": pbig, "ms_per_launch": round(ms, 4), "value": round(pbig / ms * 1e3, 1), "unit": "Choi reconstructions/s",
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_process": choi_bytes}}
        del pcb, poutb
        peng.process_prefer_dense(True)
        ms = timed(lambda: peng.lifp_dev(pc, pout, cptp=False), 5, peng)
        peng.process_prefer_dense(False)
        gemm_flop = 2.0 * pb * (16 * 36) * (2 * 256)  # [B x R] . [R x 2 D^2] on v_mfma_f64_16x16x4_f64, R = 16 x 36 rows
        tf = gemm_flop / (ms * 1e-3) / 1e12
        others["configs[2] lifp, dense-operator path"] = {
            "batch": pb, "ms_per_launch": round(ms, 4), "value": round(pb / ms * 1e3, 1), "unit": "Choi reconstructions/s",
            "mfma": {"gemm_flop": gemm_flop, "tflops_over_the_whole_call": round(tf, 2),
                     "frac_of_peak": round(tf / FP64_PEAK_TFLOPS, 4),
                     "frac_of_measured_ceiling": round(tf / FP64_MFMA_MEASURED_TFLOPS, 4),
                     "note": "the call = k_lifp_freq + k_lifp_gemm; the GEMM kernel alone: profiles/round3_*kernel_stats*"}}
        del pc, pout
        # the same at n = 3 (not a BASELINE config; 64 x 216 rows, Kronecker-factored set-up, 64 x 64 CPTP projection)
        np.random.seed(31)
        ptm3 = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 3))
        ptm3.experiment(10000, "proj-set")
        t3 = time.perf_counter()
        peng3 = ptm3._engine()
        peng3.sync()
        setup3_ms = (time.perf_counter() - t3) * 1e3
        pb3 = 64
        pc3 = torch.from_numpy(np.ascontiguousarray(np.stack([ptm3.results] * pb3))).cuda()
        pout3 = torch.empty((pb3, 64, 64), dtype=torch.complex128, device="cuda")
        for cptp in (False, True):
            ms = timed(lambda: peng3.lifp_dev(pc3, pout3, cptp=cptp), 3, peng3)
            others["n=3 process lifp" + (" + CPTP projection" if cptp else "")] = {
                "batch": pb3, "ms_per_launch": round(ms, 4), "value": round(pb3 / ms * 1e3, 1), "unit": "Choi reconstructions/s",
                "setup_ms": round(setup3_ms, 3)}
        del pc3, pout3
        # host-pointer API (NumPy in, NumPy out): H2D of the counts + kernel + D2H of rho + synchronise, MEASURED
        reps = 50
        eng.mle(counts)
        th = time.perf_counter()
        for _ in range(reps):
            eng.mle(counts)
        hp = (time.perf_counter() - th) / reps
        others["configs[1] through host pointers (PCIe-inclusive, never `value`)"] = {
            "batch": B, "ms_per_call": round(hp * 1e3, 4), "value": round(B / hp, 1), "unit": "reconstructions/s",
            "includes": "H2D 1.7 KB/trial, kernel, D2H 1 KB/trial + nit/nfev/fun/status, stream synchronise, ctypes marshalling"}
        # the reference's only published performance artefact: examples/pictures/time_test.png (state_tomography.ipynb:277-320),
        # per-call point_estimate on |0...0>, 'proj-set', 10 000 shots per setting, n = 1 ... 5 -- through the drop-in API,
        # one call at a time, beside BASELINE.md section 1's read-off values (unstated hardware, +-20 %).  The pure state
        # makes BFGS iterate: `nit` of the call is reported with it.
        plot = {"lin": {1: 2e-4, 2: 3.7e-4, 3: 1.05e-3, 4: 1.8e-2, 5: 0.65}, "mle init=lin": {1: 6e-3, 2: 5e-2, 3: 0.75, 4: 19.0},
                "mle init=mixed": {1: 7e-3, 2: 6e-2, 3: 1.3, 4: 36.0}}
        series = {}
        for nq in range(1, 6):
            np.random.seed(100 + nq)
            tq = qp.StateTomograph(qp.qobj.zero(nq))
            tq.experiment(10000, "proj-set")
            engq = tq._engine()
            for label, kw in (("lin", dict(method="lin")), ("mle init=lin", dict(method="mle", init="lin")),
                              ("mle init=mixed", dict(method="mle", init="mixed"))):
                tq.point_estimate(**kw)  # first call: POVM registration, allocations
                reps_q = 20 if nq <= 3 else 5
                tq0 = time.perf_counter()
                for _ in range(reps_q):
                    tq.point_estimate(**kw)
                sec = (time.perf_counter() - tq0) / reps_q
                entry = {"seconds_per_call": float(f"{sec:.3e}"), "reference_plot_seconds": plot[label].get(nq)}
                if label != "lin":
                    _, info_q = engq.mle(tq.results, init=kw["init"], return_info=True)
                    entry["nit"] = int(info_q["nit"])
                    entry["status"] = int(info_q["status"])
                if entry["reference_plot_seconds"]:
                    entry["speedup_vs_plot"] = round(entry["reference_plot_seconds"] / sec, 1)
                series.setdefault(label, {})[f"n={nq}"] = entry
        others["time_test_png"] = {
            "what": "examples/state_tomography.ipynb:277-320 (time_test.png): one point_estimate call on |0..0>, 'proj-set', "
                    "1e4 shots/setting, through the drop-in API (host pointers, PCIe and Python glue included)",
            "reference_values": "read off the reference's plot by eye (BASELINE.md section 1; hardware unstated)", "series": series}
        # assembly kernels (a1, a2): HBM write rate of the 5-qubit POVM tensor and Pauli basis
        e5 = qp.get_engine(5)
        t5 = torch.empty((243, 32, 1024), dtype=torch.float64, device="cuda")
        tab = torch.from_numpy(np.ascontiguousarray(qp.measurements._ONE_QUBIT["proj-set"]())).cuda()
        ms = timed(lambda: e5.povm_kron_dev(tab, t5), 10, e5)
        others["a2 qt_povm_kron n=5 (243 x 32 x 1024 f64)"] = {"ms": round(ms, 4), "GBps": round(t5.numel() * 8 / ms / 1e6, 1),
                                                               "frac_hbm": round(t5.numel() * 8 / ms / 1e6 / HBM_PEAK_GBS, 4)}
        del t5

    # ---- configs[4]: 5-qubit MLE, tensor-product Pauli POVM, 1e6 shots per setting; 256 trials PER RANK (weak) ----
    c4 = None
    if not args.no_other_configs:
        n5 = 5
        rho5 = ginibre(np.random.default_rng(1234), 2**n5)
        povm5 = qp.generate_measurement_matrix("proj-set", n5)
        shots5 = np.ones(povm5.shape[0]) * 10**6
        np.random.seed(7)  # the same eight count tensors on every rank (also the centre of the n = 5 bootstrap)
        few = simulate_counts(povm5, qp.Qobj(rho5).bloch, shots5, repeats=8)
        b5 = 256
        e5 = qp.get_engine(n5)
        e5.set_povm(povm5, shots5)
        c5 = torch.from_numpy(np.ascontiguousarray(np.concatenate([few] * (b5 // 8)))).cuda()
        r5 = torch.empty((b5, 32, 32), dtype=torch.complex128, device="cuda")
        st5 = torch.zeros(b5, dtype=torch.int32, device="cuda")
        ni5 = torch.zeros(b5, dtype=torch.int32, device="cuda")
        c4 = {"batch_per_gpu": b5, "scaling": "weak", "n_gpus": world}
        for name, fn in (("lin", lambda: e5.lin_dev(c5, r5)), ("mle", lambda: e5.mle_dev(c5, r5, status=st5, nit=ni5))):
            fn()
            e5.sync()
            torch.cuda.synchronize()
            barrier()
            tw = time.perf_counter()
            for _ in range(5):
                fn()
            e5.sync()
            torch.cuda.synchronize()
            barrier()
            dt = max_over_ranks(time.perf_counter() - tw) / 5
            c4[name] = {"ms_per_launch": round(dt * 1e3, 4), "value": round(world * b5 / dt, 1), "unit": "reconstructions/s"}
        bytes5 = 8 * 7776 + 16 * 1024
        c4["mle"]["hbm_GBps_per_gpu"] = round(bytes5 * b5 / (c4["mle"]["ms_per_launch"] * 1e-3) / 1e9, 2)
        c4["mle"]["mean_nit"] = float(ni5.float().mean().item())
        assert int(st5.sum().item()) == 0
        if rank == 0:  # the same at eight workgroups per CU (2048 trials): the per-trial rate once the launch is not one round
            c5b = c5.repeat(8, 1, 1).contiguous()
            r5b = torch.empty((8 * b5, 32, 32), dtype=torch.complex128, device="cuda")
            for name, fn in (("lin", lambda: e5.lin_dev(c5b, r5b)), ("mle", lambda: e5.mle_dev(c5b, r5b))):
                ms_b = timed(fn, 3, e5)
                c4[name]["batch_2048"] = {"ms_per_launch": round(ms_b, 4), "value": round(8 * b5 / ms_b * 1e3, 1),
                                          "hbm_GBps": round(bytes5 * 8 * b5 / (ms_b * 1e-3) / 1e9, 2)}
            del c5b, r5b
        del c5, r5

    # ---- bootstrap CIs: strong scaling over ranks (interval.py:598-612) -------------------------------
    def selection_serial_ms(engine, sorted_shard, n_tot, sim_ranks, levels):
        """What does NOT shard in a leg, measured on this GPU: the four selection launches on one shard of a
        `sim_ranks`-way split (the gathered [N][P] / [N][L][2+W] arrays are this rank's own, repeated -- same sizes, same
        work; the two all-gathers of a few hundred KB in between cannot be measured on one GPU and are not in here)."""
        plan = qd.selection_plan(n_tot, sim_ranks, len(levels))
        if plan is None:
            return None, None
        stride, n_split, width = plan
        part = sorted_shard[: -(-n_tot // sim_ranks)]
        q = torch.tensor(levels, dtype=torch.float64, device="cuda")
        spl = torch.empty((sim_ranks, n_split), dtype=torch.float64, device="cuda")
        sizes = torch.full((sim_ranks,), part.numel(), dtype=torch.int64, device="cuda")
        lo_k = torch.empty(len(levels), dtype=torch.int64, device="cuda")
        hi_k = torch.empty(len(levels), dtype=torch.int64, device="cuda")
        win = torch.empty((sim_ranks, len(levels), 2 + width), dtype=torch.float64, device="cuda")
        out = torch.empty(len(levels), dtype=torch.float64, device="cuda")
        flag = torch.zeros(2, dtype=torch.int32, device="cuda")

        def steps():
            engine.select_splitters(part, stride, n_split, spl[0])
            engine.select_bracket(spl, sizes, stride, part.numel() * sim_ranks, q, lo_k, hi_k)
            engine.select_window(part, lo_k, hi_k, width, win[0])
            engine.select_finish(win, part.numel() * sim_ranks, q, out, flag)

        engine.select_splitters(part, stride, n_split, spl[0])
        spl[1:] = spl[0]
        engine.select_bracket(spl, sizes, stride, part.numel() * sim_ranks, q, lo_k, hi_k)
        engine.select_window(part, lo_k, hi_k, width, win[0])
        win[1:] = win[0]
        ms = timed(steps, 10, engine)
        return ms, {"simulated_ranks": sim_ranks, "splitters_per_rank": n_split, "stride": stride, "window": width,
                    "exchanged_bytes_per_rank": 8 * (n_split + len(levels) * (2 + width)),
                    "sample_bytes_per_rank": 8 * part.numel()}

    def bootstrap_leg(engine, resamples, centre_matrix, dd, tile=1, device_draw=None, upload_ms=None):
        """Time this rank's shard of the loop of interval.py:598-612: [draw] + reconstruct-and-distance in one pass
        (qt_mle_dist_batch: no density matrix is written) + sort of the shard + the order statistics of the three levels
        across the ranks (ShardedSample: the distributed selection, two all-gathers of a few hundred KB; local at N = 1).
        `resamples` is the full (n, S, K) host array (identical on every rank); `tile` repeats this rank's shard on the
        device.  device_draw = (n_total, pvals (S, K), shots (S,), seed): no host array -- every rank draws ITS shard of
        the n_total resamples in HBM with qt_device_multinomial (rows keyed by their global index), inside the timed region."""
        n_tot = device_draw[0] if device_draw else len(resamples) * tile
        lo, hi = qd.shard_bounds(n_tot)
        t_up = time.perf_counter()
        if device_draw:
            _, pv, sh, seed = device_draw
            p_dev = torch.from_numpy(np.ascontiguousarray(pv)).cuda()
            n_dev = torch.from_numpy(np.asarray(sh).astype(np.int64)).cuda()
            n_set = pv.shape[0]
            shard = torch.empty((hi - lo, n_set, pv.shape[1]), dtype=torch.int64, device="cuda")
        elif tile == 1:
            shard = torch.from_numpy(np.ascontiguousarray(resamples[lo:hi])).cuda()
        else:  # the distinct resamples go up once; this rank's slice of the tiled sequence is gathered on the device
            pool_d = torch.from_numpy(np.ascontiguousarray(resamples)).cuda()
            shard = pool_d[torch.arange(lo, hi, device="cuda") % len(resamples)].contiguous()
            del pool_d
        torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - t_up) * 1e3
        dist_b = torch.empty(hi - lo, dtype=torch.float64, device="cuda")
        st_b = torch.zeros(hi - lo, dtype=torch.int32, device="cuda")
        centre_d = torch.from_numpy(np.ascontiguousarray(centre_matrix)).cuda()
        levels = [0.5, 0.9, 0.95]
        marks = {}

        def run():
            if device_draw:
                engine.device_multinomial(n_dev, p_dev, (hi - lo) * n_set, seed, first_row=lo * n_set, out=shard)
            engine.mle_dist_dev(shard, centre_d, dist_b, status=st_b)
            smp = qd.ShardedSample(dist_b, n_tot, engine=engine)  # sorts the shard in place (asynchronous)
            marks["sorted"] = smp.local
            if world > 1:
                engine.sync()
                marks["t_sel"] = time.perf_counter()
            q = smp.quantiles(levels)  # synchronises
            marks["path"] = smp.last_path
            return q

        run()  # warm-up (allocations, RCCL channel set-up)
        torch.cuda.synchronize()
        barrier()
        tb = time.perf_counter()
        q = run()
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        barrier()
        ms = max_over_ranks((time.perf_counter() - tb) * 1e3)
        assert int(st_b.abs().sum().item()) == 0, "non-zero trial status in a bootstrap leg"
        leg = {"n_points": n_tot, "wall_ms": round(ms, 3), "scaling": "strong", "n_gpus": world,
               "quantiles_hs": [round(float(x), 8) for x in q], "conf_levels": levels, "quantile_path": marks["path"],
               "timed": "reconstruct + HS distance in one pass (qt_mle_dist_batch, 8 B per resample written) + sort of the "
                        "rank's shard + order statistics across ranks (distributed selection; local at N = 1)"}
        if world > 1:  # the part that does not shrink with N, as it ran: selection launches + two small all-gathers + read-back
            leg["serial_ms"] = round(max_over_ranks((t_end - marks["t_sel"]) * 1e3), 4)
            leg["serial_ms_is"] = "measured: distributed selection incl. its two all-gathers, after the shard sort"
        elif rank == 0:
            sim = selection_serial_ms(engine, marks["sorted"], n_tot, 8, levels)
            if sim[0] is not None:
                leg["serial_ms"] = round(sim[0], 4)
                leg["serial_ms_is"] = ("the four selection launches of ONE rank of an 8-way split, run on this GPU "
                                       "(all that does not shard; excludes the two all-gathers of `exchanged_bytes_per_rank`)")
                leg["selection"] = sim[1]
                leg["serial_fraction_of_wall"] = round(sim[0] / ms, 5)
        if device_draw:
            engine.sync()
            td = time.perf_counter()
            engine.device_multinomial(n_dev, p_dev, (hi - lo) * n_set, seed, first_row=lo * n_set, out=shard)
            engine.sync()
            leg["draw_ms"] = round(max_over_ranks((time.perf_counter() - td) * 1e3), 3)
            leg["timed"] = "device draw of this rank's shard (qt_device_multinomial, Philox stream per row) + " + leg["timed"]
            leg["end_to_end_ms"] = leg["wall_ms"]
        else:
            leg["h2d_ms"] = round(max_over_ranks(h2d_ms), 3)
            if upload_ms is not None:  # resampling on the host + upload of the shard + the timed part
                leg["end_to_end_ms"] = round(upload_ms + leg["h2d_ms"] + leg["wall_ms"], 3)
        return leg

    boot = boot_dev = boot5 = boot_large = boot_large_dev = None
    if args.bootstrap_points > 0:
        from quantpy_amd.tomography.state import born_probabilities

        tmg = qp.StateTomograph(state)
        tmg.povm_matrix = povm
        tmg.results = all_counts[0]
        centre = tmg.point_estimate("mle")
        sample_runs = []
        for _ in range(3):  # the same table three times (same seed): a host-side figure, reported as the fastest + all three
            np.random.seed(4242)
            ts = time.perf_counter()
            res = simulate_counts(povm, centre.bloch, tmg.n_measurements, repeats=args.bootstrap_points)
            sample_runs.append((time.perf_counter() - ts) * 1e3)
        sample_ms = min(sample_runs)
        boot = bootstrap_leg(eng, res, centre.matrix, d, upload_ms=sample_ms)
        # the resamples themselves: NumPy's legacy stream in the reference's order (one multinomial per setting per
        # resample), drawn by qt_legacy_multinomial in one call -- serial by nature, the same on every rank
        boot["resampling_host_ms"] = round(sample_ms, 3)
        boot["resampling_host_ms_runs"] = [round(x, 3) for x in sample_runs]
        boot["resampling"] = "qt_legacy_multinomial: np.random's MT19937 stream, reference call order, bit-exact"
        boot["end_to_end_is"] = "resampling_host_ms + h2d_ms + wall_ms: the serial host draw is the Amdahl term at every N"
        # the same CI off the reference's stream (sampler='device'): this rank's shard of the resamples drawn in HBM by
        # qt_device_multinomial (one Philox stream per row: the table does not depend on the number of ranks), inside
        # the timed region -- end to end = wall
        pv_b = born_probabilities(povm, centre.bloch)
        boot_dev = bootstrap_leg(eng, None, centre.matrix, d, device_draw=(args.bootstrap_points, pv_b, tmg.n_measurements, 4242))
        # the draw alone, per call, on the handle's HIP-event pair (the driver's single perf_counter
        # pair around 20 draws read 3.49 ms per draw where every builder run read 0.07): each draw timed by itself
        lo_b, hi_b = qd.shard_bounds(args.bootstrap_points)
        n_set_b = povm.shape[0]
        dev_counts = torch.empty(((hi_b - lo_b) * n_set_b, povm.shape[1]), dtype=torch.int64, device="cuda")
        p_d = torch.from_numpy(pv_b).cuda()
        n_d = torch.from_numpy(np.asarray(tmg.n_measurements).astype(np.int64)).cuda()
        draw = lambda: eng.device_multinomial(n_d, p_d, (hi_b - lo_b) * n_set_b, 4242, first_row=lo_b * n_set_b, out=dev_counts)
        torch.cuda.synchronize()

        def per_call(k):
            ev, host = [], []
            for _ in range(k):
                th = time.perf_counter()
                eng.timer_begin()
                draw()
                ev.append(eng.timer_end())
                host.append((time.perf_counter() - th) * 1e3)
            return ev, host

        cold_ev, cold_host = per_call(20)  # as the GPU is found after the host-side work above
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
            for _ in range(16):
                draw()
            eng.sync()
        warm_ev, warm_host = per_call(20)
        stat = lambda v: {"min": round(min(v), 4), "median": round(float(np.median(v)), 4), "max": round(max(v), 4)}  # noqa: E731
        boot_dev["resampling_device_ms"] = {
            "rows": (hi_b - lo_b) * n_set_b,
            "kernel_hip_events": {"first_20_calls": stat(cold_ev), "after_preroll": stat(warm_ev)},
            "host_wall_per_call": {"first_20_calls": stat(cold_host), "after_preroll": stat(warm_host)},
            "preroll_ms": args.preroll_ms}
        boot["resampling_device_ms"] = round(max_over_ranks(float(np.median(warm_ev))), 4)
        assert bool((dev_counts.sum(1) == n_d.repeat(hi_b - lo_b)).all())
        del dev_counts
        if args.bootstrap_large > 0:
            distinct = min(args.bootstrap_large, 32768)
            tp = time.perf_counter()
            pool = vector_resamples(povm, centre.bloch, tmg.n_measurements, distinct, 99)
            pool_ms = (time.perf_counter() - tp) * 1e3
            tile = max(1, args.bootstrap_large // distinct)
            boot_large = bootstrap_leg(eng, pool, centre.matrix, d, tile=tile)
            boot_large["input"] = (f"{distinct} distinct resamples (vectorised Generator draws, {pool_ms:.0f} ms on the host) tiled "
                                   f"x{tile} on the device: no end_to_end_ms for this leg -- see bootstrap_ci_large_device_draw")
            # the same size with every resample distinct and drawn where it is used
            boot_large_dev = bootstrap_leg(eng, None, centre.matrix, d, device_draw=(
                args.bootstrap_large, pv_b, tmg.n_measurements, 99))
        if not args.no_other_configs:
            t5 = qp.StateTomograph(qp.Qobj(rho5))
            t5.povm_matrix = povm5
            t5.results = few[0]
            centre5 = t5.point_estimate("mle")
            tp = time.perf_counter()
            pool5 = vector_resamples(povm5, centre5.bloch, t5.n_measurements, args.bootstrap_points, 98)
            pool5_ms = (time.perf_counter() - tp) * 1e3
            boot5 = bootstrap_leg(e5, pool5, centre5.matrix, 32, upload_ms=pool5_ms)
            boot5["input"] = "resamples drawn setting by setting in vectorised Generator calls (not the reference's stream)"

    # ---- CPU baseline: the oracle on a bounded sample, one host core (rank 0, N = 1 only) -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        qo = _oracle()
        try:
            from threadpoolctl import threadpool_limits

            limiter = threadpool_limits(limits=1)
        except Exception:
            limiter = None
        ns = args.cpu_sample
        povm_np = np.asarray(povm)
        tc = time.perf_counter()
        worst = 0.0
        for i in range(ns):
            ref = qo.mle_estimate(counts[i % B], povm_np)
            if i < 64:
                worst = max(worst, abs(qo.infidelity(ref, rho_h[i])))
        cpu_s = time.perf_counter() - tc
        if limiter is not None and hasattr(limiter, "restore_original_limits"):
            limiter.restore_original_limits()
        cpu = {"value": round(ns / cpu_s, 3), "unit": "reconstructions/s", "cores": 1, "kind": "port",
               "sample": f"{ns} reconstructions cycling over the {B} trials of this workload, oracle/quantpy_oracle.mle_estimate "
                         f"(scipy BFGS + forward differences = the reference's algorithm), {cpu_s:.1f} s",
               "max_infidelity_gpu_vs_cpu_first64": float(f"{worst:.3e}"),
               "reference_itself": "25 reconstructions/s/core measured in the dev container (BASELINE.md section 2)",
               "all_cores": cpu_all}

    if rank == 0:
        line = {
            "metric": "state reconstructions/sec (3-qubit MLE)", "value": round(value, 1),
            "unit": "reconstructions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: 3-qubit random mixed state, 'proj-set' POVM (27 settings x 8 outcomes), "
                                   "1e5 shots/setting, point_estimate('mle'), 1k-trial batch per GPU",
                       "n_qubits": n, "batch_per_gpu": B, "shots_per_setting": args.shots, "povm": "proj-set",
                       "parallelism": f"trials sharded over {world} GPU(s), no data-path collective"},
            "rccl_ranks": rccl_ranks, "backend": (args.backend if use_dist else None),
            "preroll_ms": round(preroll_ms, 1),
            "no_preroll": (None if cold_elapsed is None else {
                "value": round(world * B * args.steps / cold_elapsed, 1), "ms_per_step": round(cold_elapsed / args.steps * 1e3, 5),
                "kernel_ms": round(cold_kernel_ms, 5),
                "note": "the same W warm-up + K timed steps run BEFORE the pre-roll, straight after the host drew the counts"}),
            "bfgs": {"mean_nit": float(nit.mean()), "mean_nfev": float(nfev.mean()),
                     "reference_equivalent_nfev": float(nfev.mean()) * (D + 1)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "saturated_batch": sat,
            "iterating": iterating,
            "two_stream_pipeline": piped,
            "other_configs": others,
            "configs4_weak": c4,
            "bootstrap_ci": boot,
            "bootstrap_ci_device_sampler": boot_dev,
            "bootstrap_ci_n5": boot5,
            "bootstrap_ci_large": boot_large,
            "bootstrap_ci_large_device_draw": boot_large_dev,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

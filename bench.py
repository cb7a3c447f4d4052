#!/usr/bin/env python3
"""Headline benchmark: state reconstructions/s, 3-qubit MLE (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: qt_mle_batch on B = 1000 independent
3-qubit trials per GPU ('proj-set' POVM, 1e5 shots per setting, Ginibre state of
np.random.default_rng(1234); counts drawn on the host from np.random.seed(7) in the
reference's call order -- SURVEY.md section 8d).  Counts are resident in HBM before the timed
region.  N > 1 (torchrun, one rank per GPU): each rank reconstructs its own 1000 trials
(weak scaling, no data-path collective); afterwards the 2000-resample bootstrap CI of
configs[3] is run strong-scaled with one RCCL all-gather and reported under "bootstrap_ci".

Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "roofline":     the dominant kernel (k_mle_batch<3>) against the HBM roofline, from HIP-event
                  timing of back-to-back launches on the engine's stream;
  "cpu_baseline": the CPU oracle (oracle/quantpy_oracle.py: scipy BFGS + forward differences,
                  i.e. the reference's algorithm) timed on one host core on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # vector FP64, SURVEY.md section 8d


def ginibre(rng, d):
    g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    rho = g @ g.conj().T
    return rho / np.trace(rho)


def _cpu_worker(job):
    """One host core of the all-cores CPU baseline: the oracle (the reference's algorithm) on its slice."""
    counts, n_trials = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import quantpy_oracle as qo

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

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import quantpy_oracle as qo

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1000, help="trials per GPU per step (configs[1]: 1000)")
    ap.add_argument("--shots", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2000, help="trials timed on the CPU oracle")
    ap.add_argument("--cpu-per-core", type=int, default=1500,
                    help="reconstructions per worker in the all-cores CPU baseline; 0 = skip it")
    ap.add_argument("--bootstrap-points", type=int, default=2000)
    ap.add_argument("--saturation-batch", type=int, default=65536, help="extra (untimed-contract) measurement; 0 = off")
    ap.add_argument("--pipelined-steps", type=int, default=0,
                    help="extra: this many steps alternated over two streams (off by default so that a rocprofv3 pass over the "
                         "default command sees single-stream launches only)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[2] / configs[4] side measurements")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N > 1 (nccl = RCCL; gloo for rehearsal)")
    ap.add_argument("--share-gpu0", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    cpu_all = None
    if (rank == 0 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_cpu_baseline
            and args.cpu_per_core > 0):
        try:
            cpu_all = cpu_all_cores(args.shots, args.cpu_per_core)
        except Exception as exc:  # the single-core figure below is the contract; this one is a supplement
            cpu_all = {"error": f"{type(exc).__name__}: {exc}"}
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist

    dev_index = 0 if args.share_gpu0 else local_rank
    torch.cuda.set_device(dev_index)
    use_dist = world > 1 or os.environ.get("QT_BENCH_FORCE_DIST") == "1"  # (1-rank group: rehearsal of the RCCL path)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    import quantpy_amd as qp
    from quantpy_amd.tomography.state import simulate_counts

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
    all_counts = np.stack([simulate_counts(povm, bloch, shots) for _ in range(B * world)])
    counts = all_counts[rank * B:(rank + 1) * B]

    eng = qp.get_engine(n, device=dev_index)
    eng.set_povm(povm, shots)
    counts_d = torch.from_numpy(np.ascontiguousarray(counts)).cuda()
    rho_d = torch.empty((B, d, d), dtype=torch.complex128, device="cuda")
    nit_d = torch.zeros(B, dtype=torch.int32, device="cuda")
    nfev_d = torch.zeros(B, dtype=torch.int32, device="cuda")
    st_d = torch.zeros(B, dtype=torch.int32, device="cuda")

    def step():
        eng.mle_dev(counts_d, rho_d, init="lin", max_iter=100, tol=1e-3, nit=nit_d, nfev=nfev_d, status=st_d)

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    eng.sync()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.timer_begin()
    for _ in range(args.steps):
        step()
    kernel_ms = eng.timer_end() / args.steps  # HIP events on the stream the kernel runs on
    eng.sync()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
    traffic, traffic_src = None, None
    pmc_file = os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")
    if B == 1000 and os.path.exists(pmc_file):  # HBM bytes per launch from the committed PMC passes
        with open(pmc_file) as fh:
            pmc = json.load(fh)
        traffic, traffic_src = pmc["traffic_bytes_per_launch"], "profiles/round1_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
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
        "fp64_dense_equivalent": {"achieved": round(fp64_tflops, 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(fp64_tflops / FP64_PEAK_TFLOPS, 6),
                                  "note": "flops of the dense (unfactorised) algorithm for the same work / FP64 vector peak"},
    }

    # ---- the same step at a batch that fills the chip (occupancy hides the single-wave latency) ----
    sat = None
    if rank == 0 and args.saturation_batch > 0:
        Bs = args.saturation_batch
        reps = (Bs + B - 1) // B
        big = torch.from_numpy(np.ascontiguousarray(np.concatenate([counts] * reps)[:Bs])).cuda()
        rho_s = torch.empty((Bs, d, d), dtype=torch.complex128, device="cuda")
        eng.mle_dev(big, rho_s)
        eng.sync()
        eng.timer_begin()
        for _ in range(10):
            eng.mle_dev(big, rho_s)
        ms = eng.timer_end() / 10
        sat = {"batch": Bs, "ms_per_step": round(ms, 4), "value": round(Bs / ms * 1e3, 1), "unit": "reconstructions/s",
               "hbm_GBps": round(bytes_per_recon * Bs / (ms * 1e-3) / 1e9, 2),
               "fp64_TFLOPs_dense_equivalent": round(flops_trial * Bs / (ms * 1e-3) / 1e12, 3)}
        del big, rho_s

    # ---- the same steps issued alternately on two handles (two HIP streams) ----------------------
    # A 1000-trial launch puts one trial-wave on each SIMD and lasts as long as its slowest trial (the
    # 25 % that need the eigenvalue clip); with a second stream the next batch starts on the SIMDs that
    # are already free.  Reported beside `value`, which stays the single-stream figure.
    piped = None
    if rank == 0 and args.pipelined_steps > 0:
        from quantpy_amd.engine import Engine

        eng2 = Engine(n, dev_index)
        eng2.set_povm(povm, shots)
        engs = (eng, eng2)
        outs = (rho_d, torch.empty_like(rho_d))
        for k in range(20):
            engs[k & 1].mle_dev(counts_d, outs[k & 1])
        eng.sync()
        eng2.sync()
        tp = time.perf_counter()
        for k in range(args.pipelined_steps):
            engs[k & 1].mle_dev(counts_d, outs[k & 1])
        eng.sync()
        eng2.sync()
        dt = time.perf_counter() - tp
        same = bool(torch.equal(outs[0], outs[1]))
        piped = {"steps": args.pipelined_steps, "streams": 2, "ms_per_step": round(dt / args.pipelined_steps * 1e3, 5),
                 "value": round(B * args.pipelined_steps / dt, 1), "unit": "reconstructions/s",
                 "outputs_identical_across_streams": same}
        eng2.close()

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
        for cptp in (False, True):
            peng.lifp_dev(pc, pout, cptp=cptp)
            peng.sync()
            peng.timer_begin()
            for _ in range(5):
                peng.lifp_dev(pc, pout, cptp=cptp)
            ms = peng.timer_end() / 5
            others["configs[2] lifp" + (" + CPTP projection" if cptp else "")] = {
                "batch": pb, "ms_per_launch": round(ms, 4), "value": round(pb / ms * 1e3, 1), "unit": "Choi reconstructions/s"}
        del pc, pout
        # configs[4]: 5-qubit MLE, tensor-product Pauli POVM, 1e6 shots per setting (per-GPU figure)
        n5 = 5
        rho5 = ginibre(np.random.default_rng(1234), 2**n5)
        povm5 = qp.generate_measurement_matrix("proj-set", n5)
        shots5 = np.ones(povm5.shape[0]) * 10**6
        np.random.seed(7)
        few = np.stack([simulate_counts(povm5, qp.Qobj(rho5).bloch, shots5) for _ in range(8)])
        b5 = 256
        e5 = qp.get_engine(n5, device=dev_index)
        e5.set_povm(povm5, shots5)
        c5 = torch.from_numpy(np.ascontiguousarray(np.concatenate([few] * (b5 // 8)))).cuda()
        r5 = torch.empty((b5, 32, 32), dtype=torch.complex128, device="cuda")
        st5 = torch.zeros(b5, dtype=torch.int32, device="cuda")
        for name, fn in (("lin", lambda: e5.lin_dev(c5, r5)), ("mle", lambda: e5.mle_dev(c5, r5, status=st5))):
            fn()
            e5.sync()
            e5.timer_begin()
            for _ in range(5):
                fn()
            ms = e5.timer_end() / 5
            others[f"configs[4] 5-qubit {name}"] = {"batch": b5, "ms_per_launch": round(ms, 4),
                                                   "value": round(b5 / ms * 1e3, 1), "unit": "reconstructions/s"}
        assert int(st5.sum().item()) == 0
        del c5, r5

    # ---- bootstrap CI (configs[3]): strong scaling over ranks, one all-gather -------------------
    boot = None
    if args.bootstrap_points > 0:
        tmg = qp.StateTomograph(state)
        tmg.povm_matrix = povm
        tmg.results = all_counts[0]
        centre = tmg.point_estimate("mle")
        np.random.seed(4242)
        # draw the resamples first (host RNG, untimed input generation), then time reconstruction
        # + distances + gather + quantiles  (what BootstrapStateInterval.setup does, split for timing)
        res = [simulate_counts(povm, centre.bloch, tmg.n_measurements) for _ in range(args.bootstrap_points)]
        res = np.stack(res)
        from quantpy_amd import distributed as qd

        lo, hi = qd.shard_bounds(len(res))
        shard_d = torch.from_numpy(np.ascontiguousarray(res[lo:hi])).cuda()
        rho_b = torch.empty((hi - lo, d, d), dtype=torch.complex128, device="cuda")
        dist_b = torch.empty(hi - lo, dtype=torch.float64, device="cuda")
        centre_d = torch.from_numpy(np.ascontiguousarray(centre.matrix)).cuda()
        torch.cuda.synchronize()
        barrier()
        tb = time.perf_counter()
        eng.mle_dev(shard_d, rho_b)
        eng.hs_dist_dev(rho_b, centre_d, dist_b)
        eng.sync()
        full = qd.allgather_concat(dist_b.cpu().numpy(), len(res))
        q = np.interp([0.5, 0.9, 0.95], np.linspace(0, 1, len(full)), np.sort(full))
        torch.cuda.synchronize()
        barrier()
        boot_ms = (time.perf_counter() - tb) * 1e3
        if use_dist:
            t = torch.tensor([boot_ms], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            boot_ms = float(t.item())
        boot = {"n_points": args.bootstrap_points, "wall_ms": round(boot_ms, 3), "scaling": "strong",
                "quantiles_hs": [round(float(x), 8) for x in q], "conf_levels": [0.5, 0.9, 0.95],
                "timed": "reconstruct + distances + all-gather + quantiles (resampling on the host RNG is input generation)"}

    # ---- CPU baseline: the oracle on a bounded sample, one host core (rank 0, N = 1 only) -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import quantpy_oracle as qo

        try:
            from threadpoolctl import threadpool_limits

            limiter = threadpool_limits(limits=1)
        except Exception:
            limiter = None
        ns = args.cpu_sample
        tc = time.perf_counter()
        worst = 0.0
        for i in range(ns):
            ref = qo.mle_estimate(counts[i % B], povm)
            if i < 64:
                worst = max(worst, abs(qo.infidelity(ref, rho_h[i])))
        cpu_s = time.perf_counter() - tc
        if limiter is not None:
            limiter.unregister() if hasattr(limiter, "unregister") else None
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
            "bfgs": {"mean_nit": float(nit.mean()), "mean_nfev": float(nfev.mean()),
                     "reference_equivalent_nfev": float(nfev.mean()) * (D + 1)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "saturated_batch": sat,
            "two_stream_pipeline": piped,
            "other_configs": others,
            "bootstrap_ci": boot,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

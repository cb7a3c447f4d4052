#!/usr/bin/env python3
"""'pgdb' at n = 3 through the factored design matrix: time per iteration and process (stop='converged', capped)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp  # noqa: E402

np.random.seed(31)
tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 3))
tmg.experiment(10000, "proj-set")
eng = tmg._engine()
for B in (1, 64, 256):
    counts = np.stack([tmg.results] * B)
    for n_iter in (1, 3):
        eng.pgdb(counts, n_iter=n_iter, stop="converged")
        t0 = time.perf_counter()
        _, iters = eng.pgdb(counts, n_iter=n_iter, stop="converged", return_iters=True)
        dt = time.perf_counter() - t0
        print(f"B={B:4d} n_iter={n_iter}: {1e3 * dt:8.2f} ms  (iterations run: {np.atleast_1d(iters)[:4]}; incl. H2D of the counts, "
              f"{counts.nbytes / 1e6:.1f} MB)", flush=True)
t0 = time.perf_counter()
for _ in range(5):
    eng.pgdb_pieces(tmg.results, np.eye(64) / 64)
print(f"qt_pgdb_pieces (one process: model, gradient, CPTP projection of the trial point): {1e3 * (time.perf_counter() - t0) / 5:.2f} ms")
# the Metropolis-Hastings process chain (MHMCProcessInterval) at n = 3: 3 launches per step, no host round trip
ch = tmg.point_estimate("lifp")
for step in (1e-5, 2e-6):
    for T in (20, 200):
        np.random.seed(1)
        deltas, uniforms = np.random.standard_normal((T, 4096)), np.random.rand(T)
        eng.mhmc_process(tmg.results, ch.choi.matrix, deltas, uniforms, step)
        t0 = time.perf_counter()
        chain, acc = eng.mhmc_process(tmg.results, ch.choi.matrix, deltas, uniforms, step)
        dt = time.perf_counter() - t0
        print(f"chain: step {step:g}, T={T:4d}: {1e3 * dt:8.2f} ms = {1e3 * dt / T:6.3f} ms per step, acceptance {acc.mean():.2f} "
              f"(incl. D2H of the {chain.nbytes / 1e6:.1f} MB chain)", flush=True)

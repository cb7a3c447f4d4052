#!/usr/bin/env python3
"""configs[3] end to end through the drop-in API: BootstrapStateInterval(n_points=2000, method='mle') on a
3-qubit tomograph -- host resampling (NumPy legacy RNG, reference order) vs the batched GPU reconstruction."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp  # noqa: E402

rng = np.random.default_rng(1234)
g = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
rho = g @ g.conj().T
rho /= np.trace(rho).real
np.random.seed(7)
tmg = qp.StateTomograph(qp.Qobj(rho))
tmg.experiment(100000, "proj-set")
tmg.point_estimate("mle")
for n_points in (200, 2000):
    t0 = time.perf_counter()
    iv = qp.BootstrapStateInterval(tmg, n_points=n_points, method="mle")
    iv.setup()
    t1 = time.perf_counter()
    boot = qp.StateTomograph(tmg.reconstructed_state)
    t2 = time.perf_counter()
    for _ in range(n_points):
        boot.experiment(tmg.n_measurements, tmg.povm_matrix)
    t3 = time.perf_counter()
    eng = tmg._engine()
    t4 = time.perf_counter()
    eng.mle(iv.boot_counts)
    t5 = time.perf_counter()
    print(f"n_points={n_points}: setup() {1e3 * (t1 - t0):8.1f} ms   of which resampling loop {1e3 * (t3 - t2):8.1f} ms, "
          f"batched mle incl. H2D/D2H {1e3 * (t5 - t4):6.2f} ms;  radii {iv([0.5, 0.9, 0.95])[0]}", flush=True)

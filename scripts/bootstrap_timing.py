#!/usr/bin/env python3
"""configs[3] end to end through the drop-in API: BootstrapStateInterval(n_points=2000, method='mle') on a
3-qubit tomograph -- host resampling (NumPy's legacy stream in the reference's order, drawn by qt_legacy_multinomial in one call) vs
the batched GPU reconstruction; the per-resample experiment() loop is timed beside it."""
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
    np.random.seed(4242)
    t0 = time.perf_counter()
    iv = qp.BootstrapStateInterval(tmg, n_points=n_points, method="mle")
    iv.setup()
    t1 = time.perf_counter()
    boot = qp.StateTomograph(tmg.reconstructed_state)
    np.random.seed(4242)
    t2 = time.perf_counter()
    one_call = boot.experiment_batch(tmg.n_measurements, tmg.povm_matrix, n_points)  # what setup() does
    t3 = time.perf_counter()
    np.random.seed(4242)
    t6 = time.perf_counter()
    for _ in range(n_points):  # the reference's form: one experiment() per resample (27 sampler calls each)
        boot.experiment(tmg.n_measurements, tmg.povm_matrix)
    t7 = time.perf_counter()
    assert np.array_equal(one_call, iv.boot_counts) and np.array_equal(boot.results, one_call[-1])
    eng = tmg._engine()
    t4 = time.perf_counter()
    eng.mle(iv.boot_counts)
    t5 = time.perf_counter()
    print(f"n_points={n_points}: setup() {1e3 * (t1 - t0):8.2f} ms   of which resampling (one qt_legacy_multinomial call) "
          f"{1e3 * (t3 - t2):7.2f} ms [per-resample experiment() loop: {1e3 * (t7 - t6):7.1f} ms], "
          f"batched mle incl. H2D/D2H {1e3 * (t5 - t4):6.2f} ms;  radii {iv([0.5, 0.9, 0.95])[0]}", flush=True)
    # opt-in: the resamples drawn on the GPU (qt_device_multinomial) -- same distribution, not the reference's stream
    for rep in range(3):
        t8 = time.perf_counter()
        dv = qp.BootstrapStateInterval(tmg, n_points=n_points, method="mle", sampler="device", seed=rep)
        dv.setup()
        t9 = time.perf_counter()
    t10 = time.perf_counter()
    boot.experiment_batch(tmg.n_measurements, tmg.povm_matrix, n_points, sampler="device", seed=1)
    t11 = time.perf_counter()
    print(f"n_points={n_points}: sampler='device': setup() {1e3 * (t9 - t8):8.2f} ms   of which resampling "
          f"(Born probabilities on the host + qt_device_multinomial + D2H) {1e3 * (t11 - t10):7.2f} ms;  "
          f"radii {dv([0.5, 0.9, 0.95])[0]}", flush=True)

#!/usr/bin/env python3
"""configs[1] / configs[4] through the drop-in API: wall-clock of ONE StateTomograph.point_estimate (POVM set-up
included) and of repeated estimates on new counts with the same POVM."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp  # noqa: E402

np.random.seed(3)
warm = qp.StateTomograph(qp.qobj.GHZ(2))
warm.experiment(1000)
warm.point_estimate("mle")  # library load, context creation
for n, shots in ((3, 100000), (5, 1000000)):
    rng = np.random.default_rng(1234)
    d = 2**n
    g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    rho = g @ g.conj().T
    state = qp.Qobj(rho / np.trace(rho).real)
    for method in ("lin", "mle"):
        tmg = qp.StateTomograph(state)
        t_e = time.perf_counter()
        tmg.experiment(shots, "proj-set")
        t0 = time.perf_counter()
        tmg.point_estimate(method)
        t1 = time.perf_counter()
        ts = []
        for _ in range(10):
            tmg.experiment(shots, "proj-set")
            t2 = time.perf_counter()
            tmg.point_estimate(method)
            ts.append(time.perf_counter() - t2)
        print(f"n = {n} point_estimate('{method}'): first call {1e3 * (t1 - t0):.2f} ms, later calls "
              f"{1e3 * np.median(ts):.2f} ms   (experiment(): {1e3 * (t0 - t_e):.1f} ms, host sampler + POVM tensor)", flush=True)

#!/usr/bin/env python3
"""configs[2] through the drop-in API: wall-clock of ONE ProcessTomograph.point_estimate('lifp') (set-up included)
and of repeated estimates on new counts with the same inputs / POVM (set-up cached)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp  # noqa: E402

np.random.seed(11)
warm = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 1))
warm.experiment(1000, "proj-set")
warm.point_estimate("lifp")  # library load, context creation
for cptp in (False, True):
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 2))
    tmg.experiment(10000, "proj-set")
    t0 = time.perf_counter()
    ch = tmg.point_estimate("lifp", cptp=cptp)
    t1 = time.perf_counter()
    ts = []
    for _ in range(20):
        tmg.experiment(10000, "proj-set")
        t2 = time.perf_counter()
        tmg.point_estimate("lifp", cptp=cptp)
        ts.append(time.perf_counter() - t2)
    print(f"2-qubit point_estimate('lifp', cptp={cptp}): first call {1e3 * (t1 - t0):.2f} ms (design matrix, left inverse, "
          f"reconstruction), later calls {1e3 * np.median(ts):.2f} ms; CPTP {ch.is_cptp(verbose=False)}", flush=True)

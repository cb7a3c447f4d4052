#!/usr/bin/env python3
"""qt_device_multinomial on configs[1]-shaped rows (27 settings x 8 outcomes, 1e5 shots): time per call, device to device."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp  # noqa: E402
from quantpy_amd.tomography.state import born_probabilities  # noqa: E402

rng = np.random.default_rng(1234)
g = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
rho = g @ g.conj().T
rho /= np.trace(rho).real
povm = qp.generate_measurement_matrix("proj-set", 3)
p = torch.from_numpy(born_probabilities(povm, qp.Qobj(rho).bloch)).cuda()
eng = qp.get_engine(3)
for shots in (100000, 1000, 20):
    n = torch.full((27,), shots, dtype=torch.int64, device="cuda")
    for reps in (2000, 65536, 2097152):
        out = torch.empty((reps * 27, 8), dtype=torch.int64, device="cuda")
        for _ in range(3):
            eng.device_multinomial(n, p, reps * 27, 7, out=out)
        eng.sync()
        t0 = time.perf_counter()
        k = 10 if reps < 10**6 else 3
        for _ in range(k):
            eng.device_multinomial(n, p, reps * 27, 7, out=out)
        eng.sync()
        ms = (time.perf_counter() - t0) * 1e3 / k
        print(f"shots {shots:7d}  resamples {reps:8d}: {ms:9.3f} ms  = {reps * 27 / ms / 1e6:8.2f} G rows/s, {reps * 27 * 7 / ms / 1e6:8.2f} G binomials/s",
              flush=True)
        del out

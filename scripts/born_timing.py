#!/usr/bin/env python3
"""qt_born_probs (a4, batched Born rule p = d A b) timing: HBM bytes = 8 (D + M) per state."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantpy_amd as qp  # noqa: E402

for n, B in ((3, 65536), (3, 1000), (2, 262144), (4, 16384), (5, 2048)):
    d, D = 2**n, 4**n
    povm = qp.generate_measurement_matrix("proj-set", n)
    M = povm.shape[0] * povm.shape[1]
    eng = qp.get_engine(n)
    eng.set_povm(povm, np.ones(povm.shape[0]) * 1000)
    rng = np.random.default_rng(0)
    bl = torch.from_numpy(rng.standard_normal((B, D)) * 0.01).cuda()
    bl[:, 0] = 1.0 / d
    out = torch.empty((B, povm.shape[0], povm.shape[1]), dtype=torch.float64, device="cuda")
    for _ in range(3):
        eng.born_probs(bl, out)
    eng.sync()
    eng.timer_begin()
    for _ in range(10):
        eng.born_probs(bl, out)
    ms = eng.timer_end() / 10
    gb = 8.0 * (D + M) * B / 1e9
    print(f"n={n} B={B}: {ms * 1e3:9.1f} us  {B / ms * 1e3 / 1e6:8.1f} M states/s  {gb / (ms * 1e-3):8.1f} GB/s "
          f"(of 8000)  {2.0 * M * D * B / (ms * 1e-3) / 1e12:6.2f} TFLOP/s", flush=True)

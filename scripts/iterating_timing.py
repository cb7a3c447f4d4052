#!/usr/bin/env python3
"""Time the iterating BFGS regime at n = 3 (configs[1] counts from the fully mixed start: ~15 iterations per trial) for a
1000-trial batch (k_mle_fused) and a 65 536-trial batch (k_mle_start + k_mle_bfgs); QTOMO_LIB selects the library."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantpy_amd as qp  # noqa: E402
from quantpy_amd.tomography.state import simulate_counts  # noqa: E402

n, d = 3, 8
rng = np.random.default_rng(1234)
g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
rho = g @ g.conj().T
rho /= np.trace(rho)
povm = qp.generate_measurement_matrix("proj-set", n)
shots = np.ones(27) * 100000
np.random.seed(7)
counts = np.stack([simulate_counts(povm, qp.Qobj(rho).bloch, shots) for _ in range(1000)])
eng = qp.get_engine(n)
eng.set_povm(povm, shots)
for B in (1000, 65536):
    cd = torch.from_numpy(np.ascontiguousarray(np.concatenate([counts] * ((B + 999) // 1000))[:B])).cuda()
    out = torch.empty((B, d, d), dtype=torch.complex128, device="cuda")
    nit = torch.zeros(B, dtype=torch.int32, device="cuda")
    for init in ("lin", "mixed"):
        eng.mle_dev(cd, out, init=init, nit=nit)
        eng.sync()
        reps = 5
        eng.timer_begin()
        for _ in range(reps):
            eng.mle_dev(cd, out, init=init, nit=nit)
        ms = eng.timer_end() / reps
        print(f"{os.environ.get('QTOMO_LIB', 'libqtomo.so'):40s} B={B:6d} init={init:5s} {ms:9.4f} ms  {B / ms * 1e3:14.1f} recon/s  "
              f"mean nit {nit.float().mean().item():.2f}", flush=True)

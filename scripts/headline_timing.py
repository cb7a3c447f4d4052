#!/usr/bin/env python3
"""configs[1] step (1000 3-qubit MLE trials, 'proj-set', 1e5 shots) timed with HIP events, with the per-trial shots
check on and off (QT_OPT_SHOTS_CHECK), plus 'lin' alone.  QTOMO_LIB selects the library."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantpy_amd as qp  # noqa: E402
from quantpy_amd import _capi  # noqa: E402
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for B in (1000, 65536):
    cd = torch.from_numpy(np.ascontiguousarray(np.concatenate([counts] * ((B + 999) // 1000))[:B])).cuda()
    out = torch.empty((B, d, d), dtype=torch.complex128, device="cuda")
    for check in (1, 0, 1, 0):
        eng.set_option(_capi.QT_OPT_SHOTS_CHECK, check)
        for name, fn in (("mle", lambda: eng.mle_dev(cd, out)), ("lin", lambda: eng.lin_dev(cd, out))):
            for _ in range(20):
                fn()
            eng.sync()
            r = reps if B == 1000 else 10
            eng.timer_begin()
            for _ in range(r):
                fn()
            ms = eng.timer_end() / r
            print(f"B={B:6d} shots_check={check} {name} {ms * 1e3:9.3f} us  {B / ms * 1e3 / 1e6:8.2f} M/s", flush=True)
eng.set_option(_capi.QT_OPT_SHOTS_CHECK, 1)

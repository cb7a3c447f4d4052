#!/usr/bin/env python3
"""Time the individual hot-path kernels at the bench workload (n=3, B trials) with HIP events."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantpy_amd as qp  # noqa: E402
from quantpy_amd import _capi  # noqa: E402
from quantpy_amd.engine import _ptr  # noqa: E402
from quantpy_amd.tomography.state import simulate_counts  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = 50
rng = np.random.default_rng(1234)
g = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
rho = g @ g.conj().T
rho /= np.trace(rho)
povm = qp.generate_measurement_matrix("proj-set", 3)
np.random.seed(7)
one = np.stack([simulate_counts(povm, qp.Qobj(rho).bloch, np.ones(27) * 100000) for _ in range(min(B, 1000))])
counts = np.concatenate([one] * ((B + len(one) - 1) // len(one)))[:B]
eng = qp.get_engine(3)
eng.set_povm(povm, np.ones(27) * 100000)
cd = torch.from_numpy(np.ascontiguousarray(counts)).cuda()
rho_d = torch.empty((B, 8, 8), dtype=torch.complex128, device="cuda")
x_d = torch.empty((B, 64), dtype=torch.float64, device="cuda")
f_d = torch.empty(B, dtype=torch.float64, device="cuda")
g_d = torch.empty((B, 64), dtype=torch.float64, device="cuda")
st = torch.zeros(B, dtype=torch.int32, device="cuda")
lib, h = eng.lib, eng._h
DEV = _capi.QT_DEVICE_PTR


def timeit(name, fn):
    for _ in range(5):
        fn()
    eng.sync()
    eng.timer_begin()
    for _ in range(reps):
        fn()
    ms = eng.timer_end() / reps
    print(f"{name:28s} {ms * 1e3:9.1f} us / launch   {B / ms / 1e3:9.3f} M trials/s")


timeit("lin (no PSD clip)", lambda: lib.qt_lin_batch(h, _ptr(cd), B, 0, _ptr(rho_d), None, _ptr(st), DEV))
timeit("lin + Jacobi PSD clip", lambda: lib.qt_lin_batch(h, _ptr(cd), B, 1, _ptr(rho_d), None, _ptr(st), DEV))
timeit("cholesky param", lambda: lib.qt_chol_param(h, _ptr(rho_d), B, _ptr(x_d), _ptr(st), DEV))
timeit("nll + gradient", lambda: lib.qt_nll_batch(h, _ptr(x_d), _ptr(cd), B, _ptr(f_d), _ptr(g_d), DEV))
timeit("mle (init lin)", lambda: eng.mle_dev(cd, rho_d))
timeit("mle (init mixed)", lambda: eng.mle_dev(cd, rho_d, init="mixed"))
timeit("hs_dist", lambda: eng.hs_dist_dev(rho_d, rho_d[0].contiguous(), f_d))

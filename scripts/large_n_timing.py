#!/usr/bin/env python3
"""Time the n = 4, 5 kernels (configs[4]: 5-qubit 'proj-set', 1e6 shots) with HIP events."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantpy_amd as qp  # noqa: E402
from quantpy_amd.tomography.state import simulate_counts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
shots = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # 0 = full rank
d = 2**n
rng = np.random.default_rng(1234 + n)
g = rng.standard_normal((d, rank or d)) + 1j * rng.standard_normal((d, rank or d))
rho = g @ g.conj().T
rho /= np.trace(rho)
povm = qp.generate_measurement_matrix("proj-set", n)
np.random.seed(7)
bl = qp.Qobj(rho).bloch
few = np.stack([simulate_counts(povm, bl, np.ones(povm.shape[0]) * shots) for _ in range(min(B, 8))])
counts = np.concatenate([few] * ((B + len(few) - 1) // len(few)))[:B]
eng = qp.get_engine(n)
eng.set_povm(povm, np.ones(povm.shape[0]) * shots)
cd = torch.from_numpy(np.ascontiguousarray(counts)).cuda()
rho_d = torch.empty((B, d, d), dtype=torch.complex128, device="cuda")
nit = torch.zeros(B, dtype=torch.int32, device="cuda")
nfev = torch.zeros(B, dtype=torch.int32, device="cuda")


def timeit(name, fn, reps=5):
    fn()
    eng.sync()
    eng.timer_begin()
    for _ in range(reps):
        fn()
    ms = eng.timer_end() / reps
    print(f"n={n} B={B} {name:24s} {ms:10.3f} ms / launch   {B / ms * 1e3:12.1f} trials/s", flush=True)


timeit("lin (physical)", lambda: eng.lin_dev(cd, rho_d))
timeit("mle (init lin)", lambda: eng.mle_dev(cd, rho_d, nit=nit, nfev=nfev))
print("   nit", nit.cpu().numpy()[:8], "nfev", nfev.cpu().numpy()[:8])
timeit("mle (init mixed)", lambda: eng.mle_dev(cd, rho_d, init="mixed", nit=nit, nfev=nfev), reps=2)
print("   nit", nit.cpu().numpy()[:8], "nfev", nfev.cpu().numpy()[:8])

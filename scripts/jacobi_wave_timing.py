"""Where the n = 4, 5 Jacobi rounds spend their time, per wavefront: clocks between barriers (work) and at the
round barrier (wait), from the profile build (QTOMO_LIB=.../libqtomo_prof.so; slots 19 / 31 of qt_large.h)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantpy_amd as qp  # noqa: E402
from quantpy_amd.tomography.state import simulate_counts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = 2**n
rng = np.random.default_rng(1234)
g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
rho = g @ g.conj().T
rho /= np.trace(rho).real
povm = qp.generate_measurement_matrix("proj-set", n)
shots = np.ones(povm.shape[0]) * (10**6 if n == 5 else 10**5)
np.random.seed(7)
base = np.stack([simulate_counts(povm, qp.Qobj(rho).bloch, shots) for _ in range(8)])
counts = np.concatenate([base] * (B // 8))
eng = qp.get_engine(n, device=0)
eng.set_povm(povm, shots)
cd_ = torch.from_numpy(counts).cuda()
out = torch.empty((B, d, d), dtype=torch.complex128, device="cuda")
waves = (d * d) // 64
prof = torch.zeros((B * waves + 8, 32), dtype=torch.int64, device="cuda")
eng.lib.qt_debug_set_prof.argtypes = [ctypes.c_void_p]
assert eng.lib.qt_debug_set_prof(prof.data_ptr()) == 0
for _ in range(2):
    eng.lin_dev(cd_, out)
eng.sync()
prof.zero_()
eng.lin_dev(cd_, out)
eng.sync()
p = prof.cpu().numpy()[: B * waves].reshape(B, waves, 32)
sweeps = (p[:, 1, 20:31] != 0).sum(axis=1)
sel = sweeps > 0
rounds = (sweeps[sel] - 1) * (d - 1)
print(f"n = {n}: {sel.sum()} of {B} trials went through the Jacobi; sweeps {np.bincount(sweeps[sel])[1:]} (index = count - 1)")
for w in range(waves):
    work = p[sel, w, 19] / rounds
    wait = p[sel, w, 31] / rounds
    print(f"  wavefront {w:2d}: work {work.mean():7.0f} clk / round   barrier wait {wait.mean():7.0f} clk / round")

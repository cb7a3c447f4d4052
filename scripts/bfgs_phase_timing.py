"""Iterating regime of the n = 3 MLE (init = 'mixed', ~16 BFGS iterations): clocks of the BFGS loop and of the
(value, gradient) evaluations inside it, per wavefront, from the profile build
(QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so; slots 21-24 of qt_small.h bfgs_iterate)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantpy_amd as qp  # noqa: E402
from quantpy_amd.tomography.state import simulate_counts  # noqa: E402

n, d, B = 3, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(1234)
g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
rho = g @ g.conj().T
rho /= np.trace(rho).real
povm = qp.generate_measurement_matrix("proj-set", n)
shots = np.ones(povm.shape[0]) * 100000
np.random.seed(7)
counts = np.stack([simulate_counts(povm, qp.Qobj(rho).bloch, shots) for _ in range(B)])
eng = qp.get_engine(n, device=0)
eng.set_povm(povm, shots)
cd_ = torch.from_numpy(counts).cuda()
out = torch.empty((B, d, d), dtype=torch.complex128, device="cuda")
prof = torch.zeros((B + 8, 32), dtype=torch.int64, device="cuda")
eng.lib.qt_debug_set_prof.argtypes = [ctypes.c_void_p]
assert eng.lib.qt_debug_set_prof(prof.data_ptr()) == 0
for _ in range(3):
    eng.mle_dev(cd_, out, init="mixed")
eng.sync()
prof.zero_()
eng.timer_begin()
eng.mle_dev(cd_, out, init="mixed")
ms = eng.timer_end()
p = prof.cpu().numpy()[:B]
loop, nll, nit, nfev = p[:, 21], p[:, 22], p[:, 23], p[:, 24]
print(f"k_mle_fused, init mixed: {ms * 1e3:.1f} us for {B} trials; iterations {nit.mean():.1f}, evaluations {nfev.mean():.1f}")
print(f"  BFGS loop      : {loop.mean():9.0f} clk per wavefront (max {loop.max()})")
print(f"  in nll_grad    : {nll.mean():9.0f} clk = {nll.mean() / (nfev.mean() - 1):.0f} per evaluation")
print(f"  everything else: {(loop - nll).mean():9.0f} clk = {(loop - nll).mean() / nit.mean():.0f} per iteration (line search, update of H, H g)")

"""Phase breakdown of k_mle_large (n = 4, 5) from in-kernel stamps; profile build + QTOMO_LIB, see phase_timing.py."""
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
shots_n = 10**6 if n == 5 else 10**5
d = 2**n
rng = np.random.default_rng(1234)
g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
rho = g @ g.conj().T
rho /= np.trace(rho).real
povm = qp.generate_measurement_matrix("proj-set", n)
shots = np.ones(povm.shape[0]) * shots_n
np.random.seed(7)
bloch = qp.Qobj(rho).bloch
base = np.stack([simulate_counts(povm, bloch, shots) for _ in range(8)])
counts = np.concatenate([base] * (B // 8))
eng = qp.get_engine(n, device=0)
eng.set_povm(povm, shots)
cd_ = torch.from_numpy(counts).cuda()
out = torch.empty((B, d, d), dtype=torch.complex128, device="cuda")
waves = (d * d) // 64
prof = torch.zeros((B * waves + 8, 32), dtype=torch.int64, device="cuda")
eng.lib.qt_debug_set_prof.argtypes = [ctypes.c_void_p]
assert eng.lib.qt_debug_set_prof(prof.data_ptr()) == 0
names = {1: "make_ctx (load)", 2: "lin_invert", 3: "cholesky #1", 6: "psd_project (sign clip)", 7: "cholesky #2",
         8: "make_feasible end", 9: "nll_grad", 10: "BFGS + build + store", 11: "(nll) entry", 12: "(nll) build L L^H",
         13: "(nll) bloch_of", 14: "(nll) fwd stages 1..n-1", 15: "(nll) stage n + log", 16: "(nll) backward stages",
         17: "(nll) matrix_of", 18: "(nll) Gt L + tail"}
for _ in range(2):
    eng.mle_dev(cd_, out)
eng.sync()
prof.zero_()
eng.timer_begin()
eng.mle_dev(cd_, out)
ms = eng.timer_end()
p = prof.cpu().numpy()[: B * waves: waves]
nonpd = p[:, 6] > 0
print(f"== k_mle_large_start<{n}>: {ms * 1e3:.1f} us for {B} trials; {nonpd.sum()} non-PD")
for label, sel in (("PD", ~nonpd), ("non-PD", nonpd)):
    if not sel.any():
        continue
    q = p[sel]
    print(f"  {label}: mean total {np.mean(q.max(1) - q[:, 0]):.0f} clk")
    prev = q[:, 0]
    for s in (1, 2, 3, 6, 7, 8, 11, 12, 13, 14, 15, 16, 17, 18, 9, 10):
        cur = q[:, s]
        have = cur > 0
        if not have.any():
            continue
        dt = (cur - prev)[have]
        print(f"    {names[s]:24s} mean {dt.mean():9.0f}  max {dt.max():9.0f} clk")
        prev = np.where(have, cur, prev)

if nonpd.any():
    q = p[nonpd][:8]
    print("Jacobi off^2/nrm^2 (x 1e30) at the start of each sweep, first non-PD trials:")
    for row in q:
        print("   ", [f"{v / 1e30:.1e}" for v in row[20:32] if v != 0])

"""Where an iteration of k_mle_large_bfgs (n = 4, 5; mixed start: 14 iterations) spends its clocks, from in-kernel stamps
of the LAST iteration of each trial (profile build + QTOMO_LIB, see phase_timing.py)."""
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
nit = torch.zeros(B, dtype=torch.int32, device="cuda")
waves = (d * d) // 64
prof = torch.zeros((B * waves + 8, 32), dtype=torch.int64, device="cuda")
eng.lib.qt_debug_set_prof.argtypes = [ctypes.c_void_p]
assert eng.lib.qt_debug_set_prof(prof.data_ptr()) == 0
for _ in range(2):
    eng.mle_dev(cd_, out, init="mixed", nit=nit)
eng.sync()
prof.zero_()
eng.timer_begin()
eng.mle_dev(cd_, out, init="mixed", nit=nit)
ms = eng.timer_end()
p = prof.cpu().numpy()[: B * waves: waves]
print(f"== n = {n}, {B} trials from the mixed state: {ms * 1e3:.1f} us, nit {nit.cpu().numpy()[:4]}")
# the last completed iteration: 21 -> 22 evaluation ... ; the final (breaking) pass overwrites 21, 22 only, so take
# differences that belong together
ev = p[:, 22] - p[:, 21]
print(f"  evaluation (last pass)             mean {ev.mean():9.0f} clk")
for a, b, name in ((23, 24, "two-loop, first loop"), (24, 19, "two-loop, second loop"), (19, 20, "g.p + line-search start")):
    dt = p[:, b] - p[:, a]
    print(f"  {name:34s} mean {dt.mean():9.0f} clk")
print("  (evaluation inside: ", {k: int((p[:, k] - p[:, k - 1]).mean()) for k in range(12, 19)}, ")")

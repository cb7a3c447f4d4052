"""Per-wave phase breakdown of the n = 3 kernels (bench workload) from in-kernel shader-clock stamps.
Needs the profile build:  python -c "from quantpy_amd.build import build_profile_library as b; b()"
and QTOMO_LIB=quantpy_amd/lib/libqtomo_prof.so in the environment."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantpy_amd as qp  # noqa: E402
from quantpy_amd.tomography.state import simulate_counts  # noqa: E402

n, d, B = 3, 8, 1000
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
names = {0: "start", 1: "load_freq", 2: "lin_invert", 3: "cholesky #1", 4: "gauss-jordan inverse", 5: "squarings",
         6: "lift tail / jacobi", 7: "cholesky #2", 8: "make_feasible end", 9: "nll_grad / end", 10: "store",
         11: "(nll) entry", 12: "(nll) build L L^H", 13: "(nll) bloch_of", 14: "(nll) fwd stages 1..n-1",
         15: "(nll) stage n + log", 16: "(nll) backward stages", 17: "(nll) matrix_of", 18: "(nll) Gt L + tail"}
ORDER = [1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 13, 14, 15, 16, 17, 18, 9, 10]
for name, fn in (("k_lin_batch", lambda: eng.lin_dev(cd_, out, physical=True)), ("k_mle_fused", lambda: eng.mle_dev(cd_, out))):
    for _ in range(3):
        fn()
    eng.sync()
    prof.zero_()
    eng.timer_begin()
    fn()
    ms = eng.timer_end()
    p = prof.cpu().numpy()[:B]
    span = (p.max(1) - p[:, 0])
    nonpd = p[:, 6] > 0
    print(f"== {name}: {ms * 1e3:.1f} us for {B} trials; slowest wave {span.max()} clk, {nonpd.sum()} non-PD trials")
    if nonpd.any():
        ks = p[nonpd, 20]
        sq = p[nonpd, 5] - p[nonpd, 4]
        for k in np.unique(ks):
            print(f"  squarings = {k}: {np.sum(ks == k)} waves, phase mean {sq[ks == k].mean():.0f} max {sq[ks == k].max()} clk")
    for label, sel in (("PD trials", ~nonpd), ("non-PD trials", nonpd)):
        if not sel.any():
            continue
        q = p[sel]
        print(f"  {label}: mean total {np.mean(q.max(1) - q[:, 0]):.0f} clk")
        prev = q[:, 0]
        for s in ORDER:
            cur = q[:, s]
            have = cur > 0
            if not have.any():
                continue
            dt = (cur - prev)[have]
            print(f"    {names[s]:24s} mean {dt.mean():8.0f}  max {dt.max():8.0f} clk  ({have.sum()} waves)")
            prev = np.where(have, cur, prev)

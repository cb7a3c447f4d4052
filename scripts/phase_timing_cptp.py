"""Where a CP projection of a 16 x 16 Choi matrix spends its clocks (profile build: QTOMO_LIB=lib/libqtomo_prof.so).
Stamps: 0 kernel entry, 2 end of the positive-definite test, 26 sign-iteration loop entry, 25 = steps taken, 27 loop exit,
1 kernel end (the LAST projection of the launch leaves its stamps).  mode 2 = one CP step, mode 0 = the Dykstra loop."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantpy_amd as qp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
np.random.seed(11)
ptm = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 2))
ptm.experiment(10000, "proj-set")
eng = ptm._engine()
raw = eng.lifp(np.stack([ptm.results] * 4), cptp=False)[0]
x = torch.from_numpy(np.ascontiguousarray(np.stack([raw] * B))).cuda()
out = torch.empty_like(x)
iters = torch.zeros(B, dtype=torch.int32, device="cuda")
prof = torch.zeros((B * 4 + 8, 32), dtype=torch.int64, device="cuda")
eng.lib.qt_debug_set_prof.argtypes = [ctypes.c_void_p]
assert eng.lib.qt_debug_set_prof(prof.data_ptr()) == 0
eng._dev_call()
from quantpy_amd import _capi  # noqa: E402
from quantpy_amd.engine import _ptr  # noqa: E402

for mode, name in ((2, "one CP step"), (0, "Dykstra CPTP")):
    for _ in range(2):
        eng._chk(eng.lib.qt_cptp_project_batch(eng._h, _ptr(x), B, mode, 1000, 1e-12, _ptr(out), _ptr(iters), _capi.QT_DEVICE_PTR))
    eng.sync()
    prof.zero_()
    eng.timer_begin()
    eng._chk(eng.lib.qt_cptp_project_batch(eng._h, _ptr(x), B, mode, 1000, 1e-12, _ptr(out), _ptr(iters), _capi.QT_DEVICE_PTR))
    ms = eng.timer_end()
    p = prof.cpu().numpy()[:B]  # (k_cptp_wave16: one wavefront = one row of stamps per matrix)
    tot = p[:, 1] - p[:, 0]
    print(f"== {name}: {ms * 1e3:.1f} us for {B} matrices; kernel clocks per wavefront (= matrix) mean {tot.mean():.0f} max {tot.max()}; "
          f"iterations {int(iters[0])}")
    print(f"   last projection of the run: PD test ends {np.mean(p[:, 2] - p[:, 0]):.0f} clk after entry (mode 2) ; sign loop "
          f"{np.mean(p[:, 27] - p[:, 26]):.0f} clk for {p[:, 25].mean():.1f} steps = {np.mean((p[:, 27] - p[:, 26]) / np.maximum(p[:, 25], 1)):.0f} clk/step ; "
          f"loop exit to kernel end {np.mean(p[:, 1] - p[:, 27]):.0f} clk")

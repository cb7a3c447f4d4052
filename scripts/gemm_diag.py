"""What bounds k_lifp_gemm (profile build, QTOMO_LIB=lib/libqtomo_prof.so): the kernel with single phases switched off.
diag bits: 1 no MFMAs, 2 no A-operand loads, 4 no slice staging, 8 no result stores."""
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
eng.process_prefer_dense(True)  # (the dense-operator path: qt_lifp_batch otherwise multiplies by the Kronecker factors, k_lifp16)
pc = torch.from_numpy(np.ascontiguousarray(np.stack([ptm.results] * B))).cuda()
out = torch.empty((B, 16, 16), dtype=torch.complex128, device="cuda")
for diag in (0, 8, 4, 2, 1, 3, 7, 15, 0):  # (the variants the profile build instantiates)
    assert eng.lib.qt_debug_set_diag(diag) == 0
    for _ in range(3):
        eng.lifp_dev(pc, out, cptp=False)
    eng.sync()
    eng.timer_begin()
    for _ in range(20):
        eng.lifp_dev(pc, out, cptp=False)
    ms = eng.timer_end() / 20
    print(f"diag {diag:2d} ({'MFMA off ' if diag & 1 else ''}{'A loads off ' if diag & 2 else ''}{'staging off ' if diag & 4 else ''}"
          f"{'stores off' if diag & 8 else ''}): k_lifp_freq + k_lifp_gemm = {ms * 1e3:6.1f} us per {B}")

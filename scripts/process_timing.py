#!/usr/bin/env python3
"""configs[2] (C3): 2-qubit process tomography of a depolarizing channel -- set-up (design matrix, MFMA Gram,
left inverse) and batched Choi reconstruction with and without the CPTP projection, HIP-event timed."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantpy_amd as qp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # 3: 64 x 216 rows, Kronecker-factored set-up (qt_process64.h)
DC = 4**NQ
np.random.seed(11)
tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, NQ))
tmg.experiment(10000, "proj-set")
t0 = time.perf_counter()
eng = tmg._engine()
eng.sync()
t1 = time.perf_counter()
print(f"n = {NQ} process set-up (qt_set_povm + qt_process_setup, {DC * tmg.tomographs[0].results.size} x {DC * DC} complex design matrix"
      f"{', kept as two Kronecker factors' if NQ == 3 else ''}): {1e3 * (t1 - t0):.2f} ms", flush=True)
boot = qp.ProcessTomograph(tmg.point_estimate("lifp"))
few = []
for _ in range(8):
    boot.experiment(10000, "proj-set")
    few.append(boot.results)
counts = np.concatenate([np.stack(few)] * (B // 8))
cd = torch.from_numpy(np.ascontiguousarray(counts)).cuda()
choi = torch.empty((B, DC, DC), dtype=torch.complex128, device="cuda")
it = torch.zeros(B, dtype=torch.int32, device="cuda")
for cptp in (False, True):
    eng.lifp_dev(cd, choi, cptp=cptp, iters=it)
    eng.sync()
    eng.timer_begin()
    for _ in range(5):
        eng.lifp_dev(cd, choi, cptp=cptp, iters=it)
    ms = eng.timer_end() / 5
    print(f"lifp B={B} cptp={cptp}: {ms:8.3f} ms / launch  {B / ms * 1e3:12.1f} processes/s   Dykstra iterations "
          f"{it.cpu().numpy()[:8]}", flush=True)

# the projection kernel by itself on the raw (unprojected) Choi matrices: Dykstra / TP step only / CP step only
from quantpy_amd import _capi  # noqa: E402

raw = torch.empty_like(choi)
eng.lifp_dev(cd, raw, cptp=False)
for mode, name in ((0, "Dykstra CPTP"), (1, "TP projection only"), (2, "CP projection only")):
    args = (eng._h, raw.data_ptr(), B, mode, 1000, 1e-12, choi.data_ptr(), it.data_ptr(), _capi.QT_DEVICE_PTR)
    eng._chk(eng.lib.qt_cptp_project_batch(*args))
    eng.sync()
    eng.timer_begin()
    for _ in range(5):
        eng._chk(eng.lib.qt_cptp_project_batch(*args))
    ms = eng.timer_end() / 5
    print(f"qt_cptp_project_batch B={B} {name:20s}: {ms:8.3f} ms / launch", flush=True)

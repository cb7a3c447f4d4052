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
np.random.seed(11)
tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 2))
tmg.experiment(10000, "proj-set")
t0 = time.perf_counter()
eng = tmg._engine()
eng.sync()
t1 = time.perf_counter()
print(f"process set-up (qt_set_povm + qt_process_setup, 576 x 256 complex design matrix): {1e3 * (t1 - t0):.2f} ms", flush=True)
boot = qp.ProcessTomograph(tmg.point_estimate("lifp"))
few = []
for _ in range(8):
    boot.experiment(10000, "proj-set")
    few.append(boot.results)
counts = np.concatenate([np.stack(few)] * (B // 8))
cd = torch.from_numpy(np.ascontiguousarray(counts)).cuda()
choi = torch.empty((B, 16, 16), dtype=torch.complex128, device="cuda")
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

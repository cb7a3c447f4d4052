"""k_lifp16 (configs[2] linear inversion through the Kronecker factors) as a stream: time and achieved bytes/s by batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantpy_amd as qp
np.random.seed(11)
ptm = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 2))
ptm.experiment(10000, "proj-set")
eng = ptm._engine()
base = torch.from_numpy(np.ascontiguousarray(np.stack([ptm.results] * 1024))).cuda()
for B in (64, 256, 1024, 3072, 4096, 16384, 65536, 262144):
    pc = base.repeat((B + 1023) // 1024, 1, 1, 1)[:B].contiguous()
    out = torch.empty((B, 16, 16), dtype=torch.complex128, device="cuda")
    for _ in range(3):
        eng.lifp_dev(pc, out, cptp=False)
    eng.sync()
    eng.timer_begin()
    reps = 20
    for _ in range(reps):
        eng.lifp_dev(pc, out, cptp=False)
    ms = eng.timer_end() / reps
    print(f"B = {B:7d}: {ms * 1e3:8.1f} us  {B / ms / 1e3:8.1f} M Choi/s  {B * 8704 / ms / 1e6:7.1f} GB/s  ({B * 8704 / ms / 1e6 / 8000:.3f} of HBM)", flush=True)
    del pc, out

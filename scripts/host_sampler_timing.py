"""The bit-exact host sampler (qt_legacy_multinomial) timed repeatedly in one process: 2000 resamples x 27 settings x 8
outcomes at 1e5 shots (the table of configs[3]'s bootstrap).  Background: bench runs read 15.5-17 ms mostly, 82-97 ms sometimes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode in ("gpu", "rccl"):
    import torch

    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
if mode == "rccl":
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29761")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    dist.all_reduce(torch.ones(1, device="cuda"))
from quantpy_amd.sampling import legacy_multinomial  # noqa: E402

rng = np.random.default_rng(1)
p = rng.random((27, 8))
p /= p.sum(1, keepdims=True)
n = np.full(27, 100000)
times = []
for k in range(12):
    np.random.seed(4242)
    t0 = time.perf_counter()
    legacy_multinomial(n, p, 2000)
    times.append((time.perf_counter() - t0) * 1e3)
print(mode, "threads", len(os.listdir("/proc/self/task")), "affinity", len(os.sched_getaffinity(0)),
      " ".join(f"{t:.1f}" for t in times), "ms")

#!/usr/bin/env python3
"""'pgdb' (projected gradient descent with backtracking, process.py:291-308) for a batch of 2-qubit processes:
wall-clock per launch through the engine (host buffers), reference stop rule and the converged one."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
np.random.seed(11)
tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 2))
tmg.experiment(10000, "proj-set")
eng = tmg._engine()
few = []
for _ in range(8):
    tmg.experiment(10000, "proj-set")
    few.append(tmg.results)
counts = np.concatenate([np.stack(few)] * ((B + 7) // 8))[:B]
for stop in ("reference", "converged"):
    eng.pgdb(counts[:8], stop=stop)
    t0 = time.perf_counter()
    choi, iters = eng.pgdb(counts, stop=stop, return_iters=True)
    dt = time.perf_counter() - t0
    print(f"pgdb B={B} stop={stop:9s}: {1e3 * dt:8.2f} ms / launch  {B / dt:10.1f} processes/s   iterations {iters[:8]}", flush=True)

"""Timing of the MomentInterval coverage study of notebooks/Verification.ipynb cell 9 (10 000 trials per state, 10 000 shots
per setting, 'proj-set') through the batched path: one sampler call, one qt_lin_dist_batch, one qt_moment_batch, SciPy's
gamma quantiles vectorised.  The reference's own loop costs ~3 ms per trial (experiment + MomentInterval + point_estimate)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import quantpy_amd as qp  # noqa: E402

N = 10000
levels = np.linspace(1e-3, 1 - 1e-3, N)[[int(c * N) for c in (0.5, 0.75, 0.9, 0.95, 0.99)]]
np.random.seed(1)
for name, state in (("zero1", qp.qobj.zero(1)), ("ghz2", qp.qobj.GHZ(2)), ("ghz3", qp.qobj.GHZ(3))):
    tmg = qp.StateTomograph(state)
    t0 = time.perf_counter()
    counts = tmg.experiment_batch(10000, "proj-set", repeats=N)
    t1 = time.perf_counter()
    eng = tmg._engine()
    eng.set_povm(tmg.povm_matrix, tmg.n_measurements)
    iv = qp.MomentInterval(tmg)
    dim, ns, own, inv = iv._design()
    eng.moments(counts[:8], ns, inv)
    eng.lin_dist(counts[:8], np.asarray(state.matrix), physical=False)
    t2 = time.perf_counter()
    dist = eng.lin_dist(counts, np.asarray(state.matrix), physical=False)
    t3 = time.perf_counter()
    mean, var = eng.moments(counts, ns, inv)
    t4 = time.perf_counter()
    radii = iv.radii_batch(counts, levels)
    t5 = time.perf_counter()
    cov = (dist[:, None] < radii).mean(0)
    print(f"{name}: sampler {1e3 * (t1 - t0):7.1f} ms | qt_lin_dist_batch {1e3 * (t3 - t2):6.2f} ms | qt_moment_batch {1e3 * (t4 - t3):6.2f} ms "
          f"(host pointers, {N} trials) | radii_batch incl. design + gamma.ppf {1e3 * (t5 - t4):7.1f} ms | coverage {np.round(cov, 4).tolist()}")

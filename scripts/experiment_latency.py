#!/usr/bin/env python3
"""Where the time of StateTomograph.experiment() goes (a4: POVM tensor, Born probabilities, draws), n = 3 and 5."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp  # noqa: E402
from quantpy_amd.sampling import legacy_multinomial  # noqa: E402
from quantpy_amd.tomography.state import born_probabilities  # noqa: E402


def best(fn, reps=5):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        out.append(time.perf_counter() - t0)
    return min(out) * 1e3, r


for n, shots in ((3, 100000), (5, 1000000)):
    d = 2**n
    rng = np.random.default_rng(1234 + n)
    g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    rho = g @ g.conj().T
    rho /= np.trace(rho).real
    state = qp.Qobj(rho)
    bloch = state.bloch
    tmg = qp.StateTomograph(state)
    np.random.seed(7)
    tmg.experiment(shots, "proj-set")
    t_all, _ = best(lambda: tmg.experiment(shots, "proj-set"))
    t_povm, povm = best(lambda: qp.generate_measurement_matrix("proj-set", n))
    t_pass, _ = best(lambda: qp.generate_measurement_matrix(povm, n))
    t_born, p = best(lambda: born_probabilities(povm, bloch))
    nm = np.ones(povm.shape[0]) * shots
    t_draw, _ = best(lambda: legacy_multinomial(nm, p, 1))
    t_bloch, _ = best(lambda: qp.Qobj(rho).bloch)
    print(f"n = {n}: experiment() {t_all:8.3f} ms = POVM tensor by name {t_povm:8.3f} (array passed through: {t_pass:.3f}) "
          f"+ Born probabilities (host einsum, reference order) {t_born:8.3f} + draws {t_draw:7.3f};  Qobj.bloch {t_bloch:.3f} ms",
          flush=True)

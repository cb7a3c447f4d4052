"""The d = 32 (and d = 16) eigenvalue clip of the state kernels against the oracle's eigh-based one (state.py:267-273) over
regimes the unit tests only sample: low shots (many large negative eigenvalues), pure / rank-deficient states, near-PD data.
Prints the worst element-wise deviation of point_estimate('lin', physical=True) per regime.  Usage: clip_accuracy_n5.py [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import quantpy_oracle as qo  # noqa: E402  (checker)

import quantpy_amd as qp  # noqa: E402
from quantpy_amd.tomography.state import simulate_counts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
d = 2**n
povm = qp.generate_measurement_matrix("proj-set", n)
povm_np = np.asarray(povm)
eng = qp.get_engine(n)
rng = np.random.default_rng(2026)
worst_all = 0.0
for shots in (30, 1000, 100000, 10000000):
    for rank in (1, 2, d // 2, d):
        g = rng.standard_normal((d, rank)) + 1j * rng.standard_normal((d, rank))
        rho = g @ g.conj().T
        rho /= np.trace(rho).real
        ns = np.ones(povm.shape[0]) * shots
        np.random.seed(int(rng.integers(0, 2**31)))
        counts = simulate_counts(povm, qp.Qobj(rho).bloch, ns, repeats=6)
        eng.set_povm(povm, ns)
        got = eng.lin(counts, physical=True)
        worst, negs = 0.0, []
        for c, r in zip(counts, got):
            ref = qo.lin_estimate(c, povm_np)
            raw = qo.lin_estimate(c, povm_np, physical=False)
            negs.append(int((np.linalg.eigvalsh(raw) < 0).sum()))
            worst = max(worst, float(np.abs(r - ref).max()))
        worst_all = max(worst_all, worst)
        print(f"n={n} shots={shots:>9d} rank={rank:>2d}: negative eigenvalues per trial {negs}  max |GPU - eigh clip| = {worst:.2e}")
print(f"worst over all regimes: {worst_all:.2e}")

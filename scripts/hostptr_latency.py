import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import quantpy_amd as qp
from quantpy_amd.tomography.state import simulate_counts
rng = np.random.default_rng(1234); g = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8)); rho = g @ g.conj().T; rho /= np.trace(rho).real
povm = qp.generate_measurement_matrix("proj-set", 3); shots = np.ones(27) * 100000
np.random.seed(7); counts = simulate_counts(povm, qp.Qobj(rho).bloch, shots, repeats=1000)
eng = qp.get_engine(3); eng.set_povm(povm, shots)
for B in (1, 1000):
    c = counts[:B]
    for _ in range(20): eng.mle(c)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(100): eng.mle(c)
        best = min(best, (time.perf_counter() - t0) / 100)
    print(f"B={B}: {best*1e6:.1f} us per host-pointer call")

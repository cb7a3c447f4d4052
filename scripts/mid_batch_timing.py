import sys, numpy as np
sys.path.insert(0, "/root/repo")
import torch, quantpy_amd as qp
from quantpy_amd import _capi
from quantpy_amd.tomography.state import simulate_counts
rng = np.random.default_rng(1234); g = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8)); rho = g @ g.conj().T; rho /= np.trace(rho).real
povm = qp.generate_measurement_matrix("proj-set", 3); shots = np.ones(27) * 100000
np.random.seed(7); counts = simulate_counts(povm, qp.Qobj(rho).bloch, shots, repeats=4096)
eng = qp.get_engine(3); eng.set_povm(povm, shots)
for B in (1000, 1500, 2000, 2048, 3000, 4096):
    cd = torch.from_numpy(counts[:B]).cuda(); out = torch.empty((B, 8, 8), dtype=torch.complex128, device="cuda")
    res = []
    for fmw in (1024, 2048, 4096):
        eng.set_option(_capi.QT_OPT_MLE_FUSED_MAX_WAVES, fmw)
        for _ in range(5): eng.mle_dev(cd, out)
        eng.sync(); eng.timer_begin()
        for _ in range(200): eng.mle_dev(cd, out)
        res.append(eng.timer_end() / 200 * 1e3)
    print(f"B={B}: fused_max_waves 1024: {res[0]:.2f} us, 2048: {res[1]:.2f} us, 4096: {res[2]:.2f} us")

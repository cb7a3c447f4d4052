"""Quick A/B aid: time 'lin'(physical) and 'mle' on the bench workload (n = 3, B = 1000) and save the
results to gpurun_out/jtol_<tag>.npz so that two builds can be compared element by element."""
import sys
import numpy as np
import torch
import quantpy_amd as qp
from quantpy_amd.tomography.state import simulate_counts

tag = sys.argv[1]
n, d, B = 3, 8, 1000
rng = np.random.default_rng(1234)
g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
rho = g @ g.conj().T
rho /= np.trace(rho).real
state = qp.Qobj(rho)
povm = qp.generate_measurement_matrix("proj-set", n)
shots = np.ones(povm.shape[0]) * 100000
np.random.seed(7)
counts = np.stack([simulate_counts(povm, state.bloch, shots) for _ in range(B)])
eng = qp.get_engine(n, device=0)
eng.set_povm(povm, shots)
cd_ = torch.from_numpy(counts).cuda()
out = torch.empty((B, d, d), dtype=torch.complex128, device="cuda")
res = {}
for name, fn in (("lin", lambda: eng.lin_dev(cd_, out, physical=True)), ("mle", lambda: eng.mle_dev(cd_, out))):
    for _ in range(5):
        fn()
    eng.sync()
    eng.timer_begin()
    for _ in range(50):
        fn()
    ms = eng.timer_end() / 50
    res[name] = out.cpu().numpy().copy()
    print(f"{tag} {name}: {ms*1e3:.2f} us / 1000 trials")
np.savez(f"gpurun_out/jtol_{tag}.npz", **res)

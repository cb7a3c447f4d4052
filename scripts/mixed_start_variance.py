import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch, quantpy_amd as qp
from quantpy_amd.tomography.state import simulate_counts
n, B, shots = 5, 256, 1000000
d = 2**n
rng = np.random.default_rng(1234 + n); g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)); rho = g @ g.conj().T; rho /= np.trace(rho)
povm = qp.generate_measurement_matrix("proj-set", n)
np.random.seed(7)
few = simulate_counts(povm, qp.Qobj(rho).bloch, np.ones(povm.shape[0]) * shots, repeats=8)
counts = np.concatenate([few] * (B // 8))
eng = qp.get_engine(n); eng.set_povm(povm, np.ones(povm.shape[0]) * shots)
cd = torch.from_numpy(np.ascontiguousarray(counts)).cuda()
out = torch.empty((B, d, d), dtype=torch.complex128, device="cuda")
nit = torch.zeros(B, dtype=torch.int32, device="cuda"); nfev = torch.zeros(B, dtype=torch.int32, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
# same sequence as large_n_timing: lin, mle(lin), then mixed
for _ in range(6): eng.lin_dev(cd, out)
for _ in range(6): eng.mle_dev(cd, out, nit=nit, nfev=nfev)
ref = None
for k in range(30):
    eng.sync(); eng.timer_begin()
    eng.mle_dev(cd, out, init="mixed", nit=nit, nfev=nfev, status=st)
    ms = eng.timer_end()
    a, b, c = nit.cpu().numpy(), nfev.cpu().numpy(), st.cpu().numpy()
    r = out.cpu().numpy()
    if ref is None: ref = (a.copy(), b.copy(), r.copy())
    same = np.array_equal(a, ref[0]) and np.array_equal(b, ref[1]) and np.array_equal(r, ref[2])
    print(f"launch {k:2d}: {ms:8.3f} ms  nit {a.min()}..{a.max()} nfev {b.min()}..{b.max()} status max {c.max()}  identical to first: {same}", flush=True)

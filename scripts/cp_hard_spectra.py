"""CP step of k_cptp_project64 on geometric spectra of alternating sign, smallest |eigenvalue| from 1e-4 to 1e-10 of the
largest: error against an eigh-based clip and the number of sign-iteration steps."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp
eng = qp.get_engine(3)
rng = np.random.default_rng(2024)
g = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
q, _ = np.linalg.qr(g)
alt = np.where(np.arange(64) % 2, 1.0, -1.0)
for lo in (1e-4, 1e-6, 1e-8, 1e-9, 1e-10):
    ev = np.geomspace(lo, 1.0, 64) * alt
    a = (q * ev) @ q.conj().T
    a = (a + a.conj().T) / 2
    w, u = np.linalg.eigh(a)
    want = (u * np.maximum(w, 1e-12)) @ u.conj().T
    r, st = eng.cptp_project(a, mode="cp", return_iters=True)
    err = r - want
    # the error in the eigenbasis: which eigen-directions carry it
    eb = np.abs(u.conj().T @ err @ u)
    i, j = np.unravel_index(np.argmax(eb), eb.shape)
    print(f"smallest |eigenvalue| {lo:.0e}: steps {int(st):2d}  max error {np.abs(err).max():.2e}  "
          f"largest eigenbasis entry {eb.max():.2e} between eigenvalues {w[i]:+.2e} and {w[j]:+.2e}")

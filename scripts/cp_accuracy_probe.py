import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp
eng = qp.get_engine(3)
def eigclip(A, eps=1e-12):
    w, U = np.linalg.eigh(A)
    return (U * np.maximum(w, eps)) @ U.conj().T
rng = np.random.default_rng(0)
def herm(n):
    G = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)); return (G + G.conj().T) / 2
H16 = herm(16)
# 1. a 16 x 16 block embedded in the identity, in each of the 4 diagonal tile positions
for pos in range(4):
    A = np.eye(64, dtype=complex) * 3.0
    A[16 * pos:16 * pos + 16, 16 * pos:16 * pos + 16] = H16
    r, st = eng.cptp_project(A, mode="cp", return_iters=True)
    print(f"block at tile {pos}: err {np.abs(r - eigclip(A)).max():.2e} steps {st}")
# 2. real symmetric, diagonal, and spectra with prescribed eigenvalues in a random basis
Q, _ = np.linalg.qr(herm(64) + 1j * herm(64))
for name, ev in (("+-1", np.where(np.arange(64) % 2, 1.0, -1.0)), ("linspace(-1,1)", np.linspace(-1, 1, 64)),
                 ("linspace(-1,1) shifted", np.linspace(-1, 1, 64) + 0.013), ("geometric 1e-4..1, alternating sign", np.geomspace(1e-4, 1, 64) * np.where(np.arange(64) % 2, 1, -1))):
    A = (Q * ev) @ Q.conj().T; A = (A + A.conj().T) / 2
    r, st = eng.cptp_project(A, mode="cp", return_iters=True)
    print(f"eigs {name:36s}: err {np.abs(r - eigclip(A)).max():.2e} steps {st}")
D = np.diag(np.linspace(-1, 1, 64)).astype(complex)
r, st = eng.cptp_project(D, mode="cp", return_iters=True)
print(f"diagonal matrix: err {np.abs(r - eigclip(D)).max():.2e} steps {st}")
S = herm(64).real.astype(complex)
r, st = eng.cptp_project(S, mode="cp", return_iters=True)
print(f"real symmetric: err {np.abs(r - eigclip(S)).max():.2e} steps {st}")

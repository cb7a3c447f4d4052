import numpy as np, sys
sys.path.insert(0, "/root/repo")
import quantpy_amd as qp
g = np.load("/root/repo/tests/golden/process3.npz")
eng = qp.get_engine(3)
def eigclip(A, eps=1e-12):
    A = np.tril(A) + np.tril(A, -1).conj().T
    w, U = np.linalg.eigh(A)
    return (U * np.maximum(w, eps)) @ U.conj().T
rng = np.random.default_rng(0)
mats = {"Q0 raw": g["Q0_choi_nocptp"], "Q1 raw": g["Q1_choi_nocptp"]}
G = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64)); H = (G + G.conj().T) / 2
mats["random hermitian"] = H
mats["psd - 0.3 I"] = G @ G.conj().T / 64 - 0.3 * np.eye(64)
for name, A in mats.items():
    (r1, steps), r2 = eng.cptp_project(A, mode="cp", return_iters=True), eng.cptp_project(A, mode="cp")
    ref = eigclip(A)
    print(f"{name:18s} err vs eigh {np.abs(r1 - ref).max():.3e}  (norm {np.abs(A).max():.2f})  deterministic {np.array_equal(r1, r2)}  sign-iteration steps {steps}  hermitian {np.abs(r1 - r1.conj().T).max():.1e}")
# the 16 x 16 clip (SignClipWG, three LDS images) on the same kinds of matrices, for comparison
eng2 = qp.get_engine(2)
G = rng.standard_normal((16, 16)) + 1j * rng.standard_normal((16, 16)); H = (G + G.conj().T) / 2
for name, A in (("16x16 random hermitian", H), ("16x16 psd - 0.3 I", G @ G.conj().T / 16 - 0.3 * np.eye(16))):
    r1 = eng2.cptp_project(A, mode="cp")
    print(f"{name:24s} err vs eigh {np.abs(r1 - eigclip(A)).max():.3e}  (norm {np.abs(A).max():.2f})")
# sensitivity: the same 64 x 64 matrices scaled (the clip level eps = 1e-12 is absolute)
for sc in (1e-3, 1.0, 1e3):
    A = mats["random hermitian"] * sc
    r1 = eng.cptp_project(A, mode="cp")
    print(f"random hermitian x {sc:g}: err vs eigh {np.abs(r1 - eigclip(A)).max():.3e}, relative {np.abs(r1 - eigclip(A)).max() / np.abs(A).max():.3e}")

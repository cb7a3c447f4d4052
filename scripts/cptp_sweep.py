"""Randomised sweep of the CPTP projection (process.py:231-278) at n = 2 (k_cptp_wave16) and n = 3 (k_cptp_project64) against a
NumPy Dykstra loop with eigh: Choi matrices of random CPTP maps of random Kraus rank plus Hermitian noise of random size.
Reports Dykstra iteration-count mismatches and the worst deviation of the projected matrix.   cptp_sweep.py [cases]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
for nq in (2, 3):
    d = 2**nq; dc = d * d
    rng = np.random.default_rng(1000 + nq)
    eye_d = np.eye(d)
    def tp(c):
        red = np.einsum("aobo->ab", c.reshape(d, d, d, d))
        return c + np.kron((eye_d - red) / d, eye_d)
    def cp(c):
        w, u = np.linalg.eigh(np.tril(c) + np.tril(c, -1).conj().T)
        return (u * np.maximum(w, 1e-12)) @ u.conj().T
    def dykstra(c, n_iter=1000, tol=1e-12):
        x = c.astype(np.complex128); p = q = y = np.zeros_like(x)
        for it in range(n_iter):
            yd = tp(x + p) - y; y = y + yd
            xd = cp(y + q) - x; x = x + xd
            crit = 2 * (abs(np.sum(yd.conj() * q)) + abs(np.sum(xd.conj() * p)))
            pd, qd = x - y, y - x; p, q = p + pd, q + qd
            crit += np.linalg.norm(pd) ** 2 + np.linalg.norm(qd) ** 2
            if crit < tol: break
        return x, it + 1
    def random_cptp(rank):
        k = rng.standard_normal((rank, d, d)) + 1j * rng.standard_normal((rank, d, d))
        s = sum(a.conj().T @ a for a in k)
        w, u = np.linalg.eigh(s)
        k = k @ ((u / np.sqrt(w)) @ u.conj().T)
        v = [a.T.reshape(-1) for a in k]
        return sum(np.outer(x, x.conj()) for x in v)
    cases = []
    for _ in range(N):
        rank = int(rng.integers(1, dc + 1))
        g = rng.standard_normal((dc, dc)) + 1j * rng.standard_normal((dc, dc))
        noise = 10.0 ** rng.uniform(-6, -0.5)
        cases.append(random_cptp(rank) + noise * (g + g.conj().T) / 2)
    batch = np.stack(cases)
    eng = qp.get_engine(nq)
    got, iters = eng.cptp_project(batch, mode="cptp", return_iters=True)
    mism, worst, its = 0, 0.0, []
    for c, gm, it in zip(cases, got, iters):
        want, wit = dykstra(c)
        its.append(wit)
        if int(it) != wit: mism += 1
        else: worst = max(worst, np.abs(gm - want).max())
    print(f"n = {nq}: {N} matrices, Dykstra iterations {min(its)} ... {max(its)} (mean {np.mean(its):.1f}); iteration-count mismatches {mism}; "
          f"worst |GPU - NumPy| over the matching ones {worst:.1e}", flush=True)

"""NumPy model of the CP step's sign iteration in k_cptp_project64 (qt_process64.h), rounding included: products as three
real ones (P1 = ArBr, P2 = AiBi, P3 = (Ar+Ai)(Br+Bi)), Hermitian products as the kernel forms them (upper 16 x 16 tiles
mirrored, diagonal tiles replaced by their Hermitian part).  Two questions:
  1. lifting polynomial: 1.9 x - 0.9 x^3 (round 2) against x (3 - 3.25 x^2 + 1.25 x^4), in tile products per SIMD (a
     Hermitian product keeps the busiest matrix pipe for 3 tiles, a general one for 4) -- on the Dykstra runs of the two
     three-qubit fixtures (same iteration counts as the reference required);
  2. what to do with the anti-Hermitian rounding of X W: leave it in X (round 2 / first round-3 kernel), mirror X without
     touching the diagonal tiles (round 2's rejected variant), mirror X with symmetrised diagonal tiles (the kernel now),
     against an eigh-based clip on spectra with tiny eigenvalues of both signs and on exactly rank-deficient matrices.
    python scripts/sign_schedule_model.py
"""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, dc = 8, 64
eye_d, I = np.eye(d), np.eye(dc)


def tp(c):
    red = np.einsum("aobo->ab", c.reshape(d, d, d, d))
    return c + np.kron((eye_d - red) / d, eye_d)


def herm(a):
    return 0.5 * (a + a.conj().T)


def mm3(a, b):
    ar, ai, br, bi = a.real, a.imag, b.real, b.imag
    p1, p2, p3 = ar @ br, ai @ bi, (ar + ai) @ (br + bi)
    return (p1 - p2) + 1j * (p3 - p1 - p2)


def mirror(m, diag=True):
    r = m.copy()
    for ti in range(4):
        for tj in range(4):
            a, b = slice(16 * ti, 16 * ti + 16), slice(16 * tj, 16 * tj + 16)
            if ti == tj and diag:
                r[a, b] = herm(m[a, b])
            elif ti > tj:
                r[a, b] = m[b, a].conj().T
    return r


LIFT3, LIFT5 = (1.9, -0.9), (3.0, -3.25, 1.25)
X_FULL, X_MIRROR_RAW_DIAG, X_MIRROR = "full", "mirror, diagonal tiles as computed", "mirror"


def clip_sign(A, eps, lift=LIFT5, x_mode=X_MIRROR):
    """-> clipped matrix, tile products per SIMD, (lifting steps, Newton-Schulz steps), |S - S^dagger|"""
    sym_x = {X_FULL: lambda x: x, X_MIRROR_RAW_DIAG: lambda x: mirror(x, False), X_MIRROR: mirror}[x_mode]
    gen = 4 if x_mode == X_FULL else 3
    cap = 24 if len(lift) == 3 else 40
    X = A / np.linalg.norm(A)
    lifting, ns_left, nl, nn, slots = True, 12, 0, 0, 0
    for k in range(64):
        Y = mirror(mm3(X, X)); slots += 3
        res = np.linalg.norm(I - Y) ** 2
        if lifting and (res < 0.5 or k >= cap):
            lifting = False
        last = (not lifting) and (res < 1e-14 or ns_left - 1 <= 0)
        if not lifting:
            ns_left -= 1
        if lifting and len(lift) == 3:
            W = mirror(lift[1] * Y + lift[2] * mm3(Y, Y)); slots += 3
            X = lift[0] * X + mm3(X, W)
        elif lifting:
            X = lift[0] * X + lift[1] * mm3(X, Y)
        else:
            X = 1.5 * X - 0.5 * mm3(X, Y)
        X = sym_x(X); slots += gen
        nl += lifting; nn += not lifting
        if last:
            break
    S = X
    R = 0.5 * (A + mirror(mm3(A, S), False)) + 0.5 * eps * (I - S); slots += 3
    return herm(R), slots, (nl, nn), np.abs(S - S.conj().T).max()


def eigclip(A, eps):
    w, u = np.linalg.eigh(A)
    return (u * np.maximum(w, eps)) @ u.conj().T


def dykstra(c, n_iter=1000, tol=1e-12, **kw):
    x = c.astype(np.complex128); p = q = y = np.zeros_like(x)
    tot, log, worst = 0, [], 0.0
    for it in range(n_iter):
        yd = tp(x + p) - y; y = y + yd
        a = y + q
        a = np.tril(a) + np.tril(a, -1).conj().T
        if np.linalg.eigvalsh(a).min() > 1e-12:
            r = a; log.append("PD")
        else:
            r, slots, st, _ = clip_sign(a, 1e-12, **kw); tot += slots; log.append(st)
            worst = max(worst, np.abs(r - eigclip(a, 1e-12)).max())
        xd = r - x; x = x + xd
        crit = 2 * (abs(np.sum(yd.conj() * q)) + abs(np.sum(xd.conj() * p)))
        pd, qd = x - y, y - x; p, q = p + pd, q + qd
        crit += np.linalg.norm(pd) ** 2 + np.linalg.norm(qd) ** 2
        if crit < tol:
            break
    return x, it + 1, tot, log, worst


if __name__ == "__main__":
    g = np.load(os.path.join(ROOT, "tests/golden/process3.npz"))
    print("1. Dykstra runs of the fixtures (process3.npz): lifting polynomial x treatment of X")
    for key in ("Q0", "Q1"):
        for name, kw in (("cubic, X in full (round 2)", dict(lift=LIFT3, x_mode=X_FULL)), ("quintic, X in full", dict(lift=LIFT5, x_mode=X_FULL)),
                         ("quintic, X mirrored (kernel)", dict(lift=LIFT5, x_mode=X_MIRROR))):
            x, its, slots, log, worst = dykstra(g[key + "_choi_nocptp"], **kw)
            print(f"  {key} {name:30s}: Dykstra {its} (reference {int(g[key + '_dykstra_iters'])}), {slots:4d} tile products per SIMD, worst clip "
                  f"error vs eigh {worst:.1e}, result vs reference {np.abs(x - g[key + '_choi_cptp']).max():.1e}, steps {log}")
    print("2. one CP step against an eigh-based clip")
    rng = np.random.default_rng(2024)
    gm = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
    q, _ = np.linalg.qr(gm)
    alt = np.where(np.arange(64) % 2, 1.0, -1.0)
    spectra = {f"geometric {lo:.0e}..1, alternating sign": np.geomspace(lo, 1.0, 64) * alt for lo in (1e-4, 1e-6, 1e-8, 1e-10)}
    spectra["rank 32 (exact zeros)"] = np.concatenate([np.zeros(32), np.linspace(-1.0, 1.0, 32)])
    spectra["one dominant eigenvalue, bulk +-1e-7"] = np.concatenate([[1.0], 1e-7 * alt[1:]])
    spectra["linspace(-1, 1)"] = np.linspace(-1.0, 1.0, 64)
    for name, ev in spectra.items():
        a = (q * ev) @ q.conj().T
        a = herm(a)
        want = eigclip(a, 1e-12)
        row = []
        for mode in (X_FULL, X_MIRROR_RAW_DIAG, X_MIRROR):
            r, _, st, asym = clip_sign(a, 1e-12, LIFT5, mode)
            row.append(f"{mode}: {np.abs(r - want).max():.1e} (|S - S^H| {asym:.0e})")
        print(f"  {name:40s} steps {st}   " + "   ".join(row))

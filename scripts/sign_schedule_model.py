"""NumPy model of the CP step's sign iteration inside the Dykstra loop of k_cptp_project64 (qt_process64.h): how many
steps each Dykstra iteration's clip takes with the degree-3 lifting polynomial (1.9 x - 0.9 x^3) and with a degree-5 one
x (a + b x^2 + c x^4), in units of the matrix pipe's work (a Hermitian product = 3 tile slots per SIMD, a general one 4),
and how far the clipped matrix is from an eigh-based clip (full matrices, rounding included).
    python scripts/sign_schedule_model.py
"""
import numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

d, dc = 8, 64
eye_d = np.eye(d)
I = np.eye(dc)
def tp(c):
    red = np.einsum("aobo->ab", c.reshape(d, d, d, d))
    return c + np.kron((eye_d - red) / d, eye_d)
def herm(a): return 0.5 * (a + a.conj().T)

def clip_sign(A, eps, lift, max_lift=40):
    """lift = (a, b) degree 3 or (a, b, c) degree 5.  Returns clipped matrix, slots, (lift steps, ns steps)."""
    nrm = np.linalg.norm(A)
    X = A / nrm
    lifting, ns_left, nl, nn, slots = True, 12, 0, 0, 0
    for k in range(64):
        Y = herm(X @ X); slots += 3
        res = np.linalg.norm(I - Y) ** 2
        if lifting and (res < 0.5 or k >= max_lift): lifting = False
        last = (not lifting) and (res < 1e-14 or ns_left - 1 <= 0)
        if not lifting: ns_left -= 1
        if lifting and len(lift) == 3:
            Z = herm(Y @ Y); slots += 3
            W = lift[0] * I + lift[1] * Y + lift[2] * Z
        elif lifting:
            W = lift[0] * I + lift[1] * Y
        else:
            W = 1.5 * I - 0.5 * Y
        X = X @ W; slots += 4
        nl += lifting; nn += (not lifting)
        if last: break
    S = X
    AS = herm(A @ S); slots += 3
    R = 0.5 * (A + AS) + 0.5 * eps * (I - S)
    return herm(R), slots, (nl, nn)

def eigclip(A, eps):
    w, u = np.linalg.eigh(A)
    return (u * np.maximum(w, eps)) @ u.conj().T

def dykstra(c, lift, n_iter=1000, tol=1e-12):
    x = c.astype(np.complex128); p = q = y = np.zeros_like(x)
    tot, log, worst = 0, [], 0.0
    for it in range(n_iter):
        yd = tp(x + p) - y; y = y + yd
        a = y + q
        a = np.tril(a) + np.tril(a, -1).conj().T
        w = np.linalg.eigvalsh(a)
        if w.min() > 1e-12: r = a; log.append("PD")
        else:
            r, slots, st = clip_sign(a, 1e-12, lift); tot += slots; log.append(st)
            worst = max(worst, np.abs(r - eigclip(a, 1e-12)).max())
        xd = r - x; x = x + xd
        crit = 2 * (abs(np.sum(yd.conj() * q)) + abs(np.sum(xd.conj() * p)))
        pd, qd = x - y, y - x; p, q = p + pd, q + qd
        crit += np.linalg.norm(pd) ** 2 + np.linalg.norm(qd) ** 2
        if crit < tol: break
    return x, it + 1, tot, log, worst

if __name__ == "__main__":
    g = np.load(os.path.join(ROOT, "tests/golden/process3.npz"))
    schemes = {"degree 3: 1.9, -0.9": (1.9, -0.9)}
    for a, s in ((3.0, -0.5), (3.1, -0.5), (3.15, -0.6)):
        c = (s + 2 * a - 3) / 2; b = 1 - a - c
        schemes[f"degree 5: {a}, {b:.3f}, {c:.3f}"] = (a, b, c)
    for key in ("Q0", "Q1"):
        for name, lift in schemes.items():
            x, its, slots, log, worst = dykstra(g[key + "_choi_nocptp"], lift)
            print(f"{key} {name:32s}: Dykstra {its} (reference {int(g[key + '_dykstra_iters'])}), {slots} slots, worst clip error vs eigh {worst:.1e}, "
                  f"result vs reference {np.abs(x - g[key + '_choi_cptp']).max():.1e}  steps {log}")

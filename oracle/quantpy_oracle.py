"""CPU oracle for the tomography hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a NumPy/SciPy restatement of the reference algorithm (nordmtr/quantpy) for the
path named in BASELINE.json:north_star.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; nothing under `quantpy_amd/` does, and the
product path raises when the HIP library is missing instead of falling back to this file.

Parity status: PINNED.  Every function below is checked in `tests/test_oracle_golden.py`
against golden vectors that `tests/golden/make_golden.py` produced by importing and running
the reference itself in the development container (versions in tests/golden/meta.json), and
against the one deterministic known-answer the reference holds
(notebooks/Moments.ipynb cells 5-7 + input.json:18-23).

Third-party arithmetic the reference delegates to and that is therefore NOT under
/root/reference (pinned by the reference at scipy 1.9.3 / numpy 1.23.4; exercised here with
the versions in meta.json):
  * scipy.optimize.minimize(method="BFGS") with 2-point forward differences
    (reference call site state.py:213)  -> used as-is here (`mle`), plus a plain restatement
    of the published algorithm (`bfgs_minimize`, MINPACK-2 dcsrch/dcstep line search with the
    Nocedal-Wright zoom fallback) that the HIP kernel's control flow is compared against;
  * scipy.linalg.eigh / inv / cholesky / solve (state.py:270, routines.py:71,86, basis.py:35);
  * numpy.random.multinomial on the legacy global stream (state.py:112).

Conventions (same as the reference): n qubits, d = 2^n, D = 4^n; POVM tensor (S, K, D) holds
Bloch rows E_{s,k} = sum_j A[s,k,j] P_j; complex matrices are complex128, row-major.
"""
import math

import numpy as np
import scipy.linalg as la
from scipy.optimize import minimize

# --------------------------------------------------------------------------------------
# a1: Pauli basis                                  (reference routines.py:6-19)
# --------------------------------------------------------------------------------------
_P1 = np.array(
    [[[1, 0], [0, 1]], [[0, 1], [1, 0]], [[0, -1j], [1j, 0]], [[1, 0], [0, -1]]],
    dtype=np.complex128,
)


def pauli_basis(n):
    """(4^n, 2^n, 2^n) complex128; index k = sum_j k_j 4^(n-1-j), P_k = P_k1 (x) ... (x) P_kn.
    Follows routines.py:14-19 (repeated np.kron of the 1-qubit list)."""
    out = _P1
    for _ in range(n - 1):
        out = np.kron(out, _P1)  # kron over all three axes == element-wise tensor product
    return out


# --------------------------------------------------------------------------------------
# a2: POVM tensors                                 (reference measurements.py:4-94)
# --------------------------------------------------------------------------------------
def _povm_1q(name):
    I_ = np.array([1.0, 0, 0, 0])
    X, Y, Z = np.eye(4)[1], np.eye(4)[2], np.eye(4)[3]
    if name == "proj":  # measurements.py:36-43
        return np.array([I_ + X, I_ - X, I_ + Y, I_ - Y, I_ + Z, I_ - Z]) / 6
    if name == "proj-set":  # :44-60
        return np.array([[I_ + X, I_ - X], [I_ + Y, I_ - Y], [I_ + Z, I_ - Z]]) / 2
    if name == "proj4":  # :61-66
        return np.array([I_ + X, I_ + Y, I_ + Z, I_ - Z]) / 4
    if name == "sic":  # :67-73
        s = 1 / np.sqrt(3)
        return np.array([[1, s, s, s], [1, s, -s, -s], [1, -s, s, -s], [1, -s, -s, s]]) / 4
    raise ValueError("Incorrect string shortcut for argument `povm`")


def measurement_matrix(povm="proj", n_qubits=1):
    """(S, K, 4^n) float64.  measurements.py:34-93."""
    if isinstance(povm, str):
        base = _povm_1q(povm)
    elif isinstance(povm, np.ndarray):
        if povm.shape[-1] == 4:
            base = povm
        elif povm.shape[-1] == 4**n_qubits:
            return povm[None, :, :] if povm.ndim == 2 else povm
        else:
            raise ValueError("Incorrect POVM matrix")
    else:
        raise ValueError("Incorrect value for argument `povm`")
    if base.ndim == 2:
        base = base[None, :, :]
    out = base
    for _ in range(n_qubits - 1):
        out = np.kron(out, base)
    return out


# --------------------------------------------------------------------------------------
# a3: matrix <-> Bloch                             (reference qobj.py:109-135, geometry.py:59-70)
# --------------------------------------------------------------------------------------
def bloch_from_matrix(m):
    """b_k = Re Tr(P_k M^dagger) / d   (qobj.py:132 with geometry.product).  Kept in the
    reference's own evaluation order (matmul, then np.trace's pairwise sum) because the
    sampler is sensitive to the last bit of p for structured states: a conditional
    probability that lands on 0.5 +- 1 ulp picks a different branch of NumPy's legacy
    binomial (see tests/test_oracle_golden.py::test_process_sampling_order)."""
    m = np.asarray(m, dtype=np.complex128)
    if m.ndim > 2:
        return np.stack([bloch_from_matrix(x) for x in m])
    n = int(round(math.log2(m.shape[-1])))
    mh = m.conj().T
    return np.array([np.real(np.trace(p @ mh, dtype=np.complex128)) for p in pauli_basis(n)]) / 2**n


def matrix_from_bloch(b):
    """M = sum_k b_k P_k   (qobj.py:114-117)."""
    b = np.asarray(b)
    n = int(round(math.log2(b.shape[-1]) / 2))
    return np.einsum("...k,kij->...ij", b.astype(np.complex128), pauli_basis(n))


# --------------------------------------------------------------------------------------
# a4: Born probabilities + multinomial sampling   (reference state.py:101-128)
# --------------------------------------------------------------------------------------
def born_probs(povm_matrix, bloch):
    """p[s,k] = d * sum_j A[s,k,j] b[j], clipped to [0,1]   (state.py:109-110)."""
    d = int(round(math.sqrt(povm_matrix.shape[-1])))
    p = np.einsum("ijk,k->ij", povm_matrix, bloch) * d
    return np.clip(p, 0, 1)


def broadcast_shots(n_measurements, n_settings):
    """state.py:104-107 (integer scalar -> per-setting float vector; length check)."""
    if np.issubdtype(type(n_measurements), np.integer):
        return np.ones(n_settings) * n_measurements
    if len(n_measurements) != n_settings:
        raise ValueError("Wrong length for argument `n_measurements`")
    return n_measurements


def sample_counts(povm_matrix, bloch, n_measurements):
    """One np.random.multinomial per POVM setting, in order, on the GLOBAL legacy stream
    (state.py:111-114)."""
    nm = broadcast_shots(n_measurements, povm_matrix.shape[0])
    p = born_probs(povm_matrix, bloch)
    return np.asarray([np.random.multinomial(n_s, p_s) for p_s, n_s in zip(p, nm)])


# --------------------------------------------------------------------------------------
# a5: left inverse                                 (reference routines.py:69-71)
# --------------------------------------------------------------------------------------
def left_inv(a):
    """inv(A^T A) A^T with a PLAIN transpose (not conjugate) -- routines.py:71."""
    return la.inv(a.T @ a) @ a.T


def weighted_povm(povm_matrix, n_meas):
    """A' = reshape(povm * N_s / sum N, (M, D))   (state.py:194-197, :222-225)."""
    n_meas = np.asarray(n_meas, dtype=float)
    return np.reshape(povm_matrix * n_meas[:, None, None] / np.sum(n_meas), (-1, povm_matrix.shape[-1]))


# --------------------------------------------------------------------------------------
# a6/a7: linear inversion + PSD clip               (reference state.py:191-202, 267-273)
# --------------------------------------------------------------------------------------
def make_feasible(m):
    """eigh; clip eigenvalues at 1e-15; U V U^dagger; divide by trace  (state.py:267-273)."""
    v, u = la.eigh(m)
    m2 = u @ np.diag(np.maximum(1e-15, v)) @ u.T.conj()
    return m2 / np.trace(m2)


def lin_estimate(counts, povm_matrix, physical=True, return_bloch=False):
    """counts (S, K) int -> rho (d, d).  state.py:191-202; n_measurements = counts.sum(-1)
    (the `results` setter, state.py:138-141)."""
    counts = np.asarray(counts)
    d = int(round(math.sqrt(povm_matrix.shape[-1])))
    n_meas = counts.sum(-1)
    flat = counts.flatten()
    freq = flat / flat.sum()
    a = weighted_povm(povm_matrix, n_meas)
    bloch = left_inv(a) @ freq / d
    rho = matrix_from_bloch(bloch)
    if physical:
        rho = make_feasible(rho)
    return (rho, bloch) if return_bloch else rho


# --------------------------------------------------------------------------------------
# a8: Cholesky parametrisation                     (reference routines.py:84-101)
# --------------------------------------------------------------------------------------
def matrix_to_tril_vec(m):
    """x = [diag(L) | Re L_(i>j) | Im L_(i>j)], strict-lower order np.tril_indices(d,-1);
    the imaginary part of the diagonal is discarded (routines.py:86-90)."""
    low = la.cholesky(m, lower=True)
    d = low.shape[0]
    off = low[np.tril_indices(d, -1)]
    return np.concatenate((np.real(np.diag(low)), np.real(off), np.imag(off)))


def tril_vec_to_lower(x):
    x = np.asarray(x, dtype=float)
    d = int(math.isqrt(len(x)))
    t = d * (d - 1) // 2
    low = np.zeros((d, d), dtype=np.complex128)
    low[np.tril_indices(d, -1)] = x[d:d + t] + 1j * x[d + t:]
    low[np.diag_indices(d)] = x[:d]
    return low


def tril_vec_to_matrix(x):
    """L L^dagger   (routines.py:93-101)."""
    low = tril_vec_to_lower(x)
    return low @ low.T.conj()


# --------------------------------------------------------------------------------------
# a9: negative log-likelihood                      (reference state.py:217-229)
# --------------------------------------------------------------------------------------
class NllProblem:
    """Holds what state.py:222-227 rebuilds on every call: A', f = counts / sum N."""

    def __init__(self, counts, povm_matrix):
        counts = np.asarray(counts)
        self.d = int(round(math.sqrt(povm_matrix.shape[-1])))
        self.n = int(round(math.log2(self.d)))
        n_meas = counts.sum(-1)
        self.a = weighted_povm(povm_matrix, n_meas)
        self.freq = counts.flatten() / sum(n_meas)
        self.basis = pauli_basis(self.n)

    def nll(self, x):
        m = tril_vec_to_matrix(x)
        rho = m / np.trace(m)
        bloch = np.real(np.einsum("kij,ij->k", self.basis, rho.conj())) / self.d
        p = self.a @ bloch * self.d
        return -np.sum(self.freq * np.log(p + 1e-10))

    def nll_and_grad(self, x):
        """Analytic gradient of `nll` (not in the reference, which differentiates numerically;
        used to quantify the finite-difference noise and as the value the HIP kernel's
        analytic gradient is compared with).  With M = L L^dagger, t = Tr M, rho = M / t:
        G = -sum_k w_k P_k, w = A'^T (f / (p + eps));  Gt = (G - Tr(G rho) I) / t;
        df/dRe L_ij = 2 Re (Gt L)_ij, df/dIm L_ij = 2 Im (Gt L)_ij, df/dL_ii = 2 Re (Gt L)_ii."""
        low = tril_vec_to_lower(x)
        m = low @ low.T.conj()
        t = np.real(np.trace(m))
        rho = m / t
        bloch = np.real(np.einsum("kij,ij->k", self.basis, rho.conj())) / self.d
        p = self.a @ bloch * self.d
        f = -np.sum(self.freq * np.log(p + 1e-10))
        w = self.a.T @ (self.freq / (p + 1e-10))
        g = -np.einsum("k,kij->ij", w.astype(np.complex128), self.basis)
        gt = (g - np.real(np.trace(g @ rho)) * np.eye(self.d)) / t
        q = gt @ low
        d = self.d
        il = np.tril_indices(d, -1)
        grad = np.concatenate((2 * np.real(np.diag(q)), 2 * np.real(q[il]), 2 * np.imag(q[il])))
        return f, grad


# --------------------------------------------------------------------------------------
# a10: MLE                                         (reference state.py:204-215)
# --------------------------------------------------------------------------------------
def mle_start(counts, povm_matrix, init="lin"):
    d = int(round(math.sqrt(povm_matrix.shape[-1])))
    if init == "mixed":
        x0 = np.eye(d, dtype=np.complex128) / d
    elif init == "lin":
        x0 = lin_estimate(counts, povm_matrix, physical=True)
    else:
        raise ValueError("Invalid value for argument `init`")
    return matrix_to_tril_vec(x0)


def mle_estimate(counts, povm_matrix, init="lin", max_iter=100, tol=1e-3, jac="fd", solver="scipy",
                 return_info=False):
    """counts -> rho.  Default (jac='fd', solver='scipy') is the reference's computation:
    scipy BFGS, forward differences, tol -> gtol (inf-norm), maxiter.
    jac='analytic' swaps in the exact gradient; solver='port' runs `bfgs_minimize` below."""
    prob = NllProblem(counts, povm_matrix)
    x0 = mle_start(counts, povm_matrix, init)
    if solver == "scipy":
        if jac == "fd":
            res = minimize(prob.nll, x0, method="BFGS", tol=tol, options={"maxiter": max_iter})
        else:
            res = minimize(prob.nll_and_grad, x0, jac=True, method="BFGS", tol=tol, options={"maxiter": max_iter})
        info = dict(nit=res.nit, nfev=res.nfev, njev=res.njev, status=res.status, x=res.x, fun=res.fun)
    else:
        info = bfgs_minimize(prob.nll_and_grad, x0, gtol=tol, maxiter=max_iter)
    m = tril_vec_to_matrix(info["x"])
    rho = m / np.trace(m)
    return (rho, info) if return_info else rho


# --------------------------------------------------------------------------------------
# Restatement of the published optimizer SciPy runs for method="BFGS"
# (scipy/optimize/_optimize.py:_minimize_bfgs, _linesearch.py, _dcsrch.py; MINPACK-2
#  dcsrch/dcstep by More' & Thuente; zoom = Nocedal & Wright Alg. 3.5/3.6).
# `fg(x) -> (f, grad)`.  Returns the same fields as OptimizeResult that the tests use.
# --------------------------------------------------------------------------------------
def _dcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax):
    sgnd = np.sign(dp) * np.sign(dx)
    with np.errstate(all="ignore"):
        if fp > fx:
            theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
            s = max(abs(theta), abs(dx), abs(dp))
            gamma = s * np.sqrt((theta / s) ** 2 - (dx / s) * (dp / s))
            if stp < stx:
                gamma = -gamma
            r = ((gamma - dx) + theta) / (((gamma - dx) + gamma) + dp)
            stpc = stx + r * (stp - stx)
            stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx)
            stpf = stpc if abs(stpc - stx) <= abs(stpq - stx) else stpc + (stpq - stpc) / 2.0
            brackt = True
        elif sgnd < 0.0:
            theta = 3 * (fx - fp) / (stp - stx) + dx + dp
            s = max(abs(theta), abs(dx), abs(dp))
            gamma = s * np.sqrt((theta / s) ** 2 - (dx / s) * (dp / s))
            if stp > stx:
                gamma = -gamma
            r = ((gamma - dp) + theta) / (((gamma - dp) + gamma) + dx)
            stpc = stp + r * (stx - stp)
            stpq = stp + (dp / (dp - dx)) * (stx - stp)
            stpf = stpc if abs(stpc - stp) > abs(stpq - stp) else stpq
            brackt = True
        elif abs(dp) < abs(dx):
            theta = 3 * (fx - fp) / (stp - stx) + dx + dp
            s = max(abs(theta), abs(dx), abs(dp))
            gamma = s * np.sqrt(max(0, (theta / s) ** 2 - (dx / s) * (dp / s)))
            if stp > stx:
                gamma = -gamma
            r = ((gamma - dp) + theta) / ((gamma + (dx - dp)) + gamma)
            if r < 0 and gamma != 0:
                stpc = stp + r * (stx - stp)
            elif stp > stx:
                stpc = stpmax
            else:
                stpc = stpmin
            stpq = stp + (dp / (dp - dx)) * (stx - stp)
            if brackt:
                stpf = stpc if abs(stpc - stp) < abs(stpq - stp) else stpq
                if stp > stx:
                    stpf = min(stp + 0.66 * (sty - stp), stpf)
                else:
                    stpf = max(stp + 0.66 * (sty - stp), stpf)
            else:
                stpf = stpc if abs(stpc - stp) > abs(stpq - stp) else stpq
                stpf = min(max(stpf, stpmin), stpmax)
        else:
            if brackt:
                theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp
                s = max(abs(theta), abs(dy), abs(dp))
                gamma = s * np.sqrt((theta / s) ** 2 - (dy / s) * (dp / s))
                if stp > sty:
                    gamma = -gamma
                r = ((gamma - dp) + theta) / (((gamma - dp) + gamma) + dy)
                stpf = stp + r * (sty - stp)
            elif stp > stx:
                stpf = stpmax
            else:
                stpf = stpmin
    if fp > fx:
        sty, fy, dy = stp, fp, dp
    else:
        if sgnd < 0:
            sty, fy, dy = stx, fx, dx
        stx, fx, dx = stp, fp, dp
    return stx, fx, dx, sty, fy, dy, stpf, brackt


def _search_wolfe1(phi_dphi, phi0, old_phi0, derphi0, c1=1e-4, c2=0.9, amax=1e100, amin=1e-100, xtol=1e-14):
    """scalar_search_wolfe1 + DCSRCH.__call__/_iterate.  Returns (stp|None, phi1, evals)."""
    if old_phi0 is not None and derphi0 != 0:
        alpha1 = min(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0)
        if alpha1 < 0:
            alpha1 = 1.0
    else:
        alpha1 = 1.0
    stp = alpha1
    # START checks (dcsrch errors -> failure)
    if stp < amin or stp > amax or derphi0 >= 0:
        return None, phi0, 0
    brackt, stage = False, 1
    finit, ginit = phi0, derphi0
    gtest = c1 * ginit
    width = amax - amin
    width1 = width / 0.5
    stx, fx, gx = 0.0, finit, ginit
    sty, fy, gy = 0.0, finit, ginit
    stmin, stmax = 0.0, stp + 4.0 * stp
    evals = 0
    f = phi0
    # iteration 0 of the 100 was the START call; 99 remain
    for _ in range(99):
        if not np.isfinite(stp):
            return None, f, evals
        f, g = phi_dphi(stp)
        evals += 1
        ftest = finit + stp * gtest
        if stage == 1 and f <= ftest and g >= 0:
            stage = 2
        warn = False
        if brackt and (stp <= stmin or stp >= stmax):
            warn = True
        if brackt and stmax - stmin <= xtol * stmax:
            warn = True
        if stp == amax and f <= ftest and g <= gtest:
            warn = True
        if stp == amin and (f > ftest or g >= gtest):
            warn = True
        if f <= ftest and abs(g) <= c2 * -ginit:
            return stp, f, evals  # CONVERGENCE (overrides a warning, as in the original)
        if warn:
            return None, f, evals
        if stage == 1 and f <= fx and f > ftest:
            fm, fxm, fym = f - stp * gtest, fx - stx * gtest, fy - sty * gtest
            gm, gxm, gym = g - gtest, gx - gtest, gy - gtest
            stx, fxm, gxm, sty, fym, gym, stp, brackt = _dcstep(
                stx, fxm, gxm, sty, fym, gym, stp, fm, gm, brackt, stmin, stmax)
            fx, fy = fxm + stx * gtest, fym + sty * gtest
            gx, gy = gxm + gtest, gym + gtest
        else:
            stx, fx, gx, sty, fy, gy, stp, brackt = _dcstep(
                stx, fx, gx, sty, fy, gy, stp, f, g, brackt, stmin, stmax)
        if brackt:
            if abs(sty - stx) >= 0.66 * width1:
                stp = stx + 0.5 * (sty - stx)
            width1 = width
            width = abs(sty - stx)
        if brackt:
            stmin, stmax = min(stx, sty), max(stx, sty)
        else:
            stmin = stp + 1.1 * (stp - stx)
            stmax = stp + 4.0 * (stp - stx)
        stp = min(max(stp, amin), amax)
        if (brackt and (stp <= stmin or stp >= stmax)) or (brackt and stmax - stmin <= xtol * stmax):
            stp = stx
    if not np.isfinite(stp):
        return None, f, evals
    return None, f, evals  # maxiter reached


def _cubicmin(a, fa, fpa, b, fb, c, fc):
    with np.errstate(divide="raise", over="raise", invalid="raise"):
        try:
            C = fpa
            db, dc = b - a, c - a
            denom = (db * dc) ** 2 * (db - dc)
            t0 = fb - fa - C * db
            t1 = fc - fa - C * dc
            A = (dc**2 * t0 - db**2 * t1) / denom
            B = (-(dc**3) * t0 + db**3 * t1) / denom
            radical = B * B - 3 * A * C
            xmin = a + (-B + np.sqrt(radical)) / (3 * A)
        except ArithmeticError:
            return None
    return xmin if np.isfinite(xmin) else None


def _quadmin(a, fa, fpa, b, fb):
    with np.errstate(divide="raise", over="raise", invalid="raise"):
        try:
            db = b - a * 1.0
            B = (fb - fa - fpa * db) / (db * db)
            xmin = a - fpa / (2.0 * B)
        except ArithmeticError:
            return None
    return xmin if np.isfinite(xmin) else None


def _search_wolfe2(phi, dphi, phi0, old_phi0, derphi0, c1=1e-4, c2=0.9, amax=1e100, maxiter=10):
    """scalar_search_wolfe2 + _zoom.  Returns (alpha|None, phi_star, derphi_star|None)."""
    alpha0 = 0.0
    if old_phi0 is not None and derphi0 != 0:
        alpha1 = min(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0)
    else:
        alpha1 = 1.0
    if alpha1 < 0:
        alpha1 = 1.0
    alpha1 = min(alpha1, amax)
    phi_a1 = phi(alpha1)
    phi_a0, derphi_a0 = phi0, derphi0

    def zoom(a_lo, a_hi, phi_lo, phi_hi, derphi_lo):
        i = 0
        phi_rec, a_rec = phi0, 0.0
        a_j = None
        while True:
            dalpha = a_hi - a_lo
            a, b = (a_hi, a_lo) if dalpha < 0 else (a_lo, a_hi)
            if i > 0:
                cchk = 0.2 * dalpha
                a_j = _cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec, phi_rec)
            if i == 0 or a_j is None or a_j > b - cchk or a_j < a + cchk:
                qchk = 0.1 * dalpha
                a_j = _quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi)
                if a_j is None or a_j > b - qchk or a_j < a + qchk:
                    a_j = a_lo + 0.5 * dalpha
            phi_aj = phi(a_j)
            if phi_aj > phi0 + c1 * a_j * derphi0 or phi_aj >= phi_lo:
                phi_rec, a_rec = phi_hi, a_hi
                a_hi, phi_hi = a_j, phi_aj
            else:
                derphi_aj = dphi(a_j)
                if abs(derphi_aj) <= -c2 * derphi0:
                    return a_j, phi_aj, derphi_aj
                if derphi_aj * (a_hi - a_lo) >= 0:
                    phi_rec, a_rec = phi_hi, a_hi
                    a_hi, phi_hi = a_lo, phi_lo
                else:
                    phi_rec, a_rec = phi_lo, a_lo
                a_lo, phi_lo, derphi_lo = a_j, phi_aj, derphi_aj
            i += 1
            if i > 10:
                return None, None, None

    for i in range(maxiter):
        if alpha1 == 0 or alpha0 > amax:
            return None, phi0, None
        if phi_a1 > phi0 + c1 * alpha1 * derphi0 or (phi_a1 >= phi_a0 and i > 0):
            return zoom(alpha0, alpha1, phi_a0, phi_a1, derphi_a0)
        derphi_a1 = dphi(alpha1)
        if abs(derphi_a1) <= -c2 * derphi0:
            return alpha1, phi_a1, derphi_a1
        if derphi_a1 >= 0:
            return zoom(alpha1, alpha0, phi_a1, phi_a0, derphi_a1)
        alpha2 = min(2 * alpha1, amax)
        alpha0, alpha1 = alpha1, alpha2
        phi_a0 = phi_a1
        phi_a1 = phi(alpha1)
        derphi_a0 = derphi_a1
    return alpha1, phi_a1, None  # maxiter: alpha accepted, gradient must be recomputed


def bfgs_minimize(fg, x0, gtol=1e-3, maxiter=100):
    """_minimize_bfgs with an analytic gradient: H0 = I, old_old_fval = f0 + |g|_2 / 2,
    stop on |g|_inf <= gtol (checked before the H update), rhok = 1000 when y.s == 0."""
    x = np.array(x0, dtype=float)
    nfev = njev = 0
    f, g = fg(x)
    nfev += 1
    njev += 1
    n = len(x)
    H = np.eye(n)
    old_old = f + np.linalg.norm(g) / 2
    k = 0
    status = 0
    alphas = []
    gnorm = np.max(np.abs(g))
    while gnorm > gtol and k < maxiter:
        pk = -H @ g
        derphi0 = g @ pk
        cache = {}

        def both(s):
            nonlocal nfev, njev
            fv, gv = fg(x + s * pk)
            nfev += 1
            njev += 1
            cache["s"], cache["f"], cache["g"] = s, fv, gv
            return fv, gv @ pk

        stp, fnew, _ = _search_wolfe1(both, f, old_old, derphi0)
        gnew = cache.get("g") if stp is not None else None
        if stp is None:
            def phi(s):
                return both(s)[0]

            def dphi(s):
                if cache.get("s") != s:
                    both(s)
                return cache["g"] @ pk

            stp, fnew, dstar = _search_wolfe2(phi, dphi, f, old_old, derphi0)
            if stp is None:
                status = 2
                break
            gnew = cache["g"] if (dstar is not None and cache.get("s") == stp) else None
        old_old, f = f, fnew
        alphas.append(stp)
        sk = stp * pk
        x = x + sk
        if gnew is None:
            _, gnew = fg(x)
            njev += 1
        yk = gnew - g
        g = gnew
        k += 1
        gnorm = np.max(np.abs(g))
        if gnorm <= gtol:
            break
        if not np.isfinite(f):
            status = 2
            break
        rinv = yk @ sk
        rho = 1000.0 if rinv == 0.0 else 1.0 / rinv
        A1 = np.eye(n) - sk[:, None] * yk[None, :] * rho
        A2 = np.eye(n) - yk[:, None] * sk[None, :] * rho
        H = A1 @ (H @ A2) + rho * sk[:, None] * sk[None, :]
    if status == 0 and k >= maxiter:
        status = 1
    elif status == 0 and (np.isnan(gnorm) or np.isnan(f) or np.isnan(x).any()):
        status = 3
    return dict(x=x, fun=f, jac=g, nit=k, nfev=nfev, njev=njev, status=status, alphas=np.array(alphas))


# --------------------------------------------------------------------------------------
# a16/a17: distances                               (reference geometry.py:5-56)
# --------------------------------------------------------------------------------------
def hs_dst(a, b):
    """sqrt(|Tr((A-B)^2)|) / sqrt(2), zero below 1e-15   (geometry.py:16-20)."""
    diff = a - b
    dist = np.sqrt(abs(np.trace(diff @ diff))) / np.sqrt(2)
    return 0 if dist < 1e-15 else dist


def _psd_sqrt(a):
    v, u = la.eigh((a + a.conj().T) / 2)
    return (u * np.sqrt(np.maximum(v, 0))) @ u.conj().T


def infidelity(a, b):
    """1 - (Tr sqrt(sqrt(A) B sqrt(A)))^2   (geometry.py:52), evaluated as the nuclear norm
    1 - (sum of singular values of sqrt(A) sqrt(B))^2 with eigh-based roots.  The reference's
    scipy.linalg.sqrtm is only ~1e-8 accurate on singular matrices, and taking
    square roots of the eigenvalues of sqrt(A) B sqrt(A) turns 1e-17 rounding noise on a zero
    eigenvalue into 3e-9; singular values carry no such amplification (error ~1e-15)."""
    sv = la.svdvals(_psd_sqrt(a) @ _psd_sqrt(b))
    return 1 - np.sum(sv) ** 2


def trace_dst(a, b):
    """|Tr sqrtm((A-B)^2)| / 2, zero below 1e-15   (geometry.py:34-38, scipy.linalg.sqrtm as there)."""
    diff = a - b
    dist = abs(np.trace(la.sqrtm(diff @ diff))) / 2
    return 0 if dist < 1e-15 else dist


def if_dst(a, b):
    """1 - |Tr sqrtm(sqrtm(A) B sqrtm(A))|^2, zero below 1e-15   (geometry.py:52-56).  The reference's own
    expression, scipy.linalg.sqrtm included: on singular arguments sqrtm is only ~1e-8 accurate, which is why the
    parity harness measures fidelity with `infidelity` above; this function pins what `if_dst` itself returns."""
    root = la.sqrtm(a)
    dist = 1 - np.abs(np.trace(la.sqrtm(root @ b @ root)) ** 2)
    return 0 if dist < 1e-15 else dist


def warm_start_stack(povm_old, results_old, povm_new, results_new):
    """experiment(..., warm_start=True)   (state.py:116-124): the POVM tensors are stacked with weights
    total-old-shots : total-new-shots, the counts are stacked; n_measurements follows from the results setter
    (state.py:138-141)."""
    n_old, n_new = np.sum(results_old.sum(-1)), np.sum(results_new.sum(-1))
    povm = np.vstack((povm_old * n_old, povm_new * n_new)) / (n_old + n_new)
    return povm, np.vstack((results_old, results_new))


# --------------------------------------------------------------------------------------
# a16: bootstrap                                    (reference interval.py:583-612)
# --------------------------------------------------------------------------------------
def bootstrap_state(counts, povm_matrix, n_points, method="lin", centre=None, **kw):
    """Serial resampling loop on the global legacy RNG; returns (sorted distances, centre,
    per-resample counts).  cl_to_dist = interp over linspace(0,1,n)  (interval.py:610-612)."""
    counts = np.asarray(counts)
    est = (lambda c: lin_estimate(c, povm_matrix)) if method == "lin" else (
        lambda c: mle_estimate(c, povm_matrix, **kw))
    if centre is None:
        centre = est(counts)
    n_meas = counts.sum(-1)
    bloch_c = bloch_from_matrix(centre)
    dist = np.empty(n_points)
    all_counts = []
    for i in range(n_points):
        c = sample_counts(povm_matrix, bloch_c, n_meas)
        all_counts.append(c)
        dist[i] = hs_dst(est(c), centre)
    order = np.sort(dist)
    return order, centre, np.stack(all_counts), dist


def quantiles(sorted_dist, conf_levels):
    """scipy interp1d(linspace(0,1,n), dist) is linear interpolation (interval.py:611-612)."""
    grid = np.linspace(0, 1, len(sorted_dist))
    return np.interp(conf_levels, grid, sorted_dist)


# --------------------------------------------------------------------------------------
# a11-a15: process tomography                       (reference process.py, basis.py, routines.py)
# --------------------------------------------------------------------------------------
def mat2vec(m):
    """column stacking  (routines.py:59-61)."""
    return m.T.reshape(-1)


def vec2mat(v):
    """routines.py:53-56."""
    s = int(math.isqrt(len(v)))
    return v.reshape(s, s).T


def out_ptrace_oper(n):
    """(D, D^2) partial trace over the OUTPUT half of a bipartite column-stacked vector
    (routines.py:47-50)."""
    eye = np.eye(2**n)
    return np.sum([np.kron(eye, np.kron(k, np.kron(eye, k))) for k in eye], axis=0)


def input_states(name_or_list, n):
    """process.py:330-339: rows of the named POVM as Bloch vectors, trace-normalised."""
    if isinstance(name_or_list, (list, tuple)):
        return [np.asarray(m) for m in name_or_list]
    rows = np.squeeze(measurement_matrix(name_or_list, n))
    mats = [matrix_from_bloch(r) for r in rows]
    return [m / np.trace(m) for m in mats]


def apply_choi(choi, rho, n):
    """Channel.transform through the Choi matrix (channel.py:139-141):
    Tr_in[(rho^T (x) I) C]."""
    d = 2**n
    c4 = (np.kron(rho.T, np.eye(d)) @ choi).reshape(d, d, d, d)
    return np.einsum("iaib->ab", c4)


def choi_from_func(func, n):
    """Channel.choi from a map (channel.py:92-100): sum_ij E_ij (x) func(E_ij)."""
    d = 2**n
    choi = np.zeros((d * d, d * d), dtype=np.complex128)
    for i in range(d):
        for j in range(d):
            e = np.zeros((d, d), dtype=np.complex128)
            e[i, j] = 1
            choi += np.kron(e, func(e))
    return choi


def lifp_operator(in_states, povm_matrix, n_meas):
    """rows vec(rho_in (x) E_m^T) over product(input states, weighted POVM rows)
    (process.py:197-208)."""
    a = weighted_povm(povm_matrix, n_meas)
    e_mats = matrix_from_bloch(a)  # (M, d, d)
    rows = [mat2vec(np.kron(rho, e.T)) for rho in in_states for e in e_mats]
    return np.array(rows)


def lifp_estimate(counts, povm_matrix, in_states, return_oper=False):
    """counts (D, S, K) -> Choi (d^2, d^2), no CPTP projection  (process.py:284-286).
    Frequencies are per-tomograph: counts / counts.sum()  (process.py:285)."""
    counts = np.asarray(counts)
    oper = lifp_operator(in_states, povm_matrix, counts[0].sum(-1))
    inv = left_inv(oper)
    freq = np.hstack([c.flatten() / c.sum() for c in counts])
    choi = vec2mat(inv @ freq)
    return (choi, oper, inv) if return_oper else choi


def decompose_single_entries(in_states):
    """process.py:75-80 with basis.py:20-35: coordinates of the d^2 single-entry matrices E_ij in the input basis,
    conj(solve(Gram, rhs)), Gram_ab = Tr(rho_a rho_b^dagger), rhs_a = Tr(rho_a E^dagger)."""
    dim = len(in_states)
    d = in_states[0].shape[0]
    gram = np.array([[np.trace(a @ b.conj().T) for b in in_states] for a in in_states], dtype=np.complex128)
    rows = []
    for i in range(d):
        for j in range(d):
            e = np.zeros((d, d))
            e[i, j] = 1
            rhs = np.array([np.trace(a @ e.conj().T) for a in in_states], dtype=np.complex128)
            rows.append(np.conj(la.solve(gram, rhs)))
    assert len(rows) == dim
    return np.array(rows)


def choi_is_cptp(choi, n, atol=1e-5):
    """channel.py:144-157: Tr_out C = I and min(eig, 0) = 0 within atol (eigenvalues through scipy.linalg.eig of
    the general matrix, real parts, as Qobj.eig does for a non-Hermitian-checked matrix)."""
    d = 2**n
    reduced = np.einsum("iaja->ij", choi.reshape(d, d, d, d))
    tp = np.allclose(reduced, np.eye(d), atol=atol)
    cp = np.allclose(np.minimum(np.real(la.eigvals(choi)), 0), 0, atol=atol)
    return tp and cp


def states_estimate(counts, povm_matrix, in_states, n, cptp=True):
    """'states' process estimator (process.py:316-327): output states by 'lin', C = sum_e E_e (x) sum_s c_es rho_out_s,
    CPTP projection only when C fails is_cptp."""
    outs = [lin_estimate(c, povm_matrix) for c in counts]
    coeff = decompose_single_entries(in_states)
    d = in_states[0].shape[0]
    choi = np.zeros((d * d, d * d), dtype=np.complex128)
    for row in coeff:
        unit = sum(c * s for c, s in zip(row, in_states))
        image = sum(c * o for c, o in zip(row, outs))
        choi += np.kron(unit, image)
    if cptp and not choi_is_cptp(choi, n):
        choi = cptp_projection(choi, n)
    return choi


def tp_projection_vec(v, n):
    """process.py:259-265."""
    d = 2**n
    p = out_ptrace_oper(n)
    return v + (p.T.conj() @ mat2vec(np.eye(d)) - (p.T.conj() @ p) @ v) / d


def cp_projection_vec(v):
    """process.py:270-277: eigh, clip at 1e-12, rebuild."""
    w, u = la.eigh(vec2mat(v))
    return mat2vec(u @ np.diag(np.maximum(1e-12, w)) @ u.T.conj())


def cptp_projection(choi, n, n_iter=1000, tol=1e-12, return_iters=False):
    """Dykstra-style alternating projection  (process.py:231-257)."""
    x = mat2vec(choi).astype(np.complex128)
    p = q = y = 0
    it = 0
    for it in range(n_iter):
        crit = 0
        y_diff = tp_projection_vec(x + p, n) - y
        y = y + y_diff
        x_diff = cp_projection_vec(y + q) - x
        x = x + x_diff
        crit += 2 * (np.abs(np.sum(y_diff.T.conj() * q)) + np.abs(np.sum(x_diff.T.conj() * p)))
        p_diff = x - y
        p = p + p_diff
        q_diff = y - x
        q = q + q_diff
        crit += la.norm(p_diff) ** 2 + la.norm(q_diff) ** 2
        if crit < tol:
            break
    out = vec2mat(x)
    return (out, it + 1) if return_iters else out


def pgdb_estimate(counts, povm_matrix, in_states, n_iter=1000, tol=1e-10, stop="reference", return_info=False):
    """'pgdb' (process.py:291-308): projected gradient descent with backtracking on the Choi vector,
    started at the fully mixed state, raw counts as weights (`_unnorm_results`, process.py:213).
    The reference's arithmetic is kept: numpy.dot(D, grad) WITHOUT conjugation (:300), log of the complex
    probabilities (:313), ordering of complex numbers as NumPy orders them (real part first).
    stop='reference' is the loop of the reference as written -- `if nll(old) - nll(new) > tol: break`
    (:303-305) leaves at the first step that lowers the NLL by more than tol and the point BEFORE that
    step is returned; stop='converged' accepts the step and stops when the decrease drops below tol."""
    counts = np.asarray(counts)
    n = int(round(math.log2(in_states[0].shape[0])))
    oper = lifp_operator(in_states, povm_matrix, counts[0].sum(-1))
    unnorm = np.hstack([c.flatten() for c in counts]).astype(float)
    dim2 = oper.shape[1]

    def nll(v):
        return -np.sum(unnorm * np.log(oper @ v + 1e-12))  # complex, as in the reference

    def gt(a, b):  # NumPy's ordering of complex scalars: lexicographic
        a, b = complex(a), complex(b)
        return (a.real, a.imag) > (b.real, b.imag)

    dch = int(round(math.sqrt(dim2)))
    v = mat2vec(np.eye(dch, dtype=np.complex128) / dch)
    mu, gamma = 1.5 / 4**n, 0.3
    it, trace = 0, []
    for it in range(n_iter):
        probas = oper @ v
        grad = -oper.T.conj() @ (unnorm / probas)
        direction = mat2vec(cptp_projection(vec2mat(v - grad / mu), n)) - v
        alpha = 1.0
        f0 = nll(v)
        dot = np.dot(direction, grad)
        while gt(nll(v + alpha * direction) - f0, gamma * alpha * dot):
            alpha /= 2
        new = v + alpha * direction
        f1 = nll(new)
        trace.append(dict(alpha=alpha, f0=f0, f1=f1, dot=dot))
        if stop == "reference":
            if gt(f0 - f1, tol):
                break
            v = new
        else:
            v = new
            if not gt(f0 - f1, tol):
                it += 1
                break
    else:
        it = n_iter
    choi = vec2mat(v)
    return (choi, dict(iters=it, trace=trace)) if return_info else choi


def mle_constr_estimate(counts, povm_matrix, init="lin", max_iter=100, tol=1e-3, jac="fd", return_info=False):
    """'mle-constr' (state.py:231-253): SLSQP on the Cholesky-parametrised NLL under Tr(L L^dagger) = 1.
    jac='fd' is the reference's call (SciPy differentiates objective and constraint by forward
    differences); jac='analytic' hands SciPy the exact gradients, which is what the HIP-backed host code
    does."""
    prob = NllProblem(counts, povm_matrix)
    x0 = mle_start(counts, povm_matrix, init)

    def unit_trace(x):  # state.py:262-265
        m = tril_vec_to_matrix(x)
        return np.trace(m) - 1

    if jac == "fd":
        import warnings

        with warnings.catch_warnings():  # the reference's constraint returns a complex number (np.trace - 1)
            warnings.simplefilter("ignore")
            res = minimize(prob.nll, x0, constraints=[{"type": "eq", "fun": unit_trace}], method="SLSQP", tol=tol,
                           options={"maxiter": max_iter})
    else:
        cons = [{"type": "eq", "fun": lambda x: float(np.dot(x, x)) - 1.0, "jac": lambda x: 2.0 * x}]
        res = minimize(prob.nll_and_grad, x0, jac=True, constraints=cons, method="SLSQP", tol=tol,
                       options={"maxiter": max_iter})
    m = tril_vec_to_matrix(res.x)
    rho = m / np.trace(m)
    info = dict(nit=res.nit, nfev=res.nfev, status=res.status, x=res.x, fun=res.fun)
    return (rho, info) if return_info else rho


def mhmc_state_chain(counts, povm_matrix, x_init, deltas, uniforms, step):
    """The Metropolis-Hastings chain of mhmc.py:80-119 with `normalized_update` (mhmc.py:116-119) and
    target -nll (interval.py:739): returns (states after every step, accepted flags)."""
    prob = NllProblem(counts, povm_matrix)
    x = np.array(x_init, dtype=float)
    f = prob.nll(x)
    chain = np.empty((len(deltas), len(x)))
    acc = np.zeros(len(deltas), dtype=np.int32)
    for t, (dl, u) in enumerate(zip(deltas, uniforms)):
        xp = x + step * dl
        xp = xp / np.linalg.norm(xp)
        fp = prob.nll(xp)
        if u <= np.exp(f - fp):
            x, f = xp, fp
            acc[t] = 1
        chain[t] = x
    return chain, acc


def mhmc_state_interval(counts, povm_matrix, state_matrix, n_points, step, burn_steps, thinning=1):
    """MHMCStateInterval.setup (interval.py:735-750) with the reference's order of random draws
    (mhmc.py:61-62, 88-89: rvs then rand, burn-in first).  Returns (sorted HS distances, samples, rate)."""
    from scipy.stats import multivariate_normal

    dim = povm_matrix.shape[-1]
    jump = multivariate_normal(mean=np.zeros(dim))
    x0 = matrix_to_tril_vec(state_matrix)
    db = jump.rvs(size=burn_steps).reshape(burn_steps, dim)
    ub = np.random.rand(burn_steps)
    total = n_points * thinning
    ds = jump.rvs(size=total).reshape(total, dim)
    us = np.random.rand(total)
    chain, acc = mhmc_state_chain(counts, povm_matrix, x0, np.concatenate([db, ds]), np.concatenate([ub, us]), step)
    samples = chain[burn_steps::thinning][:n_points]
    dist = np.sort([hs_dst(tril_vec_to_matrix(v), state_matrix) for v in samples])
    return dist, samples, float(acc[burn_steps:].mean())


def mhmc_process_interval(counts, povm_matrix, in_states, channel_choi, n_points, step, burn_steps, thinning=1):
    """MHMCProcessInterval.setup (interval.py:808-836): chain on the Choi vector with the CPTP projection as
    update rule (process.py:279-281) and target -nll with the raw counts (process.py:310-314); draws in the
    reference's order.  Returns (sorted HS distances, sample matrices, acceptance rate)."""
    from scipy.stats import multivariate_normal

    counts = np.asarray(counts)
    n = int(round(math.log2(in_states[0].shape[0])))
    oper = lifp_operator(in_states, povm_matrix, counts[0].sum(-1))
    unnorm = np.hstack([c.flatten() for c in counts]).astype(float)

    def logp(v):
        return np.sum(unnorm * np.log(oper @ v + 1e-12))  # = -nll, complex like the reference's

    dim = oper.shape[1]
    jump = multivariate_normal(mean=np.zeros(dim))
    db = jump.rvs(size=burn_steps).reshape(burn_steps, dim)
    ub = np.random.rand(burn_steps)
    total = n_points * thinning
    ds = jump.rvs(size=total).reshape(total, dim)
    us = np.random.rand(total)
    x = mat2vec(np.asarray(channel_choi, dtype=np.complex128))
    states, acc = [], []
    for dl, u in zip(np.concatenate([db, ds]), np.concatenate([ub, us])):
        xp = mat2vec(cptp_projection(vec2mat(x + step * dl), n))
        alpha = np.exp(logp(xp) - logp(x))
        ok = (u, 0.0) <= (alpha.real, alpha.imag)  # NumPy orders complex numbers lexicographically
        if ok:
            x = xp
        states.append(x)
        acc.append(ok)
    # mhmc.py:66 stores the samples in a real array: imaginary parts are dropped
    samples = [vec2mat(v.real) for v in states[burn_steps::thinning][:n_points]]
    dist = np.sort([hs_dst(m, channel_choi) for m in samples])
    return dist, samples, float(np.mean(acc[burn_steps:]))


# --------------------------------------------------------------------------------------
# f2: moments of the squared l2 error under multinomial noise      (reference stats.py:21-47, interval.py:59-110)
# --------------------------------------------------------------------------------------
# The six-index contractions of stats.py, term by term (sign, subscripts over (weights, weights, freq...)): the first
# moment's two terms (stats.py:25) and the second moment's twelve (stats.py:35-46).
_L2_FIRST = ((+1, "aiai,ai->"), (-1, "aiaj,ai,aj->"))
_L2_SECOND = ((+1, "aiaj,bkbl,ai,aj,bk,bl->"), (-1, "aiaj,bkbk,ai,aj,bk->"), (-1, "aiai,bkbl,ai,bk,bl->"), (+1, "aiai,bkbk,ai,bk->"),
              (+1, "aibj,bkal,ai,bj,bk,al->"), (-1, "aibj,bjal,ai,bj,al->"), (-1, "aibj,bkai,ai,bj,bk->"), (+1, "aibj,bjai,ai,bj->"),
              (+1, "aibj,akbl,ai,bj,ak,bl->"), (-1, "aibj,akbj,ai,bj,ak->"), (-1, "aibj,aibl,ai,bj,bl->"), (+1, "aibj,aibj,ai,bj->"))


def _l2_terms(terms, freq, weights):
    total = 0.0
    for sign, subs in terms:
        n_w = 1 if len(subs.split("->")[0].split(",")[1]) == 2 else 2  # one or two weight tensors lead the operand list
        n_f = len(subs.split("->")[0].split(",")) - n_w
        total += sign * np.einsum(subs, *([weights] * n_w + [freq] * n_f), optimize=True)
    return total


def l2_moments(freq, n_trials, inv_matrix):
    """(mean, variance) of ||P (f - p)||^2 for multinomial frequencies `freq` (S, K), `n_trials` shots per setting and
    P = `inv_matrix` (rows, S*K): stats.py:5-47 (l2_mean, l2_variance) with the weights tensor of interval.py:88,
    einsum('aij,akl->ijkl', P, P)."""
    s, k = freq.shape
    p3 = np.asarray(inv_matrix).reshape(-1, s, k)
    weights = np.einsum("aij,akl->ijkl", p3, p3)
    first = _l2_terms(_L2_FIRST, freq, weights) / n_trials
    second = _l2_terms(_L2_SECOND, freq, weights) / n_trials**2
    return first, second - first**2


def moment_radii(counts, povm_matrix, conf_levels, distr_type="gamma"):
    """MomentInterval for state tomography with the Hilbert-Schmidt distance (interval.py:72-76, 88-110)."""
    import scipy.stats as sts

    counts = np.asarray(counts)
    n_meas = counts.sum(-1).astype(float)
    dim = int(round(math.sqrt(povm_matrix.shape[-1])))
    inv = left_inv(povm_matrix.reshape(-1, povm_matrix.shape[-1])) / dim
    mean, var = l2_moments(counts / n_meas[:, None], n_meas[0], inv)
    if distr_type == "gamma":
        scale = var / mean
        distr = sts.gamma(a=mean / scale, scale=scale)
    elif distr_type == "norm":
        distr = sts.norm(loc=mean, scale=np.sqrt(var))
    else:
        distr = sts.expon(scale=mean)
    return np.sqrt(distr.ppf(conf_levels)) * np.sqrt(dim / 2)

"""CPU: the scalar line-search state machine (quantpy_amd/csrc/qt_linesearch.h, host build)
against SciPy's own scalar_search_wolfe1 -> scalar_search_wolfe2 chain, i.e. what
_line_search_wolfe12 runs inside scipy BFGS (reference call site state.py:213), on 1-D
functions chosen to hit every branch: clean convergence, bracketing, extrapolation, noisy
functions on which dcsrch gives up (fallback search), and outright failures."""
import ctypes
import os
import subprocess
import warnings

import numpy as np
import pytest
from scipy.optimize._linesearch import scalar_search_wolfe1, scalar_search_wolfe2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_ls(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ls") / "libls_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", out,
                           os.path.join(ROOT, "tests", "host", "linesearch_host.cpp")])
    lib = ctypes.CDLL(out)
    cb_t = ctypes.CFUNCTYPE(None, ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))
    lib.qt_host_line_search.argtypes = [cb_t, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                        ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.qt_host_line_search.restype = ctypes.c_int

    def run(phi, dphi, phi0, old_phi0, derphi0):
        def cb(a, pf, pg):
            pf[0] = phi(a)
            pg[0] = dphi(a)

        stp, f, n, mode = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
        ok = lib.qt_host_line_search(cb_t(cb), phi0, old_phi0, derphi0, ctypes.byref(stp), ctypes.byref(f),
                                     ctypes.byref(n), ctypes.byref(mode))
        return (stp.value if ok else None), f.value, n.value, mode.value

    return run


def scipy_wolfe12(phi, dphi, phi0, old_phi0, derphi0):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        stp, phi1, _ = scalar_search_wolfe1(phi, dphi, phi0, old_phi0, derphi0, c1=1e-4, c2=0.9,
                                            amax=1e100, amin=1e-100, xtol=1e-14)
        used = 0
        if stp is None:
            used = 1
            stp, phi1, _, dstar = scalar_search_wolfe2(phi, dphi, phi0, old_phi0, derphi0, 1e-4, 0.9, 1e100)
    return stp, phi1, used


def _functions(rng, n):
    out = []
    for i in range(n):
        kind = i % 9
        a, b, c, e = rng.uniform(0.1, 5), rng.uniform(-3, 3), rng.uniform(0.01, 30), rng.uniform(1e-9, 1e-3)
        if kind == 0:  # convex quadratic, minimum anywhere from 1e-3 to 1e3 steps away
            m = 10 ** rng.uniform(-3, 3)
            out.append((lambda t, a=a, m=m: a * (t - m) ** 2, lambda t, a=a, m=m: 2 * a * (t - m)))
        elif kind == 1:  # quartic with a far minimum: extrapolation phase
            out.append((lambda t, c=c: (t - c) ** 4 - 3 * (t - c), lambda t, c=c: 4 * (t - c) ** 3 - 3))
        elif kind == 2:  # log-barrier like the NLL: blows up at a finite step
            out.append((lambda t, c=c, a=a: -np.log(max(c - t, 1e-300)) - a * t,
                        lambda t, c=c, a=a: 1.0 / max(c - t, 1e-300) - a))
        elif kind == 3:  # oscillating
            out.append((lambda t, a=a, b=b: np.sin(a * t + b) + 0.05 * t * t - 2 * t,
                        lambda t, a=a, b=b: a * np.cos(a * t + b) + 0.1 * t - 2))
        elif kind == 4:  # inconsistent (noisy) derivative: forces warnings / the fallback search
            out.append((lambda t, a=a: a * (t - 1) ** 2 + 1e-3 * np.sin(1e4 * t),
                        lambda t, a=a, e=e: 2 * a * (t - 1) + 0.3 * np.cos(37 * t) * (1 + e)))
        elif kind == 5:  # step discontinuity inside the bracket: dcsrch stops on rounding/xtol warnings
            out.append((lambda t, a=a, c=c: a * (t - 1) ** 2 + 0.5 * (t > 0.3 + 0.01 * c),
                        lambda t, a=a: 2 * a * (t - 1)))
        elif kind == 6:  # derivative that never turns positive although phi rises: no Wolfe point
            out.append((lambda t, a=a: a * t * t - t, lambda t, e=e: -1.0 - e))
        elif kind == 7:  # value noise of the size of the decrease: sufficient-decrease test flickers
            out.append((lambda t, e=e, b=b: -e * t + e * np.sin(977.0 * t + b) + t ** 4,
                        lambda t, e=e: -e + 4 * t ** 3))
        else:  # nearly flat
            out.append((lambda t, e=e: -e * t + e * e * t * t, lambda t, e=e: -e + 2 * e * e * t))
    return out


def test_state_machine_matches_scipy_chain(host_ls):
    rng = np.random.default_rng(123)
    n_fallback = n_fail = n_ok = 0
    for phi, dphi in _functions(rng, 600):
        phi0, derphi0 = phi(0.0), dphi(0.0)
        if not derphi0 < 0:
            phi_, dphi_ = phi, dphi
            phi, dphi = (lambda t, p=phi_: p(-t)), (lambda t, d=dphi_: -d(-t))
            phi0, derphi0 = phi(0.0), dphi(0.0)
        for old in (phi0 + abs(derphi0) / 2, phi0 + rng.uniform(1e-6, 10), phi0 - 1.0):
            ref_stp, ref_phi, used = scipy_wolfe12(phi, dphi, phi0, old, derphi0)
            stp, f, n, mode = host_ls(phi, dphi, phi0, old, derphi0)
            n_fallback += used
            if ref_stp is None:
                n_fail += 1
                assert stp is None
            else:
                n_ok += 1
                assert stp is not None
                assert stp == pytest.approx(ref_stp, rel=1e-12, abs=0), (stp, ref_stp)
                assert f == pytest.approx(ref_phi, rel=1e-12, abs=1e-300)
                assert (mode != 0) == bool(used)
    # the sample must actually exercise the fallback and the failure exits
    assert n_ok > 1000 and n_fallback > 100 and n_fail > 20, (n_ok, n_fallback, n_fail)


def test_non_descent_direction_goes_to_fallback_and_fails(host_ls):
    phi, dphi = (lambda t: (t + 1) ** 2), (lambda t: 2 * (t + 1))
    ref = scipy_wolfe12(phi, dphi, phi(0.0), phi(0.0) + 1, dphi(0.0))
    got = host_ls(phi, dphi, phi(0.0), phi(0.0) + 1, dphi(0.0))
    assert ref[0] is None and got[0] is None

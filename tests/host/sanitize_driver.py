"""Run under LD_PRELOAD=libasan by tests/test_host_sanitizers.py: the sanitizer-instrumented host builds of
qt_sampler.h and qt_linesearch.h over the bit-exactness cases of the CPU suite (numpy.random / scipy are the checkers).
argv: <libsampler_san.so> <libls_san.so>.  Prints 'sanitized ok ...' when every case agreed."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sampler_cases(lib):
    lib.qt_host_legacy_multinomial.restype = ctypes.c_int
    lib.qt_host_legacy_multinomial.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_longlong, ctypes.c_int,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(1)
    checked = 0
    for trial in range(300):  # the sweep of test_c_sampler_is_numpys_legacy_multinomial_bit_for_bit
        n_set, n_out = int(rng.integers(1, 6)), int(rng.integers(1, 9))
        kind = trial % 5
        p = rng.random((n_set, n_out)) ** (1 + 3 * (kind == 1))
        if kind == 2:
            p[rng.random((n_set, n_out)) < 0.4] = 0
        p[:, 0] += 1e-9
        p /= p.sum(1, keepdims=True)
        if kind == 4 and n_out > 1:
            p[0] = 0
            p[0, int(rng.integers(0, n_out))] = 1.0
        n = rng.integers(0, [10, 100, 10**4, 10**6, 10**9][trial % 5], n_set).astype(np.int64)
        repeats = int(rng.integers(1, 30))
        np.random.seed(int(rng.integers(0, 2**31)))
        np.random.rand(int(rng.integers(0, 700)))
        start = np.random.get_state()
        want = np.asarray([[np.random.multinomial(int(n_s), p_s) for p_s, n_s in zip(p, n)] for _ in range(repeats)])
        want_state = np.random.get_state()
        key = np.ascontiguousarray(start[1], dtype=np.uint32).copy()
        pos = ctypes.c_int(int(start[2]))
        out = np.empty((repeats * n_set, n_out), dtype=np.int64)  # exactly sized: an overrun is the sanitizer's to see
        pc = np.ascontiguousarray(p)
        rc = lib.qt_host_legacy_multinomial(key.ctypes.data, ctypes.byref(pos), repeats * n_set, n_set, n.ctypes.data,
                                            pc.ctypes.data, n_out, out.ctypes.data)
        assert rc == 0
        assert np.array_equal(out.reshape(repeats, n_set, n_out), want), trial
        assert pos.value == want_state[2] and np.array_equal(key, want_state[1]), trial
        checked += 1
    # Philox4x32-10 known answers (Random123's kat_vectors)
    lib.qt_host_philox4x32_10.restype = None
    lib.qt_host_philox4x32_10.argtypes = [ctypes.c_void_p] * 3
    for ctr, key, want in (((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
                           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
                           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
                            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))):
        c, k, o = np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32), np.zeros(4, dtype=np.uint32)
        lib.qt_host_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
        assert tuple(int(x) for x in o) == want
    return checked


def linesearch_cases(lib):
    import test_linesearch_host as tl

    cb_t = ctypes.CFUNCTYPE(None, ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))
    lib.qt_host_line_search.argtypes = [cb_t, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.qt_host_line_search.restype = ctypes.c_int
    rng = np.random.default_rng(123)
    n_ok = n_fail = n_fallback = 0
    for phi, dphi in tl._functions(rng, 180):
        phi0, derphi0 = phi(0.0), dphi(0.0)
        if not derphi0 < 0:
            phi_, dphi_ = phi, dphi
            phi, dphi = (lambda t, p=phi_: p(-t)), (lambda t, d=dphi_: -d(-t))
            phi0, derphi0 = phi(0.0), dphi(0.0)
        for old in (phi0 + abs(derphi0) / 2, phi0 - 1.0):
            ref_stp, ref_phi, used = tl.scipy_wolfe12(phi, dphi, phi0, old, derphi0)

            def cb(a, pf, pg):
                pf[0] = phi(a)
                pg[0] = dphi(a)

            stp, f, n, mode = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
            ok = lib.qt_host_line_search(cb_t(cb), phi0, old, derphi0, ctypes.byref(stp), ctypes.byref(f), ctypes.byref(n),
                                         ctypes.byref(mode))
            n_fallback += used
            if ref_stp is None:
                n_fail += 1
                assert not ok
            else:
                n_ok += 1
                assert ok and abs(stp.value - ref_stp) <= 1e-12 * abs(ref_stp)
    assert n_ok > 250 and n_fallback > 20 and n_fail > 5, (n_ok, n_fallback, n_fail)
    return n_ok + n_fail


if __name__ == "__main__":
    a = sampler_cases(ctypes.CDLL(sys.argv[1]))
    b = linesearch_cases(ctypes.CDLL(sys.argv[2]))
    print(f"sanitized ok: {a} sampler sweeps, {b} line searches", flush=True)

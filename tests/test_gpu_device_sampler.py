"""The opt-in device sampler (`qt_device_multinomial`, csrc/qt_sampler.h `k_multinomial_rows`): it promises the
DISTRIBUTION of the reference's `np.random.multinomial(n_s, p_s)` draws (state.py:109-114), not their stream -- so
the tests are distributional (chi-square of binomial marginals against scipy's pmf in both sampler regimes, first
and second moments of a multinomial), structural (rows sum to n, zeros stay zero) and about the counter-based
streams (a fixed seed reproduces, any split of the rows over calls gives the same table).  Parity with the
reference's counts stays with the default sampler (tests/test_host_logic.py, test_gpu_api.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import quantpy_amd as qp

    return qp.get_engine(1)


@pytest.mark.parametrize("n,p", [(40, 0.05), (25, 0.5), (1000, 0.3), (1000, 0.93), (10**6, 0.5), (10**6, 2e-5), (7, 0.999),
                                 (400, 0.1), (2000, 0.02), (150, 0.4), (61, 0.5), (100000, 0.125)])
def test_binomial_marginal_chi_square(eng, n, p):
    """K = 2 rows are single binomials: inversion (n p <= 30), BTPE (above), and the p > 0.5 reflection."""
    from scipy import stats

    rows = 40000
    c = eng.device_multinomial([n], [[p, 1 - p]], rows, seed=1234 + n)
    assert np.all(c.sum(1) == n) and c.min() >= 0
    x = c[:, 0]
    lo, hi = int(stats.binom.ppf(1e-4, n, p)), int(stats.binom.ppf(1 - 1e-4, n, p))
    edges = np.unique(np.linspace(lo, hi + 1, min(hi - lo + 2, 40)).astype(int))
    cdf = stats.binom.cdf(edges - 1, n, p)
    expected = np.diff(np.concatenate([[0.0], cdf, [1.0]])) * rows
    observed = np.histogram(x, bins=np.concatenate([[-0.5], edges - 0.5, [n + 0.5]]))[0]
    keep = expected >= 5
    observed = np.concatenate([observed[keep], [observed[~keep].sum()]])
    expected = np.concatenate([expected[keep], [expected[~keep].sum()]])
    if expected[-1] < 5:
        observed[-2] += observed[-1]
        expected[-2] += expected[-1]
        observed, expected = observed[:-1], expected[:-1]
    chi2 = ((observed - expected) ** 2 / expected).sum()
    pval = stats.chi2.sf(chi2, len(expected) - 1)
    assert pval > 1e-5, (n, p, chi2, len(expected), pval)
    assert abs(x.mean() - n * p) < 5 * np.sqrt(n * p * (1 - p) / rows) + 1e-12


def test_multinomial_moments_and_structure(eng):
    rng = np.random.default_rng(3)
    p = rng.random((27, 8))
    p[5, 2] = 0.0  # an impossible outcome stays empty
    p[6] = [0, 0, 1, 0, 0, 0, 0, 0]  # a certain one takes every shot
    p /= p.sum(1, keepdims=True)
    n = rng.integers(1, 5000, 27)
    n[7] = 0
    reps = 4000
    c = eng.device_multinomial(n, p, reps * 27, seed=99).reshape(reps, 27, 8)
    assert np.all(c.sum(-1) == n[None, :]) and c.min() >= 0
    assert np.all(c[:, 5, 2] == 0) and np.all(c[:, 6, 2] == n[6]) and np.all(c[:, 7] == 0)
    mean = c.mean(0)
    sigma = np.sqrt(n[:, None] * p * (1 - p) / reps)
    assert np.all(np.abs(mean - n[:, None] * p) <= 5 * sigma + 1e-12)
    for s in (0, 11, 26):  # covariance n (diag p - p p^T), entry by entry within 6 standard errors (4th-moment bound)
        cov = np.cov(c[:, s].T.astype(float))
        want = n[s] * (np.diag(p[s]) - np.outer(p[s], p[s]))
        scale = n[s] * np.sqrt(np.outer(p[s], p[s]) + np.diag(p[s])) + 1.0
        assert np.all(np.abs(cov - want) < 6 * scale * np.sqrt(2.0 / reps) + 1e-9), s


def test_streams_are_counter_based(eng):
    """Row r depends on (seed, first_row + r) only: one call, or the same rows in three pieces, or written straight to
    device memory -- one table.  Another seed: another table."""
    import torch

    rng = np.random.default_rng(4)
    p = rng.random((9, 4))
    p /= p.sum(1, keepdims=True)
    n = np.full(9, 1000)
    whole = eng.device_multinomial(n, p, 900, seed=5)
    assert np.array_equal(whole, eng.device_multinomial(n, p, 900, seed=5))
    parts = [eng.device_multinomial(n, p, b - a, seed=5, first_row=a) for a, b in ((0, 100), (100, 105), (105, 900))]
    assert np.array_equal(whole, np.concatenate(parts))
    assert not np.array_equal(whole, eng.device_multinomial(n, p, 900, seed=6))
    out = torch.empty((900, 4), dtype=torch.int64, device="cuda")
    eng.device_multinomial(n, p, 900, seed=5, out=out)
    eng.sync()
    assert np.array_equal(out.cpu().numpy(), whole)
    # neighbouring rows of one setting are different draws
    assert len({tuple(r) for r in whole[::9]}) > 90


def test_argument_errors_match_numpy(eng):
    import quantpy_amd.sampling as sampling

    with pytest.raises(ValueError, match="pvals"):
        sampling.device_multinomial([10], [[0.5, 0.7, 0.1]], 1, seed=1)
    with pytest.raises(ValueError, match="pvals"):
        sampling.device_multinomial([10], [[np.nan, 0.5]], 1, seed=1)
    with pytest.raises(ValueError):
        sampling.draw_counts([10], [[0.5, 0.5]], 1, sampler="numpy", seed=3)
    with pytest.raises(ValueError, match="sampler"):
        sampling.draw_counts([10], [[0.5, 0.5]], 1, sampler="cuda")
    assert sampling.device_multinomial([10, 10], [[0.5, 0.5], [0.1, 0.9]], 0, seed=1).shape == (0, 2, 2)


def test_tomograph_and_bootstrap_with_device_sampler():
    """experiment(sampler='device') feeds the same estimators; a bootstrap CI from device draws agrees with the one
    from the reference's stream to sampling error, is reproducible from its seed, and -- with an explicit seed --
    leaves np.random's stream untouched."""
    import quantpy_amd as qp

    rng = np.random.default_rng(8)
    g = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
    rho = g @ g.conj().T
    rho /= np.trace(rho)
    t = qp.StateTomograph(qp.Qobj(rho))
    np.random.seed(10)
    before = np.random.get_state()[1].copy(), np.random.get_state()[2]
    t.experiment(10000, "proj-set", sampler="device", seed=77)
    after = np.random.get_state()[1], np.random.get_state()[2]
    assert np.array_equal(before[0], after[0]) and before[1] == after[1]
    assert t.results.shape == (27, 8) and np.all(t.results.sum(1) == 10000)
    est = t.point_estimate("mle")
    assert qp.hs_dst(est, qp.Qobj(rho)) < 0.02
    first = t.results.copy()
    t.experiment(10000, "proj-set", sampler="device", seed=77)
    assert np.array_equal(t.results, first)
    np.random.seed(11)
    t.experiment(10000, "proj-set", sampler="device")  # seed from np.random: reproducible through np.random.seed
    a = t.results.copy()
    np.random.seed(11)
    t.experiment(10000, "proj-set", sampler="device")
    assert np.array_equal(t.results, a)

    np.random.seed(12)
    t.experiment(10000, "proj-set")
    t.point_estimate("lin")
    levels = np.array([0.5, 0.9, 0.99])
    ref = qp.BootstrapStateInterval(t, n_points=4000, method="lin")
    d_ref, _ = ref(levels)
    dev = qp.BootstrapStateInterval(t, n_points=4000, method="lin", sampler="device", seed=5)
    d_dev, _ = dev(levels)
    assert np.all(np.abs(d_dev / d_ref - 1) < 0.06), (d_dev, d_ref)
    again = qp.BootstrapStateInterval(t, n_points=4000, method="lin", sampler="device", seed=5)
    assert np.array_equal(again(levels)[0], d_dev)

    ch = qp.channel.depolarizing(0.1, 1) if hasattr(qp.channel, "depolarizing") else None
    if ch is not None:
        pt = qp.ProcessTomograph(ch)
        pt.experiment(2000, "proj-set", sampler="device", seed=3)
        assert np.all(pt.results.sum(-1) == 2000)
        pt.point_estimate("lifp")
        ci = qp.BootstrapProcessInterval(pt, n_points=200, sampler="device", seed=4)
        d, _ = ci(levels)
        assert np.all(np.isfinite(d)) and np.all(np.diff(d) >= 0)


def test_device_rows_equal_the_host_instantiation_on_the_same_philox_streams(eng, tmp_path):
    """The device sampler has no reference parity by construction (its streams are not NumPy's).  What CAN be pinned: the
    kernel runs the very template the host runs -- `philox_multinomial_row` over `legacy_binomial` -- and the host
    instantiation of those binomial routines on NumPy's MT19937 words IS NumPy's sampler bit for bit
    (tests/test_host_logic.py).  Same (seed, row) streams on both sides: the counts must be identical, in the inversion
    regime, in BTPE, across the p > 0.5 reflection, with structural zeros and for ragged shot vectors.  (libm and the
    device's log / exp / sqrt differ in the last place at most: a draw flips only if a uniform falls within ~1e-16 of a
    bound.)"""
    import ctypes
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libsampler_host.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                           os.path.join(root, "tests", "host", "sampler_host.cpp")])
    host = ctypes.CDLL(so)
    host.qt_host_philox_multinomial.restype = None
    host.qt_host_philox_multinomial.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(17)
    total = 0
    for case in range(12):
        period, k = int(rng.integers(1, 28)), int(rng.integers(2, 9))
        p = rng.random((period, k)) ** (1 + 2 * (case % 3))
        if case % 4 == 1:
            p[rng.random((period, k)) < 0.3] = 0.0
            p[:, 0] += 1e-6
        if case % 4 == 2:
            p[:, 0] += 5.0  # the first conditional probability > 0.5: reflection branch
        p /= p.sum(1, keepdims=True)
        n = rng.integers(1, [30, 2000, 10**5, 10**7][case % 4], period).astype(np.int64)
        rows = int(rng.integers(1, 40)) * period + int(rng.integers(0, period))
        seed, first = int(rng.integers(0, 2**62)), int(rng.integers(0, 10**6))
        got = eng.device_multinomial(n, p, rows, seed, first_row=first)
        want = np.empty_like(got)
        pc = np.ascontiguousarray(p)
        host.qt_host_philox_multinomial(seed, first, rows, period, n.ctypes.data, pc.ctypes.data, k, want.ctypes.data)
        assert np.array_equal(got, want), (case, int((got != want).any(1).sum()), rows)
        total += rows
    # the bootstrap's own shape: 27 settings x 8 outcomes at 1e5 shots, 2000 resamples
    p = rng.random((27, 8))
    p /= p.sum(1, keepdims=True)
    n = np.full(27, 100000, dtype=np.int64)
    got = eng.device_multinomial(n, p, 54000, 4242)
    want = np.empty_like(got)
    host.qt_host_philox_multinomial(4242, 0, 54000, 27, n.ctypes.data, p.ctypes.data, 8, want.ctypes.data)
    assert np.array_equal(got, want)
    assert total > 500

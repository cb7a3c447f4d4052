"""GPU parity of a7 (`_make_feasible`, state.py:267-273) on inputs built to reach every branch of the
n = 3 clip: positive definite (no eigensolver), exactly one negative eigenvalue (certified projector
short cut), one negative with a slow / wrong-sided separation ratio (fallback), several negative ones
(Jacobi), an early Cholesky breakdown index, a tiny negative eigenvalue, and a degenerate pair.
The linear-inversion matrix is steered by the counts: counts = round(p * 1e13) with p the (possibly
negative) Born values of the target -- linear inversion is linear, signs included."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_SHOTS = 1e13


def _target(rng, spectrum, first_vector=None):
    d = len(spectrum)
    g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    if first_vector is not None:
        g[:, 0] = first_vector
    q, _ = np.linalg.qr(g)
    lam = np.asarray(spectrum, dtype=float)
    lam = lam / lam.sum()
    return (q * lam) @ q.conj().T


def _counts_for(oracle, povm, rho):
    bloch = oracle.bloch_from_matrix(rho)
    d = rho.shape[0]
    p = np.einsum("skd,d->sk", povm, bloch) * d
    c = np.rint(p * N_SHOTS).astype(np.int64)
    n_s = int(round(N_SHOTS * p[0].sum()))  # every setting sums to the same total, exactly
    for s in range(c.shape[0]):
        c[s, np.argmax(c[s])] += n_s - c[s].sum()
    return c


CASES = {
    "positive definite": ([0.01, 0.03, 0.06, 0.1, 0.15, 0.2, 0.2, 0.25], None),
    "one negative, well separated": ([-0.002, 0.012, 0.04, 0.09, 0.14, 0.17, 0.24, 0.31], None),
    "one negative, tiny": ([-1e-9, 0.012, 0.04, 0.09, 0.14, 0.17, 0.24, 0.31], None),
    "one negative, ratio 0.5": ([-0.004, 0.008, 0.04, 0.09, 0.14, 0.17, 0.24, 0.31], None),
    "one negative, lam2 < |lam1| (fallback)": ([-0.02, 0.005, 0.04, 0.09, 0.14, 0.17, 0.24, 0.31], None),
    "one negative, lam2 = |lam1| (tie)": ([-0.01, 0.01, 0.04, 0.09, 0.14, 0.17, 0.24, 0.31], None),
    "one negative, breakdown at k = 0": ([-0.05, 0.012, 0.04, 0.09, 0.14, 0.17, 0.24, 0.31],
                                         np.eye(8)[0]),
    "one negative, breakdown early": ([-0.003, 0.012, 0.04, 0.09, 0.14, 0.17, 0.24, 0.31],
                                      np.array([0.7, 0.7, 0.1, 0, 0, 0, 0, 0], dtype=complex)),
    "two negative": ([-0.002, -0.001, 0.04, 0.09, 0.14, 0.17, 0.24, 0.31], None),
    "two negative, degenerate": ([-0.002, -0.002, 0.04, 0.09, 0.14, 0.17, 0.24, 0.31], None),
    "rank one + noise": ([-0.003, -0.002, -0.001, 0.0005, 0.001, 0.002, 0.003, 1.0], None),
    "degenerate positive pair above": ([-0.002, 0.02, 0.02, 0.09, 0.14, 0.17, 0.24, 0.31], None),
}


@pytest.mark.parametrize("povm_name", ["proj-set", "sic"])
def test_clip_branches_against_oracle(oracle, povm_name):
    from quantpy_amd import get_engine

    n = 3
    povm = oracle.measurement_matrix(povm_name, n)
    rng = np.random.default_rng(99)
    rhos, names = [], []
    for name, (spec, v0) in CASES.items():
        for _ in range(3):
            rhos.append(_target(rng, spec, v0))
            names.append(name)
    counts = np.stack([_counts_for(oracle, povm, r) for r in rhos])
    eng = get_engine(n)
    shots = counts[0].sum(-1).astype(float)
    assert np.all(counts.sum(-1) == counts[0].sum(-1)[None, :])
    eng.set_povm(povm, shots)
    got_raw = eng.lin(counts, physical=False)
    got = eng.lin(counts, physical=True)
    for b, name in enumerate(names):
        want_raw = oracle.lin_estimate(counts[b], povm, physical=False)
        assert np.abs(got_raw[b] - want_raw).max() < 1e-12, name
        assert np.abs(want_raw - rhos[b]).max() < 1e-10, name  # the steering worked
        want = oracle.make_feasible(want_raw)
        assert np.abs(got[b] - want).max() < 2e-13, (name, np.abs(got[b] - want).max())
        w = np.linalg.eigvalsh(got[b])
        assert w.min() > -1e-16 and abs(np.trace(got[b]).real - 1) < 1e-13, name


def test_mle_from_clipped_start_matches_oracle(oracle):
    """'mle' started from the clipped linear-inversion estimate (state.py:204-215) on the same inputs."""
    from quantpy_amd import get_engine

    n = 3
    povm = oracle.measurement_matrix("proj-set", n)
    rng = np.random.default_rng(5)
    names = ["one negative, well separated", "one negative, lam2 < |lam1| (fallback)", "two negative",
             "positive definite"]
    rhos = [_target(rng, *CASES[k]) for k in names]
    # physical counts this time: sample them, so that the likelihood is a likelihood
    np.random.seed(17)
    counts = []
    for r in rhos:
        w, v = np.linalg.eigh(r)
        phys = (v * np.maximum(w, 1e-4)) @ v.conj().T
        phys /= np.trace(phys).real
        counts.append(oracle.sample_counts(povm, oracle.bloch_from_matrix(phys), np.ones(povm.shape[0]) * 3000))
    counts = np.stack(counts)
    eng = get_engine(n)
    eng.set_povm(povm, counts[0].sum(-1).astype(float))
    rho, info = eng.mle(counts, return_info=True)
    for b in range(len(counts)):
        want = oracle.mle_estimate(counts[b], povm, jac="analytic", solver="port", return_info=True)
        assert info["status"][b] == 0
        assert info["nit"][b] == want[1]["nit"], (names[b], info["nit"][b], want[1]["nit"])
        assert oracle.infidelity(rho[b], want[0]) < 1e-9, names[b]

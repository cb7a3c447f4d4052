"""GPU: the factorised (product-POVM) contraction path -- qt_set_povm_product -- against the dense
path, the oracle and the reference's golden vectors.  Every built-in POVM and every array whose last
axis is 4 takes this path through the drop-in API; plain (S, K, 4^n) arrays take the dense one."""
import numpy as np
import pytest
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qp():
    import quantpy_amd

    return quantpy_amd


def test_product_and_dense_paths_agree_on_lin_nll_mle(qp, oracle):
    rng = np.random.default_rng(17)
    for n in (1, 2, 3):
        d = 2**n
        eng = qp.get_engine(n)
        for name in ("proj-set", "proj", "sic"):
            a_prod = qp.generate_measurement_matrix(name, n)
            a_dense = np.array(a_prod)  # plain ndarray: no factor
            assert a_prod.valid_factor() is not None and not hasattr(a_dense, "valid_factor")
            g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
            rho = g @ g.conj().T
            rho /= np.trace(rho)
            np.random.seed(3)
            shots = 400
            counts = np.stack([oracle.sample_counts(a_dense, oracle.bloch_from_matrix(rho), shots) for _ in range(9)])
            x = np.stack([oracle.matrix_to_tril_vec(rho) + 0.03 * rng.standard_normal(d * d) for _ in range(9)])
            out = {}
            for tag, a in (("dense", a_dense), ("prod", a_prod)):
                eng.set_povm(a, counts[0].sum(-1))
                assert eng.product == (tag == "prod")
                out[tag] = (eng.lin(counts, physical=False, return_bloch=True), eng.lin(counts), eng.nll(x, counts),
                            eng.mle(counts, return_info=True), eng.left_inverse())
            (ru_d, bl_d), lin_d, (f_d, g_d), (mle_d, info_d), pinv_d = out["dense"]
            (ru_p, bl_p), lin_p, (f_p, g_p), (mle_p, info_p), pinv_p = out["prod"]
            assert np.abs(pinv_d - pinv_p).max() < 1e-12
            assert np.abs(bl_d - bl_p).max() < 1e-12 and np.abs(ru_d - ru_p).max() < 1e-12
            assert np.abs(lin_d - lin_p).max() < 1e-11
            assert np.abs(f_d - f_p).max() < 1e-12 and np.abs(g_d - g_p).max() < 1e-10
            for c, xx, f, gr in zip(counts, x, f_p, g_p):
                fo, go = oracle.NllProblem(c, a_dense).nll_and_grad(xx)
                assert abs(f - fo) < 1e-12 and np.abs(gr - go).max() < 1e-10
            same = info_d["nit"] == info_p["nit"]
            assert same.sum() >= len(same) - 1  # a borderline trial may flip an iteration count
            for r1, r2 in zip(mle_d[same], mle_p[same]):
                assert abs(oracle.infidelity(r1, r2)) < 1e-7


def test_product_path_mle_all_golden_trials(qp, oracle):
    g = load_golden("mle")
    for k in range(int(g["n_mle_cases"])):
        key = f"M{k}"
        n = int(g[key + "_n"])
        eng = qp.get_engine(n)
        counts = g[key + "_counts"]
        eng.set_povm(qp.generate_measurement_matrix(str(g[key + "_povm"]), n), counts.sum(-1))
        assert eng.product
        rho, info = eng.mle(counts, init=str(g[key + "_init"]), return_info=True)
        assert info["status"] == 0 and info["nit"] == int(g[key + "_nit"]), (key, info)
        assert info["nfev"] * (4**n + 1) == int(g[key + "_nfev"]), key
        assert abs(oracle.infidelity(g[key + "_rho"], rho)) < 1e-6, key


def test_product_path_unequal_shots_and_custom_table(qp, oracle):
    """per-setting shot vector (weights differ: the left inverse no longer factorises, the NLL still
    does) and a user table with last axis 4."""
    g = load_golden("counts_lin")
    key = "L0"  # n = 1, shots (100, 2000, 30000)
    eng = qp.get_engine(1)
    a = qp.generate_measurement_matrix("proj-set", 1)
    counts = g[key + "_counts"]
    eng.set_povm(a, counts.sum(-1))
    assert eng.product
    assert np.abs(eng.lin(counts) - g[key + "_lin"]).max() < 1e-12
    rng = np.random.default_rng(2)
    # a valid user POVM set: three projective measurements along random (non-orthogonal) directions
    dirs = rng.standard_normal((3, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    table = np.zeros((3, 2, 4))
    table[:, :, 0] = 0.5
    table[:, 0, 1:] = dirs / 2
    table[:, 1, 1:] = -dirs / 2
    for n in (2, 3):
        eng = qp.get_engine(n)
        a = qp.generate_measurement_matrix(table, n)
        assert np.array_equal(np.array(a), oracle.measurement_matrix(table, n))
        shots = (np.arange(a.shape[0]) % 3 + 1) * 500
        d = 2**n
        gm = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        rho = gm @ gm.conj().T
        rho /= np.trace(rho)
        np.random.seed(n)
        counts = np.stack([oracle.sample_counts(np.array(a), oracle.bloch_from_matrix(rho), shots) for _ in range(3)])
        eng.set_povm(a, shots)
        assert eng.product
        x = oracle.matrix_to_tril_vec(rho)
        for c, r, (f, gr) in zip(counts, eng.lin(counts), zip(*eng.nll(np.stack([x] * 3), counts))):
            assert np.abs(r - oracle.lin_estimate(c, np.array(a))).max() < 1e-10
            fo, go = oracle.NllProblem(c, np.array(a)).nll_and_grad(x)
            assert abs(f - fo) < 1e-12 and np.abs(gr - go).max() < 1e-10
        ref, ri = oracle.mle_estimate(counts[0], np.array(a), return_info=True, solver="port")
        rho_g, info = eng.mle(counts[0], return_info=True)
        assert info["nit"] == ri["nit"] and abs(oracle.infidelity(ref, rho_g)) < 1e-6


def test_in_place_edit_of_tensor_drops_the_factor(qp, oracle):
    a = qp.generate_measurement_matrix("proj-set", 2)
    a[0, 0, 0] *= 1.0001  # no longer a tensor power: the checksum must notice
    assert a.valid_factor() is None
    eng = qp.get_engine(2)
    eng.set_povm(a, np.full(9, 100))
    assert not eng.product

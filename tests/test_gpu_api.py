"""GPU: the drop-in Python API (`import quantpy_amd as qp`) driven exactly like the reference
was driven to make the golden vectors (tests/golden/make_golden.py): same seeds, same call
order -> same counts bit for bit, reconstructions within the north_star tolerances."""
import json
import os

import numpy as np
import pytest
from conftest import GOLDEN, load_golden

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


def test_config1_one_qubit_lin(qp, oracle):
    g = load_golden("counts_lin")
    np.random.seed(0)
    t = qp.StateTomograph(qp.qobj.zero(1))
    t.experiment(10000)
    assert t.results.tolist() == [[5002, 4998], [5028, 4972], [10000, 0]]
    assert np.array_equal(t.n_measurements, g["C1_n_meas"])
    rho = t.point_estimate("lin")
    assert isinstance(rho, qp.Qobj) and t.reconstructed_state is rho
    assert np.abs(rho.matrix - g["C1_lin"]).max() < 1e-13
    assert np.abs(t.point_estimate("lin", physical=False).matrix - g["C1_lin_unphys"]).max() < 1e-13
    assert np.abs(t.point_estimate("lin", physical=False).bloch - g["C1_lin_bloch_unphys"]).max() < 1e-13
    assert abs(oracle.infidelity(g["C1_mle"], t.point_estimate("mle").matrix)) < 1e-6


def test_config2_counts_and_estimates(qp, oracle):
    g = load_golden("counts_lin")
    rho_true = ginibre(np.random.default_rng(1234), 8)
    assert np.array_equal(rho_true, g["C2_rho_true"])
    np.random.seed(7)
    t = qp.StateTomograph(qp.Qobj(rho_true))
    for i in range(8):
        t.experiment(100000, "proj-set")
        assert np.array_equal(t.results, g["C2_counts"][i])
        assert np.abs(t.point_estimate("lin", physical=False).matrix - g["C2_lin_unphys"][i]).max() < 1e-12
        assert np.abs(t.point_estimate("lin").matrix - g["C2_lin"][i]).max() < 1e-12
        mle = t.point_estimate("mle")
        assert t.mle_info["nit"] == g["C2_nit"][i]
        assert abs(oracle.infidelity(g["C2_mle"][i], mle.matrix)) < 1e-10
    assert qp.hs_dst(mle, qp.Qobj(rho_true)) == pytest.approx(oracle.hs_dst(g["C2_mle"][7], rho_true), abs=1e-12)


def test_argument_errors_match_reference(qp):
    t = qp.StateTomograph(qp.qobj.zero(1))
    with pytest.raises(TypeError):
        t.experiment(1e5)  # shots must be an integer
    with pytest.raises(ValueError):
        t.experiment([10, 10])
    with pytest.raises(ValueError):
        qp.StateTomograph(qp.qobj.zero(1), dst="nope")
    with pytest.raises(ValueError):
        t.experiment(10, "unknown-povm")
    t.experiment(100)
    with pytest.raises(ValueError):
        t.point_estimate("nope")
    with pytest.raises(ValueError):
        t.point_estimate("mle", init="nope")
    with pytest.raises(ValueError):
        qp.generate_measurement_matrix(np.zeros((3, 5)), 1)
    with pytest.raises(ValueError):
        qp.ProcessTomograph(qp.channel.depolarizing(0.1, 1), input_states=[qp.qobj.zero(1)] * 3)


def test_per_setting_shots_and_custom_povm(qp):
    g = load_golden("counts_lin")
    for k in range(int(g["n_lin_cases"])):
        key = f"L{k}"
        n = int(g[key + "_n"])
        np.random.seed(int(g[key + "_seed"]))
        t = qp.StateTomograph(qp.Qobj(g[key + "_rho_true"]))
        nm = g[key + "_nmeas_arg"]
        t.experiment(int(nm) if nm.ndim == 0 else nm, str(g[key + "_povm"]))
        assert np.array_equal(t.results, g[key + "_counts"]), key
        assert np.array_equal(t.povm_matrix, g[key + "_povm_matrix"])
        assert np.abs(t.point_estimate("lin").matrix - g[key + "_lin"]).max() < 1e-12
        # the same POVM passed as an explicit array, and results injected through the setter
        t2 = qp.StateTomograph(qp.Qobj(g[key + "_rho_true"]))
        t2.povm_matrix = g[key + "_povm_matrix"]
        t2.results = g[key + "_counts"]
        assert np.abs(t2.point_estimate("lin").matrix - g[key + "_lin"]).max() < 1e-12


def test_mle_via_api_all_golden(qp, oracle):
    g = load_golden("mle")
    for k in range(0, int(g["n_mle_cases"]), 3):
        key = f"M{k}"
        np.random.seed(int(g[key + "_seed"]))
        t = qp.StateTomograph(qp.Qobj(g[key + "_rho_true"]))
        t.experiment(int(g[key + "_shots"]), str(g[key + "_povm"]))
        assert np.array_equal(t.results, g[key + "_counts"]), key
        rho = t.point_estimate("mle", init=str(g[key + "_init"]))
        assert t.mle_info["nit"] == int(g[key + "_nit"]), key
        assert abs(oracle.infidelity(g[key + "_rho"], rho.matrix)) < 1e-6, key


def test_bootstrap_state_interval(qp, oracle):
    g = load_golden("bootstrap")
    rho3 = ginibre(np.random.default_rng(1234), 8)
    for tag, n, method, shots in (("B3lin", 3, "lin", 100000), ("B3mle", 3, "mle", 100000),
                                  ("B1mle", 1, "mle", 1000), ("B2mle", 2, "mle", 200)):
        state = qp.Qobj(rho3) if n == 3 else qp.Qobj(ginibre(np.random.default_rng(50 + n), 2**n))
        assert np.array_equal(state.matrix, g[tag + "_true"])
        np.random.seed(7)
        t = qp.StateTomograph(state)
        t.experiment(shots, "proj-set")
        assert np.array_equal(t.results, g[tag + "_counts0"])
        centre = t.point_estimate(method)
        assert abs(oracle.infidelity(g[tag + "_centre"], centre.matrix)) < 1e-8
        n_points = len(g[tag + "_boot_dist"])
        np.random.seed(4242)
        iv = qp.BootstrapStateInterval(t, n_points=n_points, method=method)
        dist, cl = iv(g[tag + "_cl"])
        assert np.array_equal(iv.boot_counts, g[tag + "_boot_counts"]), tag  # same resamples, bit for bit
        tol = 1e-11 if method == "lin" else 2e-5
        assert np.abs(iv.boot_dist - g[tag + "_boot_dist"]).max() < tol, tag
        assert np.abs(dist - g[tag + "_cl_dist"]).max() < tol, tag
    d2, c2 = iv()  # default confidence levels
    assert c2.shape == (1000,) and c2[0] == 1e-3 and np.all(np.diff(d2) >= 0)


def test_process_tomography_config3_and_others(qp, oracle):
    g = load_golden("process")
    makers = {
        "P0": lambda: qp.channel.depolarizing(0.1, 1),
        "P1": lambda: qp.operator.H.as_channel(),
        "P2": lambda: qp.channel.amplitude_damping(0.3),
        "C3": lambda: qp.channel.depolarizing(0.1, 2),
        "P4": lambda: qp.operator.CNOT.as_channel(),
    }
    for key, mk in makers.items():
        n = int(g[key + "_n"])
        np.random.seed(int(g[key + "_seed"]))
        tmg = qp.ProcessTomograph(mk())
        assert np.abs(tmg.channel.choi.matrix - g[key + "_true_choi"]).max() < 1e-15
        tmg.experiment(int(g[key + "_shots"]), str(g[key + "_povm"]))
        assert np.abs(np.stack([s.matrix for s in tmg.input_basis.elements]) - g[key + "_input_states"]).max() == 0
        assert np.abs(np.stack([t.state.matrix for t in tmg.tomographs]) - g[key + "_output_states"]).max() < 1e-15
        assert np.array_equal(tmg.results, g[key + "_counts"]), key  # structured states: p on the 0.5 branch
        ch = tmg.point_estimate("lifp", cptp=False)
        assert isinstance(ch, qp.Channel)
        assert np.abs(ch.choi.matrix - g[key + "_choi_nocptp"]).max() < 1e-10, key
        if n == 1:
            assert np.abs(tmg._lifp_oper - g[key + "_lifp_oper"]).max() < 1e-15
            assert np.abs(tmg._lifp_oper_inv - g[key + "_lifp_oper_inv"]).max() < 1e-10
        else:
            assert np.abs(tmg._lifp_oper[::37] - g[key + "_lifp_oper_rows"]).max() < 1e-15
            assert np.abs(tmg._lifp_oper_inv[:, ::37] - g[key + "_lifp_oper_inv_cols"]).max() < 1e-9
        assert np.abs(tmg.tp_projection(ch).choi.matrix - g[key + "_tp_only"]).max() < 1e-10
        assert np.abs(tmg.cp_projection(ch).choi.matrix - g[key + "_cp_only"]).max() < 1e-10
        cptp = tmg.point_estimate("lifp", cptp=True)
        assert tmg.cptp_iterations == int(g[key + "_dykstra_iters"]), key
        assert np.abs(cptp.choi.matrix - g[key + "_choi_cptp"]).max() < 1e-10, key
        assert np.abs(tmg.cptp_projection(ch).choi.matrix - g[key + "_choi_cptp"]).max() < 1e-10
        assert cptp.is_cptp(verbose=False)


def test_process_known_answer_from_reference_notebook(qp):
    """notebooks/Moments.ipynb cells 3-6: SIC input states, counts of input.json:18-23 injected
    through the `results` setter, printed choi.bloch of point_estimate(cptp=False)."""
    g = load_golden("process")
    ins = [qp.Qobj(b) for b in g["NB_input_blochs"]]
    target = qp.Channel(qp.Qobj([0.5, 0, 0, 0, 0, 0, 0, 0.5, 0, 0, 0.5, 0, 0, 0.5, 0, 0]))
    tmg = qp.ProcessTomograph(target, input_states=ins)
    np.random.seed(0)
    tmg.experiment(10000, "proj-set")
    tmg.results = g["NB_counts"]
    ch = tmg.point_estimate(cptp=False)
    assert np.abs(ch.choi.bloch - g["NB_printed_bloch_nocptp"]).max() < 5e-10
    assert np.abs(ch.choi.matrix - g["NB_choi_nocptp"]).max() < 1e-12
    assert np.abs(tmg.point_estimate().choi.matrix - g["NB_choi_cptp"]).max() < 1e-10


def test_bootstrap_process_interval(qp, oracle):
    np.random.seed(21)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.2, 1))
    tmg.experiment(2000, "proj-set")
    centre = tmg.point_estimate()
    np.random.seed(22)
    iv = qp.BootstrapProcessInterval(tmg, n_points=12)
    dist, _ = iv([0.5, 0.9])
    # replay on the oracle: same resampled counts (host RNG), Choi by the oracle's own estimator
    ins = oracle.input_states("proj4", 1)
    a = oracle.measurement_matrix("proj-set", 1)
    want = []
    for c in iv.boot_counts:
        choi = oracle.cptp_projection(oracle.lifp_estimate(c, a, ins), 1)
        want.append(oracle.hs_dst(choi, centre.choi.matrix))
    assert np.abs(iv.boot_dist - np.array(want)).max() < 1e-9
    assert np.abs(dist - oracle.quantiles(np.sort(want), [0.5, 0.9])).max() < 1e-9


def test_routines_on_gpu(qp, oracle):
    from quantpy_amd import routines

    rng = np.random.default_rng(8)
    a = rng.standard_normal((30, 7))
    assert np.abs(routines._left_inv(a) - oracle.left_inv(a)).max() < 1e-12
    c = a[:, :5] + 1j * rng.standard_normal((30, 5))
    assert np.abs(routines._left_inv(c) - oracle.left_inv(c)).max() < 1e-12  # plain transpose, complex
    # >= 128 columns: the Gauss-Jordan inverse runs chip-wide (two small launches per pivot step)
    big = rng.standard_normal((400, 150))
    assert np.abs(routines._left_inv(big) - oracle.left_inv(big)).max() < 1e-10
    bigc = big[:, :130] + 1j * rng.standard_normal((400, 130))
    assert np.abs(routines._left_inv(bigc) - oracle.left_inv(bigc)).max() < 1e-10
    with pytest.raises(Exception):  # rank deficient: reported, not returned
        routines._left_inv(np.hstack([big[:, :140], big[:, :1]]))
    rho = ginibre(rng, 4)
    x = routines._matrix_to_real_tril_vec(rho)
    assert np.abs(x - oracle.matrix_to_tril_vec(rho)).max() < 1e-14
    assert np.abs(routines._real_tril_vec_to_matrix(x) - rho).max() < 1e-14
    with pytest.raises(np.linalg.LinAlgError):
        routines._matrix_to_real_tril_vec(np.diag([1.0, -1.0]).astype(complex))
    assert np.array_equal(np.asarray(qp.generate_pauli(2)), oracle.pauli_basis(2))


def test_lifp_batched_gemm_path_matches_fused_kernel_and_oracle(qp, oracle):
    """n = 2 through the DENSE left inverse (qt_process_prefer_dense; the path of POVMs with M % 4 != 0), B >= 256:
    frequencies + FP64 MFMA GEMM over the batch (+ projection kernel) against the one-process-per-workgroup kernel the
    small batches use, against the oracle on a few trials, and against the default path through the Kronecker factors of
    the left inverse (k_lifp16); a ragged batch size (not a multiple of 16) exercises the row guards."""
    np.random.seed(31)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.3, 2))
    tmg.experiment(5000, "proj-set")
    eng = tmg._engine()
    base = tmg.results.reshape(16, -1)
    nset = tmg.results.shape[1]
    rng = np.random.default_rng(5)
    b = 307
    counts = rng.multinomial(5000, np.full(4, 0.25), size=(b, 16, nset)).astype(np.int64)
    counts[0] = tmg.results  # one realistic trial
    assert base.shape[1] == nset * 4
    ins = oracle.input_states("proj4", 2)
    a = oracle.measurement_matrix("proj-set", 2)
    for cptp in (False, True):
        eng.process_prefer_dense(True)
        big, it_big = eng.lifp(counts, cptp=cptp, return_iters=True)  # GEMM path
        small = [eng.lifp(counts[lo:lo + 100], cptp=cptp, return_iters=True) for lo in range(0, b, 100)]  # fused kernel
        eng.process_prefer_dense(False)
        ref = np.concatenate([s[0] for s in small])
        it_ref = np.concatenate([s[1] for s in small])
        assert np.abs(big - ref).max() < 1e-12
        assert np.array_equal(it_big, it_ref)
        fac, it_fac = eng.lifp(counts, cptp=cptp, return_iters=True)  # the factors: one wavefront per process
        assert np.abs(fac - big).max() < (1e-9 if cptp else 1e-11), np.abs(fac - big).max()
        assert np.array_equal(it_fac, it_big)
        assert np.array_equal(eng.lifp(counts[5], cptp=cptp), fac[5])  # a process alone: the same bits
        for t in (0, 1, b - 1):
            want = oracle.lifp_estimate(counts[t], a, ins)
            if cptp:
                want = oracle.cptp_projection(want, 2)
            assert np.abs(big[t] - want).max() < 1e-9
    # enough blocks of 64 processes for a workgroup to take several of them with one staged operand slice
    b2 = 2100
    counts2 = rng.multinomial(5000, np.full(4, 0.25), size=(b2, 16, nset)).astype(np.int64)
    eng.process_prefer_dense(True)
    big = eng.lifp(counts2, cptp=False)
    ref = np.concatenate([eng.lifp(counts2[lo:lo + 200], cptp=False) for lo in range(0, b2, 200)])
    eng.process_prefer_dense(False)
    assert np.abs(big - ref).max() < 1e-12
    assert np.abs(eng.lifp(counts2, cptp=False) - big).max() < 1e-11
    # more processes than resident wavefronts (the grid is capped: wavefronts stride over the batch), and the factors
    # themselves: entry [(c d + e) D + (a d + b)][s M + m] of the dense left inverse is V_S^+[a d + c][s] V_P^+[e d + b][m]
    b3 = 4 * 768 * 3 + 37
    counts3 = rng.multinomial(5000, np.full(4, 0.25), size=(b3, 16, nset)).astype(np.int64)
    many = eng.lifp(counts3, cptp=False)
    assert np.array_equal(many[[0, 768 * 4, b3 - 1]], eng.lifp(counts3[[0, 768 * 4, b3 - 1]], cptp=False))
    assert np.isfinite(many).all()
    vs, vp = eng.process_factors()
    _, inv = eng.process_operators()
    d, dd, mm = 4, 16, nset * 4
    a_, b_, c_, e_ = np.meshgrid(np.arange(d), np.arange(d), np.arange(d), np.arange(d), indexing="ij")
    kron = np.einsum("xs,ym->xysm", vs, vp)  # [alpha][beta][s][m]
    want = np.empty((dd * dd, dd * mm), dtype=np.complex128)
    want[((c_ * d + e_) * dd + (a_ * d + b_)).ravel()] = kron[(a_ * d + c_).ravel(), (e_ * d + b_).ravel()].reshape(-1, dd * mm)
    assert np.abs(want - inv).max() < 1e-9 * np.abs(inv).max()


def test_lifp_batched_ragged_povm_and_nan_isolation(qp):
    """The batched path with a POVM whose row count R = 16 M is not a multiple of the GEMM's 64-wide K chunks
    (five-outcome one-qubit POVM -> M = 25, R = 400, padded pitch 448), and with one process that has an input
    state without counts: its own Choi matrix is NaN (as counts / 0 in the reference), its neighbours are not."""
    sic = qp.generate_measurement_matrix("sic", 1)[0]  # (4, 4) Bloch rows
    five = np.vstack([sic[:1] / 2, sic[:1] / 2, sic[1:]])  # first outcome split in two: still a complete POVM
    np.random.seed(41)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.2, 2))
    tmg.experiment(4000, five)
    assert tmg.results.shape == (16, 1, 25)
    eng = tmg._engine()
    rng = np.random.default_rng(9)
    b = 300
    counts = rng.multinomial(4000, np.full(25, 0.04), size=(b, 16, 1)).astype(np.int64)
    counts[0] = tmg.results
    counts[137, 5] = 0  # process 137, input state 5: no counts at all
    big = eng.lifp(counts, cptp=False)
    ref = np.concatenate([eng.lifp(counts[lo:lo + 100], cptp=False) for lo in range(0, b, 100)])
    assert np.isnan(big[137]).all() and np.isnan(ref[137]).all()
    keep = np.arange(b) != 137
    assert np.isfinite(big[keep]).all()
    assert np.abs(big[keep] - ref[keep]).max() < 1e-12


@pytest.mark.parametrize("pieces", [1, 5, 13])
def test_lifp_factor_kernel_generic_row_counts(qp, oracle, pieces):
    """k_lifp16 with M / 4 not the compile-time 9 of the 'proj-set' POVM: all 36 'proj-set' elements as ONE measurement
    (each / 9) with the last one split into `pieces` equal parts -- M = 36 (the specialised kernel), 40 and 48 (the
    run-time loop) -- against the dense-operator path of the same set-up and the factor formula in NumPy."""
    np.random.seed(17)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.2, 2))
    rows = np.asarray(qp.generate_measurement_matrix("proj-set", 2), dtype=np.float64).reshape(-1, 16) / 9.0
    rows = np.concatenate([rows[:-1]] + [rows[-1:] / pieces] * pieces)
    m = rows.shape[0]
    assert m == 35 + pieces and m % 4 == 0
    eng = qp.get_engine(2)
    eng.set_povm(rows[None], 9 * 4000)
    eng.process_setup(np.stack([np.asarray(s.matrix, dtype=np.complex128) for s in tmg.input_basis.elements]))
    rng = np.random.default_rng(pieces)
    b = 333
    counts = rng.integers(0, 3000, size=(b, 16, 1, m)).astype(np.int64)
    fac = eng.lifp(counts, cptp=False)
    eng.process_prefer_dense(True)
    dense = eng.lifp(counts, cptp=False)
    eng.process_prefer_dense(False)
    assert np.abs(fac - dense).max() < 1e-10 * np.abs(dense).max()
    vs, vp = eng.process_factors()
    d = 4
    a_, b_, c_, e_ = np.meshgrid(np.arange(d), np.arange(d), np.arange(d), np.arange(d), indexing="ij")
    for k in (0, b - 1):
        f = counts[k, :, 0, :] / counts[k, :, 0, :].sum(axis=1, keepdims=True)
        x = vs @ f @ vp.T
        want = np.empty((16, 16), dtype=np.complex128)
        want[(a_ * d + b_).ravel(), (c_ * d + e_).ravel()] = x[(a_ * d + c_).ravel(), (e_ * d + b_).ravel()]
        assert np.abs(fac[k] - want).max() < 1e-11 * max(1.0, np.abs(want).max())

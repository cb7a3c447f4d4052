"""CPU: the oracle (oracle/quantpy_oracle.py) against the golden vectors that
tests/golden/make_golden.py produced from the imported reference, and against the one
known-answer the reference itself holds (notebooks/Moments.ipynb cells 5-7)."""
import warnings

import numpy as np
import pytest
from conftest import load_golden

warnings.filterwarnings("ignore")


def test_pauli_and_povm_tensors_bit_exact(oracle):
    g = load_golden("operators")
    for n in (1, 2, 3):
        assert np.array_equal(oracle.pauli_basis(n), g[f"pauli_n{n}"])
        for povm in ("proj", "proj-set", "proj4", "sic"):
            assert np.array_equal(oracle.measurement_matrix(povm, n), g[f"povm_{povm}_n{n}"])
    assert np.array_equal(oracle.measurement_matrix(g["povm_custom2d_in"], 2), g["povm_custom2d_n2"])
    for n in (4,):
        a = oracle.measurement_matrix("proj-set", n)
        idx = g[f"povm_proj-set_n{n}_idx"]
        assert tuple(g[f"povm_proj-set_n{n}_shape"]) == a.shape
        assert np.array_equal(a[idx[:, 0], idx[:, 1], idx[:, 2]], g[f"povm_proj-set_n{n}_val"])


def test_bloch_matrix_conversions(oracle):
    g = load_golden("states_born")
    for n in (1, 2, 3, 4):
        assert np.abs(oracle.bloch_from_matrix(g[f"rho_n{n}"]) - g[f"bloch_n{n}"]).max() < 1e-15
        assert np.abs(oracle.matrix_from_bloch(g[f"bloch_n{n}"]) - g[f"rho_from_bloch_n{n}"]).max() < 1e-15
        assert np.abs(oracle.bloch_from_matrix(g[f"nonherm_n{n}"]) - g[f"nonherm_bloch_n{n}"]).max() < 1e-14


def test_born_probabilities(oracle):
    g = load_golden("states_born")
    for n in (1, 2, 3):
        d = 2**n
        for povm in ("proj", "proj-set", "sic"):
            a = oracle.measurement_matrix(povm, n)
            for b, p in zip(g[f"bloch_n{n}"], g[f"born_{povm}_n{n}"]):
                assert np.array_equal(np.einsum("ijk,k->ij", a, b) * d, p)
                assert np.allclose(oracle.born_probs(a, b).sum(-1), 1, atol=1e-12)


def test_counts_bit_exact_rng_order(oracle):
    g = load_golden("counts_lin")
    np.random.seed(0)
    c = oracle.sample_counts(oracle.measurement_matrix("proj-set", 1), np.array([0.5, 0, 0, 0.5]), 10000)
    assert c.tolist() == [[5002, 4998], [5028, 4972], [10000, 0]]  # SURVEY 8d / BASELINE.md C1
    assert np.array_equal(c, g["C1_counts"])
    np.random.seed(7)
    a = oracle.measurement_matrix("proj-set", 3)
    b = oracle.bloch_from_matrix(g["C2_rho_true"])
    for i in range(8):
        assert np.array_equal(oracle.sample_counts(a, b, 100000), g["C2_counts"][i])
    with pytest.raises(TypeError):
        oracle.broadcast_shots(1e5, 27)  # a float scalar is not accepted (state.py:104-106)
    with pytest.raises(ValueError):
        oracle.broadcast_shots([1, 2], 3)


def test_linear_inversion(oracle):
    g = load_golden("counts_lin")
    a1 = oracle.measurement_matrix("proj-set", 1)
    assert np.abs(oracle.lin_estimate(g["C1_counts"], a1) - g["C1_lin"]).max() < 1e-14
    assert np.abs(oracle.lin_estimate(g["C1_counts"], a1, physical=False) - g["C1_lin_unphys"]).max() < 1e-14
    a3 = oracle.measurement_matrix("proj-set", 3)
    for c, r, ru in zip(g["C2_counts"], g["C2_lin"], g["C2_lin_unphys"]):
        assert np.abs(oracle.lin_estimate(c, a3) - r).max() < 1e-13
        assert np.abs(oracle.lin_estimate(c, a3, physical=False) - ru).max() < 1e-13
    for k in range(int(g["n_lin_cases"])):
        key = f"L{k}"
        a = g[key + "_povm_matrix"]
        n_meas = g[key + "_counts"].sum(-1)
        assert np.abs(oracle.left_inv(oracle.weighted_povm(a, n_meas)) - g[key + "_leftinv"]).max() < 1e-9
        rho, bloch = oracle.lin_estimate(g[key + "_counts"], a, physical=False, return_bloch=True)
        assert np.abs(bloch - g[key + "_lin_bloch"]).max() < 1e-13
        assert np.abs(rho - g[key + "_lin_unphys"]).max() < 1e-13
        assert np.abs(oracle.lin_estimate(g[key + "_counts"], a) - g[key + "_lin"]).max() < 1e-12


def test_cholesky_param_and_nll(oracle):
    g = load_golden("chol_nll")
    for k in range(int(g["n_nll_cases"])):
        key = f"N{k}"
        assert np.abs(oracle.matrix_to_tril_vec(g[key + "_rho"]) - g[key + "_x"]).max() < 1e-14
        assert np.abs(oracle.tril_vec_to_matrix(g[key + "_x"]) - g[key + "_LLh"]).max() < 1e-15
        assert np.abs(oracle.tril_vec_to_matrix(g[key + "_xr"]) - g[key + "_LLh_r"]).max() < 1e-15
        prob = oracle.NllProblem(g[key + "_counts"], g[key + "_povm_matrix"])
        assert abs(prob.nll(g[key + "_x"]) - g[key + "_nll_x"]) < 1e-13
        assert abs(prob.nll(g[key + "_xr"]) - g[key + "_nll_xr"]) < 1e-13
        f, grad = prob.nll_and_grad(g[key + "_xr"])
        assert abs(f - g[key + "_nll_xr"]) < 1e-13
        # analytic gradient vs the reference's differentiated NLL (central: ~1e-9; SciPy's
        # forward difference as BFGS sees it: ~1e-7)
        assert np.abs(grad - g[key + "_cgrad_xr"]).max() < 5e-8
        assert np.abs(grad - g[key + "_fdgrad_xr"]).max() < 5e-6


def _mle_cases(g, pred):
    return [k for k in range(int(g["n_mle_cases"])) if pred(int(g[f"M{k}_n"]), k)]


def test_mle_reference_algorithm_subset(oracle):
    """scipy BFGS + forward differences, exactly as state.py:213: same nit/nfev; fidelity 1e-6
    (north_star).  n=3 iterating cases take ~1 s each in the oracle, so a subset runs here."""
    g = load_golden("mle")
    for k in _mle_cases(g, lambda n, k: n <= 2 or k in (44, 46, 54, 58, 62)):
        key = f"M{k}"
        a = oracle.measurement_matrix(str(g[key + "_povm"]), int(g[key + "_n"]))
        rho, info = oracle.mle_estimate(g[key + "_counts"], a, init=str(g[key + "_init"]), return_info=True)
        assert info["nit"] == int(g[key + "_nit"]) and info["nfev"] == int(g[key + "_nfev"]), key
        assert abs(oracle.infidelity(g[key + "_rho"], rho)) < 1e-6, key
        assert oracle.hs_dst(g[key + "_rho"], rho) < 1e-4, key


def test_mle_optimizer_restatement_all_cases(oracle):
    """The plain restatement of SciPy's BFGS (oracle.bfgs_minimize, analytic gradient) on all
    70 golden trials: identical iteration count to the reference and fidelity within 1e-6;
    and bit-identical to scipy's own BFGS given the same analytic gradient (n <= 2)."""
    g = load_golden("mle")
    for k in range(int(g["n_mle_cases"])):
        key = f"M{k}"
        n = int(g[key + "_n"])
        a = oracle.measurement_matrix(str(g[key + "_povm"]), n)
        kw = dict(init=str(g[key + "_init"]), return_info=True)
        rho, info = oracle.mle_estimate(g[key + "_counts"], a, solver="port", **kw)
        assert info["nit"] == int(g[key + "_nit"]), key
        assert info["status"] == int(g[key + "_status"]), key
        # scipy counts the d^2 forward-difference evaluations of each gradient in nfev
        assert info["nfev"] + 4**n * info["njev"] == int(g[key + "_nfev"]), key
        assert abs(oracle.infidelity(g[key + "_rho"], rho)) < 1e-6, key
        if n <= 2:
            rho2, info2 = oracle.mle_estimate(g[key + "_counts"], a, jac="analytic", **kw)
            assert np.array_equal(rho, rho2) and info2["nit"] == info["nit"], key


def test_process_known_answer_from_reference_notebook(oracle):
    """notebooks/Moments.ipynb cells 5-7: counts (= input.json:18-23), SIC input states,
    'proj-set' -> printed choi.bloch (cptp=False)."""
    g = load_golden("process")
    ins = [oracle.matrix_from_bloch(b) for b in g["NB_input_blochs"]]
    a = oracle.measurement_matrix("proj-set", 1)
    choi, oper, inv = oracle.lifp_estimate(g["NB_counts"], a, ins, return_oper=True)
    assert np.abs(oracle.bloch_from_matrix(choi) - g["NB_printed_bloch_nocptp"]).max() < 5e-10
    assert np.abs(choi - g["NB_choi_nocptp"]).max() < 1e-13
    assert np.abs(oper - g["NB_lifp_oper"]).max() < 1e-15
    assert np.abs(inv - g["NB_lifp_oper_inv"]).max() < 1e-11
    cptp = oracle.cptp_projection(choi, 1)
    # (cell 7's printed cptp=True matrix is stale notebook output from other counts: the
    #  reference itself, run on cell 5's counts, differs from it by 1.3e-2, so it pins nothing)
    assert np.abs(cptp - g["NB_choi_cptp"]).max() < 1e-12


def test_process_cases(oracle):
    g = load_golden("process")
    for key in ("P0", "P1", "P2", "C3", "P4"):
        n = int(g[key + "_n"])
        a = oracle.measurement_matrix(str(g[key + "_povm"]), n)
        ins = oracle.input_states("proj4", n)
        assert np.abs(np.stack(ins) - g[key + "_input_states"]).max() < 1e-15
        outs = np.stack([oracle.apply_choi(g[key + "_true_choi"], r, n) for r in ins])
        assert np.abs(outs - g[key + "_output_states"]).max() < 1e-12
        choi, oper, inv = oracle.lifp_estimate(g[key + "_counts"], a, ins, return_oper=True)
        if n == 1:
            assert np.abs(oper - g[key + "_lifp_oper"]).max() < 1e-15
        else:
            assert np.abs(oper[::37] - g[key + "_lifp_oper_rows"]).max() < 1e-15
            assert np.abs(inv[:, ::37] - g[key + "_lifp_oper_inv_cols"]).max() < 1e-9
        assert np.abs(choi - g[key + "_choi_nocptp"]).max() < 1e-11, key
        v = oracle.mat2vec(choi)
        assert np.abs(oracle.vec2mat(oracle.tp_projection_vec(v, n)) - g[key + "_tp_only"]).max() < 1e-11
        assert np.abs(oracle.vec2mat(oracle.cp_projection_vec(v)) - g[key + "_cp_only"]).max() < 1e-11
        cptp, iters = oracle.cptp_projection(choi, n, return_iters=True)
        assert iters == int(g[key + "_dykstra_iters"]), key
        assert np.abs(cptp - g[key + "_choi_cptp"]).max() < 1e-10, key


def test_process_sampling_order(oracle):
    """ProcessTomograph.experiment draws input state by input state, setting by setting
    (process.py:124-129) on the one global stream."""
    g = load_golden("process")
    for key in ("P0", "C3"):
        n = int(g[key + "_n"])
        a = oracle.measurement_matrix(str(g[key + "_povm"]), n)
        np.random.seed(int(g[key + "_seed"]))
        counts = np.stack([oracle.sample_counts(a, oracle.bloch_from_matrix(r), int(g[key + "_shots"]))
                           for r in g[key + "_output_states"]])
        assert np.array_equal(counts, g[key + "_counts"])


def test_bootstrap_small(oracle):
    g = load_golden("bootstrap")
    for tag, n, method in (("B3lin", 3, "lin"), ("B1mle", 1, "mle"), ("B2mle", 2, "mle")):
        a = oracle.measurement_matrix("proj-set", n)
        np.random.seed(4242)
        n_points = len(g[tag + "_boot_dist"])
        # solver='port' (exact gradient): the forward-difference restatement flips one borderline
        # B2mle resample (19 vs 18 iterations, HS 1.3e-3) on rounding noise alone -- the
        # reference's loosely converged BFGS is path-dependent.
        kw = {} if method == "lin" else dict(solver="port")
        srt, centre, counts, dist = oracle.bootstrap_state(g[tag + "_counts0"], a, n_points, method=method, **kw)
        assert np.abs(centre - g[tag + "_centre"]).max() < 1e-6
        assert np.array_equal(counts, g[tag + "_boot_counts"]), tag
        assert np.abs(dist - g[tag + "_boot_dist"]).max() < (1e-12 if method == "lin" else 2e-5)
        q = oracle.quantiles(srt, g[tag + "_cl"])
        assert np.abs(q - g[tag + "_cl_dist"]).max() < (1e-12 if method == "lin" else 2e-5)


def test_large_n(oracle):
    g = load_golden("large")
    a = oracle.measurement_matrix("proj-set", 4)
    assert np.abs(oracle.lin_estimate(g["n4_counts"], a, physical=False) - g["n4_lin_unphys"]).max() < 1e-12
    assert np.abs(oracle.lin_estimate(g["n4_counts"], a) - g["n4_lin"]).max() < 1e-12
    prob = oracle.NllProblem(g["n4_counts"], a)
    assert abs(prob.nll(g["n4_x"]) - g["n4_nll"]) < 1e-12


def test_pgdb_against_reference(oracle):
    """'pgdb' (process.py:291-308): the reference's first iteration piece by piece, and what it returns.
    In every case the projected-gradient direction from the (non-CPTP, trace 1) fully mixed start is an
    ascent direction, the backtracking loop halves alpha down to ~5.6e-17 and the estimate never moves:
    the reference returns its starting point."""
    g = load_golden("pgdb")
    for key in ("P0", "P2", "C3"):
        n = int(g[key + "_n"])
        povm = oracle.measurement_matrix(str(g[key + "_povm"]), n)
        ins = list(g[key + "_input_states"])
        choi, info = oracle.pgdb_estimate(g[key + "_counts"], povm, ins, n_iter=2, return_info=True)
        t0 = info["trace"][0]
        assert t0["alpha"] == float(g[key + "_it0_alpha"]), key
        assert abs(t0["dot"] - complex(g[key + "_it0_dot"])) < 1e-9 * abs(complex(g[key + "_it0_dot"])), key
        assert abs(t0["f0"] - g[key + "_it0_nll"][0]) < 1e-12 * abs(g[key + "_it0_nll"][0]), key
        assert np.abs(choi - g[key + "_returned"]).max() < 1e-13, key
        conv = oracle.pgdb_estimate(g[key + "_counts"], povm, ins, n_iter=int(g[key + "_conv_cap"]), stop="converged")
        assert np.abs(conv - g[key + "_conv_choi"]).max() < 1e-13, key


def test_mle_constr_against_reference(oracle):
    """'mle-constr' (state.py:231-253): the oracle's SLSQP call reproduces the reference's results, and
    with exact gradients (what the HIP-backed host code hands SciPy) stays within 1e-9 infidelity."""
    g = load_golden("constr")
    for k in range(int(g["n_cases"])):
        key = f"K{k}"
        n = int(g[key + "_n"])
        povm = oracle.measurement_matrix(str(g[key + "_povm"]), n)
        for init in ("lin", "mixed"):
            want = g[key + "_" + init]
            got = oracle.mle_constr_estimate(g[key + "_counts"], povm, init=init)
            # forward differences with h = 1.5e-8 turn the last-bit differences between two NLL
            # implementations into ~1e-8 of gradient noise, so the iterates agree to ~1e-7, not to rounding
            assert np.abs(got - want).max() < 1e-5 and oracle.infidelity(got, want) < 1e-9, (key, init)
            exact = oracle.mle_constr_estimate(g[key + "_counts"], povm, init=init, jac="analytic")
            assert oracle.infidelity(exact, want) < 1e-9, (key, init)


def test_mhmc_state_interval_against_reference(oracle):
    """MHMCStateInterval (interval.py:689-750, mhmc.py): same global RNG seed, same order of draws -> the
    oracle's chain ends where the reference's ended and gives the same distance quantiles."""
    g = load_golden("mhmc")
    for k in range(int(g["n_cases"])):
        key = f"H{k}"
        n = int(g[key + "_n"])
        povm = oracle.measurement_matrix(str(g[key + "_povm"]), n)
        n_points, burn, thin = (int(v) for v in g[key + "_args"])
        np.random.seed(int(g[key + "_rng_seed"]))
        dist, samples, rate = oracle.mhmc_state_interval(g[key + "_counts"], povm, g[key + "_state"], n_points,
                                                        float(g[key + "_step"]), burn, thin)
        assert np.abs(dist - g[key + "_all_dist"]).max() < 1e-12, key
        q = np.interp(g["conf_levels"], np.linspace(0, 1, len(dist)), dist)
        assert np.abs(q - g[key + "_radii"]).max() < 1e-12, key
        assert 0.0 < rate <= 1.0


def test_mhmc_process_interval_against_reference(oracle):
    """MHMCProcessInterval (interval.py:763-850): the oracle's chain reproduces the reference's samples."""
    g = load_golden("mhmc")
    for key in ("Q0", "Q1"):
        n = int(g[key + "_n"])
        povm = oracle.measurement_matrix("proj-set", n)
        ins = oracle.input_states("proj4", n)
        n_points, burn = (int(v) for v in g[key + "_args"])
        np.random.seed(200 + int(g[key + "_seed"]))
        dist, samples, rate = oracle.mhmc_process_interval(g[key + "_counts"], povm, ins, g[key + "_channel"], n_points,
                                                          float(g[key + "_step"]), burn)
        assert np.abs(np.stack(samples) - g[key + "_samples"]).max() < 1e-10, key
        assert np.abs(dist - g[key + "_dist"]).max() < 1e-10, key
        assert abs(rate - float(g[key + "_rate"])) < 1e-12, key


# ---- round 2: leftovers.npz (if_dst / trace_dst, warm_start, BootstrapProcessInterval) -----------------------
def test_geometry_distances_against_reference(oracle):
    """a17: geometry.py:5-56 on 24 pairs (full rank, rank deficient, pure, identical, nearly equal).  The
    restatement uses scipy.linalg.sqrtm exactly as the reference does, so it must agree to rounding; the eigh /
    SVD based `infidelity` the parity harness uses agrees to sqrtm's accuracy on singular arguments (~1e-8)."""
    g = load_golden("leftovers")
    worst_eigh = 0.0
    for i in range(int(g["geo_n_pairs"])):
        a, b = g[f"geo{i}_a"], g[f"geo{i}_b"]
        assert abs(oracle.hs_dst(a, b) - float(g[f"geo{i}_hs"])) < 1e-15
        assert abs(oracle.trace_dst(a, b) - float(g[f"geo{i}_trace"])) < 1e-13
        assert abs(oracle.if_dst(a, b) - float(g[f"geo{i}_if"])) < 1e-13
        worst_eigh = max(worst_eigh, abs(oracle.infidelity(a, b) - float(g[f"geo{i}_if"])))
    assert worst_eigh < 2e-7, worst_eigh


def test_warm_start_accumulation_against_reference(oracle):
    """state.py:116-124 / process.py:122-129: counts bit-exact through the RNG call order, the stacked POVM, and
    the estimators on the accumulated data."""
    g = load_golden("leftovers")
    for tag, n, first, seed in (("W1", 1, 1000, 101), ("W2", 2, 1000, 102), ("W3", 3, 5000, 103)):
        a = oracle.measurement_matrix("proj-set", n)
        bloch = oracle.bloch_from_matrix(g[tag + "_state"])
        np.random.seed(seed)
        c1 = oracle.sample_counts(a, bloch, first)
        second = g[tag + "_second"]
        c2 = oracle.sample_counts(a, bloch, int(second) if second.ndim == 0 else second)
        povm, res = oracle.warm_start_stack(a, c1, a, c2)
        assert np.array_equal(res, g[tag + "_results"]) and np.array_equal(povm, g[tag + "_povm"])
        assert np.array_equal(res.sum(-1), g[tag + "_nmeas"])
        assert np.abs(oracle.lin_estimate(res, povm, physical=False) - g[tag + "_lin_unphys"]).max() < 1e-11
        assert np.abs(oracle.lin_estimate(res, povm) - g[tag + "_lin"]).max() < 1e-11
        rho, info = oracle.mle_estimate(res, povm, return_info=True)
        assert info["nit"] == int(g[tag + "_mle_nit"]) and abs(oracle.infidelity(rho, g[tag + "_mle"])) < 1e-9
        c3 = oracle.sample_counts(a, bloch, first)
        povm3, res3 = oracle.warm_start_stack(povm, res, a, c3)
        assert np.array_equal(res3, g[tag + "_results3"]) and np.array_equal(povm3, g[tag + "_povm3"])
        assert np.abs(oracle.lin_estimate(res3, povm3) - g[tag + "_lin3"]).max() < 1e-11
    for tag, n in (("WP1", 1), ("WP2", 2)):
        counts, povm = g[tag + "_results"], g[tag + "_povm"]
        ins = oracle.input_states("proj4", n)
        raw = oracle.lifp_estimate(counts, povm, ins)
        assert np.abs(raw - g[tag + "_choi_raw"]).max() < 1e-10
        assert np.abs(oracle.cptp_projection(raw, n) - g[tag + "_choi"]).max() < 1e-10


def test_bootstrap_process_interval_against_reference(oracle):
    """interval.py:615-685 run by the reference itself (n = 2 'lifp'; n = 1 'lifp', 'states', 'pgdb'): the oracle's
    estimators on the recorded resamples reproduce every bootstrap Choi matrix and distance."""
    g = load_golden("leftovers")
    for tag, n, method in (("BP2lifp", 2, "lifp"), ("BP1lifp", 1, "lifp"), ("BP1states", 1, "states"), ("BP1pgdb", 1, "pgdb")):
        povm = oracle.measurement_matrix("proj-set", n)
        ins = oracle.input_states("proj4", n)
        centre = g[tag + "_centre"]
        dists = []
        for counts, want in zip(g[tag + "_boot_counts"][: (2 if method == "pgdb" else None)], g[tag + "_boot_choi"]):
            if method == "lifp":
                est = oracle.cptp_projection(oracle.lifp_estimate(counts, povm, ins), n)
            elif method == "states":
                est = oracle.states_estimate(counts, povm, ins, n)
            else:
                est = oracle.pgdb_estimate(counts, povm, ins, n_iter=20)  # the estimate never moves
            assert np.abs(est - want).max() < 1e-9, tag
            dists.append(oracle.hs_dst(est, centre))
        if method != "pgdb":
            assert np.abs(np.array(dists) - g[tag + "_boot_dist"]).max() < 1e-9
            assert np.abs(oracle.quantiles(np.sort(dists), g["conf_levels"]) - g[tag + "_cl_dist"]).max() < 1e-9

"""GPU: the rows SURVEY 8f ranks "next" -- MomentInterval (closed-form CI), the 'states' process
estimator, and the JSON command-line front ends -- against golden vectors from the reference and the
reference-held known answers (notebooks/Moments.ipynb cell 4 radii, cell 6 Bloch vector)."""
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


def test_moment_interval_state(qp):
    g = load_golden("moment")
    cls = g["conf_levels"]
    for k in range(int(g["n_state_cases"])):
        key = f"S{k}"
        n = int(g[key + "_n"])
        t = qp.StateTomograph(qp.qobj.fully_mixed(n))
        t.povm_matrix = qp.generate_measurement_matrix(str(g[key + "_povm"]), n)
        t.results = g[key + "_counts"]
        for distr in ("gamma", "norm", "exp"):
            got, _ = qp.MomentInterval(t, distr_type=distr)(cls)
            assert np.allclose(got, g[key + "_" + distr], rtol=1e-9, atol=0), (key, distr)
    with pytest.raises(NotImplementedError):
        qp.MomentInterval(t, distr_type="cauchy")()


def test_moment_interval_process_known_answer(qp):
    g = load_golden("moment")
    gp = load_golden("process")
    ins = [qp.Qobj(b) for b in gp["NB_input_blochs"]]
    tmg = qp.ProcessTomograph(qp.Channel(qp.Qobj([0.5, 0, 0, 0, 0, 0, 0, 0.5, 0, 0, 0.5, 0, 0, 0.5, 0, 0])),
                              input_states=ins)
    np.random.seed(0)
    tmg.experiment(10000, "proj-set")
    tmg.results = gp["NB_counts"]
    radii, _ = qp.MomentInterval(tmg)([0.5, 0.75, 0.9])
    assert np.abs(radii - g["NB_process_printed"]).max() < 5e-9  # the notebook's printed 8 digits
    assert np.allclose(radii, g["NB_process_radii"], rtol=1e-9)
    np.random.seed(11)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 2))
    tmg.experiment(10000, "proj-set")
    assert np.array_equal(tmg.results, g["C3_counts"])
    assert np.allclose(qp.MomentInterval(tmg)(g["conf_levels"])[0], g["C3_process_radii"], rtol=1e-8)


def test_process_states_method(qp, oracle):
    g = load_golden("process")
    makers = {"P0": lambda: qp.channel.depolarizing(0.1, 1), "P1": lambda: qp.operator.H.as_channel(),
              "C3": lambda: qp.channel.depolarizing(0.1, 2)}
    for key, mk in makers.items():
        np.random.seed(int(g[key + "_seed"]))
        tmg = qp.ProcessTomograph(mk())
        tmg.experiment(int(g[key + "_shots"]), str(g[key + "_povm"]))
        assert np.array_equal(tmg.results, g[key + "_counts"])
        assert np.abs(tmg.point_estimate("states", cptp=False).choi.matrix - g[key + "_states_lin"]).max() < 1e-10
        assert np.abs(tmg.point_estimate("states").choi.matrix - g[key + "_states_lin_cptp"]).max() < 1e-9
        if int(g[key + "_n"]) == 1:
            got = tmg.point_estimate("states", states_est_method="mle").choi.matrix
            assert np.abs(got - g[key + "_states_mle_cptp"]).max() < 1e-6
    with pytest.raises(ValueError):
        tmg.point_estimate("nope")


def test_cli_wire_format(qp, tmp_path):
    """the reference's own input.json (tests/golden/reference_input.json) through both front ends."""
    from quantpy_amd import cli

    src = os.path.join(GOLDEN, "reference_input.json")
    out = tmp_path / "out.json"
    cli.process_interval(["-i", src, "-o", str(out)])
    res = json.load(open(out))
    gp, gm = load_golden("process"), load_golden("moment")
    # input.json truncates the SIC input states to 4 digits: 1.3e-4 from the notebook's full-precision run
    assert np.abs(np.array(res["process"]) - gp["NB_printed_bloch_nocptp"]).max() < 3e-4
    assert np.abs(np.array(res["hs_radius"]) - gm["NB_process_printed"]).max() < 1e-4
    assert len(res["hs_radius"]) == 3
    # a state file in the same format
    data = {"povm_matrix": qp.generate_measurement_matrix("proj-set", 1).tolist(),
            "outcomes": [[5002, 4998], [5028, 4972], [10000, 0]], "conf_levels": [0.5, 0.9]}
    sfile = tmp_path / "state.json"
    sfile.write_text(json.dumps(data))
    res = cli.state_interval(["-i", str(sfile), "--no-ci"])
    assert "hs_radius" not in res
    assert np.abs(np.array(res["state"]) - load_golden("counts_lin")["C1_lin_bloch_unphys"]).max() < 1e-13
    res = cli.state_interval(["-i", str(sfile)])
    assert len(res["hs_radius"]) == 2 and res["hs_radius"][0] < res["hs_radius"][1]


def test_pgdb_matches_reference_and_oracle(oracle):
    """'pgdb' (process.py:291-308) through the C ABI: against what the reference returned (golden) and
    against the oracle with the same iteration cap; both stop rules; and through ProcessTomograph."""
    from quantpy_amd import get_engine
    import quantpy_amd as qp

    g = load_golden("pgdb")
    for key in ("P0", "P2", "C3"):
        n = int(g[key + "_n"])
        povm = oracle.measurement_matrix(str(g[key + "_povm"]), n)
        ins = g[key + "_input_states"]
        counts = g[key + "_counts"]
        eng = get_engine(n)
        eng.set_povm(povm, counts[0].sum(-1).astype(float))
        eng.process_setup(ins)
        for stop in ("reference", "converged"):
            got, iters = eng.pgdb(counts, n_iter=3, stop=stop, return_iters=True)
            want = oracle.pgdb_estimate(counts, povm, list(ins), n_iter=3, stop=stop)
            assert np.abs(got - want).max() < 1e-12, (key, stop)
            assert np.abs(got - g[key + "_returned"]).max() < 1e-12, (key, stop)
            assert 1 <= iters <= 3
        batch = eng.pgdb(np.stack([counts, counts]), n_iter=2)
        assert np.abs(batch[0] - batch[1]).max() == 0.0
    # drop-in surface: ProcessTomograph.point_estimate('pgdb') returns a Channel
    np.random.seed(11)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 1))
    tmg.experiment(10000, "proj-set")
    ch = tmg.point_estimate("pgdb", n_iter=2)
    assert isinstance(ch, qp.Channel)
    assert np.abs(ch.choi.matrix - np.eye(4) / 4).max() < 1e-12


def test_mle_constr_matches_reference(qp, oracle):
    """StateTomograph.point_estimate('mle-constr') -- SciPy's SLSQP on the host, NLL + exact gradient from
    qt_nll_batch -- against the reference's results (golden), both starting points."""
    g = load_golden("constr")
    for k in range(int(g["n_cases"])):
        key = f"K{k}"
        n = int(g[key + "_n"])
        tmg = qp.StateTomograph(qp.Qobj(np.eye(2**n) / 2**n))
        tmg.experiment(10, str(g[key + "_povm"]))
        tmg.results = g[key + "_counts"]
        for init in ("lin", "mixed"):
            got = tmg.point_estimate("mle-constr", init=init)
            assert isinstance(got, qp.Qobj)
            assert oracle.infidelity(got.matrix, g[key + "_" + init]) < 1e-9, (key, init)
            assert np.abs(got.matrix - g[key + "_" + init]).max() < 1e-5, (key, init)
    with pytest.raises(ValueError):
        tmg.point_estimate("mle-constr", init="nope")


def test_mhmc_state_interval_matches_reference(qp, oracle):
    """MHMCStateInterval through qt_mhmc_state: the reference's radii (golden, same seed), the oracle's
    chain state by state, and warm_start continuing the chain."""
    g = load_golden("mhmc")
    for k in range(int(g["n_cases"])):
        key = f"H{k}"
        n = int(g[key + "_n"])
        povm_name = str(g[key + "_povm"])
        n_points, burn, thin = (int(v) for v in g[key + "_args"])
        step = float(g[key + "_step"])
        tmg = qp.StateTomograph(qp.Qobj(np.eye(2**n) / 2**n))
        tmg.experiment(10, povm_name)
        tmg.results = g[key + "_counts"]
        tmg.reconstructed_state = qp.Qobj(g[key + "_state"])
        np.random.seed(int(g[key + "_rng_seed"]))
        iv = qp.MHMCStateInterval(tmg, n_points=n_points, step=step, burn_steps=burn, thinning=thin)
        radii = iv(g["conf_levels"])[0]
        assert np.abs(radii - g[key + "_radii"]).max() < 1e-10, (key, radii, g[key + "_radii"])
        assert np.abs(iv.cl_to_dist(np.linspace(0, 1, n_points)) - g[key + "_all_dist"]).max() < 1e-10, key
        np.random.seed(int(g[key + "_rng_seed"]))
        _, samples, rate = oracle.mhmc_state_interval(g[key + "_counts"], oracle.measurement_matrix(povm_name, n),
                                                      g[key + "_state"], n_points, step, burn, thin)
        assert np.abs(iv.samples - samples).max() < 1e-12, key
        assert abs(iv.acceptance_rate - rate) < 1e-12
    # warm start: the second call continues from the last state without a new burn-in
    np.random.seed(5)
    iv = qp.MHMCStateInterval(tmg, n_points=20, step=0.001, burn_steps=10, warm_start=True)
    iv.setup()
    last = iv._x_t.copy()
    iv.setup()
    assert iv._burned and not np.array_equal(last, iv._x_t)
    with pytest.raises(NotImplementedError):
        np.random.seed(11)
        ptm = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 1))
        ptm.experiment(100, "proj-set")
        qp.MHMCStateInterval(ptm).setup()


def test_sugiyama_and_holder_intervals(qp):
    """SugiyamaInterval (interval.py:219-265) and HolderInterval (interval.py:421-539) against the
    reference's numbers; the unprovided (cvxopt) intervals exist under the reference's names and say so."""
    g = load_golden("holder")
    cls = g["conf_levels"]
    for k in range(int(g["n_state_cases"])):
        key = f"S{k}"
        n = int(g[key + "_n"])
        tmg = qp.StateTomograph(qp.Qobj(np.eye(2**n) / 2**n), str(g[key + "_dst"]))
        tmg.experiment(10, str(g[key + "_povm"]))
        tmg.results = g[key + "_counts"]
        radii = qp.SugiyamaInterval(tmg, n_points=400)(cls)[0]
        assert np.abs(radii - g[key + "_radii"]).max() < 1e-10, key
    np.random.seed(21)
    ptm = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 1))
    ptm.experiment(2000, "proj-set")
    assert np.array_equal(ptm.results, g["H_counts"])
    ptm.point_estimate("states")
    for t, want in zip(ptm.tomographs, g["H_states"]):
        assert np.abs(t.reconstructed_state.matrix - want).max() < 1e-10
    dist, cl = qp.HolderInterval(ptm, n_points=300, kind="sugiyama")(cls)
    assert np.abs(dist - g["H_sugiyama_dist"]).max() < 1e-10 and np.abs(cl - g["H_sugiyama_cl"]).max() < 1e-15
    np.random.seed(31)
    dist, cl = qp.HolderInterval(ptm, n_points=40, kind="bootstrap", method="lin")(cls)
    assert np.abs(dist - g["H_bootstrap_dist"]).max() < 1e-10
    np.random.seed(41)
    dist, cl = qp.HolderInterval(ptm, n_points=60, kind="mhmc", step=0.01, burn_steps=20)(cls)
    assert np.abs(dist - g["H_mhmc_dist"]).max() < 1e-9
    with pytest.raises(ValueError):
        qp.HolderInterval(ptm)(cls)  # the reference's default kind='wang' has no branch either
    with pytest.raises(TypeError):
        qp.HolderInterval(ptm, kind="moment")(cls)
    assert str(g["H_error_wang"]) == "ValueError" and str(g["H_error_moment"]) == "TypeError"
    for name in ("MomentFidelityStateInterval", "MomentFidelityProcessInterval", "PolytopeStateInterval",
                 "PolytopeProcessInterval"):
        with pytest.raises(NotImplementedError):
            getattr(qp, name)(ptm)


def test_mhmc_process_interval_matches_reference(qp):
    """MHMCProcessInterval through qt_mhmc_process against the reference's samples (golden, same seeds)."""
    g = load_golden("mhmc")
    makers = {"Q0": lambda: qp.channel.depolarizing(0.1, 1), "Q1": lambda: qp.channel.depolarizing(0.2, 2)}
    for key, mk in makers.items():
        np.random.seed(int(g[key + "_seed"]))
        tmg = qp.ProcessTomograph(mk())
        tmg.experiment(int(g[key + "_shots"]), "proj-set")
        assert np.array_equal(tmg.results, g[key + "_counts"])
        ch = tmg.point_estimate("lifp")
        assert np.abs(ch.choi.matrix - g[key + "_channel"]).max() < 1e-9
        n_points, burn = (int(v) for v in g[key + "_args"])
        np.random.seed(200 + int(g[key + "_seed"]))
        iv = qp.MHMCProcessInterval(tmg, n_points=n_points, step=float(g[key + "_step"]), burn_steps=burn,
                                    return_samples=True)
        dist, cl, rate, mats = iv.setup()
        assert np.abs(np.stack(mats) - g[key + "_samples"]).max() < 1e-8, key
        assert np.abs(dist - g[key + "_dist"]).max() < 1e-8, key
        assert abs(rate - float(g[key + "_rate"])) < 1e-12, key
        np.random.seed(200 + int(g[key + "_seed"]))
        radii = qp.MHMCProcessInterval(tmg, n_points=n_points, step=float(g[key + "_step"]), burn_steps=burn)([0.5])[0]
        assert np.abs(radii - np.interp(0.5, np.linspace(0, 1, n_points), g[key + "_dist"])).max() < 1e-8

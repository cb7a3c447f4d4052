"""GPU: the drop-in API against reference-generated fixtures for what round 1 left unpinned
(tests/golden/leftovers.npz: geometry.py distances, warm_start accumulation, BootstrapProcessInterval incl. n = 2 and
the 'states' / 'pgdb' methods), n = 5 MLE against the oracle on 8 trials at 1e6 shots, and the properties of the
2000-resample bootstrap of configs[3]."""
import numpy as np
import pytest
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _ginibre(rng, d, rank=None):
    g = rng.standard_normal((d, rank or d)) + 1j * rng.standard_normal((d, rank or d))
    r = g @ g.conj().T
    return r / np.trace(r)


def test_distances_against_reference():
    """a17 (geometry.py:5-56): hs_dst on the GPU kernel, trace_dst / if_dst with the reference's own expression."""
    import quantpy_amd as qp

    g = load_golden("leftovers")
    for i in range(int(g["geo_n_pairs"])):
        a, b = g[f"geo{i}_a"], g[f"geo{i}_b"]
        assert abs(qp.hs_dst(a, b) - float(g[f"geo{i}_hs"])) < 1e-14
        assert abs(qp.hs_dst(qp.Qobj(a), qp.Qobj(b)) - float(g[f"geo{i}_hs"])) < 1e-14
        assert abs(qp.trace_dst(qp.Qobj(a), qp.Qobj(b)) - float(g[f"geo{i}_trace"])) < 1e-12
        assert abs(qp.if_dst(a, b) - float(g[f"geo{i}_if"])) < 1e-12
    t = qp.StateTomograph(qp.Qobj(g["geo0_a"]), dst="if")
    assert abs(t.dst(qp.Qobj(g["geo0_a"]), qp.Qobj(g["geo0_b"])) - float(g["geo0_if"])) < 1e-12


@pytest.mark.parametrize("tag,n,first,seed", [("W1", 1, 1000, 101), ("W2", 2, 1000, 102), ("W3", 3, 5000, 103)])
def test_state_warm_start_against_reference(oracle, tag, n, first, seed):
    """experiment(..., warm_start=True) (state.py:116-124): same RNG stream, same stacked POVM, estimators on the
    accumulated data (the stacked tensor is a plain array: dense operand path, 2S then 3S settings)."""
    import quantpy_amd as qp

    g = load_golden("leftovers")
    second = g[tag + "_second"]
    np.random.seed(seed)
    t = qp.StateTomograph(qp.Qobj(g[tag + "_state"]))
    t.experiment(first, "proj-set")
    t.experiment(int(second) if second.ndim == 0 else second, "proj-set", warm_start=True)
    assert np.array_equal(t.results, g[tag + "_results"])
    assert np.array_equal(np.asarray(t.povm_matrix), g[tag + "_povm"])
    assert np.array_equal(t.n_measurements, g[tag + "_nmeas"])
    assert np.abs(t.point_estimate("lin", physical=False).matrix - g[tag + "_lin_unphys"]).max() < 1e-10
    lin = t.point_estimate("lin").matrix
    assert np.abs(lin - g[tag + "_lin"]).max() < 1e-10 and abs(oracle.infidelity(lin, g[tag + "_lin"])) < 1e-10
    mle = t.point_estimate("mle").matrix
    assert t.mle_info["nit"] == int(g[tag + "_mle_nit"])
    assert abs(oracle.infidelity(mle, g[tag + "_mle"])) < 1e-6
    t.experiment(first, "proj-set", warm_start=True)
    assert np.array_equal(t.results, g[tag + "_results3"])
    assert np.array_equal(np.asarray(t.povm_matrix), g[tag + "_povm3"])
    assert np.abs(t.point_estimate("lin").matrix - g[tag + "_lin3"]).max() < 1e-10


@pytest.mark.parametrize("tag,n,p", [("WP1", 1, 0.15), ("WP2", 2, 0.1)])
def test_process_warm_start_against_reference(tag, n, p):
    import quantpy_amd as qp

    g = load_golden("leftovers")
    np.random.seed(200 + n)
    pt = qp.ProcessTomograph(qp.channel.depolarizing(p, n))
    pt.experiment(2000, "proj-set")
    pt.experiment(1000, "proj-set", warm_start=True)
    assert np.array_equal(pt.results, g[tag + "_results"])
    assert np.array_equal(np.asarray(pt.tomographs[0].povm_matrix), g[tag + "_povm"])
    assert np.abs(pt.point_estimate("lifp", cptp=False).choi.matrix - g[tag + "_choi_raw"]).max() < 1e-10
    assert np.abs(pt.point_estimate("lifp", cptp=True).choi.matrix - g[tag + "_choi"]).max() < 1e-10


@pytest.mark.parametrize("tag,n,method,n_points,shots", [("BP2lifp", 2, "lifp", 6, 3000), ("BP1lifp", 1, "lifp", 40, 1000),
                                                         ("BP1states", 1, "states", 24, 1000), ("BP1pgdb", 1, "pgdb", 3, 1000)])
def test_bootstrap_process_interval_against_reference(tag, n, method, n_points, shots):
    """interval.py:615-685 as the reference itself ran it: same resampled counts (RNG order), same distances and
    quantiles, for every estimator the reference's class accepts."""
    import quantpy_amd as qp

    g = load_golden("leftovers")
    cls = g["conf_levels"]
    np.random.seed(300 + n)
    pt = qp.ProcessTomograph(qp.channel.depolarizing(0.1, n))
    pt.experiment(shots, "proj-set")
    assert np.array_equal(pt.results, g[tag + "_counts0"])
    centre = pt.point_estimate(method)
    assert np.abs(centre.choi.matrix - g[tag + "_centre"]).max() < 1e-10
    np.random.seed(5150)
    iv = qp.BootstrapProcessInterval(pt, n_points=n_points, method=method)
    dist, cl = iv(cls)
    assert np.array_equal(iv.boot_counts, g[tag + "_boot_counts"])
    assert np.abs(iv.boot_dist - g[tag + "_boot_dist"]).max() < 1e-9
    assert np.abs(np.asarray(dist) - g[tag + "_cl_dist"]).max() < 1e-9 and np.array_equal(cl, cls)


def test_n5_mle_against_oracle_eight_trials_1e6_shots(oracle):
    """configs[4] size (5 qubits, 'proj-set' 243 x 32, 1e6 shots per setting): eight trials -- six full-rank states, a
    rank-2 and a rank-1 state -- from the 'lin' start, and two of them from the fully mixed start (BFGS iterates),
    against the oracle's restatement of SciPy's BFGS: identical iteration counts, infidelity < 1e-6 (north-star bar)."""
    import quantpy_amd as qp

    n, d = 5, 32
    a = qp.generate_measurement_matrix("proj-set", n)
    ad = np.asarray(a)
    rng = np.random.default_rng(2025)
    states = [_ginibre(rng, d) for _ in range(6)] + [_ginibre(rng, d, rank=2), _ginibre(rng, d, rank=1)]
    np.random.seed(31)
    counts = np.stack([oracle.sample_counts(ad, oracle.bloch_from_matrix(s), 10**6) for s in states])
    eng = qp.get_engine(n)
    eng.set_povm(a, counts[0].sum(-1))
    rho, info = eng.mle(counts, return_info=True)
    assert np.all(info["status"] == 0)
    for c, r, nit in zip(counts, rho, info["nit"]):
        ref, ri = oracle.mle_estimate(c, ad, return_info=True, solver="port")
        assert nit == ri["nit"], (nit, ri["nit"])
        assert abs(oracle.infidelity(ref, r)) < 1e-6
    # at 1e6 shots the projected linear inversion already meets gtol for every one of them (as the reference's
    # own runs do at n = 3); from the fully mixed start BFGS has to walk the whole way
    pick = [0, 7]
    rho_m, info_m = eng.mle(counts[pick], init="mixed", return_info=True)
    assert np.all(info_m["status"] == 0) and np.all(info_m["nit"] > 5)
    for c, r, nit in zip(counts[pick], rho_m, info_m["nit"]):
        ref, ri = oracle.mle_estimate(c, ad, init="mixed", return_info=True, solver="port")
        assert nit == ri["nit"], (nit, ri["nit"])
        assert abs(oracle.infidelity(ref, r)) < 1e-6


def test_bootstrap_2000_resamples_properties():
    """configs[3] at its full size through the drop-in class: 2000 'mle' resamples.  Size-independent properties:
    finite, non-negative, sorted quantile table, invariant under a permutation of the resamples, equal to the
    engine's own batch on the recorded counts, and centred where 24 reference resamples of the same law sit."""
    import quantpy_amd as qp

    g = load_golden("bootstrap")
    np.random.seed(7)
    t = qp.StateTomograph(qp.Qobj(g["B3mle_true"]))
    t.experiment(100000, "proj-set")
    assert np.array_equal(t.results, g["B3mle_counts0"])
    centre = t.point_estimate("mle")
    np.random.seed(4242)
    iv = qp.BootstrapStateInterval(t, n_points=2000, method="mle")
    dist, cl = iv([0.5, 0.9, 0.95])
    d_all = iv.boot_dist
    assert d_all.shape == (2000,) and np.all(np.isfinite(d_all)) and np.all(d_all >= 0)
    assert np.array_equal(iv.boot_counts[:24], g["B3mle_boot_counts"])  # the same stream as the reference's first 24
    assert np.abs(d_all[:24] - g["B3mle_boot_dist"]).max() < 2e-5
    grid = iv.cl_to_dist.y
    assert np.all(np.diff(grid) >= 0) and np.array_equal(np.sort(d_all), grid)
    assert dist[0] <= dist[1] <= dist[2]
    eng = t._engine()
    perm = np.random.default_rng(0).permutation(2000)
    rho_p = eng.mle(iv.boot_counts[perm])
    d_p = eng.hs_dist(rho_p, centre.matrix)
    assert np.array_equal(np.sort(d_p), grid)
    srt, q = eng.sort_quantiles(d_all, [0.5, 0.9, 0.95])  # device sort + interp1d semantics
    assert np.array_equal(srt, grid) and np.abs(q - np.asarray(dist)).max() < 1e-16
    ref24 = np.median(g["B3mle_boot_dist"])
    assert 0.5 * ref24 < np.median(d_all) < 2 * ref24

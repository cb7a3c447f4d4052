"""GPU: process tomography of THREE qubits (64 input states x 216 POVM rows, 64 x 64 Choi matrix) against fixtures the
reference itself produced (tests/golden/process3.npz, make_golden.py:gen_process3 -- there the 13824 x 4096 complex
design matrix and its left inverse are built densely; here the design matrix stays Kronecker-factored,
csrc/qt_process64.h)."""
import numpy as np
import pytest
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g3():
    return load_golden("process3")


def _tomograph(qp, g, key):
    np.random.seed(int(g[key + "_seed"]))
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 3))
    tmg.experiment(int(g[key + "_shots"]), "proj-set")
    return tmg


def test_n3_experiment_is_bit_exact(g3):
    import quantpy_amd as qp

    tmg = _tomograph(qp, g3, "Q0")
    assert np.abs(tmg.channel.choi.matrix - g3["Q_true_choi"]).max() < 1e-14
    assert np.abs(np.stack([s.matrix for s in tmg.input_basis.elements]) - g3["Q_input_states"]).max() < 1e-15
    assert np.array_equal(tmg.results, g3["Q0_counts"])  # 64 x 27 multinomials on the reference's stream


def test_n3_factored_left_inverse_is_the_references(g3):
    """Sampled columns of the reference's 4096 x 13824 `_lifp_oper_inv` against the Kronecker product of the two
    factor inverses (include/qtomo.h: qt_process_get_factors), and sampled rows of `_lifp_oper` against the factors
    themselves."""
    import quantpy_amd as qp
    from quantpy_amd import _capi

    tmg = _tomograph(qp, g3, "Q0")
    eng = tmg._engine()
    d, D, M = 8, 64, 216
    vs = np.empty((D, D), dtype=np.complex128)
    vp = np.empty((D, M), dtype=np.complex128)
    assert eng.lib.qt_process_get_factors(eng._h, vs.ctypes.data, vp.ctypes.data, _capi.QT_HOST_PTR) == 0
    cols = np.arange(0, D * M, 1999)
    want = g3["Q_lifp_oper_inv_cols"]  # [4096][len(cols)], row index v = col * D + row of the Choi matrix
    a, b, c, e = np.meshgrid(np.arange(d), np.arange(d), np.arange(d), np.arange(d), indexing="ij")
    v = (c * d + e) * D + (a * d + b)
    got = np.empty_like(want)
    for k, col in enumerate(cols):
        s, m = divmod(int(col), M)
        got[v.ravel(), k] = (vs[(a * d + c).ravel(), s] * vp[(e * d + b).ravel(), m])
    assert np.abs(got - want).max() < 1e-9 * np.abs(want).max()
    with pytest.raises(RuntimeError):
        eng.process_operators()  # the dense form is not materialised at n = 3


@pytest.mark.parametrize("key", ["Q0", "Q1"])
def test_n3_lifp_and_projections_match_the_reference(g3, key):
    import quantpy_amd as qp

    tmg = _tomograph(qp, g3, key)
    assert np.array_equal(tmg.results, g3[key + "_counts"])
    raw = tmg.point_estimate("lifp", cptp=False)
    assert np.abs(raw.choi.matrix - g3[key + "_choi_nocptp"]).max() < 1e-10
    fixed = tmg.point_estimate("lifp", cptp=True)
    assert tmg.cptp_iterations == int(g3[key + "_dykstra_iters"])
    assert np.abs(fixed.choi.matrix - g3[key + "_choi_cptp"]).max() < 1e-9
    assert fixed.is_cptp(verbose=False)
    ref_raw = qp.Channel(g3[key + "_choi_nocptp"])
    assert np.abs(tmg.tp_projection(ref_raw).choi.matrix - g3[key + "_tp_only"]).max() < 1e-12
    assert np.abs(tmg.cp_projection(ref_raw).choi.matrix - g3[key + "_cp_only"]).max() < 1e-11
    assert np.abs(tmg.cptp_projection(ref_raw).choi.matrix - g3[key + "_choi_cptp"]).max() < 1e-9


def test_n3_states_estimator_and_batch(g3):
    import quantpy_amd as qp

    tmg = _tomograph(qp, g3, "Q0")
    st = tmg.point_estimate("states", cptp=False)
    assert np.abs(st.choi.matrix - g3["Q0_states_lin"]).max() < 1e-9
    st = tmg.point_estimate("states", cptp=True)
    assert np.abs(st.choi.matrix - g3["Q0_states_lin_cptp"]).max() < 1e-8
    # a batch: every process is reconstructed independently of its neighbours
    counts = np.stack([g3["Q0_counts"], g3["Q1_counts"], g3["Q0_counts"]])
    eng = tmg._engine()
    choi, iters = eng.lifp(counts, cptp=True, return_iters=True)
    assert list(iters) == [int(g3["Q0_dykstra_iters"]), int(g3["Q1_dykstra_iters"]), int(g3["Q0_dykstra_iters"])]
    assert np.array_equal(choi[0], choi[2])
    assert np.abs(choi[1] - g3["Q1_choi_cptp"]).max() < 1e-9
    # a positive-definite Choi matrix is left alone by the CP step (Cholesky short cut), a clipped one is PSD
    assert np.abs(eng.cptp_project(g3["Q_true_choi"] + 1e-3 * np.eye(64), mode="cp") - (g3["Q_true_choi"] + 1e-3 * np.eye(64))).max() < 1e-15
    assert np.linalg.eigvalsh(eng.cptp_project(g3["Q0_choi_nocptp"], mode="cp")).min() > 0
    # a chain step whose proposal is the current point itself is accepted for every u <= 1 (exp(0) = 1)
    chain, accepted = eng.mhmc_process(g3["Q0_counts"], choi[0], np.zeros((2, 4096)), np.array([1.0, 0.3]), 0.01)
    assert list(accepted) == [1, 1] and np.abs(chain - choi[0]).max() < 1e-9


def test_n3_bootstrap_process_interval_runs(g3):
    """BootstrapProcessInterval on three qubits: resamples in the reference's stream order (one sampler call), batched
    'lifp' + CPTP, sorted finite distances; the same seed gives the same interval."""
    import quantpy_amd as qp

    tmg = _tomograph(qp, g3, "Q0")
    tmg.point_estimate("lifp")
    out = []
    for _ in range(2):
        np.random.seed(5)
        iv = qp.BootstrapProcessInterval(tmg, n_points=6)
        radii, _ = iv([0.1, 0.5, 0.9])
        assert iv.boot_counts.shape == (6, 64, 27, 8) and np.all(iv.boot_counts.sum(-1) == 10000)
        assert np.all(np.isfinite(radii)) and np.all(np.diff(radii) >= 0) and radii[0] > 0
        out.append(radii)
    assert np.array_equal(out[0], out[1])


def test_n3_process_without_counts_is_nan_and_alone(g3):
    """An input state without counts makes that process's frequencies 0 / 0 (process.py:285): its Choi matrix is NaN,
    with and without the projection, and its neighbours in the batch are untouched."""
    import quantpy_amd as qp

    tmg = _tomograph(qp, g3, "Q0")
    eng = tmg._engine()
    counts = np.stack([g3["Q0_counts"], g3["Q0_counts"], g3["Q1_counts"]])
    counts[1, 17] = 0
    good = eng.lifp(counts[[0, 2]], cptp=False)
    raw = eng.lifp(counts, cptp=False)
    assert np.isnan(raw[1]).all() and np.array_equal(raw[0], good[0]) and np.array_equal(raw[2], good[1])
    fixed, iters = eng.lifp(counts, cptp=True, return_iters=True)
    assert np.isnan(fixed[1]).all() and not np.isnan(fixed[[0, 2]]).any()
    assert iters[0] == int(g3["Q0_dykstra_iters"]) and iters[2] == int(g3["Q1_dykstra_iters"])


@pytest.mark.parametrize("nq", [3, 2])
def test_cptp_projection_regimes_against_eigh_dykstra(oracle, nq):
    """k_cptp_project64 (n = 3) and k_cptp_wave16 (n = 2: one wavefront per process, the matrix in the registers of the
    matrix-core tile) on Choi matrices the fixtures do not reach -- already CPTP, CP but not TP, TP but far from CP,
    a unitary channel (rank one) with noise, large noise -- against a NumPy Dykstra loop with eigh (process.py:237-278;
    its TP step checked once against the oracle's operator form): same iteration counts, Choi to 1e-9."""
    import quantpy_amd as qp

    d = 2**nq
    dc = d * d
    rng = np.random.default_rng(dc)
    eye_d = np.eye(d)

    def tp(c):
        red = np.einsum("aobo->ab", c.reshape(d, d, d, d))
        return c + np.kron((eye_d - red) / d, eye_d)

    def cp(c):
        w, u = np.linalg.eigh(np.tril(c) + np.tril(c, -1).conj().T)
        return (u * np.maximum(w, 1e-12)) @ u.conj().T

    def dykstra(c, n_iter=1000, tol=1e-12):
        x = c.astype(np.complex128)
        p = q = y = np.zeros_like(x)
        for it in range(n_iter):
            yd = tp(x + p) - y
            y = y + yd
            xd = cp(y + q) - x
            x = x + xd
            crit = 2 * (abs(np.sum(yd.conj() * q)) + abs(np.sum(xd.conj() * p)))
            pd, qd = x - y, y - x
            p, q = p + pd, q + qd
            crit += np.linalg.norm(pd) ** 2 + np.linalg.norm(qd) ** 2
            if crit < tol:
                break
        return x, it + 1

    def herm(scale):
        g = rng.standard_normal((dc, dc)) + 1j * rng.standard_normal((dc, dc))
        return scale * (g + g.conj().T) / 2

    def random_cptp(rank):
        k = rng.standard_normal((rank, d, d)) + 1j * rng.standard_normal((rank, d, d))
        s = sum(a.conj().T @ a for a in k)
        w, u = np.linalg.eigh(s)
        k = k @ ((u / np.sqrt(w)) @ u.conj().T)  # sum K^dagger K = 1
        v = [a.T.reshape(-1) for a in k]  # v[(in, out)] = K[out][in]
        return sum(np.outer(x, x.conj()) for x in v)  # C[(a, o), (b, o')] = sum_k K[o][a] conj(K[o'][b])

    base = random_cptp(3)
    assert np.abs(np.einsum("aobo->ab", base.reshape(d, d, d, d)) - eye_d).max() < 1e-12  # the construction is TP in this layout
    some = herm(0.05)
    assert np.abs(tp(some) - oracle.vec2mat(oracle.tp_projection_vec(oracle.mat2vec(some), nq))).max() < 1e-12
    cases = {
        "already CPTP": random_cptp(dc),
        "CP, not TP": 1.3 * random_cptp(d),
        "unitary channel + noise": random_cptp(1) + herm(0.01),
        "rank 3 + large noise": base + herm(0.2),
        "indefinite": herm(1.0),
    }
    eng = qp.get_engine(nq)
    batch = np.stack(list(cases.values()))
    got, iters = eng.cptp_project(batch, mode="cptp", return_iters=True)
    for (name, c), g, it in zip(cases.items(), got, iters):
        want, want_it = dykstra(c)
        assert int(it) == want_it, (name, int(it), want_it)
        assert np.abs(g - want).max() < 1e-9, (name, np.abs(g - want).max())
        assert np.linalg.eigvalsh(g).min() > -1e-9 and np.abs(np.einsum("aobo->ab", g.reshape(d, d, d, d)) - eye_d).max() < 1e-5


def test_n3_pgdb_matches_the_reference():
    """'pgdb' at n = 3 (process.py:291-314) through the factored design matrix (k_pgdb64_grad / k_cptp_project64 /
    k_pgdb64_step) against the reference's own run on the dense 13824 x 4096 operator (make_golden.py:gen_pgdb3): what it
    returns (it leaves inside its first iteration, at the fully mixed start), and its loop with the step accepted --
    every iterate of three iterations, i.e. gradient, Dykstra projection, backtracking step and NLL of each."""
    import quantpy_amd as qp

    g = load_golden("pgdb3")
    np.random.seed(31)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 3))
    tmg.experiment(10000, "proj-set")
    assert np.array_equal(tmg.results, g["R0_counts"])
    ch = tmg.point_estimate("pgdb", n_iter=2)
    assert isinstance(ch, qp.Channel)
    assert np.abs(ch.choi.matrix - g["R0_returned"]).max() < 1e-14
    assert tmg.pgdb_iterations == 2  # (its backtracking ends at a step of 2^-54, which changes nothing: it never leaves)
    eng = tmg._engine()
    # the pieces of the first iteration against the reference's dense operator (qt_pgdb_pieces)
    start = np.eye(64) / 64
    probas, grad, proj = eng.pgdb_pieces(tmg.results, start)
    want_p, want_g = g["R0_it0_probas"], g["R0_it0_grad"].reshape(64, 64).T  # column-stacked vector -> matrix
    assert np.abs(want_p.imag).max() < 1e-12 and np.abs(probas - want_p.real).max() < 1e-13 * np.abs(want_p).max() + 1e-15
    assert np.abs(grad - want_g).max() < 1e-11 * np.abs(want_g).max()
    # The trial point c - g / mu is ~1e9 in size and far outside the CPTP set: Dykstra's 1000 iterations do not
    # converge from there (neither the reference's nor these), the result is the last CP step -- a clip of a 1e9-sized
    # matrix whose surviving eigenvalues are its rounding noise.  Agreement is therefore asked for at that noise level.
    want_d = g["R0_it0_direction"].reshape(64, 64).T
    trial_size = np.abs(start - grad / (1.5 / 64)).max()
    assert trial_size > 1e6
    assert np.abs((proj - start) - want_d).max() < 64 * 2.3e-16 * trial_size
    dot = np.dot((proj - start).T.reshape(-1), grad.T.reshape(-1))
    assert abs(dot - g["R0_it0_dot"]) < 1e-6 * abs(g["R0_it0_dot"])
    iterates, trace = g["R0_conv_iterates"], g["R0_conv_trace"]
    for k in range(len(iterates)):
        got, iters = eng.pgdb(tmg.results, n_iter=k + 1, stop="converged", return_iters=True)
        assert iters == k + 1
        err = np.abs(got - iterates[k]).max()
        assert err < 1e-9, (k, err)
        assert np.abs(np.trace(got) - 1) < 1e-9 and np.linalg.eigvalsh(got).min() > -1e-9
    # a batch: independent loop states; the reference's stop rule and the converging one side by side
    batch = eng.pgdb(np.stack([tmg.results, g["R0_counts"][::-1].copy(), tmg.results]), n_iter=2, stop="converged")
    assert np.array_equal(batch[0], batch[2]) and np.abs(batch[0] - iterates[-1]).max() < 1e-9
    assert np.all(np.isfinite(batch[1]))
    assert np.abs(eng.pgdb(tmg.results, n_iter=0) - np.eye(64) / 64).max() == 0.0
    # the NLL the line search saw (gen_pgdb3 records the reference's f0, f1 and alpha per step): recomputed from the iterates
    alphas = trace[:, 2].real
    assert np.all((alphas > 0) & (alphas <= 1))


def test_n3_mhmc_process_interval_matches_the_reference():
    """MHMCProcessInterval at n = 3 (interval.py:763-850) through the factored chain kernels (k_mhmc64_propose /
    k_cptp_project64 / k_mhmc64_accept) against two short chains the reference ran on its dense operator
    (make_golden.py:gen_mhmc3): same proposals and uniforms from the same seeds, the samples, the sorted distances and
    the acceptance rate."""
    import quantpy_amd as qp

    g = load_golden("mhmc3")
    np.random.seed(31)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 3))
    tmg.experiment(10000, "proj-set")
    assert np.array_equal(tmg.results, g["M_counts"])
    ch = tmg.point_estimate("lifp")
    assert np.abs(ch.choi.matrix - g["M_channel"]).max() < 1e-9
    for key in ("M0", "M1"):
        n_points, burn = (int(v) for v in g[key + "_args"])
        np.random.seed(int(g[key + "_seed"]))
        iv = qp.MHMCProcessInterval(tmg, n_points=n_points, step=float(g[key + "_step"]), burn_steps=burn, return_samples=True)
        dist, cl, rate, mats = iv.setup()
        assert abs(rate - float(g[key + "_rate"])) < 1e-12, (key, rate, float(g[key + "_rate"]))
        assert np.abs(np.stack(mats) - g[key + "_samples"]).max() < 1e-8, key
        assert np.abs(dist - g[key + "_dist"]).max() < 1e-8, key


def test_n3_pgdb_pieces_at_general_points(g3):
    """The factored model and gradient away from the fully mixed start: (a) probabilities at a random Choi matrix
    against rows of the reference's dense 13824 x 4096 operator (golden `Q_lifp_oper_rows`, every 997th row);
    (b) the adjoint identity <L d, w> = -<g, d> with g = -L^H w from one call and L d from another -- model and
    gradient kernels are each other's transposes to rounding."""
    import quantpy_amd as qp

    tmg = _tomograph(qp, g3, "Q0")
    tmg.point_estimate("lifp", cptp=False)
    eng = tmg._engine()
    rng = np.random.default_rng(17)

    def random_choi():
        k = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
        c = k @ k.conj().T
        return c / np.trace(c) * 8

    c, dd = random_choi(), random_choi()
    p_c, g_c, _ = eng.pgdb_pieces(tmg.results, c)
    rows = g3["Q_lifp_oper_rows"]  # [14][4096], column v = col * 64 + row of the Choi matrix
    want = (rows @ c.T.reshape(-1)).real
    assert np.abs(p_c[::997] - want).max() < 1e-13 * max(1.0, np.abs(want).max())
    p_d, _, _ = eng.pgdb_pieces(tmg.results, dd)
    w = tmg.results.reshape(-1) / p_c
    lhs = float(np.dot(p_d, w))
    rhs = -float(np.real(np.sum(np.conj(g_c) * dd)))
    assert abs(lhs - rhs) < 1e-11 * abs(lhs), (lhs, rhs)


@pytest.mark.parametrize("pieces", [1, 3, 5, 9])
def test_n3_lifp_both_kernel_paths_against_the_factor_formula(g3, pieces):
    """qt_lifp_batch at n = 3 for POVMs with M % 4 == 0 (k_lifp64: both products of a process in one matrix-core kernel;
    M / 4 a multiple of its look-ahead chunk or not) and M % 4 != 0 (k_lifp_freq + k_gemm + k_lifp_kron_finish), against
    X = V_S^+ F V_P^+^T and Choi[(a, b)][(c, e)] = X[(a, c)][(e, b)] evaluated in NumPy with the engine's own factors
    (which test_n3_factored_left_inverse_is_the_references pins to the reference for the 'proj-set' POVM).  The POVM: all
    216 'proj-set' elements as ONE measurement (each / 27), its last element split into `pieces` equal parts --
    M = 216 (54 k-steps, no tail), 218 (fallback), 220 (55 k-steps: one in the tail loop), 224 (56: two)."""
    import quantpy_amd as qp
    from quantpy_amd import _capi

    tmg = _tomograph(qp, g3, "Q0")
    first = tmg.tomographs[0]
    rows = np.asarray(first.povm_matrix, dtype=np.float64).reshape(-1, 64) / 27.0
    rows = np.concatenate([rows[:-1]] + [rows[-1:] / pieces] * pieces)
    M = rows.shape[0]
    assert M == 215 + pieces
    eng = qp.get_engine(3)
    eng.set_povm(rows[None], 10000 * 27)
    eng.process_setup(np.stack([np.asarray(s.matrix, dtype=np.complex128) for s in tmg.input_basis.elements]))
    d, D = 8, 64
    vs = np.empty((D, D), dtype=np.complex128)
    vp = np.empty((D, M), dtype=np.complex128)
    assert eng.lib.qt_process_get_factors(eng._h, vs.ctypes.data, vp.ctypes.data, _capi.QT_HOST_PTR) == 0
    rng = np.random.default_rng(pieces)
    B = 5
    counts = rng.integers(0, 4000, size=(B, D, 1, M)).astype(np.int64)
    counts[3, :, 0, ::7] = 0  # zeros among the counts are ordinary numbers
    got, iters = eng.lifp(counts, cptp=False, return_iters=True)
    assert list(iters) == [0] * B
    a, b, c, e = np.meshgrid(np.arange(d), np.arange(d), np.arange(d), np.arange(d), indexing="ij")
    for k in range(B):
        f = counts[k, :, 0, :] / counts[k, :, 0, :].sum(axis=1, keepdims=True)
        x = vs @ f @ vp.T
        want = np.empty((D, D), dtype=np.complex128)
        want[(a * d + b).ravel(), (c * d + e).ravel()] = x[(a * d + c).ravel(), (e * d + b).ravel()]
        assert np.abs(got[k] - want).max() < 1e-11 * max(1.0, np.abs(want).max()), (pieces, k, np.abs(got[k] - want).max())
    # one process of a batch alone gives the same bits, and an input state without counts is NaN for its process only
    assert np.array_equal(eng.lifp(counts[2], cptp=False), got[2])
    counts[1, 40] = 0
    again = eng.lifp(counts, cptp=False)
    assert np.isnan(again[1]).all() and np.array_equal(again[[0, 2, 3, 4]], got[[0, 2, 3, 4]])


def test_n3_cp_step_on_hard_spectra():
    """The CP step of k_cptp_project64 (sign-function clip: degree-5 lifting, Newton-Schulz finish) against an eigh-based
    clip (process.py:270-277) on spectra the fixtures do not have: eigenvalues down to 1e-10 of the norm with both signs,
    exact zeros (rank-deficient, where sign(0) = 0 leaves eps / 2 instead of eps: 5e-13), negative definite, a multiple
    of the identity, one dominant eigenvalue over a tiny bulk, and the zero matrix."""
    import quantpy_amd as qp

    rng = np.random.default_rng(2024)
    g = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
    q, _ = np.linalg.qr(g)
    alt = np.where(np.arange(64) % 2, 1.0, -1.0)
    spectra = {
        "geometric 1e-10..1, alternating": np.geomspace(1e-10, 1.0, 64) * alt,
        "geometric 1e-6..1, all negative but one": -np.geomspace(1e-6, 1.0, 64) * np.where(np.arange(64) == 63, -1.0, 1.0),
        "rank 32 (exact zeros)": np.concatenate([np.zeros(32), np.linspace(-1.0, 1.0, 32)]),
        "negative definite": -np.linspace(0.1, 1.0, 64),
        "-0.3 I": -0.3 * np.ones(64),
        "one dominant eigenvalue, bulk +-1e-7": np.concatenate([[1.0], 1e-7 * alt[1:]]),
        "clustered at +-1e-3": 1e-3 * alt * (1 + 1e-9 * np.arange(64)),
    }
    mats = []
    for ev in spectra.values():
        a = (q * ev) @ q.conj().T
        mats.append((a + a.conj().T) / 2)
    mats.append(np.zeros((64, 64), dtype=np.complex128))
    batch = np.stack(mats)
    eng = qp.get_engine(3)
    got, steps = eng.cptp_project(batch, mode="cp", return_iters=True)
    for name, a, r, st in zip(list(spectra) + ["zero matrix"], batch, got, steps):
        w, u = np.linalg.eigh(a)
        want = (u * np.maximum(w, 1e-12)) @ u.conj().T
        scale = max(np.linalg.norm(a), 1e-12)
        assert np.abs(r - want).max() < 2e-12 * max(scale, 1.0) + 1e-12, (name, np.abs(r - want).max(), int(st))
        assert np.abs(r - r.conj().T).max() == 0.0, name  # exactly Hermitian
        assert int(st) <= 40, (name, int(st))


def test_n2_cp_step_on_rank_deficient_and_tiny_spectra():
    """The 16 x 16 CP step (k_cptp_project<16>, SignClipWG<16>) on what test_n3_cp_step_on_hard_spectra gives the 64 x 64 one:
    the Choi matrix of a unitary channel (rank one, exact zeros), eigenvalues down to 1e-10 of the norm with both signs."""
    import quantpy_amd as qp

    rng = np.random.default_rng(16)
    g = rng.standard_normal((16, 16)) + 1j * rng.standard_normal((16, 16))
    q, _ = np.linalg.qr(g)
    alt = np.where(np.arange(16) % 2, 1.0, -1.0)
    u4, _ = np.linalg.qr(rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)))
    v = u4.T.reshape(-1)
    mats = [np.outer(v, v.conj())]  # Choi matrix of rho -> U rho U^dagger
    for ev in (np.geomspace(1e-10, 1.0, 16) * alt, np.concatenate([np.zeros(8), np.linspace(-1.0, 1.0, 8)]),
               np.concatenate([[1.0], 1e-7 * alt[1:]])):
        a = (q * ev) @ q.conj().T
        mats.append((a + a.conj().T) / 2)
    batch = np.stack(mats)
    got = qp.get_engine(2).cptp_project(batch, mode="cp")
    for a, r in zip(batch, got):
        w, u = np.linalg.eigh(a)
        want = (u * np.maximum(w, 1e-12)) @ u.conj().T
        assert np.abs(r - want).max() < 3e-12 * max(np.linalg.norm(a), 1.0) + 1e-12, np.abs(r - want).max()

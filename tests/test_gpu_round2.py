"""GPU: the round-1 review items -- the engine leaves the caller's device and stream order alone, the process
set-up cache follows POVM re-uploads, the POVM guard sees permutations, device sort + quantiles (a16)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_engine_calls_leave_current_device_and_follow_torch_stream(oracle):
    """ADVICE r1 (high, low): an engine call must not move torch's current device, and device-pointer calls
    are ordered with the torch work around them without an explicit sync (the engine binds to torch's
    current stream on its first device-pointer call)."""
    import torch

    import quantpy_amd as qp
    from quantpy_amd.engine import Engine

    before = torch.cuda.current_device()
    eng = Engine(2)  # default device = torch's current device
    assert eng.device == before
    a = qp.generate_measurement_matrix("proj-set", 2)
    eng.set_povm(a, np.full(9, 1000.0))
    assert torch.cuda.current_device() == before
    rng = np.random.default_rng(2)
    g = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
    rho = g @ g.conj().T
    rho /= np.trace(rho)
    np.random.seed(3)
    counts = np.stack([oracle.sample_counts(np.array(a), oracle.bloch_from_matrix(rho), 1000) for _ in range(4096)])
    want = eng.lin(counts)
    cd = torch.from_numpy(counts).cuda()
    for _ in range(3):
        # producer (a large fill + copy), engine kernel, consumer: all on torch's current stream, no sync between
        scratch = torch.zeros((4096, 9, 4), dtype=torch.int64, device="cuda")
        scratch.copy_(cd)
        out = torch.full((4096, 4, 4), 7.0, dtype=torch.complex128, device="cuda")
        eng.lin_dev(scratch, out)
        total = out.sum()
        assert np.array_equal(out.cpu().numpy(), want)
        assert abs(total.item() - want.sum()) < 1e-9
    assert torch.cuda.current_device() == before
    eng.close()


def test_state_tomography_between_process_estimates_keeps_the_process_setup(oracle):
    """A -> B -> A on the shared engine: the second process estimate must redo
    qt_process_setup instead of trusting a key the POVM re-upload invalidated."""
    import quantpy_amd as qp

    np.random.seed(21)
    ptm = qp.ProcessTomograph(qp.channel.depolarizing(0.2, 2))
    ptm.experiment(2000, "proj-set")
    first = ptm.point_estimate("lifp", cptp=False).choi.matrix
    stm = qp.StateTomograph(qp.qobj.GHZ(2))
    stm.experiment(777, "proj-set")  # another shot count: a different POVM key on the same engine
    stm.point_estimate("lin")
    again = ptm.point_estimate("lifp", cptp=False).choi.matrix
    assert np.array_equal(first, again)


def test_in_place_permutation_of_tensor_drops_the_factor(oracle):
    import quantpy_amd as qp

    a = qp.generate_measurement_matrix("proj-set", 2)
    a[0, [0, 1]] = a[0, [1, 0]]  # outcome swap inside one setting: same sum, same sum of squares
    assert a.valid_factor() is None
    eng = qp.get_engine(2)
    shots = np.full(9, 500.0)
    eng.set_povm(a, shots)
    assert not eng.product
    rng = np.random.default_rng(8)
    g = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
    rho = g @ g.conj().T
    rho /= np.trace(rho)
    np.random.seed(4)
    c = oracle.sample_counts(np.array(a), oracle.bloch_from_matrix(rho), 500)
    assert np.abs(eng.lin(c) - oracle.lin_estimate(c, np.array(a))).max() < 1e-10  # reconstructs with the PERMUTED POVM


@pytest.mark.parametrize("n", [1, 2, 7, 2000, 100003])
def test_device_sort_and_quantiles_match_numpy_and_interp1d(n):
    """a16, interval.py:610-612: dist.sort(); interp1d(linspace(0, 1, n), dist)(conf_levels)."""
    import torch
    from scipy.interpolate import interp1d

    import quantpy_amd as qp

    eng = qp.get_engine(1)
    rng = np.random.default_rng(n)
    x = np.abs(rng.standard_normal(n)) * 10.0 ** rng.integers(-12, 3, n)
    x[:: max(1, n // 5)] = 0.0  # exact zeros (hs_dst clamps below 1e-15) and ties
    levels = np.concatenate(([0.0, 1.0, 0.5, 1e-3, 1 - 1e-3], rng.uniform(0, 1, 40)))
    if n > 2:
        levels = np.concatenate((levels, np.linspace(0, 1, n)[[1, n // 2, n - 2]]))  # exactly on grid points
    srt, q = eng.sort_quantiles(x, levels)
    assert np.array_equal(srt, np.sort(x))
    if n > 1:
        want = interp1d(np.linspace(0, 1, n), np.sort(x))(levels)
        assert np.array_equal(q, want), np.abs(q - want).max()
    else:
        assert np.all(q == x[0])
    xd = torch.from_numpy(x).cuda()
    qd = eng.sort_quantiles(xd, levels)
    assert np.array_equal(xd.cpu().numpy(), srt) and np.array_equal(qd, q)
    with pytest.raises(ValueError):
        eng.sort_quantiles(x, [1.5])


def test_mismatched_shots_are_flagged_not_silently_reweighted(oracle):
    """the reference takes N_s / sum N from each trial's own results (state.py:138-141,
    194-197); counts whose per-setting totals are not proportional to the registered shots must raise / carry
    QT_TRIAL_SHOTS, proportional ones (all settings scaled alike) must pass with the same answer."""
    import torch

    import quantpy_amd as qp
    from quantpy_amd import _capi

    for n in (2, 4, 5):
        d = 2**n
        a = qp.generate_measurement_matrix("proj-set", n)
        ad = np.asarray(a)
        rng = np.random.default_rng(n)
        g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        rho = g @ g.conj().T
        rho /= np.trace(rho)
        np.random.seed(5)
        bloch = oracle.bloch_from_matrix(rho)
        good = np.stack([oracle.sample_counts(ad, bloch, 1000) for _ in range(5)])
        eng = qp.get_engine(n)
        eng.set_povm(a, good[0].sum(-1))
        want_lin, want_mle = eng.lin(good), eng.mle(good)
        for setting, outcome in ((good.shape[1] - 1, good.shape[2] - 1), (good.shape[1] // 2, 1), (2, 0)):
            bad = good.copy()
            bad[3, setting, outcome] += 1  # one stray count in one setting of one trial
            for call in (eng.lin, eng.mle):
                with pytest.raises(ValueError, match="per-setting totals"):
                    call(bad)
        st = torch.zeros(5, dtype=torch.int32, device="cuda")
        out = torch.empty((5, d, d), dtype=torch.complex128, device="cuda")
        eng.mle_dev(torch.from_numpy(bad).cuda(), out, status=st)
        eng.sync()
        assert st.cpu().tolist() == [0, 0, 0, _capi.TRIAL_SHOTS, 0]
        assert np.array_equal(out.cpu().numpy()[[0, 1, 2, 4]], want_mle[[0, 1, 2, 4]])  # the other trials are untouched
        doubled = np.stack([oracle.sample_counts(ad, bloch, 2000) for _ in range(2)])  # proportional: same weights
        assert np.abs(eng.lin(doubled)[0] - oracle.lin_estimate(doubled[0], ad)).max() < 1e-10
        assert np.array_equal(eng.lin(good), want_lin)
    t = qp.StateTomograph(qp.Qobj(rho))
    np.random.seed(6)
    t.experiment(1000, "proj-set")
    with pytest.raises(ValueError, match="per-setting totals"):
        t.point_estimate_batch(bad)


def test_product_povm_at_n5_registers_without_dense_operands(oracle):
    """qt_set_povm_product(n = 5) used to allocate 6 x 63.7 MB and assemble A, A^T, A', A'^T that
    the factorised estimators never read.  Now < 100 MB, and the dense operands / left inverse appear on demand."""
    import torch

    import quantpy_amd as qp
    from quantpy_amd.engine import Engine

    torch.cuda.synchronize()
    eng = Engine(5)
    a = qp.generate_measurement_matrix("proj-set", 5)
    free0, _ = torch.cuda.mem_get_info()
    eng.set_povm(a, np.full(243, 1000.0))
    assert eng.product
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 100e6, (free0 - free1) / 1e6
    rng = np.random.default_rng(3)
    g = rng.standard_normal((32, 32)) + 1j * rng.standard_normal((32, 32))
    rho = g @ g.conj().T
    rho /= np.trace(rho)
    bl = oracle.bloch_from_matrix(rho)
    p = eng.born_probs(bl)  # factorised Born kernel, no dense operand
    assert np.abs(p - np.clip(np.einsum("skd,d->sk", np.asarray(a), bl) * 32, 0, 1)).max() < 1e-14
    free2, _ = torch.cuda.mem_get_info()
    assert free0 - free2 < 120e6
    eng.close()
    e4 = Engine(4)
    a4 = qp.generate_measurement_matrix("proj-set", 4)
    ns = np.full(81, 500.0)
    e4.set_povm(a4, ns)
    inv = e4.left_inverse()  # on demand: dense tensor, transposes, Gram GEMM, Gauss-Jordan
    aw = oracle.weighted_povm(np.asarray(a4), ns)
    assert np.abs(inv @ aw - np.eye(256)).max() < 1e-9
    assert np.abs(inv - oracle.left_inv(aw)).max() < 1e-8
    e4.close()


def _three_qubit_mubs():
    """The 9 mutually unbiased bases of 3 qubits as joint eigenbases of 9 disjoint maximal commuting sets of Pauli
    operators (a symplectic spread of the 63 non-identity Pauli strings), found by backtracking."""
    import itertools

    def commute(p, q):  # p, q = (x, z) bit masks
        return (bin(p[0] & q[1]).count("1") + bin(p[1] & q[0]).count("1")) % 2 == 0

    paulis = [(x, z) for x in range(8) for z in range(8) if (x, z) != (0, 0)]
    groups = []
    for a, b, c in itertools.combinations(paulis, 3):  # all maximal abelian subgroups: 3 independent commuting generators
        if not (commute(a, b) and commute(a, c) and commute(b, c)):
            continue
        span = set()
        for ca, cb, cc in itertools.product((0, 1), repeat=3):
            el = (ca * a[0] ^ cb * b[0] ^ cc * c[0], ca * a[1] ^ cb * b[1] ^ cc * c[1])
            span.add(el)
        if len(span) == 8:
            groups.append(frozenset(span - {(0, 0)}))
    groups = sorted(set(groups), key=sorted)

    def search(chosen, used):
        if len(chosen) == 9:
            return chosen
        first = next(p for p in paulis if p not in used)
        for grp in groups:
            if first in grp and not (grp & used):
                got = search(chosen + [grp], used | grp)
                if got:
                    return got
        return None

    spread = search([], frozenset())
    assert spread is not None
    one = [np.eye(2), np.array([[0, 1], [1, 0]]), np.array([[0, -1j], [1j, 0]]), np.array([[1, 0], [0, -1]])]

    def matrix(p):
        m = np.eye(1)
        for q in (2, 1, 0):
            xb, zb = (p[0] >> q) & 1, (p[1] >> q) & 1
            m = np.kron(m, one[(1 if xb and not zb else 2 if xb and zb else 3 if zb else 0)])
        return m

    bases = []
    for grp in spread:
        gens = sorted(grp)[:7]
        rng = np.random.default_rng(5)
        h = sum(rng.standard_normal() * matrix(p) for p in gens)  # generic element of the commuting algebra
        _, vecs = np.linalg.eigh(h)
        bases.append(vecs)
    return bases


def test_true_mub_povm_as_custom_array_n3(oracle):
    """configs[1] says "MUB POVM"; the reference has no MUB generator but accepts a (9, 8, 64) array
    (measurements.py:79-83, SURVEY 8 note +).  Nine mutually unbiased bases of 3 qubits -- NOT a tensor power of a
    one-qubit table -- through experiment / 'lin' / 'mle' on the dense operand path (qt_set_povm)."""
    import quantpy_amd as qp

    bases = _three_qubit_mubs()
    for a, b in ((0, 1), (2, 7), (4, 8)):
        assert np.allclose(np.abs(bases[a].conj().T @ bases[b]) ** 2, 1 / 8, atol=1e-12)  # mutually unbiased
    pauli = oracle.pauli_basis(3)
    povm = np.empty((9, 8, 64))
    for s, vecs in enumerate(bases):
        for o in range(8):
            proj = np.outer(vecs[:, o], vecs[:, o].conj())
            povm[s, o] = np.real(np.einsum("kij,ji->k", pauli, proj)) / 8  # Bloch row: Tr(P_k E) / d
    assert np.allclose(povm.sum(1)[:, 0], 1.0) and np.allclose(povm.sum(1)[:, 1:], 0.0, atol=1e-12)
    rng = np.random.default_rng(9)
    g = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
    rho = g @ g.conj().T
    rho /= np.trace(rho)
    np.random.seed(12)
    t = qp.StateTomograph(qp.Qobj(rho))
    t.experiment(100000, povm)
    assert t.results.shape == (9, 8) and np.all(t.results.sum(-1) == 100000)
    np.random.seed(12)
    assert np.array_equal(t.results, oracle.sample_counts(povm, oracle.bloch_from_matrix(rho), 100000))
    lin = t.point_estimate("lin")
    assert not t._engine().product
    assert np.abs(lin.matrix - oracle.lin_estimate(t.results, povm)).max() < 1e-11
    mle = t.point_estimate("mle")
    ref, ri = oracle.mle_estimate(t.results, povm, return_info=True, solver="port")
    assert t.mle_info["nit"] == ri["nit"] and abs(oracle.infidelity(ref, mle.matrix)) < 1e-6
    ref_s = oracle.mle_estimate(t.results, povm)  # the reference's computation: scipy BFGS + forward differences
    assert abs(oracle.infidelity(ref_s, mle.matrix)) < 1e-6
    assert abs(oracle.infidelity(rho, mle.matrix)) < 1e-3
    np.random.seed(3)
    t.experiment(300, povm)  # low shots: BFGS iterates
    mle = t.point_estimate("mle")
    ref, ri = oracle.mle_estimate(t.results, povm, return_info=True, solver="port")
    assert t.mle_info["nit"] == ri["nit"] and ri["nit"] > 0 and abs(oracle.infidelity(ref, mle.matrix)) < 1e-6


def test_options_and_iteration_limits(oracle):
    """qt_set_option argument checks; max_iter limits of the n = 3 BFGS kernels (LDS holds the two-loop scalars):
    <= 256 iterations in one launch, <= 2000 through the split pair, beyond that a clear error -- and a long run
    (max_iter = 300, tol = 0: every trial iterates until the line search gives up or the cap) agrees between the forms."""
    import quantpy_amd as qp
    from quantpy_amd import _capi

    eng = qp.Engine(3)
    with pytest.raises(qp.EngineError):
        eng.set_option(99, 1)
    with pytest.raises(qp.EngineError):
        eng.set_option(_capi.QT_OPT_MLE_FUSED_MAX_WAVES, -1)
    a = qp.generate_measurement_matrix("proj-set", 3)
    rng = np.random.default_rng(4)
    g = rng.standard_normal((8, 1)) + 1j * rng.standard_normal((8, 1))
    rho = g @ g.conj().T
    rho /= np.trace(rho)
    np.random.seed(8)
    counts = np.stack([oracle.sample_counts(np.asarray(a), oracle.bloch_from_matrix(rho), 200) for _ in range(3)])
    eng.set_povm(a, counts[0].sum(-1))
    with pytest.raises(qp.EngineError) as ei:
        eng.mle(counts, max_iter=2001)
    assert ei.value.code == _capi.QT_ERR_UNSUPPORTED
    r_fused, i_fused = eng.mle(counts, max_iter=40, tol=1e-6, return_info=True)  # > 24 pairs: LDS pairs + global pairs
    eng.set_option(_capi.QT_OPT_MLE_FUSED_MAX_WAVES, 0)
    r_split, i_split = eng.mle(counts, max_iter=40, tol=1e-6, return_info=True)
    r_long, i_long = eng.mle(counts, max_iter=300, tol=1e-6, return_info=True)  # > 256: always the split pair
    eng.set_option(_capi.QT_OPT_MLE_FUSED_MAX_WAVES, 1024)
    assert np.array_equal(i_fused["nit"], i_split["nit"]) and np.array_equal(i_fused["status"], i_split["status"])
    assert np.abs(r_fused - r_split).max() < 1e-12
    for c, r, nit, st in zip(counts, r_fused, i_fused["nit"], i_fused["status"]):
        ref, ri = oracle.mle_estimate(c, np.asarray(a), max_iter=40, tol=1e-6, return_info=True, solver="port")
        assert nit == ri["nit"] and abs(oracle.infidelity(ref, r)) < 1e-6
        assert nit > 24 or st != 3  # the case is meant to go past the LDS-resident pairs
    assert np.all(i_long["nit"] >= i_fused["nit"])
    eng.close()


def test_hs_dist_of_any_square_size_and_split_timer():
    """qt_hs_dist_dim (Choi matrices are 4^n x 4^n, not the handle's 2^n x 2^n) against the reference formula
    sqrt(|Tr((A - B)^2)|) / sqrt(2) (geometry.py:16-20), and the two halves of qt_timer_end."""
    import quantpy_amd as qp

    rng = np.random.default_rng(12)
    eng = qp.get_engine(3)
    for dim in (4, 16, 64):
        a = rng.standard_normal((5, dim, dim)) + 1j * rng.standard_normal((5, dim, dim))
        b = rng.standard_normal((dim, dim)) + 1j * rng.standard_normal((dim, dim))
        want = np.array([np.sqrt(np.abs(np.trace((x - b) @ (x - b)))) / np.sqrt(2) for x in a])
        assert np.abs(eng.hs_dist(a, b) - want).max() < 1e-12 * want.max()
    assert qp.hs_dst(qp.Qobj(np.eye(64) / 64), qp.Qobj(np.eye(64) / 64)) == 0  # a 6-qubit object on the 3-qubit engine
    eng.timer_begin()
    eng.timer_stop()
    assert 0.0 <= eng.timer_elapsed() < 50.0

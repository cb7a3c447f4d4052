"""CPU: host-side logic of the product (no GPU, no compute call into the library):
Qobj conversions in the reference's rounding order, containers, gate zoo, sharding helpers,
and that libqtomo.so loads and exports every symbol include/qtomo.h declares."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
from conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def qp():
    import quantpy_amd

    return quantpy_amd


def test_library_exports_every_declared_symbol():
    from quantpy_amd import _capi

    header = open(os.path.join(ROOT, "include", "qtomo.h")).read()
    declared = set(re.findall(r"\b(qt_[a-z0-9_]+)\s*\(", header))
    declared -= {"qt_handle_t"}
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    from quantpy_amd.build import LIB, build_library

    if not os.path.exists(LIB):  # clean checkout: hipcc cross-compiles without a GPU
        build_library(force=True)
    lib = _capi.load()  # attaches prototypes: AttributeError if a symbol is missing
    for name in declared:
        assert hasattr(lib, name)
    assert lib.qt_version() >= 100


def test_no_gpu_means_loud_failure_not_fallback(qp):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(qp.EngineUnavailable):
        qp.get_engine(1)
    t = qp.StateTomograph(qp.qobj.zero(1))
    np.random.seed(0)
    t.experiment(10000)  # sampling is host logic
    assert t.results.tolist() == [[5002, 4998], [5028, 4972], [10000, 0]]
    with pytest.raises(qp.EngineUnavailable):
        t.point_estimate("lin")


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "quantpy_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(import|from)\s+.*oracle", text, re.M) or "quantpy_oracle" in text:
                    bad.append(f)
    assert not bad, bad


def test_qobj_conversions_bit_exact_with_reference(qp):
    g = load_golden("states_born")
    for n in (1, 2, 3, 4):
        for rho, bloch, back in zip(g[f"rho_n{n}"], g[f"bloch_n{n}"], g[f"rho_from_bloch_n{n}"]):
            assert np.array_equal(qp.Qobj(rho).bloch, bloch)
            assert np.array_equal(qp.Qobj(bloch).matrix, back)
        assert np.array_equal(qp.Qobj(g[f"nonherm_n{n}"]).bloch, g[f"nonherm_bloch_n{n}"])


def test_qobj_container_semantics(qp):
    q = qp.Qobj([0.5, 0, 0, 0.5])
    assert q.n_qubits == 1 and np.allclose(q.matrix, [[1, 0], [0, 0]])
    assert np.allclose(qp.Qobj([0.1, 0.2, 0.3]).bloch, [0.5, 0.1, 0.2, 0.3])  # identity component prepended
    assert np.allclose(qp.Qobj([1, 0], is_ket=True).matrix, [[1, 0], [0, 0]])
    q.matrix = np.eye(2) / 2
    assert np.allclose(q.bloch, [0.5, 0, 0, 0])  # setter invalidated the cached Bloch vector
    q.bloch = [0.5, 0.5, 0, 0]
    assert np.allclose(q.matrix, [[0.5, 0.5], [0.5, 0.5]])
    c = qp.Qobj(q)
    c.matrix[0, 0] = 7
    assert q.matrix[0, 0] == 0.5  # deep copy
    with pytest.raises(ValueError):
        qp.Qobj(np.zeros((2, 2, 2)))
    ghz = qp.qobj.GHZ(3)
    assert ghz.is_pure() and ghz.is_density_matrix() and abs(ghz.impurity()) < 1e-12
    assert np.allclose(ghz.ptrace([0]).matrix, np.eye(2) / 2)
    assert np.allclose(qp.qobj.GHZ(2).ptrace([0, 1]).matrix, qp.qobj.GHZ(2).matrix)
    u, s, vh = qp.qobj.GHZ(2).schmidt()
    assert np.allclose(s, [2**-0.5, 2**-0.5])
    assert not qp.qobj.fully_mixed(2).is_pure()
    with pytest.raises(ValueError):
        qp.qobj.fully_mixed(1).ket()
    assert (q + q) == q * 2 and (q - q) == q * 0 and (q / 2) == q * 0.5 and (-q) == q * -1 and (2 * q) == q * 2
    assert q != q * 3
    with pytest.raises(ValueError):
        q * "a"
    k = q.kron(qp.qobj.zero(1))
    assert k.n_qubits == 2 and np.allclose(k.matrix, np.kron(q.matrix, qp.qobj.zero(1).matrix))
    assert "Quantum object" in repr(q) and "equation" in q._repr_latex_()
    assert np.allclose(qp.qobj.zero(1).T.matrix, qp.qobj.zero(1).matrix)
    assert np.allclose(q.H.matrix, q.matrix.conj().T) and np.allclose(q.conj().matrix, q.matrix.conj())
    assert np.allclose(qp.product(q, q), np.trace(q.matrix @ q.matrix.conj().T))


def test_gates_and_channels(qp):
    op = qp.operator
    for gate in (op.X, op.Y, op.Z, op.H, op.T, op.S, op.CNOT, op.CY, op.CZ, op.SWAP, op.ISWAP, op.MS, op.Toffoli,
                 op.Fredkin, op.RX(0.3), op.RY(0.4), op.RZ(0.5), op.PHASE(0.6)):
        assert np.allclose(gate.matrix @ gate.matrix.conj().T, np.eye(gate.matrix.shape[0]))
    assert np.allclose(op.CNOT.matrix[2:, 2:], op.X.matrix) and np.allclose(op.Toffoli.matrix[6:, 6:], op.X.matrix)
    assert np.allclose(op.Fredkin.matrix[5:7, 5:7], op.X.matrix)
    assert np.allclose((op.H @ op.Z @ op.H).matrix, op.X.matrix)
    assert np.allclose(qp.join_gates([op.X, op.Z]).matrix, op.Z.matrix @ op.X.matrix)
    plus = op.H.transform(qp.qobj.zero(1))
    assert np.allclose(plus.matrix, np.full((2, 2), 0.5))
    ch = op.H.as_channel()
    assert ch.is_cptp(verbose=False) and ch.n_qubits == 1
    assert np.allclose(qp.Channel(ch.choi).transform(qp.qobj.zero(1)).matrix, plus.matrix)  # Choi form
    assert np.allclose(qp.Channel(ch.kraus).transform(qp.qobj.zero(1)).matrix, plus.matrix)  # Kraus form
    dep = qp.channel.depolarizing(0.25, 1)
    out = dep.transform(qp.qobj.zero(1))
    assert np.allclose(out.matrix, [[0.875, 0], [0, 0.125]])
    assert np.allclose(dep.choi.ptrace([0]).matrix, np.eye(2))
    assert np.allclose(qp.channel.dephasing(0.5).transform(plus).matrix, np.eye(2) / 2)
    ad = qp.channel.amplitude_damping(0.3)
    assert ad.is_cptp(verbose=False)
    assert np.allclose(ad.transform(qp.Qobj([0, 1], is_ket=True)).matrix, [[0.3, 0], [0, 0.7]])
    assert qp.channel.walsh_hadamard(2).is_cptp(verbose=False)
    assert np.allclose(qp.channel.depolarize(ch, 1.0).choi.matrix, qp.channel.depolarizing(1, 1).choi.matrix)
    assert not qp.Channel(np.diag([1.0, 0, 0, -1.0])).is_cptp(verbose=False)
    with pytest.raises(ValueError):
        qp.Channel(lambda r: r)  # n_qubits is compulsory for a map
    assert (dep + dep) == dep * 2 and (dep - dep) == dep * 0


def test_basis_and_routines(qp, oracle):
    from quantpy_amd import routines
    from quantpy_amd.tomography.process import _generate_input_states

    states = _generate_input_states("proj4", 1)
    b = qp.basis.Basis(states)
    assert b.dim == 4 and np.allclose(b.gram, b.gram.conj().T)
    target = qp.Qobj(np.array([[0.3, 0.1 - 0.2j], [0.1 + 0.2j, 0.7]]))
    coeffs = b.decompose(target)
    assert np.allclose(b.compose(coeffs).matrix, target.matrix)
    for n in (1, 2):
        assert np.array_equal(routines._out_ptrace_oper(n), oracle.out_ptrace_oper(n))
    m = np.arange(16).reshape(4, 4) + 1j
    assert np.array_equal(routines._vec2mat(routines._mat2vec(m)), m)
    assert np.array_equal(routines._mat2vec(m), oracle.mat2vec(m))
    units = routines.generate_single_entries(2)
    assert len(units) == 4 and units[1][0, 1] == 1 and units[1].sum() == 1
    z = np.arange(6.0)
    assert np.array_equal(routines._complex_to_real(routines._real_to_complex(z)), z)
    assert len(qp.generate_pauli(1)) == 4  # the reference's one-qubit special case: a plain list
    for name in ("proj", "proj-set", "proj4", "sic"):
        assert np.array_equal(qp.generate_measurement_matrix(name, 1), oracle.measurement_matrix(name, 1))
    full = np.ones((5, 16))
    assert qp.generate_measurement_matrix(full, 2).shape == (1, 5, 16)


def test_shard_bounds_cover_and_balance():
    from quantpy_amd.distributed import shard_bounds

    for n in (0, 1, 7, 8, 2000, 2001):
        for ws in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


_GLOO_WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from quantpy_amd import distributed as qd
dist.init_process_group("gloo")
rank, ws = qd.world()
n = int(sys.argv[2])
items = np.arange(n * 3, dtype=np.int64).reshape(n, 3) if rank == 0 else np.zeros((n, 3), dtype=np.int64)
items = qd.broadcast_array(items)            # every rank now holds rank 0's "counts"
seen = []
def fn(shard):
    seen.append(len(shard))
    return shard.sum(axis=1).astype(np.float64) * 0.5
out = qd.sharded_map(items, fn)
want = np.arange(n * 3).reshape(n, 3).sum(axis=1) * 0.5
lo, hi = qd.shard_bounds(n)
assert out.shape == (n,) and np.array_equal(out, want), (rank, out, want)
assert sum(seen) == hi - lo
import torch
full = qd.allgather_device(torch.from_numpy(want[lo:hi].copy()), n)   # the tensor form bench.py uses (gloo: host path)
assert full.shape == (n,) and np.array_equal(full.numpy(), want), (rank, full)
print(f"rank {rank}/{ws} ok {hi - lo}", flush=True)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("n_items", [2000, 7, 1])
def test_sharded_bootstrap_gather_gloo_world2(tmp_path, n_items):
    """the N > 1 path: contiguous shards, one all-gather, identical result on every rank
    (world_size 2, gloo, CPU)."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + n_items % 50), str(script), ROOT, str(n_items)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "rank 0/2 ok" in res.stdout and "rank 1/2 ok" in res.stdout


def test_bench_launcher_starts_n_ranks_and_refuses_a_mislabelled_world():
    """`python bench.py --gpus 2` (the driver's form, no WORLD_SIZE) must start 2 ranks itself -- child
    torchrun -- and the ranks must see a group of 2; a WORLD_SIZE that contradicts --gpus is an error, not a
    1-GPU number with rc 0.  (--launch-check forms the group on gloo / CPU and stops before any GPU call.)"""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"launch_check": True, "n_gpus": 2, "rank_sum": 3}
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode == 2 and "refusing" in bad.stderr


def test_default_device_is_the_ranks_own_gpu(monkeypatch):
    """every API-level caller used get_engine(n) with device 0.  Without an initialised torch GPU
    context the engine key follows LOCAL_RANK (QTOMO_DEVICE overrides); an explicit device wins."""
    from quantpy_amd import engine

    monkeypatch.delenv("QTOMO_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert engine.engine_key(3) == (3, 0)
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert engine.default_device() == 1 and engine.engine_key(3) == (3, 1)
    assert engine.engine_key(3, device=5) == (3, 5)
    monkeypatch.setenv("QTOMO_DEVICE", "2")
    assert engine.engine_key(2) == (2, 2)
    # a launcher that masks every rank down to ONE visible GPU (HIP_VISIBLE_DEVICES) leaves LOCAL_RANK = 3
    # pointing at device 0 -- the default is taken modulo the visible devices, and resolved once per LOCAL_RANK value
    monkeypatch.delenv("QTOMO_DEVICE")

    class OneGpu:
        @staticmethod
        def qt_device_count():
            return 1

    from quantpy_amd import _capi

    monkeypatch.setattr(_capi, "load", lambda: OneGpu)
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert engine.default_device() == 0 and engine.engine_key(3) == (3, 0)
    monkeypatch.setattr(_capi, "load", lambda: type("FourGpus", (), {"qt_device_count": staticmethod(lambda: 4)}))
    monkeypatch.setenv("LOCAL_RANK", "6")
    assert engine.default_device() == 2


def test_povm_tensor_guard_sees_permutations(qp, oracle):
    """sum / sum-of-squares checksums are blind to permutations of exact dyadic entries.  The
    position-weighted digest must drop the factor for every in-place outcome or setting swap that changes the
    tensor (host logic only: PovmTensor is built here from the oracle's tensor, no GPU call)."""
    from quantpy_amd.measurements import _ONE_QUBIT, PovmTensor

    table = _ONE_QUBIT["proj-set"]()
    full = oracle.measurement_matrix("proj-set", 3)
    assert PovmTensor(full.copy(), table).valid_factor() is not None
    changed = 0
    for s in range(27):
        for k1 in range(8):
            for k2 in range(k1 + 1, 8):
                t = PovmTensor(full.copy(), table)
                t[s, [k1, k2]] = t[s, [k2, k1]]
                if not np.array_equal(np.asarray(t), full):
                    changed += 1
                    assert t.valid_factor() is None, (s, k1, k2)
    assert changed == 27 * 28
    for s in range(26):
        t = PovmTensor(full.copy(), table)
        t[[s, s + 1]] = t[[s + 1, s]]
        assert t.valid_factor() is None, s
    t = PovmTensor(full.copy(), table)
    t[3, 2, 5] += 1e-9
    assert t.valid_factor() is None
    assert PovmTensor(full.copy(), table)[1:].valid_factor() is None  # derived arrays carry no factor


def test_oracle_moment_sums_against_reference_radii(oracle):
    """The oracle's restatement of stats.py:21-47 (the reference's einsums term by term) with its CPU left inverse must
    reproduce the reference's MomentInterval radii (tests/golden/moment.npz, written by the reference itself); the GPU
    kernel qt_moment_batch is checked against both in tests/test_gpu_moments.py."""
    g = load_golden("moment")
    for k in range(int(g["n_state_cases"])):
        key = f"S{k}"
        n = int(g[key + "_n"])
        a = oracle.measurement_matrix(str(g[key + "_povm"]), n)
        radii = oracle.moment_radii(g[key + "_counts"], a, g["conf_levels"])
        assert np.allclose(radii, g[key + "_gamma"], rtol=1e-9), key


def test_header_is_plain_c(tmp_path):
    """include/qtomo.h is the drop-in boundary: it has to compile as C99 (no C++ or torch types in the
    signatures) -- tests/host/abi_header_check.c calls a few entry points with plain pointers."""
    import subprocess

    src = os.path.join(ROOT, "tests", "host", "abi_header_check.c")
    out = tmp_path / "abi_header_check.o"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-c", src, "-o", str(out)])
    assert out.exists()


def _numpy_draws(n, p, repeats):
    """The reference's sampling loop itself (state.py:112 inside interval.py:598-604): numpy.random is the checker."""
    return np.asarray([[np.random.multinomial(int(n_s), p_s) for p_s, n_s in zip(p, n)] for _ in range(repeats)])


def test_c_sampler_is_numpys_legacy_multinomial_bit_for_bit():
    """qt_legacy_multinomial (csrc/qt_sampler.h) against numpy.random.multinomial on the same MT19937 state: the
    counts AND the generator state afterwards (so whatever the caller draws next is unchanged too).  The sweep
    covers both binomial branches (inversion at n p <= 30, BTPE above), the p > 0.5 reflection, exact zeros and
    ones in p, n = 0, categories that exhaust n early, and generator positions either side of a 624-word refill."""
    from quantpy_amd.sampling import legacy_multinomial

    rng = np.random.default_rng(1)
    for trial in range(300):
        n_set, n_out = int(rng.integers(1, 6)), int(rng.integers(1, 9))
        kind = trial % 5
        p = rng.random((n_set, n_out)) ** (1 + 3 * (kind == 1))
        if kind == 2:
            p[rng.random((n_set, n_out)) < 0.4] = 0
        p[:, 0] += 1e-9
        p /= p.sum(1, keepdims=True)
        if kind == 4 and n_out > 1:
            p[0] = 0
            p[0, int(rng.integers(0, n_out))] = 1.0
        n = rng.integers(0, [10, 100, 10**4, 10**6, 10**9][trial % 5], n_set)
        repeats = int(rng.integers(1, 30))
        np.random.seed(int(rng.integers(0, 2**31)))
        np.random.rand(int(rng.integers(0, 700)))
        start = np.random.get_state()
        want = _numpy_draws(n, p, repeats)
        want_state = np.random.get_state()
        np.random.set_state(start)
        got = legacy_multinomial(n, p, repeats)
        got_state = np.random.get_state()
        assert got.dtype == np.int64 and np.array_equal(got, want), (trial, n, p)
        assert got_state[2] == want_state[2] and np.array_equal(got_state[1], want_state[1]), trial
        assert got_state[3:] == want_state[3:]  # the cached Gaussian is not ours to touch


def test_c_sampler_rejects_what_numpy_rejects():
    from quantpy_amd.sampling import legacy_multinomial

    np.random.seed(3)
    before = np.random.get_state()
    for bad in ([[0.5, 0.6, 0.1]], [[-0.1, 0.6, 0.5]], [[np.nan, 0.5, 0.5]]):
        with pytest.raises(ValueError):
            np.random.multinomial(10, bad[0])
        with pytest.raises(ValueError):
            legacy_multinomial([10], bad)
    after = np.random.get_state()
    assert after[2] == before[2] and np.array_equal(after[1], before[1])  # nothing drawn


def test_bootstrap_resamples_in_one_call_equal_the_reference_loop(qp, oracle):
    """The 64 resamples tests/golden holds for a 3-qubit bootstrap (drawn by the reference's own
    StateTomograph.experiment loop, make_golden.py:gen_bootstrap) come out of the one-call path: probabilities by
    the reference's einsum, all draws by one qt_legacy_multinomial call."""
    from quantpy_amd.tomography.state import simulate_counts

    g = load_golden("bootstrap")
    povm = oracle.measurement_matrix("proj-set", 3)
    np.random.seed(4242)
    got = simulate_counts(povm, qp.Qobj(g["B3lin_centre"]).bloch, g["B3lin_nmeas"], repeats=g["B3lin_boot_counts"].shape[0])
    assert np.array_equal(got, g["B3lin_boot_counts"])
    np.random.seed(4242)
    one = simulate_counts(povm, qp.Qobj(g["B3lin_centre"]).bloch, g["B3lin_nmeas"])
    assert one.shape == (27, 8) and np.array_equal(one, got[0])


def test_c_sampler_other_routes_to_the_generator(monkeypatch):
    """Same draws when the MT19937 state is copied out and back instead of advanced in place (the route taken if
    NumPy's state layout were not the one checked for), and the reference's own loop when np.random has been given
    a generator that is not MT19937."""
    from quantpy_amd import sampling

    p = np.random.default_rng(5).random((4, 6))
    p /= p.sum(1, keepdims=True)
    n = [1000, 50, 10**6, 7]
    np.random.seed(11)
    want = sampling.legacy_multinomial(n, p, 9)
    want_next = np.random.rand()
    monkeypatch.setattr(sampling, "_mt19937_address", lambda bitgen: None)
    np.random.seed(11)
    assert np.array_equal(sampling.legacy_multinomial(n, p, 9), want) and np.random.rand() == want_next
    monkeypatch.undo()
    if hasattr(np.random, "set_bit_generator"):
        old = np.random.get_bit_generator()
        try:
            np.random.set_bit_generator(np.random.PCG64(3))
            got = sampling.legacy_multinomial(n, p, 2)
            np.random.set_bit_generator(np.random.PCG64(3))
            assert np.array_equal(got, _numpy_draws(n, p, 2))
        finally:
            np.random.set_bit_generator(old)


def test_c_sampler_argument_errors_are_codes_not_crashes():
    """qt_legacy_multinomial is plain host code: bad arguments come back as QT_ERR_ARG with a message, no GPU needed."""
    import ctypes

    from quantpy_amd import _capi

    lib = _capi.load()
    key = np.zeros(624, dtype=np.uint32)
    pos = ctypes.c_int(624)
    n = np.array([5], dtype=np.int64)
    p = np.array([[0.5, 0.5]])
    out = np.zeros((1, 2), dtype=np.int64)
    ok = (key.ctypes.data, ctypes.byref(pos), 1, 1, n.ctypes.data, p.ctypes.data, 2, out.ctypes.data)
    assert lib.qt_legacy_multinomial(*ok) == 0 and out.sum() == 5
    for bad in ((None,) + ok[1:], ok[:2] + (-1,) + ok[3:], ok[:3] + (0,) + ok[4:], ok[:6] + (0,) + ok[7:], ok[:7] + (None,)):
        assert lib.qt_legacy_multinomial(*bad) == -1 and _capi.last_error()
    pos.value = 700
    assert lib.qt_legacy_multinomial(*ok) == -1 and "624" in _capi.last_error()
    pos.value = 0
    n[0] = -3
    assert lib.qt_legacy_multinomial(*ok) == -1


def test_philox4x32_10_known_answers():
    """The block function of the opt-in device sampler (csrc/qt_sampler.h) against the known-answer vectors published
    with the algorithm (Random123 kat_vectors, philox4x32 with 10 rounds): zero, all-ones and the digits of pi."""
    from quantpy_amd import _capi

    lib = _capi.load()
    cases = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in cases:
        c, k, out = np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32), np.zeros(4, dtype=np.uint32)
        lib.qt_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
        assert out.tolist() == want, [hex(v) for v in out]


def test_chain_proposal_increments_are_the_references_draws():
    """interval._proposal_increments: above dim 256 the draws skip scipy's frozen multivariate_normal (identity
    covariance: an eigen-decomposition and an SVD that return the identity) -- same numbers, same stream position."""
    from scipy.stats import multivariate_normal

    from quantpy_amd.tomography.interval import _proposal_increments

    for dim in (16, 320):
        np.random.seed(dim)
        want = multivariate_normal(mean=np.zeros(dim)).rvs(size=7).reshape(7, dim)
        after_want = np.random.rand()
        np.random.seed(dim)
        got = _proposal_increments(dim)(7)
        assert np.array_equal(got, want) and np.random.rand() == after_want
        np.random.seed(dim + 1)
        one = _proposal_increments(dim)(1)
        np.random.seed(dim + 1)
        assert np.array_equal(one, multivariate_normal(mean=np.zeros(dim)).rvs(size=1).reshape(1, dim))

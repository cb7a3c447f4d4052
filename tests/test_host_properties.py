"""CPU: property tests (hypothesis) of the host containers against plain NumPy and against the oracle
on random inputs -- the golden files pin a handful of states, these cover the space around them.
No compute call into the library."""
import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st


@pytest.fixture(scope="module")
def qp():
    import quantpy_amd

    return quantpy_amd


def _random_state(seed, n_qubits, rank=None):
    rng = np.random.RandomState(seed)
    d = 2**n_qubits
    r = rank or d
    g = rng.randn(d, r) + 1j * rng.randn(d, r)
    rho = g @ g.conj().T
    return rho / np.trace(rho).real


def _random_kraus(seed, n_qubits, n_ops):
    """A CPTP map from a random isometry: V (d*n_ops x d), V^dagger V = 1, cut into n_ops blocks."""
    rng = np.random.RandomState(seed)
    d = 2**n_qubits
    g = rng.randn(d * n_ops, d) + 1j * rng.randn(d * n_ops, d)
    v, _ = np.linalg.qr(g)
    return [v[i * d:(i + 1) * d] for i in range(n_ops)]


seeds = st.integers(min_value=0, max_value=2**31 - 1)


@settings(max_examples=40, deadline=None)
@given(seed=seeds, n=st.integers(1, 4))
def test_bloch_matrix_round_trip_and_oracle_agreement(qp, oracle, seed, n):
    rho = _random_state(seed, n)
    q = qp.Qobj(rho)
    assert np.array_equal(q.bloch, oracle.bloch_from_matrix(rho))  # same rounding order as the reference
    back = qp.Qobj(q.bloch).matrix
    assert np.array_equal(back, oracle.matrix_from_bloch(q.bloch))
    assert np.abs(back - rho).max() < 1e-14
    assert abs(q.bloch[0] - 2.0**-n) < 1e-15  # identity component of a unit-trace operator
    # purity through the Bloch vector: Tr rho^2 = d * |b|^2
    assert abs(2**n * np.dot(q.bloch, q.bloch) - np.trace(rho @ rho).real) < 1e-13


@settings(max_examples=30, deadline=None)
@given(seed=seeds, na=st.integers(1, 2), nb=st.integers(1, 2))
def test_kron_and_partial_trace(qp, seed, na, nb):
    a, b = qp.Qobj(_random_state(seed, na)), qp.Qobj(_random_state(seed + 1, nb))
    ab = a.kron(b)
    assert ab.n_qubits == na + nb
    assert np.allclose(ab.matrix, np.kron(a.matrix, b.matrix), atol=1e-15)
    # Pauli strings are ordered like kron, so the Bloch vector of a product is the product of Bloch vectors
    assert np.allclose(ab.bloch, np.kron(a.bloch, b.bloch), atol=1e-15)
    assert np.allclose(ab.ptrace(list(range(na))).matrix, a.matrix, atol=1e-14)
    assert np.allclose(ab.ptrace(list(range(na, na + nb))).matrix, b.matrix, atol=1e-14)
    assert ab.is_density_matrix() and not (na + nb > 1 and ab.is_pure())


@settings(max_examples=30, deadline=None)
@given(seed=seeds, n=st.integers(1, 2), n_ops=st.integers(1, 4))
def test_channel_forms_agree(qp, oracle, seed, n, n_ops):
    kraus = _random_kraus(seed, n, n_ops)
    rho = _random_state(seed + 7, n)
    want = sum(k @ rho @ k.conj().T for k in kraus)
    from_kraus = qp.Channel([qp.Operator(k) for k in kraus])
    assert np.allclose(from_kraus.transform(qp.Qobj(rho)).matrix, want, atol=1e-13)
    choi = from_kraus.choi
    assert np.allclose(choi.matrix, oracle.choi_from_func(lambda r: sum(k @ r @ k.conj().T for k in kraus), n),
                       atol=1e-13)
    from_choi = qp.Channel(choi)
    assert np.allclose(from_choi.transform(qp.Qobj(rho)).matrix, want, atol=1e-13)
    assert np.allclose(oracle.apply_choi(choi.matrix, rho, n), want, atol=1e-13)
    assert from_choi.is_cptp(verbose=False)
    # Kraus operators recovered from the Choi matrix describe the same map
    again = qp.Channel(from_choi.kraus)
    assert np.allclose(again.transform(qp.Qobj(rho)).matrix, want, atol=1e-12)
    # trace preservation as the partial-trace condition the reference's TP projection enforces
    assert np.allclose(choi.ptrace(list(range(n))).matrix, np.eye(2**n), atol=1e-13)


@settings(max_examples=25, deadline=None)
@given(seed=seeds, shots=st.integers(1, 5000), povm=st.sampled_from(["proj", "proj-set", "sic"]))
def test_experiment_counts_are_a_pure_function_of_the_seed(qp, oracle, seed, shots, povm):
    n = 1  # the n-qubit POVM tensor is assembled on the GPU (qt_povm_kron); tests/test_gpu_state.py covers n > 1
    rho = _random_state(seed, n, rank=1)
    t = qp.StateTomograph(qp.Qobj(rho))
    np.random.seed(seed % 2**31)
    t.experiment(shots, povm)
    np.random.seed(seed % 2**31)
    want = oracle.sample_counts(oracle.measurement_matrix(povm, n), oracle.bloch_from_matrix(rho),
                                oracle.broadcast_shots(shots, t.povm_matrix.shape[0]))
    assert np.array_equal(t.results, want)  # NumPy's legacy RNG in the reference's call order
    assert np.array_equal(t.results.sum(axis=1), np.full(t.povm_matrix.shape[0], shots))


@settings(max_examples=200, deadline=None)
@given(n_items=st.integers(0, 10**7), world=st.integers(1, 64))
def test_shard_bounds_partition(n_items, world):
    from quantpy_amd.distributed import shard_bounds

    spans = [shard_bounds(n_items, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n_items
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [hi - lo for lo, hi in spans]
    assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1

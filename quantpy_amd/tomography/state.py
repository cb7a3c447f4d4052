"""Quantum state tomography with quantpy's API (reference quantpy/tomography/state.py).

`experiment()` simulates counts on the host with NumPy's legacy global RNG in the reference's
call order (one multinomial per POVM setting), so a seed reproduces the reference's counts
bit for bit; `point_estimate()` hands the counts to the HIP engine (qt_lin_batch /
qt_mle_batch).  There is no CPU estimator in this package.
"""
import numpy as np

from ..engine import get_engine
from ..geometry import hs_dst, if_dst, trace_dst
from ..measurements import generate_measurement_matrix
from ..qobj import Qobj
from ..sampling import draw_counts

_DISTANCES = {"hs": hs_dst, "trace": trace_dst, "if": if_dst}


def _resolve_dst(dst):
    if isinstance(dst, str):
        if dst not in _DISTANCES:
            raise ValueError("Invalid value for argument `dst`")
        return _DISTANCES[dst]
    return dst


def born_probabilities(povm_matrix, bloch):
    """p[s, k] that feed the sampler (reference state.py:109-111), evaluated on the host with the reference's own
    NumPy expression: the binomial sampler behind np.random.multinomial branches on p <= 0.5, and structured states
    put conditional probabilities exactly there, so the last bit of p decides the draw.  (The GPU Born-rule kernel
    qt_born_probs serves batched probability evaluation, where that bit does not matter.)"""
    dim = int(round(np.sqrt(povm_matrix.shape[-1])))
    probas = np.einsum("ijk,k->ij", povm_matrix, bloch) * dim
    return np.clip(probas, 0, 1)


def simulate_counts(povm_matrix, bloch, n_measurements, repeats=None, sampler="numpy", seed=None):
    """Born probabilities + multinomial draws for one state (reference state.py:109-114): one draw per POVM setting,
    in order, on NumPy's global legacy stream -- made by `qt_legacy_multinomial`, which restates NumPy's sampler bit
    for bit, so a seed reproduces the reference's counts and leaves `np.random` where the reference leaves it.
    `repeats=R` gives the (R, S, K) counts of R successive experiments (a bootstrap's resamples) from one call.
    sampler='device' (opt-in): the same distribution drawn on the GPU from Philox streams keyed by `seed`
    (sampling.device_multinomial) -- not the reference's counts for a given np.random.seed."""
    counts = draw_counts(n_measurements, born_probabilities(povm_matrix, bloch), 1 if repeats is None else repeats,
                         sampler, seed)
    return counts[0] if repeats is None else counts


class StateTomograph:
    """Simulate measurements of `state` and reconstruct its density matrix.

    Parameters
    ----------
    state : Qobj
    dst : 'hs' | 'trace' | 'if' | callable(Qobj, Qobj) -> float

    Attributes set by `experiment()`: povm_matrix (S, K, 4^n), results (S, K) int,
    n_measurements (S,), and by `point_estimate()`: reconstructed_state.
    """

    def __init__(self, state, dst="hs"):
        self.state = state
        self.dst = _resolve_dst(dst)
        self._results = None

    # ---- data ---------------------------------------------------------------------------------
    def _experiment_arguments(self, n_measurements, povm):
        povm_matrix = generate_measurement_matrix(povm, self.state.n_qubits)
        n_settings = povm_matrix.shape[0]
        if np.issubdtype(type(n_measurements), np.integer):
            n_measurements = np.ones(n_settings) * n_measurements
        elif len(n_measurements) != n_settings:
            raise ValueError("Wrong length for argument `n_measurements`")
        return povm_matrix, n_measurements

    def experiment(self, n_measurements, povm="proj-set", warm_start=False, sampler="numpy", seed=None):
        """Draw measurement outcomes.

        n_measurements : integer (shots per POVM setting) or one entry per setting.  A float
            scalar is rejected exactly like the reference does (TypeError from len()).
        povm : name or array, see `generate_measurement_matrix`.
        warm_start : append to the data of the previous call instead of replacing it.
        sampler, seed : 'numpy' (default) = the reference's draws on np.random's stream; 'device' = GPU sampler.
        """
        povm_matrix, n_measurements = self._experiment_arguments(n_measurements, povm)
        counts = simulate_counts(povm_matrix, self.state.bloch, n_measurements, sampler=sampler, seed=seed)
        if warm_start:
            old_total, new_total = np.sum(self.n_measurements), np.sum(n_measurements)
            self.povm_matrix = np.vstack((self.povm_matrix * old_total, povm_matrix * new_total)) / (old_total + new_total)
            self.results = np.vstack((self.results, counts))  # the setter recomputes n_measurements
        else:
            self.povm_matrix = povm_matrix
            self.results = counts
            self.n_measurements = np.asarray(n_measurements)

    def experiment_batch(self, n_measurements, povm="proj-set", repeats=1, sampler="numpy", seed=None):
        """The counts (repeats, S, K) of `repeats` successive `experiment(n_measurements, povm)` calls -- same global
        stream, same order -- drawn in one call (the resampling loop of a bootstrap, reference interval.py:598-604).
        Leaves the tomograph as the last of those calls would."""
        povm_matrix, n_measurements = self._experiment_arguments(n_measurements, povm)
        counts = simulate_counts(povm_matrix, self.state.bloch, n_measurements, repeats=repeats, sampler=sampler, seed=seed)
        if repeats:
            self.povm_matrix = povm_matrix
            self.results = counts[-1]
            self.n_measurements = np.asarray(n_measurements)
        return counts

    @property
    def results(self):
        return self._results

    @results.setter
    def results(self, results):
        self._results = results
        self.n_measurements = results.sum(-1)

    @property
    def flat_results(self):
        return self.results.flatten()

    # ---- estimators ---------------------------------------------------------------------------
    def _engine(self):
        eng = get_engine(self.state.n_qubits)
        eng.set_povm(self.povm_matrix, self.n_measurements)
        return eng

    def point_estimate(self, method="lin", physical=True, init="lin", max_iter=100, tol=1e-3):
        """Reconstruct the density matrix on the GPU.

        method : 'lin' -- linear inversion in the Pauli basis (+ eigenvalue clip at 1e-15 and
                 trace renormalisation when `physical`);
                 'mle' -- Cholesky-parametrised maximum likelihood, BFGS with scipy's control
                 flow (gtol = `tol` in the inf-norm, `max_iter` iterations), started from the
                 physical 'lin' estimate (`init='lin'`) or the fully mixed state ('mixed').
        Returns the Qobj, also stored in `reconstructed_state`.
        """
        if method == "lin":
            rho = self._engine().lin(self.results, physical=physical)
        elif method == "mle":
            if init not in ("lin", "mixed"):
                raise ValueError("Invalid value for argument `init`")
            rho, info = self._engine().mle(self.results, init=init, max_iter=max_iter, tol=tol, return_info=True)
            if info["status"] == 1:  # the reference fails inside scipy.linalg.cholesky here
                raise np.linalg.LinAlgError("starting point of the MLE is not positive definite")
            self.mle_info = info
        elif method == "mle-constr":
            rho = self._point_estimate_mle_constr(init, max_iter, tol)
        else:
            raise ValueError("Invalid value for argument `method`")
        self.reconstructed_state = Qobj(rho)
        return self.reconstructed_state

    def _point_estimate_mle_constr(self, init, max_iter, tol):
        """'mle-constr' (reference state.py:231-253): the same Cholesky-parametrised NLL minimised by
        SciPy's SLSQP under Tr(L L^dagger) = 1.  The optimiser is the one the reference calls (its own
        dependency, on the host); what it evaluates -- the NLL and, instead of the reference's D + 1
        finite-difference evaluations, its exact gradient -- comes from `qt_nll_batch` on the GPU, and the
        start point from `qt_lin_batch` / `qt_chol_param`.  Same iterates to ~1e-7, infidelity to the
        reference's result ~1e-12 (tests/test_gpu_widening.py)."""
        from scipy.optimize import minimize

        eng = self._engine()
        d = 2**self.state.n_qubits
        if init == "mixed":
            start = np.eye(d, dtype=np.complex128) / d
        elif init == "lin":
            start = eng.lin(self.results, physical=True)
        else:
            raise ValueError("Invalid value for argument `init`")
        x0, status = eng.chol_param(start)
        if status == 1:  # the reference fails inside scipy.linalg.cholesky here
            raise np.linalg.LinAlgError("starting point of the MLE is not positive definite")
        counts = self.results
        unit_trace = {"type": "eq", "fun": lambda x: float(np.dot(x, x)) - 1.0, "jac": lambda x: 2.0 * x}
        res = minimize(lambda x: eng.nll(x, counts, grad=True), x0, jac=True, constraints=[unit_trace], method="SLSQP",
                       tol=tol, options={"maxiter": max_iter})
        self.mle_info = dict(nit=res.nit, nfev=res.nfev, fun=res.fun, status=res.status)
        m = eng.chol_unparam(res.x)
        return m / np.trace(m)

    def point_estimate_batch(self, counts, method="lin", physical=True, init="lin", max_iter=100, tol=1e-3):
        """Extension: reconstruct many count tensors (B, S, K) measured with this tomograph's POVM
        and shots in one launch.  Returns (rho (B, d, d), info dict or None)."""
        eng = self._engine()
        if method == "lin":
            return eng.lin(np.asarray(counts), physical=physical), None
        if method == "mle":
            return eng.mle(np.asarray(counts), init=init, max_iter=max_iter, tol=tol, return_info=True)
        raise ValueError("Invalid value for argument `method`")

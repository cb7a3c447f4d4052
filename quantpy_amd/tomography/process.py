"""Quantum process tomography with quantpy's API (reference quantpy/tomography/process.py).

A process is probed by preparing each state of an input basis, applying the channel and doing
state tomography on the output.  Counts are simulated on the host (same RNG order as the
reference: input state by input state, POVM setting by setting); the estimator -- design
matrix, its left inverse, Choi linear inversion and the Dykstra CPTP projection -- runs on the
GPU (qt_process_setup / qt_lifp_batch / qt_cptp_project_batch).
"""
import numpy as np

from ..basis import Basis
from ..channel import Channel
from ..engine import get_engine
from ..measurements import generate_measurement_matrix
from ..qobj import Qobj
from ..routines import _mat2vec, _out_ptrace_oper, _vec2mat, generate_single_entries
from ..sampling import draw_counts
from .state import StateTomograph, _resolve_dst, born_probabilities


def _generate_input_states(input_states, n_qubits):
    """Named POVM -> its rows as trace-normalised states; a list is taken as given."""
    if isinstance(input_states, list):
        return input_states
    states = []
    for bloch in np.squeeze(generate_measurement_matrix(input_states, n_qubits)):
        state = Qobj(bloch)
        state /= state.trace()
        states.append(state)
    return states


class ProcessTomograph:
    """Simulate and reconstruct the tomography of `channel`.

    Parameters
    ----------
    channel : Channel
    input_states : name of a POVM whose rows define the input states ('proj4' default) or a list
        of Qobj; must span the operator space (4^n elements), else ValueError.
    dst : 'hs' | 'trace' | 'if' | callable
    """

    def __init__(self, channel, input_states="proj4", dst="hs"):
        self.channel = channel
        self.dst = _resolve_dst(dst)
        self.input_states = input_states
        self.input_basis = Basis(_generate_input_states(input_states, channel.n_qubits))
        if self.input_basis.dim != 4**channel.n_qubits:
            raise ValueError("Input states do not constitute a basis")
        self._decomposed_single_entries = np.array(
            [self.input_basis.decompose(Qobj(unit)) for unit in generate_single_entries(2**channel.n_qubits)])
        self._ptrace_oper = _out_ptrace_oper(channel.n_qubits)
        self._ptrace_dag_ptrace = self._ptrace_oper.T.conj() @ self._ptrace_oper

    # ---- data -------------------------------------------------------------------------------------
    def experiment(self, n_measurements, povm="proj-set", warm_start=False, sampler="numpy", seed=None):
        """State tomography of channel(rho_in) for every input state, in basis order.  sampler='device' (opt-in, see
        StateTomograph.experiment): input state i draws from the Philox streams of seed + i."""
        if not warm_start:
            self.tomographs = [StateTomograph(self.channel.transform(state)) for state in self.input_basis.elements]
        for i, tmg in enumerate(self.tomographs):
            tmg.experiment(n_measurements, povm, warm_start=warm_start, sampler=sampler,
                           seed=None if seed is None else int(seed) + i)

    def experiment_batch(self, n_measurements, povm="proj-set", repeats=1, sampler="numpy", seed=None):
        """Counts (repeats, n_inputs, S, K) of `repeats` successive `experiment(n_measurements, povm)` calls, drawn in
        one call in the same order on the same global stream (resample, input state, setting: the loop of reference
        interval.py:673-676).  Leaves the tomographs as the last of those calls would."""
        self.tomographs = [StateTomograph(self.channel.transform(state)) for state in self.input_basis.elements]
        first = self.tomographs[0]
        povm_matrix, shots = first._experiment_arguments(n_measurements, povm)
        probas = np.concatenate([born_probabilities(povm_matrix, tmg.state.bloch) for tmg in self.tomographs])
        n_in, n_set = len(self.tomographs), povm_matrix.shape[0]
        counts = draw_counts(np.tile(shots, n_in), probas, repeats, sampler, seed).reshape(repeats, n_in, n_set, -1)
        if repeats:
            for tmg, last in zip(self.tomographs, counts[-1]):
                tmg.povm_matrix, tmg.results, tmg.n_measurements = povm_matrix, last, np.asarray(shots)
        return counts

    @property
    def results(self):
        assert hasattr(self, "tomographs"), "No results"
        return np.asarray([tmg.results for tmg in self.tomographs])

    @results.setter
    def results(self, results):
        assert hasattr(self, "tomographs"), "Call experiment first"
        for tmg, counts in zip(self.tomographs, results):
            tmg.results = counts

    # ---- estimators -------------------------------------------------------------------------------
    def _engine(self):
        first = self.tomographs[0]
        eng = get_engine(self.channel.n_qubits)
        eng.set_povm(first.povm_matrix, first.n_measurements)
        eng.process_setup(np.stack([np.asarray(s.matrix, dtype=np.complex128) for s in self.input_basis.elements]))
        return eng

    def point_estimate(self, method="lifp", cptp=True, n_iter=1000, tol=1e-10, states_est_method="lin",
                       states_physical=True, states_init="lin", *, pgdb_stop="reference"):
        """method 'lifp': Choi matrix by linear inversion of all frequencies at once, then (if
        `cptp`) the alternating projection onto completely positive trace-preserving maps.
        'states': from the reconstructed output states.  'pgdb': projected gradient descent with
        backtracking (reference process.py:291-308).  The reference's loop leaves at the first step
        that lowers the NLL by more than `tol` and returns the point before it -- normally the fully
        mixed start; that behaviour is the default here (`pgdb_stop='reference'`, a drop-in), and
        `pgdb_stop='converged'` (keyword-only extension) iterates to convergence instead."""
        if method == "states":
            return self._point_estimate_states(cptp, states_est_method, states_physical, states_init, n_iter, tol)
        if method == "pgdb":
            eng = self._engine()
            self._unnorm_results = np.hstack([tmg.flat_results for tmg in self.tomographs])
            choi, iters = eng.pgdb(self.results, n_iter=n_iter, tol=tol, stop=pgdb_stop, return_iters=True)
            self.pgdb_iterations = int(iters)
            self.reconstructed_channel = Channel(choi)
            return self.reconstructed_channel
        if method != "lifp":
            raise ValueError("Incorrect value for argument `method`")
        eng = self._engine()
        self._unnorm_results = np.hstack([tmg.flat_results for tmg in self.tomographs])
        self.frequencies = np.hstack([tmg.flat_results / tmg.flat_results.sum() for tmg in self.tomographs])
        choi, iters = eng.lifp(self.results, cptp=cptp, return_iters=True)
        self.cptp_iterations = int(iters)
        self.reconstructed_channel = Channel(choi)
        return self.reconstructed_channel

    def _point_estimate_states(self, cptp, method, physical, init, n_iter, tol):
        """Choi matrix from the reconstructed output states (reference process.py:316-327): all 4^n
        output tomographs are reconstructed in ONE batched launch, then
        C = sum_ij E_ij (x) sum_s c_ij,s rho_out,s with c_ij the coordinates of E_ij in the input basis;
        projected onto CPTP only if it is not CPTP already."""
        first = self.tomographs[0]
        eng = get_engine(self.channel.n_qubits)
        eng.set_povm(first.povm_matrix, first.n_measurements)
        if method == "lin":
            outs = eng.lin(self.results, physical=physical)
        elif method == "mle":
            outs, info = eng.mle(self.results, init=init, max_iter=n_iter, tol=tol, return_info=True)
            if np.any(info["status"] == 1):
                raise np.linalg.LinAlgError("starting point of the MLE is not positive definite")
        else:
            raise ValueError("Invalid value for argument `method`")
        for tmg, rho in zip(self.tomographs, outs):
            tmg.reconstructed_state = Qobj(rho)
        ins = np.stack([np.asarray(s.matrix, dtype=np.complex128) for s in self.input_basis.elements])
        coeff = self._decomposed_single_entries  # (d^2, D): row ij = coordinates of E_ij
        units = np.einsum("es,sab->eab", coeff, ins)           # the single-entry matrices, recomposed
        images = np.einsum("es,sab->eab", coeff, np.asarray(outs))  # their images under the channel
        choi = np.zeros((ins.shape[1] ** 2,) * 2, dtype=np.complex128)
        for unit, image in zip(units, images):
            choi += np.kron(unit, image)
        self.reconstructed_channel = Channel(choi)
        if cptp and not self.reconstructed_channel.is_cptp(verbose=False):
            self.reconstructed_channel = self.cptp_projection(self.reconstructed_channel)
        return self.reconstructed_channel

    def point_estimate_batch(self, counts, method="lifp", cptp=True, n_iter=1000, tol=1e-10, states_est_method="lin",
                             states_physical=True, states_init="lin", pgdb_stop="reference"):
        """Extension: counts (B, D, S, K) measured with this tomograph's POVM, shots and input states -> Choi
        matrices (B, D, D), every resample in the same launch(es).  Same estimators and defaults as
        `point_estimate`; this is what BootstrapProcessInterval runs."""
        counts = np.asarray(counts)
        if method == "lifp":
            return self._engine().lifp(counts, cptp=cptp)
        if method == "pgdb":
            return self._engine().pgdb(counts, n_iter=n_iter, tol=tol, stop=pgdb_stop)
        if method != "states":
            raise ValueError("Incorrect value for argument `method`")
        first = self.tomographs[0]
        eng = get_engine(self.channel.n_qubits)
        eng.set_povm(first.povm_matrix, first.n_measurements)
        b, dd = counts.shape[:2]
        flat = counts.reshape((b * dd,) + counts.shape[2:])
        if states_est_method == "lin":  # all B * 4^n output states in one launch
            outs = eng.lin(flat, physical=states_physical)
        elif states_est_method == "mle":
            outs, info = eng.mle(flat, init=states_init, max_iter=n_iter, tol=tol, return_info=True)
            if np.any(info["status"] == 1):
                raise np.linalg.LinAlgError("starting point of the MLE is not positive definite")
        else:
            raise ValueError("Invalid value for argument `method`")
        outs = outs.reshape((b, dd) + outs.shape[1:])
        ins = np.stack([np.asarray(s.matrix, dtype=np.complex128) for s in self.input_basis.elements])
        coeff = self._decomposed_single_entries
        units = np.einsum("es,sab->eab", coeff, ins)
        images = np.einsum("es,bsac->beac", coeff, outs)
        dim = ins.shape[1]
        choi = np.einsum("eij,bekl->bikjl", units, images).reshape(b, dim * dim, dim * dim)  # sum_e unit_e (x) image_e
        if cptp:
            bad = [i for i in range(b) if not Channel(choi[i]).is_cptp(verbose=False)]
            if bad:
                choi[bad] = eng.cptp_project(choi[bad], mode="cptp")
        return choi

    @property
    def _lifp_oper(self):
        return self._engine().process_operators()[0]

    @property
    def _lifp_oper_inv(self):
        return self._engine().process_operators()[1]

    def _project(self, channel, mode, vectorized, **kw):
        eng = get_engine(channel.n_qubits)
        out = eng.cptp_project(np.asarray(channel.choi.matrix, dtype=np.complex128), mode=mode, **kw)
        return _mat2vec(out) if vectorized else Channel(out)

    def cptp_projection(self, channel, n_iter=1000, tol=1e-12):
        """Dykstra alternating projection of `channel` onto CPTP maps."""
        return self._project(channel, "cptp", False, n_iter=n_iter, tol=tol)

    def _cptp_projection_vec(self, choi_vec, n_iter=1000, tol=1e-12):
        return self._project(Channel(_vec2mat(np.asarray(choi_vec))), "cptp", True, n_iter=n_iter, tol=tol)

    def tp_projection(self, channel, vectorized=False):
        """Affine projection onto trace-preserving maps: Tr_out C = I."""
        return self._project(channel, "tp", vectorized)

    def cp_projection(self, channel, vectorized=False):
        """Projection onto completely positive maps: eigenvalues of C clipped at 1e-12."""
        return self._project(channel, "cp", vectorized)

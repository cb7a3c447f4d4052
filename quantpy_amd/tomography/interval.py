"""Bootstrap confidence intervals (reference quantpy/tomography/interval.py:19-56, 542-685).

The reference resamples in a serial Python loop: experiment -> point_estimate -> distance.  Here
the loop only draws the counts (host RNG, same call order, so a seed gives the same resamples);
all resamples are then reconstructed in one batched launch, sharded over the ranks of the
process group when there is one, and the distances are all-gathered (quantpy_amd.distributed).

`MomentInterval` (reference interval.py:59-110 with stats.py:21-47) is the closed-form interval the
reference's CLI scripts use: the first two moments of the squared Hilbert-Schmidt error of the
linear-inversion estimate under multinomial noise, matched to a gamma / normal / exponential
law.  Its one heavy step, the left inverse of the (state or process) design matrix, runs on the GPU
(qt_left_inverse), and so do the moment sums (qt_moment_batch: the reference's six-operand einsums collected into
O(M^2) matrix form, batched over trials).

`MHMCStateInterval` / `MHMCProcessInterval` run their chains on the GPU (qt_mhmc_state / qt_mhmc_process);
`SugiyamaInterval` and `HolderInterval` are closed-form / compositions over those.  The four intervals that
need cvxopt's SOCP / LP solvers (MomentFidelity{State,Process}Interval, Polytope{State,Process}Interval) are
names that raise NotImplementedError: SURVEY.md section 2 row 11 puts them out of scope and the image has no
cvxopt to validate against.
"""
from abc import ABC, abstractmethod
from enum import Enum, auto

import numpy as np
from scipy.interpolate import interp1d

import scipy.stats as sts

from .. import distributed as qdist
from ..engine import get_engine
from ..geometry import hs_dst, trace_dst
from ..routines import _left_inv


class Mode(Enum):
    STATE = auto()
    CHANNEL = auto()


def _proposal_increments(dim):
    """size -> (size, dim) proposal increments of a chain, drawn as the reference draws them: scipy's frozen
    `multivariate_normal(mean=zeros(dim))` (mhmc.py / interval.py:735-738, :822-825) on np.random's global stream.
    That call ends in `RandomState.multivariate_normal`, which draws standard_normal((size, dim)) and multiplies by
    sqrt(s) v of svd(cov); for the identity covariance LAPACK returns s = 1, v = I exactly, so the product is the draws
    themselves.  Up to dim 256 the reference's own call is made; above (three-qubit processes: dim 4096, where the
    eigen-decomposition in the constructor and the SVD per call cost ~20 s each) the draws are taken directly -- the
    n = 3 chains of tests/golden/mhmc3.npz, made by the reference through the full call, pin the equivalence."""
    if dim <= 256:
        from scipy.stats import multivariate_normal

        frozen = multivariate_normal(mean=np.zeros(dim))
        return lambda size: frozen.rvs(size=size).reshape(size, dim)
    return lambda size: np.random.standard_normal((size, dim))


def _pop_hidden_keys(kwargs):
    return {k: v for k, v in kwargs.items() if k not in ("self", "tmg") and not k.startswith("__")}


class ConfidenceInterval(ABC):
    """Functor: `interval(conf_levels)` -> (distances, conf_levels)."""

    EPS = 1e-15

    def __init__(self, tmg, **kwargs):
        self.tmg = tmg
        if hasattr(tmg, "state"):
            self.mode = Mode.STATE
        elif hasattr(tmg, "channel"):
            self.mode = Mode.CHANNEL
        else:
            raise ValueError()
        for name, value in kwargs.items():
            setattr(self, name, value)

    def __call__(self, conf_levels=None):
        if conf_levels is None:
            conf_levels = np.linspace(1e-3, 1 - 1e-3, 1000)
        if not hasattr(self, "cl_to_dist"):
            self.setup()
        return self.cl_to_dist(conf_levels), conf_levels

    @abstractmethod
    def setup(self):
        """Build `self.cl_to_dist`."""

    def _finish(self, dist):
        dist = np.sort(dist)
        self.cl_to_dist = interp1d(np.linspace(0, 1, len(dist)), dist)


class MomentInterval(ConfidenceInterval):
    """Closed-form interval from the first two moments of the squared HS error of linear inversion.
    distr_type : 'gamma' (default) | 'norm' | 'exp'.

    Both heavy steps run on the GPU: the left inverse of the design matrix (qt_left_inverse: MFMA Gram, pivoted
    Gauss-Jordan) and the moment sums of stats.py:21-47 (qt_moment_batch).  `radii_batch` evaluates the interval for
    a whole batch of count tensors of the same experiment in one launch -- the coverage study of
    notebooks/Verification.ipynb (10 000 trials per state) is that call."""

    def __init__(self, tmg, distr_type="gamma"):
        super().__init__(tmg, **_pop_hidden_keys(locals()))

    def _design(self):
        """(dim, n_measurements (S,), counts (S, K), inv_matrix (rows, S*K)) as interval.py:72-87 builds them."""
        tmg = self.tmg
        if self.mode == Mode.STATE:
            dim = 2**tmg.state.n_qubits
            n_measurements = tmg.n_measurements
            counts = np.asarray(tmg.results)
            povm = np.asarray(tmg.povm_matrix)
            design = povm.reshape(-1, povm.shape[-1])
        else:
            dim = 4**tmg.channel.n_qubits
            first = tmg.tomographs[0]
            n_measurements = np.tile(first.n_measurements, len(tmg.tomographs))
            counts = np.vstack([t.results for t in tmg.tomographs])
            povm = np.asarray(first.povm_matrix)
            povm_rows = povm.reshape(-1, povm.shape[-1])
            states = np.asarray([rho.T.bloch for rho in tmg.input_basis.elements])
            design = np.einsum("sd,pi->spdi", states, povm_rows).reshape(states.shape[0] * povm_rows.shape[0], -1)
        inv_matrix = np.asarray(_left_inv(design)) / dim  # GPU: Gram GEMM (MFMA), pivoted Gauss-Jordan, GEMM
        return dim, np.asarray(n_measurements, dtype=np.float64), counts, inv_matrix

    def _engine(self):
        tmg = self.tmg
        return get_engine(tmg.state.n_qubits if self.mode == Mode.STATE else tmg.channel.n_qubits)

    def _distribution(self, mean, variance):
        if self.distr_type == "norm":
            return sts.norm(loc=mean, scale=np.sqrt(variance))
        if self.distr_type == "gamma":
            scale = variance / mean
            return sts.gamma(a=mean / scale, scale=scale)
        if self.distr_type == "exp":
            return sts.expon(scale=mean)
        raise NotImplementedError(f"Unsupported distribution type {self.distr_type}")

    def _alpha(self, dim):
        if self.tmg.dst == hs_dst:
            return np.sqrt(dim / 2)
        if self.tmg.dst == trace_dst:
            return dim / 2
        raise NotImplementedError()

    def setup(self):
        dim, n_measurements, counts, inv_matrix = self._design()
        mean, variance = self._engine().moments(counts, n_measurements, inv_matrix)
        distr = self._distribution(mean, variance)
        alpha = self._alpha(dim)
        self.mean, self.variance = mean, variance
        self.cl_to_dist = lambda cl: np.sqrt(distr.ppf(cl)) * alpha

    def radii_batch(self, counts, conf_levels):
        """The interval's radii for B count tensors of this tomograph's experiment (same POVM, same shots): counts
        (B, S, K) for a state, (B, 4^n, S, K) for a process -> radii (B, len(conf_levels)).  One moment launch for the
        batch; the gamma / normal quantiles are SciPy's, vectorised over the batch."""
        dim, n_measurements, own, inv_matrix = self._design()
        c = np.asarray(counts)
        c = c.reshape((c.shape[0],) + own.shape)
        mean, variance = self._engine().moments(c, n_measurements, inv_matrix)
        distr = self._distribution(mean[:, None], variance[:, None])
        return np.sqrt(distr.ppf(np.asarray(conf_levels, dtype=np.float64)[None, :])) * self._alpha(dim)


class BootstrapStateInterval(ConfidenceInterval):
    """Parametric bootstrap around `state` (default: the tomograph's reconstructed state) with the
    tomograph's own POVM and shots; distances tmg.dst(resampled estimate, state)."""

    def __init__(self, tmg, n_points=1000, method="lin", physical=True, init="lin", tol=1e-3, max_iter=100,
                 state=None, sampler="numpy", seed=None):
        super().__init__(tmg, **_pop_hidden_keys(locals()))

    def setup(self):
        if self.mode == Mode.CHANNEL:
            raise NotImplementedError("This interval works only for state tomography")
        tmg = self.tmg
        if self.state is None:
            if hasattr(tmg, "reconstructed_state"):
                self.state = tmg.reconstructed_state
            else:
                self.state = tmg.point_estimate(method=self.method, physical=self.physical, init=self.init,
                                                tol=self.tol, max_iter=self.max_iter)
        boot = tmg.__class__(self.state, tmg.dst)
        if self.n_points and tmg.dst is hs_dst and self.method in ("lin", "mle"):
            self._setup_fused(boot)
            return
        # every resample's counts, one global RNG stream in the reference's order (resample after resample, setting
        # after setting): ONE call of the C restatement of NumPy's sampler instead of n_points x S Python calls
        # (sampler='device', opt-in: the same distribution drawn on the GPU, off the reference's stream)
        counts = boot.experiment_batch(tmg.n_measurements, tmg.povm_matrix, self.n_points, sampler=self.sampler,
                                       seed=self.seed)
        counts = qdist.broadcast_array(counts) if self.n_points else np.empty((0,) + tmg.results.shape)
        self.boot_counts = counts
        centre = self.state

        def reconstruct(shard):
            rho, info = boot.point_estimate_batch(shard, method=self.method, physical=self.physical, init=self.init,
                                                  tol=self.tol, max_iter=self.max_iter)
            if info is not None and np.any(info["status"] == 1):
                raise np.linalg.LinAlgError("starting point of the MLE is not positive definite")
            if tmg.dst is hs_dst:
                return get_engine(centre.n_qubits).hs_dist(rho, centre.matrix)
            from ..qobj import Qobj

            return np.array([tmg.dst(Qobj(r), centre) for r in rho], dtype=np.float64)

        self.boot_dist = qdist.sharded_map(counts, reconstruct)
        self._finish(self.boot_dist)

    def _setup_fused(self, boot):
        """The loop of interval.py:598-609 for the Hilbert-Schmidt distance and the 'lin' / 'mle' estimators, as this
        rank's shard of it: the shard's counts go to (sampler='numpy': one C call on np.random's stream on rank 0,
        broadcast) or are drawn in (sampler='device': rows keyed by their global index, each rank draws only its own)
        HBM, ONE launch family reconstructs them and writes the distance to `state` (qt_lin_dist_batch /
        qt_mle_dist_batch: 8 bytes per resample leave the kernel, no density matrices), the shard is sorted where it is,
        and `cl_to_dist` evaluates interp1d's order statistics across the ranks (quantpy_amd.distributed.ShardedSample)
        -- no all-gather of the sample unless somebody reads `boot_dist` / `cl_to_dist.y`."""
        import torch

        from .. import _capi
        from ..sampling import resolve_seed
        from .state import born_probabilities

        tmg = self.tmg
        rank, world = qdist.world()
        lo, hi = qdist.shard_bounds(self.n_points)
        povm_matrix, shots = boot._experiment_arguments(tmg.n_measurements, tmg.povm_matrix)
        boot.povm_matrix, boot.n_measurements = povm_matrix, np.asarray(shots)
        eng = boot._engine()
        dev = torch.device("cuda", eng.device)
        n_set, n_out = np.asarray(povm_matrix).shape[:2]
        d = 2 ** self.state.n_qubits
        if self.sampler == "device":
            pvals = born_probabilities(povm_matrix, self.state.bloch)
            if not (np.all(pvals >= 0) and np.all(pvals[:, :-1].sum(1) <= 1.0 + 1e-12)):
                raise ValueError("sum(pvals[:-1]) > 1.0")
            seed = resolve_seed(self.seed)
            if world > 1:  # one Philox key for the whole table: rank 0's
                seed = int(qdist.broadcast_array(np.array([seed], dtype=np.uint64).view(np.int64)).view(np.uint64)[0])
            counts = torch.empty((hi - lo, n_set, n_out), dtype=torch.int64, device=dev)
            if hi > lo:
                eng.device_multinomial(np.asarray(shots).astype(np.int64), pvals, (hi - lo) * n_set, seed,
                                       first_row=lo * n_set, out=counts)
            self._boot_counts_host, self._boot_counts_device = None, counts
        elif self.sampler == "numpy":
            host = boot.experiment_batch(tmg.n_measurements, tmg.povm_matrix, self.n_points) if rank == 0 or world == 1 else \
                np.empty((self.n_points, n_set, n_out), dtype=np.int64)
            host = qdist.broadcast_array(host)
            self.boot_counts = host
            counts = torch.from_numpy(np.ascontiguousarray(host[lo:hi])).to(dev)
        else:
            raise ValueError(f"sampler must be 'numpy' or 'device', not {self.sampler!r}")
        dist = torch.empty(hi - lo, dtype=torch.float64, device=dev)
        status = torch.zeros(hi - lo, dtype=torch.int32, device=dev)
        centre = torch.from_numpy(np.ascontiguousarray(self.state.matrix, dtype=np.complex128)).to(dev)
        if hi > lo:
            if self.method == "lin":
                eng.lin_dist_dev(counts, centre, dist, physical=self.physical, status=status)
            else:
                eng.mle_dist_dev(counts, centre, dist, init=self.init, max_iter=self.max_iter, tol=self.tol, status=status)
        eng.sync()
        st = status.cpu().numpy()
        bad = np.array([int(np.any(st == 1)), int(np.any(st == _capi.TRIAL_SHOTS))])
        if world > 1:  # every rank raises, or none
            bad = qdist.allgather_equal(bad).max(axis=0)
        if bad[0]:
            raise np.linalg.LinAlgError("starting point of the MLE is not positive definite")
        if bad[1]:
            raise ValueError("per-setting totals of a trial do not match the registered shots")
        if self.sampler == "device" and world == 1:
            boot.results = counts[-1].cpu().numpy()  # the tomograph is left as the last experiment() would leave it
        self._dist_shard = dist.clone()  # resample order (boot_dist); the sample below is sorted in place
        self._boot_dist = None
        self.sample = qdist.ShardedSample(dist, self.n_points, engine=eng)
        self.cl_to_dist = _SampleInterp(self.sample)

    @property
    def boot_dist(self):
        """The distances in resample order, on the host (all-gathered on first use when the ranks hold shards)."""
        if getattr(self, "_boot_dist", None) is None and getattr(self, "_dist_shard", None) is not None:
            self._boot_dist = qdist.allgather_device(self._dist_shard, self.n_points).cpu().numpy()
        return getattr(self, "_boot_dist", None)

    @boot_dist.setter
    def boot_dist(self, value):
        self._boot_dist, self._dist_shard = value, None

    @property
    def boot_counts(self):
        """Every resample's counts on the host.  sampler='device': this rank's shard only (rows lo .. hi of the table),
        fetched when somebody reads it."""
        if getattr(self, "_boot_counts_host", None) is None and getattr(self, "_boot_counts_device", None) is not None:
            self._boot_counts_host = self._boot_counts_device.cpu().numpy()
        return getattr(self, "_boot_counts_host", None)

    @boot_counts.setter
    def boot_counts(self, value):
        self._boot_counts_host, self._boot_counts_device = value, None


class _SampleInterp:
    """`interp1d(np.linspace(0, 1, n), sorted_dist)` (interval.py:611-612) over a `ShardedSample`: calling it evaluates
    the order statistics where the shards are (collective when there are several ranks); `.x` / `.y` are interp1d's
    attributes, `.y` gathering the sorted sample on first use."""

    def __init__(self, sample):
        self.sample = sample

    def __call__(self, conf_levels):
        cl = np.asarray(conf_levels, dtype=np.float64)
        return self.sample.quantiles(cl.reshape(-1)).reshape(cl.shape)

    @property
    def y(self):
        full = self.sample.gather_sorted()
        if not isinstance(full, np.ndarray):
            self.sample.engine.sync()
            full = full.cpu().numpy()
        return full

    @property
    def x(self):
        return np.linspace(0, 1, self.sample.n_total)


class MHMCStateInterval(ConfidenceInterval):
    """Metropolis-Hastings samples of the likelihood on the Cholesky parameters around the point estimate
    (reference interval.py:689-750, mhmc.py): distances tmg.dst(sample, state), sorted.
    The random numbers are drawn here exactly as the reference draws them -- proposal increments from
    scipy's frozen `multivariate_normal(zeros(4^n))`, then `numpy.random.rand`, first for the burn-in,
    then for the samples -- and the chain itself (one likelihood evaluation per step, inherently serial)
    runs in one launch of `qt_mhmc_state`."""

    def __init__(self, tmg, n_points=1000, step=0.01, burn_steps=1000, thinning=1, warm_start=False,
                 use_new_estimate=False, state=None, verbose=False):
        super().__init__(tmg, **_pop_hidden_keys(locals()))

    def setup(self):

        if self.mode == Mode.CHANNEL:
            raise NotImplementedError("This interval works only for state tomography")
        tmg = self.tmg
        if not self.use_new_estimate:
            self.state = tmg.reconstructed_state
        elif self.state is None:
            self.state = tmg.point_estimate(method="mle", physical=True)
        eng = tmg._engine()
        dim = 4**tmg.state.n_qubits
        if not (self.warm_start and hasattr(self, "_x_t")):
            x0, status = eng.chol_param(np.asarray(self.state.matrix, dtype=np.complex128))
            if status == 1:  # the reference fails inside scipy.linalg.cholesky here
                raise np.linalg.LinAlgError("the state the chain starts from is not positive definite")
            self._x_t, self._burned = x0, False
        jump = _proposal_increments(dim)
        parts = []
        if not self._burned:
            parts.append((jump(self.burn_steps), np.random.rand(self.burn_steps)))
        total = self.n_points * self.thinning
        parts.append((jump(total), np.random.rand(total)))
        deltas = np.concatenate([p[0] for p in parts])
        uniforms = np.concatenate([p[1] for p in parts])
        chain, accepted = eng.mhmc_state(tmg.results, self._x_t, deltas, uniforms, self.step)
        skip = 0 if self._burned else self.burn_steps
        self._burned = True
        self._x_t = chain[-1].copy() if len(chain) else self._x_t
        self.samples = chain[skip::self.thinning][: self.n_points]
        self.acceptance_rate = float(accepted[skip:].mean()) if total else 0.0
        mats = eng.chol_unparam(self.samples)
        if tmg.dst is hs_dst:
            dist = eng.hs_dist(mats, np.asarray(self.state.matrix, dtype=np.complex128))
        else:
            dist = np.array([tmg.dst(m, self.state.matrix) for m in mats], dtype=np.float64)
        self._finish(dist)


class SugiyamaInterval(ConfidenceInterval):
    """Hoeffding-type interval of Sugiyama et al., arXiv:1306.4191 (reference interval.py:219-265): closed
    form in the left inverse of the rescaled POVM matrix (computed on the GPU, `qt_left_inverse`)."""

    def __init__(self, tmg, n_points=1000, max_confidence=0.999):
        super().__init__(tmg, **_pop_hidden_keys(locals()))

    def setup(self):
        from ..geometry import if_dst, trace_dst

        if self.mode == Mode.CHANNEL:
            raise NotImplementedError("Sugiyama interval works only for state tomography")
        tmg = self.tmg
        dim = 2**tmg.state.n_qubits
        dist = np.linspace(0, 1, self.n_points)
        settings, outcomes, width = tmg.povm_matrix.shape
        scaled = np.reshape(np.asarray(tmg.povm_matrix), (-1, width)) * dim / np.sqrt(2 * dim)
        inverse = get_engine(tmg.state.n_qubits).left_inverse_of(scaled).reshape(-1, settings, outcomes)
        ratios = tmg.n_measurements.sum() / tmg.n_measurements
        spread = np.max(inverse, axis=-1) - np.min(inverse, axis=-1)
        c_alpha = np.sum(spread**2 * ratios[None, :], axis=-1) + self.EPS
        if tmg.dst == hs_dst:
            b = 8 / (dim**2 - 1)
        elif tmg.dst == trace_dst:
            b = 16 / (dim**2 - 1) / dim
        elif tmg.dst == if_dst:
            b = 4 / (dim**2 - 1) / dim
        else:
            raise NotImplementedError("Unsupported distance")
        conf_levels = 1 - 2 * np.sum(np.exp(-b * dist[:, None] ** 2 * np.sum(tmg.n_measurements) / c_alpha[None, :]),
                                     axis=1)
        self.cl_to_dist = interp1d(conf_levels, dist)


class HolderInterval(ConfidenceInterval):
    """Process interval assembled from one state interval per input state (reference interval.py:421-539):
    conf_level^(number of inputs) and sqrt(sum_ij |c_i . conj(c_j)| delta_i delta_j) over the
    coordinates c of the single-entry matrices in the input basis.  `kind` in {'mhmc', 'bootstrap',
    'sugiyama'}; as in the reference, the default 'wang' (and 'boot') is rejected by `setup` with a
    ValueError and 'moment' fails on MomentInterval's signature."""

    def __init__(self, tmg, n_points=1000, kind="wang", max_confidence=0.999, method="lin", method_boot="lin",
                 physical=True, init="lin", tol=1e-3, max_iter=100, step=0.01, burn_steps=1000, thinning=1):
        super().__init__(tmg, **_pop_hidden_keys(locals()))

    def __call__(self, conf_levels=None):
        if conf_levels is None:
            conf_levels = np.linspace(1e-3, 1 - 1e-3, 1000)
        if not hasattr(self, "intervals"):
            self.setup()
        results = [interval(conf_levels) for interval in self.intervals]
        deltas = np.asarray([r[0] for r in results])
        conf_levels = results[0][1] ** self.tmg.input_basis.dim
        entries = self.tmg._decomposed_single_entries
        coef = np.abs(np.einsum("ij,ik->jk", entries, entries.conj()))
        dist = np.sqrt(np.einsum("ik,jk,ij->k", deltas, deltas, coef))
        return dist, conf_levels

    def setup(self):
        if self.mode == Mode.STATE:
            raise NotImplementedError("Holder interval works only for process tomography")
        parts = self.tmg.tomographs
        if self.kind == "moment":
            self.intervals = [MomentInterval(t, self.n_points, self.max_confidence) for t in parts]  # TypeError, as there
        elif self.kind == "mhmc":
            self.intervals = [MHMCStateInterval(t, self.n_points, self.step, self.burn_steps, self.thinning) for t in parts]
        elif self.kind == "bootstrap":
            self.intervals = [BootstrapStateInterval(t, self.n_points, self.method, physical=self.physical, init=self.init,
                                                     tol=self.tol, max_iter=self.max_iter) for t in parts]
        elif self.kind == "sugiyama":
            self.intervals = [SugiyamaInterval(t, self.n_points, self.max_confidence) for t in parts]
        else:
            raise ValueError("Incorrect value for argument `kind`.")
        for interval in self.intervals:
            interval.setup()


def _needs_cvxopt(name, lines):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(f"{name} (reference interval.py:{lines}) poses a cone / linear program for cvxopt; "
                                  "that solver is outside the tomography hot path and is not provided")

    return type(name, (ConfidenceInterval,), {"__init__": __init__, "setup": lambda self: None,
                                              "__doc__": f"Placeholder: the reference's {name} needs cvxopt."})


MomentFidelityStateInterval = _needs_cvxopt("MomentFidelityStateInterval", "113-160")
MomentFidelityProcessInterval = _needs_cvxopt("MomentFidelityProcessInterval", "163-216")
PolytopeStateInterval = _needs_cvxopt("PolytopeStateInterval", "268-335")
PolytopeProcessInterval = _needs_cvxopt("PolytopeProcessInterval", "338-418")


class MHMCProcessInterval(ConfidenceInterval):
    """Metropolis-Hastings samples of the process likelihood on the Choi vector (reference
    interval.py:763-850): every proposal x + step * delta is projected onto the CPTP set
    (`_cptp_update_rule`, process.py:279-281), the target is exp(-nll) with the raw counts.  Draws on the
    host in the reference's order, the chain in one launch of `qt_mhmc_process`.  As in the reference,
    `return_samples=True` makes `setup()` return (dist, conf_levels, acceptance_rate, matrices) instead of
    preparing the functor."""

    def __init__(self, tmg, n_points=1000, step=0.01, burn_steps=1000, thinning=1, warm_start=False, method="lifp",
                 states_est_method="lin", states_physical=True, states_init="lin", use_new_estimate=False,
                 channel=None, verbose=False, return_samples=False):
        super().__init__(tmg, **_pop_hidden_keys(locals()))

    def setup(self):

        if self.mode == Mode.STATE:
            raise NotImplementedError("This interval works only for process tomography")
        tmg = self.tmg
        if not self.use_new_estimate:
            self.channel = tmg.reconstructed_channel
        elif self.channel is None:
            self.channel = tmg.point_estimate(self.method, states_est_method=self.states_est_method,
                                              states_physical=self.states_physical, states_init=self.states_init)
        eng = tmg._engine()
        dim = 16**tmg.channel.n_qubits
        centre = np.asarray(self.channel.choi.matrix, dtype=np.complex128)
        if not (self.warm_start and hasattr(self, "_x_t")):
            self._x_t, self._burned = centre.copy(), False
        jump = _proposal_increments(dim)
        parts = []
        if not self._burned:
            parts.append((jump(self.burn_steps), np.random.rand(self.burn_steps)))
        total = self.n_points * self.thinning
        parts.append((jump(total), np.random.rand(total)))
        deltas = np.concatenate([p[0] for p in parts])
        uniforms = np.concatenate([p[1] for p in parts])
        chain, accepted = eng.mhmc_process(tmg.results, self._x_t, deltas, uniforms, self.step)
        skip = 0 if self._burned else self.burn_steps
        self._burned = True
        if len(chain):
            self._x_t = chain[-1].copy()
        # mhmc.py:66 collects the samples in a REAL array: the imaginary parts of the chain states are
        # dropped there (NumPy's ComplexWarning), and the distances are those of the real parts
        self.samples = np.ascontiguousarray(chain[skip::self.thinning][: self.n_points].real)
        self.acceptance_rate = float(accepted[skip:].mean()) if total else 0.0
        if tmg.dst is hs_dst:
            dist = get_engine(self.channel.n_qubits).hs_dist(self.samples, centre)
        else:
            dist = np.array([tmg.dst(m, centre) for m in self.samples], dtype=np.float64)
        dist = np.sort(dist)
        conf_levels = np.linspace(0, 1, len(dist))
        if self.return_samples:
            return dist, conf_levels, self.acceptance_rate, list(self.samples)
        self.cl_to_dist = interp1d(conf_levels, dist)


class BootstrapProcessInterval(ConfidenceInterval):
    """Parametric bootstrap of a process tomography around `channel` (default: the reconstructed
    channel): resampled counts for every input state, batched Choi reconstruction."""

    def __init__(self, tmg, n_points=1000, method="lifp", cptp=True, tol=1e-10, channel=None,
                 states_est_method="lin", states_physical=True, states_init="lin", sampler="numpy", seed=None):
        super().__init__(tmg, **_pop_hidden_keys(locals()))

    def setup(self):
        if self.mode == Mode.STATE:
            raise NotImplementedError("This interval works only for process tomography")
        tmg = self.tmg
        if self.channel is None:
            if hasattr(tmg, "reconstructed_channel"):
                self.channel = tmg.reconstructed_channel
            else:
                self.channel = tmg.point_estimate(method=self.method, states_physical=self.states_physical,
                                                  states_init=self.states_init, cptp=self.cptp)
        boot = tmg.__class__(self.channel, tmg.input_states, tmg.dst)
        shots, povm = tmg.tomographs[0].n_measurements, tmg.tomographs[0].povm_matrix
        counts = qdist.broadcast_array(boot.experiment_batch(shots, povm=povm, repeats=self.n_points,
                                                             sampler=self.sampler, seed=self.seed))
        self.boot_counts = counts
        centre = self.channel.choi

        def reconstruct(shard):
            # the reference's loop (interval.py:675-680) passes method, states_physical, states_init and cptp;
            # `tol` and `states_est_method` stay at point_estimate's defaults there, and so they do here
            choi = boot.point_estimate_batch(shard, method=self.method, cptp=self.cptp,
                                             states_physical=self.states_physical, states_init=self.states_init)
            if tmg.dst is hs_dst:
                return get_engine(tmg.channel.n_qubits).hs_dist(choi, centre.matrix)  # 4^n x 4^n matrices on the channel's engine
            from ..qobj import Qobj

            return np.array([tmg.dst(Qobj(c), centre) for c in choi], dtype=np.float64)

        self.boot_dist = qdist.sharded_map(counts, reconstruct)
        self._finish(self.boot_dist)

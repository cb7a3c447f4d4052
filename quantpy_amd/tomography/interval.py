"""Bootstrap confidence intervals (reference quantpy/tomography/interval.py:19-56, 542-685).

The reference resamples in a serial Python loop: experiment -> point_estimate -> distance.  Here
the loop only draws the counts (host RNG, same call order, so a seed gives the same resamples);
all resamples are then reconstructed in one batched launch, sharded over the ranks of the
process group when there is one, and the distances are all-gathered (quantpy_amd.distributed).

The closed-form / convex-programming intervals of the reference (Moment*, Sugiyama, Polytope*,
Holder, MHMC*) are outside this package's hot path and are not provided.
"""
from abc import ABC, abstractmethod
from enum import Enum, auto

import numpy as np
from scipy.interpolate import interp1d

from .. import distributed as qdist
from ..engine import get_engine
from ..geometry import hs_dst


class Mode(Enum):
    STATE = auto()
    CHANNEL = auto()


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


class BootstrapStateInterval(ConfidenceInterval):
    """Parametric bootstrap around `state` (default: the tomograph's reconstructed state) with the
    tomograph's own POVM and shots; distances tmg.dst(resampled estimate, state)."""

    def __init__(self, tmg, n_points=1000, method="lin", physical=True, init="lin", tol=1e-3, max_iter=100,
                 state=None):
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
        counts = []
        for _ in range(self.n_points):  # serial on purpose: one global RNG stream, reference order
            boot.experiment(tmg.n_measurements, tmg.povm_matrix)
            counts.append(boot.results)
        counts = qdist.broadcast_array(np.stack(counts)) if self.n_points else np.empty((0,) + tmg.results.shape)
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


class BootstrapProcessInterval(ConfidenceInterval):
    """Parametric bootstrap of a process tomography around `channel` (default: the reconstructed
    channel): resampled counts for every input state, batched Choi reconstruction."""

    def __init__(self, tmg, n_points=1000, method="lifp", cptp=True, tol=1e-10, channel=None,
                 states_est_method="lin", states_physical=True, states_init="lin"):
        super().__init__(tmg, **_pop_hidden_keys(locals()))

    def setup(self):
        if self.mode == Mode.STATE:
            raise NotImplementedError("This interval works only for process tomography")
        tmg = self.tmg
        if self.method != "lifp":
            raise NotImplementedError("only method='lifp' is on the GPU hot path")
        if self.channel is None:
            if hasattr(tmg, "reconstructed_channel"):
                self.channel = tmg.reconstructed_channel
            else:
                self.channel = tmg.point_estimate(method=self.method, states_physical=self.states_physical,
                                                  states_init=self.states_init, cptp=self.cptp)
        boot = tmg.__class__(self.channel, tmg.input_states, tmg.dst)
        shots, povm = tmg.tomographs[0].n_measurements, tmg.tomographs[0].povm_matrix
        counts = []
        for _ in range(self.n_points):
            boot.experiment(shots, povm=povm)
            counts.append(boot.results)
        counts = qdist.broadcast_array(np.stack(counts))
        self.boot_counts = counts
        centre = self.channel.choi

        def reconstruct(shard):
            choi = boot.point_estimate_batch(shard, cptp=self.cptp)
            if tmg.dst is hs_dst:
                return get_engine(centre.n_qubits).hs_dist(choi, centre.matrix)
            from ..qobj import Qobj

            return np.array([tmg.dst(Qobj(c), centre) for c in choi], dtype=np.float64)

        self.boot_dist = qdist.sharded_map(counts, reconstruct)
        self._finish(self.boot_dist)

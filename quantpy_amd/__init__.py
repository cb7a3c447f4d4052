"""quantpy_amd -- MI355X-native drop-in for the tomography hot path of nordmtr/quantpy.

`import quantpy_amd as qp` exposes the reference's public names (quantpy/__init__.py:1-23) for
the path this package implements: Qobj / Channel / Operator containers on the host, and
StateTomograph / ProcessTomograph / Bootstrap*Interval whose estimators are HIP kernels behind
libqtomo.so (include/qtomo.h).  There is no CPU estimator: without the library or a GPU the
estimators raise `EngineUnavailable`.
"""
from . import basis, channel, engine, operator, qobj  # noqa: F401
from ._capi import EngineUnavailable  # noqa: F401
from .base_quantum import BaseQuantum  # noqa: F401
from .channel import Channel  # noqa: F401
from .engine import Engine, EngineError, get_engine  # noqa: F401
from .geometry import hs_dst, if_dst, product, trace_dst  # noqa: F401
from .measurements import generate_measurement_matrix  # noqa: F401
from .operator import Operator  # noqa: F401
from .qobj import Qobj  # noqa: F401
from .routines import generate_pauli, join_gates, kron  # noqa: F401
from .tomography.interval import (  # noqa: F401
    BootstrapProcessInterval,
    BootstrapStateInterval,
    ConfidenceInterval,
    HolderInterval,
    MHMCProcessInterval,
    MHMCStateInterval,
    MomentFidelityProcessInterval,
    MomentFidelityStateInterval,
    MomentInterval,
    PolytopeProcessInterval,
    PolytopeStateInterval,
    SugiyamaInterval,
)
from .tomography.process import ProcessTomograph  # noqa: F401
from .tomography.state import StateTomograph  # noqa: F401

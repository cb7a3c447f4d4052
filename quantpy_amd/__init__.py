"""quantpy_amd -- MI355X-native drop-in for the tomography hot path of nordmtr/quantpy.

Host-side objects (Qobj, Channel, Operator, ...) keep quantpy's API; the estimators
(`point_estimate`, bootstrap, CPTP projection) run as HIP kernels through libqtomo.so.
"""
from . import engine  # noqa: F401
from ._capi import EngineUnavailable  # noqa: F401
from .engine import Engine, EngineError, get_engine  # noqa: F401

from .interval import (  # noqa: F401
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
from .process import ProcessTomograph  # noqa: F401
from .state import StateTomograph  # noqa: F401

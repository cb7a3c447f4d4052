from .interval import (  # noqa: F401
    BootstrapProcessInterval,
    BootstrapStateInterval,
    ConfidenceInterval,
    MHMCStateInterval,
    MomentInterval,
)
from .process import ProcessTomograph  # noqa: F401
from .state import StateTomograph  # noqa: F401

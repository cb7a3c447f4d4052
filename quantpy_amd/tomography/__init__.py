from .interval import BootstrapProcessInterval, BootstrapStateInterval, ConfidenceInterval  # noqa: F401
from .process import ProcessTomograph  # noqa: F401
from .state import StateTomograph  # noqa: F401

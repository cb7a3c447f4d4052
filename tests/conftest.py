import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests need lib/libqtomo.so: build it once (hipcc, gfx950) if this checkout has none -- the
    library is the only implementation there is, so a missing one must not turn into skips."""
    if not any(item.get_closest_marker("gpu") for item in items):
        return
    from quantpy_amd.build import LIB, build_library

    if not os.path.exists(LIB):  # (a stale-by-mtime check would misfire on a freshly copied tree)
        build_library(force=True)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only; see oracle/quantpy_oracle.py header)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import quantpy_oracle

    return quantpy_oracle


@pytest.fixture(scope="session")
def engine():
    """The HIP engine on cuda:0.  Fails (does not skip) when the library or a GPU is missing."""
    import quantpy_amd

    return quantpy_amd.engine

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")
    config.addinivalue_line("markers", "slow: minutes of oracle time (whole-frame parity of the full-size configurations); opt in with -m \"gpu and slow\" or TRG_RUN_SLOW=1")


def pytest_collection_modifyitems(config, items):
    # slow tests run only when asked for by name: `-m "gpu and slow"` (or TRG_RUN_SLOW=1) -- a plain `-m gpu` stays a one-minute suite
    if "slow" in (config.getoption("-m") or "") or os.environ.get("TRG_RUN_SLOW"):
        return
    skip = pytest.mark.skip(reason="slow: run with -m \"gpu and slow\" or TRG_RUN_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure only)."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def built():
    """Make sure the native libraries exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def cornell(O):
    return O.OracleScene.cornell_box()

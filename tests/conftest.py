import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    # A test that hangs (a wedged GPU, a collective whose peer died) must not hold the box until the lease's own limit:
    # with pytest-timeout present, ten minutes per test, enforced by a watchdog THREAD that dumps the stacks and ends the
    # process (the signal method cannot interrupt a blocked runtime call).  The longest test takes about half a minute.
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 600.0
        config.option.timeout_method = "thread"


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle (and, when possible, the product library) are built."""
    import __graft_entry__ as g
    g.build(quiet=True)

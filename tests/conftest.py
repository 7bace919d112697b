import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """No test may stall a run for good: where pytest-timeout is installed every test that does not ask for its own limit gets
    six minutes (the slowest takes half a minute); a stalled one then fails with the threads' stacks instead of hanging."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(360))


@pytest.fixture(scope="session")
def ctx():
    """One device context for the whole GPU session (one exec-domain thread)."""
    import schroedinger_amd as sa
    if sa.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible")
    c = sa.Context(0)
    yield c
    c.close()

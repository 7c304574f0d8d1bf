import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ra():
    import rt_amd
    return rt_amd


@pytest.fixture(scope="session")
def gpu_ctx(ra):
    """One context for the whole GPU session (tests run in one process)."""
    ctx = ra.Context(0)
    yield ctx
    ctx.close()

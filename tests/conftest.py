import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the in-tree native pieces once (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as entry

    entry.build()
    yield


def has_gpu():
    try:
        import dsgd_amd

        return dsgd_amd.device_count() > 0
    except Exception:
        return False

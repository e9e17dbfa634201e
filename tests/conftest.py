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


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Which branch of every conditional parity assertion ran (tests/waivers.py)."""
    import json

    import waivers

    t = waivers.table()
    if not t:
        return
    terminalreporter.write_sep("-", "conditional parity assertions: strict / waived")
    for fam, v in t.items():
        terminalreporter.write_line("%-58s strict %4d   waived %4d  %s" % (fam, v["strict"], v["waived"], "; ".join(v["reasons"][:3])))
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "waivers.json"), "w") as f:
            json.dump(t, f, indent=1)
    except OSError:
        pass

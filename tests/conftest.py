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
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _native_handles_released_before_exit():
    """Module-scoped model fixtures live until the session ends; release their native handles here, while the HIP runtime
    is certainly intact, instead of leaving it to interpreter shutdown (the shim's atexit hook is the second line)."""
    yield
    mod = sys.modules.get("nl_vsgg_amd.lib.sttran")
    if mod is not None:
        mod._destroy_live_handles()

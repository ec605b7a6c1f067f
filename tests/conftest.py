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


@pytest.fixture(autouse=True)
def _library_options_restored():
    """Whatever a test does to the library's process-wide options (gm_set_option: reduction strategy, kernel forms, graph-build
    experiments, debug flags), the next test starts from the documented defaults -- also when the test failed half way."""
    yield
    so = os.path.join(ROOT, "graphmat_amd", "libgraphmat_hip.so")
    if "graphmat_amd._lib" in sys.modules and os.path.exists(so):
        mod = sys.modules["graphmat_amd._lib"]
        if getattr(mod, "_lib", None) is not None:
            mod._lib.gm_reset_options()

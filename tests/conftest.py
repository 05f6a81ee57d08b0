import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def czo():
    import czo as m      # oracle binding (test infrastructure)
    m.lib()
    return m


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(autouse=True)
def _shipped_process_knobs(request):
    """Process-wide tuning knobs (cz_debug_set keys backed by globals of the kernel files: mid_nw, dec_attn_split_max, ...) go back to the
    values the library ships with after EVERY GPU test, so that no test sees another test's configuration (the suite runs in one process)."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import candlezip_amd
        candlezip_amd.reset_process_knobs()

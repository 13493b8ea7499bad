import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The parity cases have a handful of series; the engine reserves the wavefront-per-series kernels of dlm_wave48.hip for
# batches above 256 series.  Lift that rule so that the small cases exercise them (tests/test_full_size_gpu.py runs the
# production rule at N = 2000; DLM_NO_WAVE48 inside a test selects the other implementation).
os.environ.setdefault("DLM_FORCE_WAVE48", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files():
    """The sampler fixtures (ref_build_covariances.npz pins the covariance builder: tests/test_covariances.py;
    model_pn_trace_*.npz are the model call traces: tests/test_model_trace_gpu.py)."""
    return sorted(f for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and not f.startswith("ref_build_") and not f.startswith("model_pn_trace_"))


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load the HIP library; never skips -- a missing library is a failure."""
    import pigs_amd
    pigs_amd.build()
    from pigs_amd import _lib
    return _lib.load()

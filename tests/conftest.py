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
    model_pn_trace_*.npz are the model call traces: tests/test_model_trace_gpu.py; ref_loss_curve_*.npz the loss
    curves of the training loop through the reference's functions: tests/test_training_*.py)."""
    return sorted(f for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and not f.startswith(("ref_build_", "model_pn_trace_", "ref_loss_curve_")))


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load the HIP library; never skips -- a missing library is a failure."""
    import pigs_amd
    pigs_amd.build()
    from pigs_amd import _lib
    return _lib.load()


def grads_within_accumulation_bound(got, args, grads, ulps=1e-6, floor=1e-6):
    """The gradient bar that replaces a loosened global tolerance: every entry of the float32 gradients
    ``got`` = (g_means, g_conics_flat, g_values) must lie within ``ulps`` (1e-6 = ~8 float32 ulp) of the sum of
    the ABSOLUTE per-pair contributions to that entry plus ``floor`` of the largest entry (what the cut-off
    may drop), both from the float64 oracle on ``args`` = (means, conics_flat, values, samples) and the
    incoming gradients ``grads`` = {order: array}.  An entry that is a small difference of large
    contributions cannot be summed to 1e-5 of itself in float32 in any order; this is the bar a float32
    dense sum can meet.  Returns [(name, worst error / bound)] of the entries that exceed it."""
    import numpy as np
    from oracle import c_oracle
    want, bound = c_oracle.accumulation_bound(*args, grads, ulps=ulps, floor=floor)
    bad = []
    for name, g, w, b in zip(("means", "conics", "values"), got, want, bound):
        g = g.detach().cpu().double().numpy() if hasattr(g, "detach") else np.asarray(g, dtype=np.float64)
        ratio = np.abs(g.reshape(w.shape) - w) / b
        if not (ratio <= 1.0).all():
            bad.append((name, float(ratio.max())))
    return bad

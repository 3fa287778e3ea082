"""GPU fuzz: the binned path against the dense HIP path (itself pinned to the oracle) on random
problem shapes -- sizes, scales over several orders of magnitude, strong anisotropy, clustered
and duplicated points, Gaussians far outside the sampled region."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Sampler(hip_lib):
    assert torch.cuda.is_available()
    from diff_gaussian_sampling import GaussianSampler
    return GaussianSampler


def make_case(rng):
    N = int(rng.integers(1, 3000))
    M = int(rng.integers(1, 6000))
    c = int(rng.integers(1, 3))
    span = 10.0 ** rng.uniform(-1, 1.5)                       # domain size 0.1 .. 30
    centre = rng.uniform(-5, 5, 2)
    means = centre + rng.uniform(-span, span, (N, 2))
    sig_lo = 10.0 ** rng.uniform(-3, -1) * span
    sig = sig_lo * 10.0 ** rng.uniform(0, rng.uniform(0.1, 2.5), (N, 2))      # up to 300x spread
    rho = np.tanh(rng.normal(0, rng.uniform(0.1, 2.0), N))                  # up to |rho| ~ 0.99
    s0, s1, tau = sig[:, 0] ** 2, sig[:, 1] ** 2, rho * sig[:, 0] * sig[:, 1]
    det = s0 * s1 - tau ** 2
    con = np.stack((s1 / det, -tau / det, s0 / det), -1)
    values = rng.uniform(-1, 1, (N, c))
    kind = rng.integers(0, 4)
    if kind == 0:
        pts = centre + rng.uniform(-1.3 * span, 1.3 * span, (M, 2))
    elif kind == 1:                                             # clusters + duplicates
        k = max(1, M // 50)
        cl = centre + rng.uniform(-span, span, (k, 2))
        pts = cl[rng.integers(0, k, M)] + rng.normal(0, sig_lo * 0.01, (M, 2)) * rng.integers(0, 2, (M, 1))
    elif kind == 2:                                             # regular grid
        r = max(1, int(np.sqrt(M)))
        gx, gy = np.meshgrid(np.linspace(-span, span, r), np.linspace(-span, span, r), indexing="xy")
        pts = centre + np.stack((gx, gy), -1).reshape(-1, 2)
    else:                                                       # a thin line
        t = rng.uniform(-span, span, M)
        pts = centre + np.stack((t, 0.3 * t + 1e-3 * span), -1)
    return means, values, con, pts


SEEDS = int(os.environ.get("PIGS_FUZZ_SEEDS", "60"))     # a longer campaign: PIGS_FUZZ_SEEDS=20000 pytest ... (passes, 80 s)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_binned_matches_dense(Sampler, seed):
    rng = np.random.default_rng(1000 + seed)
    means, values, con, pts = make_case(rng)
    orders = (0, 1, "lap") if seed % 3 == 2 else (0, 1, 2, 3)
    outs, grads = {}, {}
    for backend in ("dense", "binned"):
        t = [torch.tensor(a, dtype=torch.float32, device="cuda") for a in (means, values, con, pts)]
        for x in t[:3]:
            x.requires_grad_(True)
        s = Sampler(True, backend=backend, fuse="all")
        s.preprocess(t[0], t[1], None, t[2], t[3])
        o = s.sample(orders)
        torch.manual_seed(seed)
        loss = sum((x * torch.randn_like(x)).sum() for x in o)
        loss.backward()
        outs[backend] = [x.detach() for x in o]
        grads[backend] = [x.grad for x in t[:3]]
    # The cut-off drops terms below e^(-q_max/2) poly(q_max) of the TERM's own scale, which for
    # derivative order k is |v| lambda^(k/2) (lambda = the conic's larger eigenvalue = 1 / sigma_min^2).
    # The bars are relative to the outputs' maxima or to that scale, whichever is larger: the two agree
    # when some sample point sits in the core of the sharpest Gaussian, and where none does (sparse
    # points, every point in the tails) the outputs' maxima understate what was truncated.
    lam = (con[:, 0] + con[:, 2]) / 2 + np.sqrt(((con[:, 0] - con[:, 2]) / 2) ** 2 + con[:, 1] ** 2)
    vmax = np.abs(values).max(axis=1)
    term = {k: float((vmax * lam ** (k / 2)).max()) for k in (0, 1, 2, 3)}
    term["lap"] = term[2]
    # float32 sums of many overlapping terms of random sign: the two paths add them in different
    # orders, so their difference scales with sum |term| / |sum term| -- measured on order 0 with
    # |values| -- times the float32 epsilon; the bars widen once that ratio passes 50
    sa = Sampler(False, backend="dense")
    ta = [torch.tensor(a, dtype=torch.float32, device="cuda") for a in (means, np.abs(values), con, pts)]
    sa.preprocess(ta[0], ta[1], None, ta[2], ta[3])
    cond = float(sa.sample_gaussians().max()) / max(float(outs["dense"][0].abs().max()), 1e-30)
    slack = max(1.0, cond / 50.0)
    under = 1.0
    for o, a, b in zip(orders, outs["dense"], outs["binned"]):
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), ("order", o, seed)
        top = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-5 * slack * max(top, term[o]) + 1e-30, ("order", o, seed)
        under = max(under, term[o] / max(top, 1e-30))
    for k, (a, b) in enumerate(zip(grads["dense"], grads["binned"])):
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), ("grad", k, seed)
        # random-sign weights cancel in the gradient sums: fp32 accumulation-order noise only.  The bar is
        # that of a self-comparison, not of parity: in the one seed of 20 000 that came nearest (13186: points
        # 26 units from the origin, sigma ~ 0.01; the two paths 8.1e-5 apart in the means' gradient) BOTH
        # paths are 2.8e-3 from the float64 result on the same float32 inputs (tools/fuzz_one.py 13186 --oracle):
        # what separates them is the order of the additions, not the cut-off
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) * under + 1e-30, ("grad", k, seed)

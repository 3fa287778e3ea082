"""GPU parity at the BASELINE.json sizes: oracle comparison where the oracle finishes in seconds,
size-independent properties beyond that."""
import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu
F32_TOL = 1e-5


def rel(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)   # float32 cannot hold less


@pytest.fixture(scope="module")
def Sampler(hip_lib):
    assert torch.cuda.is_available()
    from diff_gaussian_sampling import GaussianSampler
    return GaussianSampler


@pytest.mark.parametrize("kappa", [0.5, 1.3])
def test_config2_8k_x_256sq_fwd_bwd(Sampler, kappa):
    """BASELINE.json configs[1]: 8k Gaussians x 256^2 grid, fwd + deriv + bwd, vs the oracle on a
    4096-point subsample (forward) and on the whole grid through linearity (backward)."""
    from pigs_amd import synthetic
    gs, pts = synthetic.CONFIGS["c2"](kappa)
    t = {k: v.float().cuda() for k, v in gs.items()}
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    pts = pts.float().cuda()
    s = Sampler(False, fuse="all")
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
    u, ux, uxx = s.sample((0, 1, 2))
    idx = torch.arange(0, pts.shape[0], 16, device="cuda")
    args = [t[k].detach().cpu().double().numpy() for k in ("means", "conics", "values")]
    exp = c_oracle.forward(*args, pts[idx].cpu().double().numpy(), orders=(0, 1, 2))
    for o, out in enumerate((u, ux, uxx)):
        assert rel(out[idx], exp[o]) < F32_TOL, o
    # backward of a loss supported on the subsample only equals the oracle's backward there
    g = torch.Generator(device="cpu").manual_seed(5)
    rs = [torch.rand(e.shape, generator=g, dtype=torch.float64) * 2 - 1 for e in (exp[0], exp[1], exp[2])]
    loss = sum((out[idx] * r.float().cuda()).sum() for out, r in zip((u, ux, uxx), rs))
    loss.backward()
    em, ec, ev = c_oracle.backward(*args, pts[idx].cpu().double().numpy(),
                                   {o: r.float().double().numpy() for o, r in enumerate(rs)})
    assert rel(t["means"].grad, em) < F32_TOL
    assert rel(t["values"].grad, ev) < F32_TOL
    assert rel(t["conics"].grad, ec) < F32_TOL

"""GaussianSampler.residual(): r = a0 u + a1 . grad u + lap (u_xx + u_yy) - target in one launch (extension,
SURVEY.md 8f-4; the reference's diffusion / wave residuals model_pn.py:612-617, 834-849, test_no_mlp.py:127-144)
against the oracle's outputs composed the same way, forward and backward, dense and binned, both hosts."""
import numpy as np
import pytest
import torch

from conftest import grads_within_accumulation_bound
from oracle import c_oracle
from pigs_amd import synthetic

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def compose(exp, coeffs, target, d):
    a0, a1, aL = coeffs
    r = a0 * exp[0]
    for i in range(d):
        r = r + a1[i] * exp[1][:, i]
    lap = sum(exp[2][:, i, i] for i in range(d))
    return r + aL * lap - (0 if target is None else target)


def incoming(w, coeffs, d, c):
    """The gradients that arrive at orders 0, 1, 2 when w [M, c] arrives at r."""
    a0, a1, aL = coeffs
    M = w.shape[0]
    g1 = np.stack([a1[i] * w for i in range(d)], 1)
    g2 = np.zeros((M, d, d, c))
    for i in range(d):
        g2[:, i, i] = aL * w
    return {0: a0 * w, 1: g1, 2: g2}


@pytest.mark.parametrize("host", ["native", "ctypes"])
@pytest.mark.parametrize("dtype,backend,d,c", [(torch.float32, "dense", 2, 1), (torch.float32, "binned", 2, 1),
                                               (torch.float32, "binned", 2, 2), (torch.float64, "dense", 2, 2),
                                               (torch.float32, "dense", 1, 1), (torch.float64, "dense", 1, 2)])
def test_residual_matches_composed_oracle(hip_lib, host, dtype, backend, d, c):
    from diff_gaussian_sampling import GaussianSampler
    rng = np.random.default_rng(7 * d + c)
    N, M = 400, 3000
    means = rng.uniform(-1, 1, (N, d))
    if d == 2:
        s0 = np.exp(2 * rng.normal(-3.0, 0.4, (N, 2)))
        tau = np.tanh(rng.normal(0, 0.6, N)) * np.sqrt(s0[:, 0] * s0[:, 1])
        det = s0[:, 0] * s0[:, 1] - tau ** 2
        con = np.stack((s0[:, 1] / det, -tau / det, s0[:, 0] / det), -1)
    else:
        con = 1.0 / np.exp(2 * rng.normal(-3.0, 0.4, (N, 1)))
    values = rng.uniform(-1, 1, (N, c))
    pts = rng.uniform(-1, 1, (M, d))
    target = rng.uniform(-1, 1, (M, c))
    coeffs = (1.7, tuple(rng.uniform(-0.2, 0.2, d)), -0.003)
    t = [torch.as_tensor(a, dtype=dtype, device="cuda") for a in (means, values, con, pts, target)]
    for x in t[:3] + [t[4]]:
        x.requires_grad_(True)
    s = GaussianSampler(True, backend=backend, host=host)
    s.preprocess(t[0], t[1], None, t[2], t[3])
    assert (s._plan is not None) == (backend == "binned")
    r = s.residual(a0=coeffs[0], a1=coeffs[1], lap=coeffs[2], target=t[4])
    assert tuple(r.shape) == (M, c) and r.dtype == dtype
    args = [x.detach().cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]
    tg = t[4].detach().cpu().double().numpy()
    exp = c_oracle.forward(*args, orders=(0, 1, 2))
    want = compose(exp, coeffs, tg, d)
    # the bar is on the scale of the TERMS: the residual is a difference of them (lap u alone is ~1e3 u here)
    scale = max(abs(coeffs[0]) * np.abs(exp[0]).max(), abs(coeffs[2]) * np.abs(exp[2]).max(), np.abs(tg).max())
    tol = 1e-5 if dtype == torch.float32 else 1e-11
    assert np.abs(r.detach().cpu().double().numpy() - want).max() / scale < tol
    w = rng.uniform(-1, 1, (M, c))
    (r * torch.as_tensor(w, dtype=dtype, device="cuda")).sum().backward()
    g64 = incoming(w, coeffs, d, c)
    if dtype == torch.float32:
        bad = grads_within_accumulation_bound((t[0].grad, t[2].grad, t[1].grad), args, g64)
        assert not bad, bad
    else:
        gm, gc, gv = c_oracle.backward(*args, g64)
        assert rel(t[0].grad, gm) < tol and rel(t[2].grad, gc) < tol and rel(t[1].grad, gv) < tol
    assert rel(t[4].grad, -w) < 1e-6
    # without a target, without a1
    r2 = s.residual(a0=0.5, lap=1.0)
    want2 = compose(exp, (0.5, (0.0,) * d, 1.0), None, d)
    assert np.abs(r2.detach().cpu().double().numpy() - want2).max() / np.abs(want2).max() < tol


def test_residual_equals_composed_outputs_at_c3_size(hip_lib):
    """BASELINE configs[2] size (65 536 x 1024^2, binned): the one-launch residual equals the composition of the
    sampler's own u, grad u and Hessian-trace outputs at every point, and its backward equals theirs
    (linearity: the same launch arithmetic, gradients formed on the fly)."""
    from diff_gaussian_sampling import GaussianSampler
    gs, pts = synthetic.CONFIGS["c3"](0.5)
    t = {k: v.float().cuda() for k, v in gs.items()}
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    pts = pts.float().cuda()
    s = GaussianSampler(False, backend="binned")
    s.preprocess(t["means"], t["values"], None, t["conics"], pts)
    a0, a1, aL = 2.0, (0.3, -0.1), -0.01
    gen = torch.Generator().manual_seed(0)
    target = torch.rand((pts.shape[0], 1), generator=gen).cuda()
    w = (torch.rand((pts.shape[0], 1), generator=gen) * 2 - 1).cuda()
    r = s.residual(a0=a0, a1=a1, lap=aL, target=target)
    g_r = torch.autograd.grad((r * w).sum(), (t["means"], t["values"], t["conics"]))
    u, du, lap = s.sample((0, 1, "lap"))
    comp = a0 * u + a1[0] * du[:, 0] + a1[1] * du[:, 1] + aL * lap - target
    g_c = torch.autograd.grad((comp * w).sum(), (t["means"], t["values"], t["conics"]))
    scale = float(max(a0 * u.detach().abs().max(), abs(aL) * lap.detach().abs().max()))
    assert float((r - comp).detach().abs().max()) / scale < 2e-6
    for a, b in zip(g_r, g_c):
        assert float((a - b).abs().max() / b.abs().max()) < 5e-6


def test_diffusion_loss_through_residual(hip_lib):
    """test_no_mlp.py:127-144: mean(((u - u_prev) / dt - D lap u)^2) written with the three sample_*() calls and
    with residual(): same loss, same gradients."""
    from diff_gaussian_sampling import GaussianSampler
    gs = synthetic.lattice_gaussians(20, 20, 1.1, seed=5)
    t = {k: v.float().cuda() for k, v in gs.items()}
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    gen = torch.Generator().manual_seed(1)
    pts = (torch.rand((1024, 2), generator=gen) * 2 - 1).cuda()
    u_prev = torch.rand((1024, 1), generator=gen).cuda()
    dt, D = 0.01, 0.05
    s = GaussianSampler(False)
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
    u, uxx = s.sample_gaussians(), s.sample_gaussians_laplacian()
    loss_ref = (((u - u_prev) / dt - D * (uxx[:, 0, 0] + uxx[:, 1, 1])) ** 2).mean()
    g_ref = torch.autograd.grad(loss_ref, (t["means"], t["values"], t["conics"]))
    loss = s.residual(a0=1 / dt, lap=-D, target=u_prev / dt).pow(2).mean()
    g = torch.autograd.grad(loss, (t["means"], t["values"], t["conics"]))
    assert abs(float(loss) - float(loss_ref)) / float(loss_ref) < 1e-5
    for a, b in zip(g, g_ref):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-5

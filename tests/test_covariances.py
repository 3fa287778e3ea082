"""Covariance builder (gaussians.py:163-193): the numpy oracle against the reference's own outputs
(CPU), and the fused HIP operator against the oracle and the golden vectors (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import covariances_numpy as oracle

FIX = os.path.join(GOLDEN, "ref_build_covariances.npz")


def rel(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_oracle_matches_reference_outputs():
    d = np.load(FIX)
    cov, con = oracle.build_covariances(d["scaling"], d["transform"])
    assert rel(cov, d["cov"]) < 1e-14
    assert np.abs(con / d["conic"] - 1).max() < 1e-12          # element-wise: conics span 6 decades
    # the flat triples are entries [0, 1, 3] of the reference's full matrices
    assert np.array_equal(d["cov"], d["full_cov"].reshape(-1, 4)[:, [0, 1, 3]])
    assert np.allclose(d["conic"], d["full_conic"].reshape(-1, 4)[:, [0, 1, 3]], rtol=1e-13, atol=0)


def test_oracle_backward_matches_reference_autograd():
    d = np.load(FIX)
    gs, gt = oracle.build_covariances_backward(d["scaling"], d["transform"], d["r_cov"], d["r_conic"])
    assert rel(gs, d["g_scaling"]) < 1e-12
    assert rel(gt, d["g_transform"]) < 1e-12
    # one-sided incoming gradients
    gs0, gt0 = oracle.build_covariances_backward(d["scaling"], d["transform"], d["r_cov"], None)
    gs1, gt1 = oracle.build_covariances_backward(d["scaling"], d["transform"], None, d["r_conic"])
    assert rel(gs0 + gs1, d["g_scaling"]) < 1e-12 and rel(gt0 + gt1, d["g_transform"]) < 1e-12


def test_oracle_conic_is_the_inverse():
    rng = np.random.default_rng(0)
    s = np.exp(rng.normal(-3, 1, (50, 2)))
    t = rng.normal(0, 2, (50, 1))
    cov, con = oracle.build_covariances(s, t)
    for k in range(50):
        S = np.array([[cov[k, 0], cov[k, 1]], [cov[k, 1], cov[k, 2]]])
        C = np.array([[con[k, 0], con[k, 1]], [con[k, 1], con[k, 2]]])
        assert np.allclose(S @ C, np.eye(2), atol=1e-9)


def test_cpu_tensors_are_rejected(hip_lib):
    from pigs_amd import covariances
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        covariances.build_covariances(torch.ones(4, 2), torch.zeros(4, 1))
    with pytest.raises(NotImplementedError):
        covariances.build_covariances(torch.ones(4, 3), torch.zeros(4, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 1e-5)])
def test_hip_matches_golden_and_oracle(hip_lib, dtype, tol):
    from pigs_amd import covariances
    d = np.load(FIX)
    s = torch.as_tensor(d["scaling"], dtype=dtype, device="cuda").requires_grad_(True)
    t = torch.as_tensor(d["transform"], dtype=dtype, device="cuda").requires_grad_(True)
    cov, con = covariances.build_covariances(s, t)
    assert cov.shape == (s.shape[0], 3) and con.shape == (s.shape[0], 3)
    assert rel(cov, d["cov"]) < tol and rel(con, d["conic"]) < tol
    # element-wise too (the global-max metric hides the small conics): float32 keeps 1e-5 per element
    assert np.abs(con.detach().cpu().double().numpy() / d["conic"] - 1).max() < max(tol, 2e-6) * 5
    r1 = torch.as_tensor(d["r_cov"], dtype=dtype, device="cuda")
    r2 = torch.as_tensor(d["r_conic"], dtype=dtype, device="cuda")
    gs, gt = torch.autograd.grad((cov * r1).sum() + (con * r2).sum(), (s, t))
    assert gt.shape == t.shape
    assert rel(gs, d["g_scaling"]) < tol and rel(gt, d["g_transform"]) < tol
    # only one output used: the other incoming gradient is absent (NULL at the C ABI)
    cov, con = covariances.build_covariances(s, t)
    (gs1,) = torch.autograd.grad((con * r2).sum(), (s,))
    e1, _ = oracle.build_covariances_backward(d["scaling"], d["transform"], None, d["r_conic"])
    assert rel(gs1, e1) < tol


@pytest.mark.gpu
def test_hip_full_matrices_and_sampler_round_trip(hip_lib):
    """build_full_covariances -> flatten_covariances -> preprocess, as test_gaussian_sampling.py:36-56 does."""
    from pigs_amd import covariances, synthetic
    from diff_gaussian_sampling import GaussianSampler
    from oracle import c_oracle
    g = torch.Generator().manual_seed(2)
    N = 300
    s = torch.exp(torch.randn((N, 2), generator=g) * 0.3 - 5.0).cuda()
    t = (torch.randn((N, 1), generator=g) * 0.5).cuda()
    full_cov, full_con = covariances.build_full_covariances(s, t)
    assert full_cov.shape == (N, 2, 2) and torch.equal(full_cov[:, 0, 1], full_cov[:, 1, 0])
    eye = torch.eye(2, device="cuda").expand(N, 2, 2)
    assert torch.allclose(full_cov.double() @ full_con.double(), eye.double(), atol=1e-4)
    cov, con = covariances.flatten_covariances(full_cov, full_con)
    ecov, econ = oracle.build_covariances(s.cpu().double().numpy(), t.cpu().double().numpy())
    assert rel(cov, ecov) < 1e-6 and rel(con, econ) < 1e-5
    means = (torch.rand((N, 2), generator=g) * 2 - 1).cuda()
    values = (torch.rand((N, 1), generator=g) * 2 - 1).cuda()
    pts = synthetic.grid_samples(40).float().cuda()
    smp = GaussianSampler(True)
    smp.preprocess(means, values, cov, con, pts)
    u = smp.sample_gaussians()
    exp = c_oracle.forward(means.cpu().double().numpy(), econ, values.cpu().double().numpy(),
                           pts.cpu().double().numpy(), orders=(0,))
    assert rel(u, exp[0]) < 1e-5


@pytest.mark.gpu
def test_hip_extreme_correlation_and_empty(hip_lib):
    from pigs_amd import covariances
    s = torch.tensor([[1e-4, 2e-4], [3.0, 1e-6], [1e-8, 1e-8]], device="cuda")
    t = torch.tensor([[6.0], [-7.5], [0.0]], device="cuda")
    cov, con = covariances.build_covariances(s, t)
    ecov, econ = oracle.build_covariances(s.cpu().double().numpy(), t.cpu().double().numpy())
    def elementwise(a, e):          # relative where the expectation is not zero, exact zero where it is
        a = a.cpu().double().numpy()
        return np.abs(np.where(e == 0, a, a / np.where(e == 0, 1, e) - 1)).max()
    assert elementwise(cov, ecov) < 1e-5
    assert elementwise(con, econ) < 1e-4      # 1 - tanh^2 without cancellation (naive float32: 1e-2 at t = 6)
    cov, con = covariances.build_covariances(torch.empty((0, 2), device="cuda"), torch.empty((0, 1), device="cuda"))
    assert cov.shape == (0, 3) and con.shape == (0, 3)

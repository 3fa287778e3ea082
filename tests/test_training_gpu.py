"""GPU end-to-end: an MLP-free PINN training loop driven through the sampler, against the same
loop driven through the torch oracle on the CPU (loss-curve parity; BASELINE.json configs[4] in
the form that needs only the sampler).

Loop shape follows /root/reference/test_no_mlp.py:84-186: the Gaussians are the parameters
(raw means through tanh, log-variances through exp, raw correlation through tanh), Adam with
lr 1e-2, 1024 random collocation points per step; step 0..9 fit an initial condition
(:107-121), later steps minimise the diffusion residual u_t = u_xx + u_yy against the frozen
previous state (:91-97, :124-144)."""
import math

import numpy as np
import pytest
import torch

from oracle import dense_torch
from pigs_amd import synthetic

pytestmark = pytest.mark.gpu


class OracleSampler:
    """Stand-in with the sampler's surface, computing through oracle/dense_torch.py (checker)."""

    def preprocess(self, means, values, covariances, conics, samples):
        self.args = (means, conics, values, samples)
        self.cache = None

    def _get(self, o):
        if self.cache is None:
            self.cache = dense_torch.forward(*self.args, orders=(0, 1, 2), chunk=1024)
        return self.cache[o]

    def sample_gaussians(self):
        return self._get(0)

    def sample_gaussians_derivative(self):
        return self._get(1)

    def sample_gaussians_laplacian(self):
        return self._get(2)


def run_loop(sampler, device, steps=30, n=16, scale=2.5, dt=0.1):
    g = torch.Generator(device="cpu").manual_seed(7)
    tx = torch.linspace(-1, 1, n) * 0.6
    gx, gy = torch.meshgrid((tx, tx), indexing="ij")
    raw_means = torch.atanh(torch.stack((gx, gy), dim=-1).reshape(n * n, 2)).to(device).requires_grad_(True)
    raw_scaling = torch.full((n * n, 2), -3.0, device=device, requires_grad=True)
    transform = torch.zeros((n * n, 1), device=device, requires_grad=True)
    values = (0.1 * torch.rand((n * n, 1), generator=g)).to(device).requires_grad_(True)
    optim = torch.optim.Adam([raw_means, values, raw_scaling, transform], lr=1e-2)

    def gaussians():
        means = torch.tanh(raw_means) * scale
        cov, con = synthetic.covariances_from_raw(torch.exp(raw_scaling), transform)
        return means, cov, con

    losses, prev = [], None
    for it in range(steps):
        samples = ((torch.rand((1024, 2), generator=g) * 2 - 1) * scale).to(device)
        if it == 10:                      # freeze the fitted state as the previous time level
            with torch.no_grad():
                means, cov, con = gaussians()
                prev = (means.clone(), values.detach().clone(), cov.clone(), con.clone())
        means, cov, con = gaussians()
        sampler.preprocess(means, values, cov, con, samples)
        u = sampler.sample_gaussians()
        if it < 10:
            desired = torch.exp(-0.5 * (samples ** 2).sum(-1) / (0.1 * scale))
            loss = torch.mean((u[:, 0] - desired) ** 2)
        else:
            uxx = sampler.sample_gaussians_laplacian()
            ux = sampler.sample_gaussians_derivative()
            with torch.no_grad():
                sampler2 = type(sampler)() if isinstance(sampler, OracleSampler) else sampler.__class__(False, backend=sampler.backend)
                sampler2.preprocess(*prev, samples)
                u_prev = sampler2.sample_gaussians()
            ut = (u - u_prev) / dt
            loss = torch.mean((ut[:, 0] - (uxx[:, 0, 0, 0] + uxx[:, 1, 1, 0])) ** 2) + 1e-3 * torch.mean(ux ** 2)
        optim.zero_grad()
        loss.backward()
        optim.step()
        losses.append(float(loss.detach()))
    return np.array(losses)


def test_training_loss_curve_matches_oracle(hip_lib):
    assert torch.cuda.is_available()
    from diff_gaussian_sampling import GaussianSampler
    gpu = run_loop(GaussianSampler(True), torch.device("cuda"))
    torch.set_num_threads(8)
    cpu = run_loop(OracleSampler(), torch.device("cpu"))
    assert np.isfinite(gpu).all()
    assert gpu[9] < gpu[0]                                   # the fit converges
    assert gpu[-1] < gpu[10]                                 # the residual decreases
    rel = np.abs(gpu - cpu) / np.maximum(np.abs(cpu), 1e-12)
    assert rel.max() < 2e-3, (rel.max(), gpu, cpu)           # fp32 trajectories stay together over 30 steps


def test_training_loss_curve_matches_the_reference_functions(hip_lib):
    """The same loop against tests/golden/ref_loss_curve_no_mlp.npz: the loss per step with the REFERENCE's own
    gaussians.sample_gaussians / gaussian_derivative / gaussian_derivative2 as the sampler (tools/gen_loss_curve.py,
    float32 on the CPU, recorded in the build container) -- BASELINE config 5's "loss-curve parity vs reference" in
    the form that needs only the sampler."""
    import os
    from conftest import GOLDEN
    from diff_gaussian_sampling import GaussianSampler
    ref = np.load(os.path.join(GOLDEN, "ref_loss_curve_no_mlp.npz"))["losses"]
    gpu = run_loop(GaussianSampler(False), torch.device("cuda"), steps=len(ref))
    rel = np.abs(gpu - ref) / np.maximum(np.abs(ref), 1e-12)
    assert rel.max() < 2e-3, (rel.max(), gpu, ref)           # float32 trajectories stay together over 30 steps


def test_training_loss_curve_through_the_binned_path(hip_lib):
    """The optimiser loop on the BINNED path (plans rebuilt every step on new random collocation points, the culled
    forward and the two-cut-off backward inside an Adam trajectory) against the reference-function loss curve: the
    truncation at q = 36 / 40 must not move a 30-step float32 trajectory more than the dense path's own 2e-3."""
    import os
    from conftest import GOLDEN
    from diff_gaussian_sampling import GaussianSampler
    ref = np.load(os.path.join(GOLDEN, "ref_loss_curve_no_mlp.npz"))["losses"]
    s = GaussianSampler(False, backend="binned")
    gpu = run_loop(s, torch.device("cuda"), steps=len(ref))
    assert s._plan is not None
    rel = np.abs(gpu - ref) / np.maximum(np.abs(ref), 1e-12)
    assert rel.max() < 2e-3, (rel.max(), gpu, ref)

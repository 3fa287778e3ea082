"""Seeded synthetic workloads of the benchmark configurations (BASELINE.json).

Everything is generated on the CPU with a seeded ``torch.Generator`` (bit-identical across
machines) in float64 and cast/moved by the caller.  The parameterisation follows the reference:
variances ``s`` and raw off-diagonal ``t`` -> covariance [[s0, tau], [tau, s1]] with
tau = tanh(t) sqrt(s0 s1), conic = inverse (/root/reference/gaussians.py:163-183), flattened
to [xx, xy, yy] (:186-189); means on an ``indexing="ij"`` lattice (model_pn.py:338-342); sample
grids ``linspace(-1, 1, res)`` with ``indexing="xy"`` (test_gaussian_sampling.py:43-46).
"""
import math

import torch


def covariances_from_raw(s, t):
    """s [N,2] variances, t [N,1] raw correlation -> (cov_flat [N,3], conic_flat [N,3])."""
    tau = torch.tanh(t[:, 0]) * torch.sqrt(s[:, 0] * s[:, 1])
    det = s[:, 0] * s[:, 1] - tau * tau
    cov = torch.stack((s[:, 0], tau, s[:, 1]), dim=-1)
    con = torch.stack((s[:, 1] / det, -tau / det, s[:, 0] / det), dim=-1)
    return cov, con


def lattice_gaussians(nx, ny, kappa, seed=0, c=1, jitter=True):
    """nx*ny anisotropic Gaussians on a jittered lattice over [-1,1]^2.

    sigma ~ kappa * spacing (spacing = sqrt(4/N), the equal-area cell side) with a log-normal
    spread; kappa = 0.5 is the sparse regime (about 28 Gaussians within q <= 36 of a point),
    kappa = 1.3 is reference-like (e^-4 variances on a 20x20 lattice, model_pn.py:344).
    Returns dict(means [N,2], values [N,c], covariances [N,3], conics [N,3]) in float64.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    N = nx * ny
    tx = torch.linspace(-1, 1, nx, dtype=torch.float64)
    ty = torch.linspace(-1, 1, ny, dtype=torch.float64)
    gx, gy = torch.meshgrid((tx, ty), indexing="ij")
    means = torch.stack((gx, gy), dim=-1).reshape(N, 2)
    if jitter:
        step = torch.tensor([2.0 / max(nx - 1, 1), 2.0 / max(ny - 1, 1)], dtype=torch.float64)
        means = means + (torch.rand((N, 2), generator=g, dtype=torch.float64) - 0.5) * step
    spacing = math.sqrt(4.0 / N)
    logvar = 2.0 * math.log(kappa * spacing) + 0.25 * torch.randn((N, 2), generator=g, dtype=torch.float64)
    s = torch.exp(logvar)
    t = 0.5 * torch.randn((N, 1), generator=g, dtype=torch.float64)
    values = torch.rand((N, c), generator=g, dtype=torch.float64) * 2 - 1
    cov, con = covariances_from_raw(s, t)
    return dict(means=means, values=values, covariances=cov, conics=con)


def grid_samples(res_x, res_y=None, lo=-1.0, hi=1.0, row0=0, rows=None):
    """[rows*res_x, 2] points of a res_x x res_y grid, x fastest (``indexing='xy'``); ``row0`` /
    ``rows`` select a contiguous block of grid rows (the multi-GPU shard of SURVEY.md 8e)."""
    res_y = res_x if res_y is None else res_y
    rows = res_y - row0 if rows is None else rows
    tx = torch.linspace(lo, hi, res_x, dtype=torch.float64)
    ty = torch.linspace(lo, hi, res_y, dtype=torch.float64)[row0:row0 + rows]
    gx, gy = torch.meshgrid((tx, ty), indexing="xy")
    return torch.stack((gx, gy), dim=-1).reshape(rows * res_x, 2)


def line_gaussians_1d(n, log_variance=-5.0):
    """test_1d.py:11-24: n Gaussians on linspace(-1,1), variance e^log_variance, values exp(-(4x)^2)."""
    means = torch.linspace(-1, 1, n, dtype=torch.float64).reshape(n, 1)
    values = torch.exp(-(means * 4) ** 2)
    cov = torch.full((n, 1), math.exp(log_variance), dtype=torch.float64)
    return dict(means=means, values=values, covariances=cov, conics=1.0 / cov)


CONFIGS = {
    # BASELINE.json configs[0]: 1-D, 256 Gaussians x 4096 points
    "c1": lambda: (line_gaussians_1d(256), torch.linspace(-1, 1, 4096, dtype=torch.float64).reshape(-1, 1)),
    # configs[1]: 8k Gaussians x 256^2 grid
    "c2": lambda kappa=0.5: (lattice_gaussians(128, 64, kappa), grid_samples(256)),
    # configs[2]: 65k Gaussians x 1024^2 grid (roofline run)
    "c3": lambda kappa=0.5: (lattice_gaussians(256, 256, kappa), grid_samples(1024)),
}

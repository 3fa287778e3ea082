#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own PyTorch functions.

Runs ONLY in the build container, where the reference is mounted read-only at
/root/reference.  The reference is imported, never copied: the fixtures hold data
(inputs + expected outputs), not source.  Outputs are produced by

* ``gaussians.sample_gaussians``      (/root/reference/gaussians.py:48-58)   order 0
* ``gaussians.gaussian_derivative``   (/root/reference/gaussians.py:89-101)  order 1
* ``gaussians.gaussian_derivative2``  (/root/reference/gaussians.py:103-116) order 2
  (its stray ``torch.ones(..., device="cuda")`` at :110 is redirected to CPU at call time)
* order 3: ``torch.autograd`` of ``gaussian_derivative2`` wrt the sample points
* parameter gradients: ``torch.autograd`` of L = sum(out * r) wrt (means, values, full conics),
  the comparison the reference makes at test_derivatives.py:122-124, 208-220, 340-356.

Input recipes follow the reference drivers: test_gaussian_sampling.py:13-46,
test_derivatives.py:13-70, test_1d.py:11-27, test_density.py:10-38, test_torus.py:10-33,
gaussians.build_full_covariances (:163-183).
``ref_build_covariances.npz`` pins the covariance builder itself (gaussians.py:163-193).

Usage:  MPLBACKEND=Agg python tools/gen_golden.py
"""
import os
import sys
import unittest.mock

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import gaussians as ref  # the reference module (read-only import)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

_real_ones = torch.ones


def _ones_cpu(*a, **k):
    k.pop("device", None)
    return _real_ones(*a, **k)


def ref_order2(means, full_conics, values, samples):
    with unittest.mock.patch.object(torch, "ones", _ones_cpu):
        return ref.gaussian_derivative2(means, full_conics, values, samples)


def ref_order3(means, full_conics, values, samples):
    """[M,d,d,d,c] by differentiating the reference Hessian wrt the sample points."""
    M, d = samples.shape
    H = ref_order2(means, full_conics, values, samples)  # [M,d,d,c]
    c = H.shape[-1]
    rows = []
    for i in range(d):
        for j in range(d):
            for ch in range(c):
                (gk,) = torch.autograd.grad(H[:, i, j, ch].sum(), samples, create_graph=True)
                rows.append(gk)  # [M, d(k)]
    T = torch.stack(rows, 0).reshape(d, d, c, M, d)      # i j c m k
    return T.permute(3, 0, 1, 4, 2).contiguous()         # m i j k c


def flat_grad(G, d):
    iu = np.triu_indices(d)
    out = G[:, iu[0], iu[1]].copy()
    off = iu[0] != iu[1]
    out[:, off] += G[:, iu[1][off], iu[0][off]]
    return out


def flat_sym(A, d):
    iu = np.triu_indices(d)
    return A[:, iu[0], iu[1]]


def run_case(name, means, values, full_cov, full_conics, samples, seed):
    d = means.shape[1]
    rec = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        mu = means.to(dt).clone().requires_grad_(True)
        v = values.to(dt).clone().requires_grad_(True)
        C = full_conics.to(dt).clone().requires_grad_(True)
        s = samples.to(dt).clone().requires_grad_(True)
        outs = {
            0: ref.sample_gaussians(mu, C, v, s),
            1: ref.gaussian_derivative(mu, C, v, s),
            2: ref_order2(mu, C, v, s),
            3: ref_order3(mu, C, v, s),
        }
        gen = torch.Generator().manual_seed(seed)
        for o, out in outs.items():
            rec[f"out{o}_{tag}"] = out.detach().numpy()
            r = torch.rand(out.shape, generator=gen, dtype=torch.float64).to(dt) * 2 - 1
            if tag == "f64":
                rec[f"r{o}"] = r.numpy()
            gm, gv, gC = torch.autograd.grad((out * r).sum(), (mu, v, C), retain_graph=True)
            rec[f"gmeans{o}_{tag}"] = gm.numpy()
            rec[f"gvalues{o}_{tag}"] = gv.numpy()
            rec[f"gconics_full{o}_{tag}"] = gC.numpy()
            rec[f"gconics{o}_{tag}"] = flat_grad(gC.numpy(), d)
    rec["means"] = means.double().numpy()
    rec["values"] = values.double().numpy()
    rec["conics_full"] = full_conics.double().numpy()
    rec["covariances_full"] = full_cov.double().numpy()
    rec["conics"] = flat_sym(rec["conics_full"], d)
    rec["covariances"] = flat_sym(rec["covariances_full"], d)
    rec["samples"] = samples.double().numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: N={means.shape[0]} M={samples.shape[0]} d={d} c={values.shape[1]} -> "
          f"{os.path.getsize(path) / 1024:.0f} KiB")


def grid_samples(res, scale=1.0):
    # test_gaussian_sampling.py:42-46 / test_derivatives.py:60-64
    tx = torch.linspace(-1, 1, res, dtype=torch.float64) * scale
    ty = torch.linspace(-1, 1, res, dtype=torch.float64) * scale
    gx, gy = torch.meshgrid((tx, ty), indexing="xy")
    return torch.stack((gx, gy), dim=-1).reshape(res * res, 2)


def case_gaussian_sampling():
    # test_gaussian_sampling.py:13-46
    nx = ny = 10
    d = 2
    tx = torch.linspace(-1, 1, nx, dtype=torch.float64)
    ty = torch.linspace(-1, 1, ny, dtype=torch.float64)
    gx, gy = torch.meshgrid((tx, ty), indexing="ij")
    means = torch.stack((gx, gy), dim=-1).reshape(nx * ny, d)
    scaling = torch.ones((nx * ny, d), dtype=torch.float64) * -4.0
    transform = torch.zeros((nx * ny, 1), dtype=torch.float64)
    conics = torch.inverse(torch.diag(torch.ones(d, dtype=torch.float64)) * 0.1)
    sample_mean = torch.tensor([0.1, 0.4], dtype=torch.float64).reshape(1, d, 1)
    x = means.unsqueeze(-1) - sample_mean
    values = torch.exp(-0.5 * (x.transpose(-1, -2) @ (conics @ x))).squeeze(-1)
    scaling = torch.exp(scaling)
    transform = torch.tanh(transform)
    full_cov, full_con = ref.build_full_covariances(scaling, transform)
    samples = grid_samples(256)
    idx = torch.randperm(samples.shape[0], generator=torch.Generator().manual_seed(1))[:1536]
    idx, _ = torch.sort(idx)
    run_case("ref_test_gaussian_sampling", means, values, full_cov, full_con, samples[idx], seed=11)


def case_derivatives():
    # test_derivatives.py:13-70
    nx = ny = 10
    d = 2
    tx = torch.linspace(-1, 1, nx, dtype=torch.float64)
    ty = torch.linspace(-1, 1, ny, dtype=torch.float64)
    gx, gy = torch.meshgrid((tx, ty), indexing="ij")
    means = torch.stack((gx, gy), dim=-1).reshape(nx, ny, d)
    scaling = torch.ones((nx, ny, d), dtype=torch.float64) * -3.5
    transform = torch.full((nx, ny, 1), 0.5, dtype=torch.float64)
    s4 = means.unsqueeze(-1) * 4
    values = torch.exp(-(s4.transpose(-1, -2) @ s4)).squeeze(-1)
    scaling = torch.exp(scaling)
    full_cov, full_con = ref.build_full_covariances(scaling, transform)
    samples = grid_samples(64)
    idx = torch.arange(0, samples.shape[0], 3)
    run_case("ref_test_derivatives", means.reshape(-1, d), values.reshape(-1, 1),
             full_cov.reshape(-1, d, d), full_con.reshape(-1, d, d), samples[idx], seed=12)


def case_1d():
    # test_1d.py:11-27
    n, d = 20, 1
    means = torch.linspace(-1, 1, n, dtype=torch.float64).reshape(-1, 1)
    scaling = torch.ones((n, d), dtype=torch.float64) * -5.0
    s4 = means.unsqueeze(-1) * 4
    values = torch.exp(-(s4.transpose(-1, -2) @ s4)).squeeze(-1)
    cov = torch.exp(scaling)
    con = 1.0 / cov
    samples = torch.linspace(-1, 1, 200, dtype=torch.float64).reshape(-1, 1)
    run_case("ref_test_1d", means, values, cov.reshape(n, 1, 1), con.reshape(n, 1, 1), samples, seed=13)


def case_density():
    # test_density.py:10-38: a jittered 10 x 10 lattice of isotropic Gaussians (variance e^-4), values 0.5, 64^2 grid
    nx = ny = 10
    d = 2
    g = torch.Generator().manual_seed(21)
    tx = torch.linspace(-1, 1, nx, dtype=torch.float64)
    ty = torch.linspace(-1, 1, ny, dtype=torch.float64)
    gx, gy = torch.meshgrid((tx, ty), indexing="ij")
    means = torch.stack((gx, gy), dim=-1).reshape(nx * ny, d) \
        + (torch.rand((nx * ny, d), generator=g, dtype=torch.float64) * 2.0 - 1.0) * 0.1
    scaling = torch.exp(torch.ones((nx * ny, d), dtype=torch.float64) * -4.0)
    transform = torch.tanh(torch.zeros((nx * ny, d * (d - 1) // 2), dtype=torch.float64))
    full_cov, full_con = ref.build_full_covariances(scaling, transform)
    values = torch.ones((nx * ny, 1), dtype=torch.float64) * 0.5
    samples = grid_samples(64)
    run_case("ref_test_density", means, values, full_cov, full_con, samples[::3], seed=14)


def case_torus():
    # test_torus.py:10-33: ten Gaussians (variance e^-3) in a column at x = -0.95, right at the border of the 128^2
    # sample grid.  The script's name says what the native sampler may do there (wrap around); the reference's PyTorch
    # twin -- what these fixtures pin -- does not: dense, non-periodic sums (parity unpinned (2)).
    n, d = 10, 2
    ty = torch.linspace(-1, 1, n, dtype=torch.float64)
    means = torch.stack((torch.ones(n, dtype=torch.float64) * -0.95, ty), dim=-1)
    scaling = torch.exp(torch.ones((n, d), dtype=torch.float64) * -3.0)
    transform = torch.zeros((n, d * (d - 1) // 2), dtype=torch.float64)
    full_cov, full_con = ref.build_full_covariances(scaling, transform)
    values = torch.ones((n, 1), dtype=torch.float64) * 0.5
    samples = grid_samples(128)
    run_case("ref_test_torus", means, values, full_cov, full_con, samples[::11], seed=15)


def case_random(name, N, M, d, c, seed):
    g = torch.Generator().manual_seed(seed)
    means = torch.rand((N, d), generator=g, dtype=torch.float64) * 2 - 1
    s = torch.exp(-3.0 + 0.5 * torch.randn((N, d), generator=g, dtype=torch.float64))
    t = 0.7 * torch.randn((N, d * (d - 1) // 2), generator=g, dtype=torch.float64)
    values = torch.rand((N, c), generator=g, dtype=torch.float64) * 2 - 1
    full_cov, full_con = ref.build_full_covariances(s, t)
    samples = torch.rand((M, d), generator=g, dtype=torch.float64) * 2.2 - 1.1
    run_case(name, means, values, full_cov, full_con, samples, seed=seed + 100)


def case_build_covariances():
    """gaussians.build_covariances / build_full_covariances (/root/reference/gaussians.py:163-193):
    (scaling [N,2] > 0, transform [N,1]) -> flat covariances and conics [N,3], with the gradients of
    L = sum(cov * r1) + sum(conic * r2) wrt both inputs (torch.autograd through the reference)."""
    g = torch.Generator().manual_seed(11)
    N = 257
    s = torch.exp(torch.randn((N, 2), generator=g, dtype=torch.float64) * 1.5 - 4.0).requires_grad_(True)
    t = (torch.randn((N, 1), generator=g, dtype=torch.float64) * 1.2).requires_grad_(True)
    full_cov, full_con = ref.build_full_covariances(s, t)
    cov, con = ref.flatten_covariances(full_cov, full_con)
    r1 = torch.rand(cov.shape, generator=g, dtype=torch.float64) * 2 - 1
    r2 = torch.rand(con.shape, generator=g, dtype=torch.float64) * 2 - 1
    gs, gt = torch.autograd.grad((cov * r1).sum() + (con * r2).sum(), (s, t))
    cov32, con32 = ref.build_covariances(s.detach().float(), t.detach().float())
    path = os.path.join(OUT, "ref_build_covariances.npz")
    np.savez_compressed(path, scaling=s.detach().numpy(), transform=t.detach().numpy(),
                        cov=cov.detach().numpy(), conic=con.detach().numpy(),
                        full_cov=full_cov.detach().numpy(), full_conic=full_con.detach().numpy(),
                        cov_f32=cov32.numpy(), conic_f32=con32.numpy(),
                        r_cov=r1.numpy(), r_conic=r2.numpy(), g_scaling=gs.numpy(), g_transform=gt.numpy())
    print("wrote", path)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--only-covariances" in sys.argv:      # keeps the other fixtures' bytes untouched
        case_build_covariances()
        return
    if "--only-density-torus" in sys.argv:
        case_density()
        case_torus()
        return
    case_build_covariances()
    case_gaussian_sampling()
    case_derivatives()
    case_1d()
    case_density()
    case_torus()
    case_random("random_d2_c2", N=97, M=301, d=2, c=2, seed=3)
    case_random("random_d2_c1", N=300, M=777, d=2, c=1, seed=4)
    case_random("random_d1_c2", N=41, M=130, d=1, c=2, seed=5)
    case_random("random_d3_c1", N=23, M=57, d=3, c=1, seed=6)


if __name__ == "__main__":
    main()

"""GPU: preprocess_aggregate / aggregate_neighbors on the HIP kernels (pigs_amd/csrc/aggregate.hip) through
the sampler surface.  PARITY UNPINNED (the reference's arithmetic is not visible): what
is checked is (a) the kernels against the dense torch statement of this repo's definition
(oracle/aggregate_torch.py), forward and all six gradients, float64 and float32, at the reference
test's sizes (E = 21, F = 5, K = 4, L = 2: test_neighbor_aggregation.py:69-98) and at the model's
(E = 25, F = 6, L = 16, K = 16, N ~ 1 600: model_pn.py:44-49, 199-230); (b) the reference's own check,
torch.autograd.gradcheck in float64 on all six arguments (test_neighbor_aggregation.py:96-98); (c) that
nothing of size N x N is allocated."""
import math

import numpy as np
import pytest
import torch

from oracle import aggregate_torch

pytestmark = pytest.mark.gpu


def gaussians(n_side, dtype, spread=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    t = torch.linspace(-1, 1, n_side, dtype=torch.float64)
    gx, gy = torch.meshgrid((t, t), indexing="ij")
    N = n_side * n_side
    means = torch.stack((gx, gy), dim=-1).reshape(N, 2) + (torch.rand((N, 2), generator=g, dtype=torch.float64) - 0.5) * 0.1
    s = torch.exp(torch.randn((N, 2), generator=g, dtype=torch.float64) * 0.3 + math.log(spread * (2.0 / n_side) ** 2))
    tau = torch.tanh(torch.randn(N, generator=g, dtype=torch.float64) * 0.5) * torch.sqrt(s[:, 0] * s[:, 1])
    det = s[:, 0] * s[:, 1] - tau * tau
    conics = torch.stack((s[:, 1] / det, -tau / det, s[:, 0] / det), dim=-1)
    cov = torch.stack((s[:, 0], tau, s[:, 1]), dim=-1)
    return means.to(dtype).cuda(), cov.to(dtype).cuda(), conics.to(dtype).cuda()


def arguments(N, L, K, E, dtype, seed=1):
    g = torch.Generator().manual_seed(seed)
    F = (E - 1) // 4
    shapes = [(N, L), (L, L), (N, K), (N, K), (F,), (L, 2 * E)]
    args = [torch.rand(s, generator=g, dtype=torch.float64) for s in shapes]
    args[4] = torch.randn(F, generator=g, dtype=torch.float64) * 10
    return [a.to(dtype).cuda().requires_grad_(True) for a in args]


def sampler_for(means, cov, conics, **kw):
    from diff_gaussian_sampling import GaussianSampler
    s = GaussianSampler(True, unpinned_aggregate=True, **kw)
    values = torch.ones((means.shape[0], 1), dtype=means.dtype, device=means.device)
    s.preprocess(means, values, cov, conics, means)          # model_pn.py:648: sampling at the means
    s.preprocess_aggregate()
    return s


@pytest.mark.parametrize("n_side,L,K,E,dtype,tol", [
    (5, 2, 4, 21, torch.float64, 1e-11), (5, 2, 4, 21, torch.float32, 2e-5),
    (12, 16, 16, 25, torch.float64, 1e-11), (40, 16, 16, 25, torch.float32, 5e-5), (40, 16, 16, 25, torch.float64, 1e-10)])
def test_kernels_match_dense_definition(hip_lib, n_side, L, K, E, dtype, tol):
    means, cov, conics = gaussians(n_side, dtype, spread=2.0)
    s = sampler_for(means, cov, conics)
    args = arguments(means.shape[0], L, K, E, dtype)
    out = s.aggregate_neighbors(*args)
    r = torch.randn(out.shape, dtype=dtype, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    grads = torch.autograd.grad((out * r).sum(), args)
    # the dense checker, in float64 on the same (rounded) inputs -- on the CPU: torch's float64 einsum /
    # matmul chain of the checker is only good to ~3e-8 on this GPU stack (measured: the kernels agree with
    # the CPU run of the checker to 2e-14, the GPU run of the checker with its own CPU run to 3e-8)
    m64, c64 = means.double().cpu(), conics.double().cpu()
    a64 = [a.detach().double().cpu().requires_grad_(True) for a in args]
    mask, delta, g = aggregate_torch.neighbor_structure(m64, c64, 36.0)
    assert mask.sum(1).max() > 1 and not mask.all()           # a real, sparse neighbour relation
    exp = aggregate_torch.aggregate(mask, delta, g, *a64)
    egrads = torch.autograd.grad((exp * r.double().cpu()).sum(), a64)

    def rel(got, want):
        return float((got.detach().double().cpu() - want.detach()).abs().max() / want.detach().abs().max().clamp_min(1e-300))
    assert rel(out, exp) < tol, rel(out, exp)
    for name, got, want in zip(("features", "transform", "queries", "keys", "frequencies", "distance_transform"), grads, egrads):
        assert got.shape == want.shape
        assert rel(got, want) < tol * (10 if name == "frequencies" else 1), (name, rel(got, want))


def test_gradcheck_all_six_arguments_float64(hip_lib):
    """The reference's own test: torch.autograd.gradcheck(test_func, (features, transform, queries, keys,
    frequencies, distance_transform)) in float64 (test_neighbor_aggregation.py:50-57, 96-98)."""
    means, cov, conics = gaussians(5, torch.float64, spread=1.5)
    s = sampler_for(means, cov, conics)
    args = arguments(25, 2, 4, 21, torch.float64)
    assert torch.autograd.gradcheck(lambda *a: s.aggregate_neighbors(*a), args)


def test_no_dense_pair_tensor_is_allocated(hip_lib):
    """N = 6 400 with L = 16, E = 25: an [N, N, 2E] float32 tensor would be 8 GB; the sparse lists are N x cap int32."""
    means, cov, conics = gaussians(80, torch.float32, spread=1.0)
    N = means.shape[0]
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    s = sampler_for(means, cov, conics, backend="dense")
    args = arguments(N, 16, 16, 25, torch.float32)
    out = s.aggregate_neighbors(*args)
    out.sum().backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < 3 * N * N * 4 + (64 << 20), peak           # the two index slabs (+ the sampler's own outputs), far below N*N*2E*4
    assert peak < N * N * 50 * 4 / 8
    assert all(a.grad is not None and torch.isfinite(a.grad).all() for a in args)


def test_warns_once_that_parity_is_unpinned(hip_lib):
    import warnings
    from diff_gaussian_sampling import GaussianSampler
    from pigs_amd import sampler as S
    S.GaussianSampler._warned_aggregate = False
    means, cov, conics = gaussians(4, torch.float32)
    s = GaussianSampler(False)
    s.preprocess(means, torch.ones((16, 1), device="cuda"), cov, conics, means)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s.preprocess_aggregate()
        s.preprocess_aggregate()
    assert sum("parity" in str(x.message) or "own definition" in str(x.message) for x in w) == 1

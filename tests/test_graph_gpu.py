"""hipGraph capture of the hot path.  include/pigs_amd.h promises that no entry point allocates,
frees or synchronises; here preprocess + fused sample + backward are captured once into a
`torch.cuda.CUDAGraph` (a hipGraph on ROCm) and replayed on NEW parameter values written into
the captured buffers, and every replay is checked against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from pigs_amd import synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(a, b):
    a = a.detach().cpu().double().numpy()
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("backend,n,res", [("dense", 12, 32), ("binned", 48, 96)])
def test_captured_step_replays_on_new_parameters(hip_lib, backend, n, res):
    from diff_gaussian_sampling import GaussianSampler
    dev = torch.device("cuda")
    gs = synthetic.lattice_gaussians(n, n, 0.8, seed=2)
    means = gs["means"].float().to(dev).requires_grad_(True)
    values = gs["values"].float().to(dev).requires_grad_(True)
    conics = gs["conics"].float().to(dev).requires_grad_(True)
    samples = synthetic.grid_samples(res).float().to(dev)
    M = samples.shape[0]
    gen = torch.Generator().manual_seed(5)
    rs = [torch.rand(s, generator=gen).to(dev) * 2 - 1 for s in ((M, 1), (M, 2, 1), (M, 2, 2, 1))]
    sampler = GaussianSampler(False, backend=backend, fuse="all")

    def step():
        sampler.preprocess(means, values, None, conics, samples)
        outs = sampler.sample((0, 1, 2))
        loss = sum((o * r).sum() for o, r in zip(outs, rs))
        grads = torch.autograd.grad(loss, (means, values, conics))
        return outs, grads

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # warm up on the capture stream: library load, allocator, autograd
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        outs, grads = step()
    assert (sampler._plan is not None) == (backend == "binned")

    rng = np.random.default_rng(0)
    for trial in range(12):
        with torch.no_grad():              # new parameter values in the captured buffers
            means.add_(torch.as_tensor(rng.normal(0, 0.01, means.shape), dtype=torch.float32, device=dev))
            values.copy_(torch.as_tensor(rng.uniform(-1, 1, values.shape), dtype=torch.float32, device=dev))
            conics.mul_(1.0 + 0.02 * trial)
        graph.replay()
        torch.cuda.synchronize()
        args = [x.detach().cpu().double().numpy() for x in (means, conics, values, samples)]
        exp = c_oracle.forward(*args, orders=(0, 1, 2))
        for o in (0, 1, 2):
            assert rel(outs[o], exp[o]) < TOL, (trial, "order", o, rel(outs[o], exp[o]))
        em, ec, ev = c_oracle.backward(*args, {o: r.cpu().double().numpy() for o, r in enumerate(rs)})
        assert rel(grads[0], em) < TOL, (trial, "means", rel(grads[0], em))
        assert rel(grads[1], ev) < TOL, (trial, "values", rel(grads[1], ev))
        assert rel(grads[2], ec) < TOL, (trial, "conics", rel(grads[2], ec))


def test_graphed_step_helper(hip_lib):
    """pigs_amd.graphs.GraphedStep: the capture recipe as a helper, replayed on updated parameters."""
    from diff_gaussian_sampling import GaussianSampler
    from pigs_amd.graphs import GraphedStep
    dev = torch.device("cuda")
    gs = synthetic.lattice_gaussians(10, 10, 0.9, seed=4)
    samples = synthetic.grid_samples(24).float().to(dev)
    sampler = GaussianSampler(False)

    def make_inputs():
        return tuple(gs[k].float().to(dev).requires_grad_(True) for k in ("means", "values", "conics"))

    def fn(means, values, conics):
        sampler.preprocess(means, values, None, conics, samples)
        u, ux, lap = sampler.sample((0, 1, "lap"))
        loss = ((u - 0.1 * lap) ** 2).mean() + (ux ** 2).mean()
        return (loss,) + torch.autograd.grad(loss, (means, values, conics))

    step = GraphedStep(fn, make_inputs)
    rng = np.random.default_rng(3)
    for trial in range(4):
        with torch.no_grad():
            step.inputs[1].copy_(torch.as_tensor(rng.uniform(-1, 1, step.inputs[1].shape), dtype=torch.float32, device=dev))
        loss, gm, gv, gc = step()
        torch.cuda.synchronize()
        means, values, conics = (x.detach().clone().requires_grad_(True) for x in step.inputs)
        eager = fn(means, values, conics)          # the same step issued eagerly on the new values
        for a, b in zip((loss, gm, gv, gc), eager):
            a, b = a.detach(), b.detach()
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-30, trial


@pytest.mark.parametrize("host", ["native", "ctypes"])
def test_graphed_step_with_static_samples_replays_a_warm_step(hip_lib, host):
    """GraphedStep(samplers=..., static_samples=True): the capture reuses the sorted sample structure its warm-up
    runs built (no samples build in the graph: a replay is a warm step), keeps it alive, and stays right on new
    parameter values -- also after eager traffic on the same sampler has evicted everything it remembered."""
    from diff_gaussian_sampling import GaussianSampler
    from pigs_amd.graphs import GraphedStep
    dev = torch.device("cuda")
    gs = synthetic.lattice_gaussians(32, 32, 0.7, seed=6)
    gen = torch.Generator().manual_seed(17)
    samples = (torch.rand((6000, 2), generator=gen) * 2 - 1).float().to(dev)       # points in no order: a real sort
    sampler = GaussianSampler(False, backend="binned", fuse="all", host=host)

    def make_inputs():
        return tuple(gs[k].float().to(dev).requires_grad_(True) for k in ("means", "values", "conics"))

    def fn(means, values, conics):
        sampler.preprocess(means, values, None, conics, samples)
        u, du, h = sampler.sample((0, 1, 2))
        loss = ((u[:, 0] - (h[:, 0, 0, 0] + h[:, 1, 1, 0])) ** 2).mean() + (du ** 2).mean()
        return (u, du, h) + torch.autograd.grad(loss, (means, values, conics))

    step = GraphedStep(fn, make_inputs, samplers=[sampler], static_samples=True)
    assert not sampler.static_samples                       # restored
    captured = sampler._plan
    assert captured.recorded_only and captured.samples.built          # the Gaussian half recorded, the samples half reused
    assert any(captured.samples is p for keep in step._keep for p in keep)
    other = GaussianSampler(False, backend="binned", fuse="all", host=host)
    rng = np.random.default_rng(5)
    for trial in range(3):
        with torch.no_grad():
            step.inputs[1].copy_(torch.as_tensor(rng.uniform(-1, 1, step.inputs[1].shape), dtype=torch.float32, device=dev))
        for k in range(6):        # eager traffic: evicts the remembered sample plans, recycles pooled workspaces
            sampler.preprocess(step.inputs[0].detach(), step.inputs[1].detach(), None, step.inputs[2].detach(),
                               (torch.rand((6000, 2), generator=gen) * 2 - 1).float().to(dev))
            sampler.sample_gaussians()
        outs = step()
        torch.cuda.synchronize()
        m, v, c = (x.detach().clone().requires_grad_(True) for x in step.inputs)
        other.preprocess(m, v, None, c, samples)
        u, du, h = other.sample((0, 1, 2))
        loss = ((u[:, 0] - (h[:, 0, 0, 0] + h[:, 1, 1, 0])) ** 2).mean() + (du ** 2).mean()
        exp = (u, du, h) + torch.autograd.grad(loss, (m, v, c))
        for k, (a, b) in enumerate(zip(outs, exp)):
            a, b = a.detach(), b.detach()
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-30, (trial, k)

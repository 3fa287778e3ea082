"""The two host sides of GaussianSampler -- the native C++ torch extension (pigs_amd/_pigs_host.so, the
default) and the ctypes host (pigs_amd/sampler.py) -- drive the same C ABI and must agree; and what a
hipGraph capture may and may not share with eager calls (the sample-plan cache and the workspace
pool stay out of a capture)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from pigs_amd import synthetic

pytestmark = pytest.mark.gpu
HOSTS = ("native", "ctypes")


def rel(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else b
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def case(n, res, kappa=0.8, seed=3, c=1, dev="cuda"):
    gs = synthetic.lattice_gaussians(n, n, kappa, seed=seed, c=c)
    t = {k: v.float().to(dev) for k, v in gs.items()}
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    return t, synthetic.grid_samples(res).float().to(dev)


def test_native_extension_is_the_default_and_loaded(hip_lib):
    import sys
    from diff_gaussian_sampling import GaussianSampler
    s = GaussianSampler(False)
    assert s.host == "native" and s._core is not None
    assert "pigs_amd._pigs_host" in sys.modules


@pytest.mark.parametrize("backend", ["dense", "binned"])
def test_hosts_agree_forward_and_backward(hip_lib, backend):
    from diff_gaussian_sampling import GaussianSampler
    t, pts = case(24, 64)
    res = {}
    for host in HOSTS:
        s = GaussianSampler(False, backend=backend, host=host)
        s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
        outs = (s.sample_gaussians(), s.sample_gaussians_derivative(), s.sample_gaussians_laplacian(),
                s.sample_gaussians_third_derivative())
        gen = torch.Generator().manual_seed(1)
        loss = sum((o * torch.rand(o.shape, generator=gen).to(o.device)).sum() for o in outs)
        grads = torch.autograd.grad(loss, (t["means"], t["values"], t["conics"]))
        res[host] = [o.detach() for o in outs] + list(grads)
        assert (s._plan is not None) == (backend == "binned")
    for k, (a, b) in enumerate(zip(res["native"], res["ctypes"])):
        if backend == "dense" and k < 4:
            assert torch.equal(a, b)          # same forward kernel, same launch geometry: bit-identical
        else:
            # binned: list order depends on the build's atomics; dense backward: its cross-workgroup sums are
            # atomics too -- last-bit differences between any two runs
            assert rel(a, b) < 2e-6
    args = [t[k].detach().cpu().double().numpy() for k in ("means", "conics", "values")]
    exp = c_oracle.forward(*args, pts.cpu().double().numpy(), orders=(0, 1, 2, 3))
    for o in range(4):
        assert rel(res["native"][o], exp[o]) < 1e-5


@pytest.mark.parametrize("host", HOSTS)
def test_node_survives_nonretaining_backward_and_checks_versions(hip_lib, host):
    """test_derivatives.py:214-215, 349-352: one output after the other, the last without retain_graph,
    then another output of the same launch; and an in-place update of a bound tensor is an error."""
    from diff_gaussian_sampling import GaussianSampler
    t, pts = case(8, 16)
    s = GaussianSampler(True, fuse="all", host=host)
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
    u, du = s.sample_gaussians(), s.sample_gaussians_derivative()
    g1 = torch.autograd.grad(u.sum(), t["means"])[0]                 # not retained
    g2 = torch.autograd.grad(du[:, 0].sum(), t["means"])[0]          # same node, again
    args = [t[k].detach().cpu().double().numpy() for k in ("means", "conics", "values")]
    p64 = pts.cpu().double().numpy()
    e1 = c_oracle.backward(*args, p64, {0: np.ones(tuple(u.shape))})[0]
    go = np.zeros(tuple(du.shape)); go[:, 0] = 1.0
    e2 = c_oracle.backward(*args, p64, {1: go})[0]
    assert rel(g1, e1) < 1e-5 and rel(g2, e2) < 1e-5
    with torch.no_grad():
        t["values"].mul_(2.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        torch.autograd.grad(du[:, 1].sum(), t["means"])


@pytest.mark.parametrize("host", HOSTS)
def test_no_grad_and_detached_inputs_build_no_graph(hip_lib, host):
    from diff_gaussian_sampling import GaussianSampler
    t, pts = case(8, 16)
    s = GaussianSampler(False, host=host)
    with torch.no_grad():
        s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
        assert not s.sample_gaussians().requires_grad
    s.preprocess(t["means"].detach(), t["values"].detach(), None, t["conics"].detach(), pts)
    assert s.sample_gaussians().grad_fn is None
    s.preprocess(t["means"], t["values"].detach(), None, t["conics"].detach(), pts)
    u = s.sample_gaussians()
    assert u.requires_grad
    u.sum().backward()
    assert t["means"].grad is not None and t["values"].grad is None


@pytest.mark.parametrize("host", HOSTS)
def test_graphed_binned_step_resorts_updated_samples(hip_lib, host):
    """A captured binned step whose SAMPLES are a static input: after `copy_` of new points a replay
    must sample at the new points (the capture records the samples build; it neither reuses an eagerly
    built sample plan nor leaves one behind), also after enough other preprocess calls to evict
    anything the warm-up runs remembered."""
    from diff_gaussian_sampling import GaussianSampler
    from pigs_amd.graphs import GraphedStep
    dev = torch.device("cuda")
    gs = synthetic.lattice_gaussians(32, 32, 0.7, seed=6)
    sampler = GaussianSampler(False, backend="binned", fuse="all", host=host)
    M = 4096
    gen = torch.Generator().manual_seed(11)

    def make_inputs():
        m = gs["means"].float().to(dev).requires_grad_(True)
        v = gs["values"].float().to(dev).requires_grad_(True)
        c = gs["conics"].float().to(dev).requires_grad_(True)
        pts = (torch.rand((M, 2), generator=gen) * 2 - 1).float().to(dev)
        return m, v, c, pts

    def fn(m, v, c, pts):
        sampler.preprocess(m, v, None, c, pts)
        u, du, h = sampler.sample((0, 1, 2))
        loss = ((u[:, 0] - (h[:, 0, 0, 0] + h[:, 1, 1, 0])) ** 2).mean() + (du ** 2).mean()
        return (u, du, h) + torch.autograd.grad(loss, (m, v, c))

    step = GraphedStep(fn, make_inputs)
    other = GaussianSampler(False, backend="binned", fuse="all", host=host)
    for trial in range(3):
        new_pts = (torch.rand((M, 2), generator=gen) * 2 - 1).float().to(dev)
        with torch.no_grad():
            step.inputs[3].copy_(new_pts)
        # eager traffic between replays: evicts remembered sample plans, recycles pooled workspaces
        for k in range(5):
            sampler.preprocess(step.inputs[0].detach(), step.inputs[1].detach(), None, step.inputs[2].detach(),
                               (torch.rand((M, 2), generator=gen) * 2 - 1).float().to(dev))
            sampler.sample_gaussians()
        outs = step()
        torch.cuda.synchronize()
        m, v, c = (x.detach().clone().requires_grad_(True) for x in step.inputs[:3])
        other.preprocess(m, v, None, c, new_pts)
        u, du, h = other.sample((0, 1, 2))
        loss = ((u[:, 0] - (h[:, 0, 0, 0] + h[:, 1, 1, 0])) ** 2).mean() + (du ** 2).mean()
        exp = (u, du, h) + torch.autograd.grad(loss, (m, v, c))
        for k, (a, b) in enumerate(zip(outs, exp)):
            assert rel(a, b) < 2e-6, (trial, k, rel(a, b))


@pytest.mark.parametrize("host", HOSTS)
def test_capture_leaves_no_unbuilt_sample_plan_behind(hip_lib, host):
    """A sample plan first met inside a capture was only recorded, not built: an eager preprocess on the
    same samples tensor afterwards must build its own."""
    from diff_gaussian_sampling import GaussianSampler
    dev = torch.device("cuda")
    t, pts = case(16, 48)
    sampler = GaussianSampler(False, backend="binned", host=host)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    warm = synthetic.grid_samples(40).float().to(dev)
    with torch.cuda.stream(side), torch.no_grad():
        sampler.preprocess(t["means"], t["values"], None, t["conics"], warm)      # warm-up on OTHER points
        sampler.sample_gaussians()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side), torch.no_grad():
        sampler.preprocess(t["means"], t["values"], None, t["conics"], pts)     # first time these points are seen
        sampler.sample_gaussians()
    # never replayed: whatever the capture recorded has not run
    assert all(p.source is not pts for p in sampler._sample_plans)
    with torch.no_grad():
        sampler.preprocess(t["means"], t["values"], None, t["conics"], pts)
        u = sampler.sample_gaussians()
    args = [t[k].detach().cpu().double().numpy() for k in ("means", "conics", "values")]
    exp = c_oracle.forward(*args, pts.cpu().double().numpy(), orders=(0,))
    assert rel(u, exp[0]) < 1e-5


@pytest.mark.parametrize("host", HOSTS)
def test_eager_calls_behind_a_capture_do_not_sample_a_recorded_plan(hip_lib, host):
    """a capture records preprocess + sample; an EAGER sample_*() afterwards, with no new
    preprocess, must not run on what was only recorded (a samples workspace that never executed) -- the
    order-3 call builds its second plan, the order-1 call samples the first one: both against the oracle.
    And a SamplePlan that was built eagerly stays built when a capture records another plan on top of it."""
    from diff_gaussian_sampling import GaussianSampler
    dev = torch.device("cuda")
    t, pts = case(16, 48)
    sampler = GaussianSampler(False, backend="binned", fuse="none", host=host)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    warm = synthetic.grid_samples(40).float().to(dev)
    with torch.cuda.stream(side), torch.no_grad():
        sampler.preprocess(t["means"], t["values"], None, t["conics"], warm)
        sampler.sample_gaussians()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side), torch.no_grad():
        sampler.preprocess(t["means"], t["values"], None, t["conics"], pts)
        sampler.sample_gaussians()
    assert sampler._plan.recorded_only and not sampler._plan.samples.built
    with torch.no_grad():                        # never replayed: nothing the capture recorded has run
        d3 = sampler.sample_gaussians_third_derivative()
        d1 = sampler.sample_gaussians_derivative()
    assert not sampler._plan.recorded_only and sampler._plan.samples.built
    args = [t[k].detach().cpu().double().numpy() for k in ("means", "conics", "values")]
    exp = c_oracle.forward(*args, pts.cpu().double().numpy(), orders=(1, 3))
    assert rel(d1, exp[1]) < 1e-5 and rel(d3, exp[3]) < 1e-5
    # the mirror: an eagerly built SamplePlan keeps `built` when a capture builds the order-3 plan on it
    sp = sampler._plan.samples
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        torch.cuda.current_stream().wait_stream(torch.cuda.default_stream())
    with torch.no_grad():
        sampler.preprocess(t["means"], t["values"], None, t["conics"], pts)      # eager, remembered points
        sp = sampler._plan.samples
        assert sp.built
    torch.cuda.synchronize()
    with torch.cuda.graph(graph2, stream=side), torch.no_grad():
        sampler.sample_gaussians_third_derivative()                              # records plan3 on the built samples
    assert sp.built


def test_plan_used_on_another_stream_is_not_recycled(hip_lib):
    from diff_gaussian_sampling import GaussianSampler
    t, pts = case(16, 48)
    for host in HOSTS:
        s = GaussianSampler(False, backend="binned", host=host)
        with torch.no_grad():
            s.preprocess(t["means"], t["values"], None, t["conics"], pts)
            plan = s._plan
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                s.sample_gaussians()
            torch.cuda.synchronize()
        assert plan.other_stream_used


@pytest.mark.parametrize("N", [300, 3000])
def test_hosts_agree_on_aggregate_neighbors(hip_lib, N):
    """preprocess_aggregate / aggregate_neighbors (parity unpinned: this repository's own definition) through
    both hosts: N = 300 takes the all-pairs list build, N = 3000 the grid build with its counting pass."""
    from diff_gaussian_sampling import GaussianSampler
    rng = np.random.default_rng(N)
    side = int(np.sqrt(N))
    gs = synthetic.lattice_gaussians(side, N // side, 1.1, seed=N)
    n = gs["means"].shape[0]
    means, conics, values = (gs[k].float().cuda() for k in ("means", "conics", "values"))
    L, K, F = 8, 4, 3
    E = 4 * F + 1
    gen = torch.Generator().manual_seed(2)
    mk = lambda *s: torch.randn(*s, generator=gen).cuda().requires_grad_(True)
    args = [mk(n, L), mk(L, L), mk(n, K), mk(n, K), mk(F), mk(L, 2 * E)]
    gout = torch.randn((n, L), generator=gen).cuda()
    res = {}
    for host in HOSTS:
        s = GaussianSampler(True, unpinned_aggregate=True, host=host)
        s.preprocess(means, values, None, conics, means)
        s.preprocess_aggregate()
        out = s.aggregate_neighbors(*args)
        res[host] = [out.detach()] + list(torch.autograd.grad(out, args, grad_outputs=gout))
        nb = s._neighbors
        assert int(nb.overflow.item()) == 0 and int(nb.row_counts.max()) <= nb.cap
        res[host + "_pairs"] = int(nb.row_counts.sum())
        assert int(nb.col_counts.sum()) == res[host + "_pairs"]
    assert res["native_pairs"] == res["ctypes_pairs"]
    for a, b in zip(res["native"], res["ctypes"]):
        assert rel(a, b) < 2e-5

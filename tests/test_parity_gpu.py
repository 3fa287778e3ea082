"""GPU parity: the HIP path (through the GaussianSampler surface -> C ABI) against the fixtures
produced by the reference and against the CPU oracle on the same seeded inputs.

Bar (north_star): forward values, derivatives and parameter gradients within 1e-5 relative
(max-abs error over the tensor / max-abs of the expected tensor) in float32; 1e-11 in float64.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_files, grads_within_accumulation_bound
from oracle import c_oracle

pytestmark = pytest.mark.gpu

F32_TOL = 1e-5
F64_TOL = 1e-11
D12 = [f for f in golden_files() if "d3" not in f]


def rel(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)   # float32 cannot hold less


def elementwise_ok(a, b, rtol=1e-5, floor=1e-6):
    """Every element on its own: |a - b| <= rtol |b| + floor max|b|.  The global metric above never looks at
    the small outputs; this one holds each of them to 1e-5 relative, with an absolute floor ten times
    below the old bar (float32 sums that nearly cancel cannot do better than a few 1e-7 of the largest
    output: tools/elem_err.py measured <= 7.5e-7 of the maximum, dense and binned alike)."""
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return bool((np.abs(a - b) <= rtol * np.abs(b) + floor * max(np.abs(b).max(), 1e-30)).all())


def input_rounding(exp_rounded, ref):
    """Distance between the float64 oracle on the float32-rounded inputs and the reference's float64 outputs
    on the unrounded ones: what the rounding of the INPUTS costs, whatever the kernel does."""
    return np.abs(np.asarray(exp_rounded) - ref).max() / max(np.abs(ref).max(), 1e-30)


def dev(a, dtype):
    return torch.as_tensor(np.asarray(a), dtype=dtype, device="cuda")


@pytest.fixture(scope="module")
def Sampler(hip_lib):
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from diff_gaussian_sampling import GaussianSampler
    return GaussianSampler


def load_case(name, dtype):
    z = np.load(os.path.join(GOLDEN, name))
    t = {k: dev(z[k], dtype) for k in ("means", "values", "covariances", "conics", "samples")}
    return z, t


@pytest.mark.parametrize("fuse", ["none", "all"])
@pytest.mark.parametrize("name", D12)
def test_forward_f64_matches_reference(Sampler, name, fuse):
    z, t = load_case(name, torch.float64)
    s = Sampler(True, fuse=fuse)
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], t["samples"])
    outs = (s.sample_gaussians(), s.sample_gaussians_derivative(), s.sample_gaussians_laplacian(),
            s.sample_gaussians_third_derivative())
    for o, out in enumerate(outs):
        assert tuple(out.shape) == z[f"out{o}_f64"].shape
        assert rel(out, z[f"out{o}_f64"]) < F64_TOL, (name, o)


def backends_for(name):
    """The reference-generated fixtures run through the dense path (what "auto" picks at their sizes) and,
    where the binned path takes the case (float32, d = 2, c <= 2), through it as well."""
    d2 = ("ref_test_derivatives.npz", "ref_test_gaussian_sampling.npz", "ref_test_density.npz", "ref_test_torus.npz")
    return ["auto", "binned"] if "d2" in name or name in d2 else ["auto"]


F32_CASES = [(n, b) for n in D12 for b in backends_for(n)]


@pytest.mark.parametrize("fuse", ["none", "all"])
@pytest.mark.parametrize("name,backend", F32_CASES)
def test_forward_f32_matches_reference(Sampler, name, backend, fuse):
    z, t = load_case(name, torch.float32)
    s = Sampler(True, fuse=fuse, backend=backend)
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], t["samples"])
    assert (s._plan is not None) == (backend == "binned")
    outs = (s.sample_gaussians(), s.sample_gaussians_derivative(), s.sample_gaussians_laplacian(),
            s.sample_gaussians_third_derivative())
    # expected: float64 oracle on the float32-rounded inputs (isolates the kernel's arithmetic) ...
    exp = c_oracle.forward(*(t[k].cpu().double().numpy() for k in ("means", "conics", "values", "samples")),
                           orders=(0, 1, 2, 3))
    for o, out in enumerate(outs):
        assert out.dtype == torch.float32
        assert rel(out, exp[o]) < F32_TOL, (name, o)
        assert elementwise_ok(out, exp[o]), (name, o)
        # ... and the reference's own float64 outputs on the unrounded inputs: the same bar plus exactly
        # what rounding the inputs to float32 costs (measured on the oracle, not granted as a factor)
        assert rel(out, z[f"out{o}_f64"]) < F32_TOL + input_rounding(exp[o], z[f"out{o}_f64"]), (name, o)


@pytest.mark.parametrize("name,backend,dtype,tol",
                         [(n, "auto", torch.float64, F64_TOL) for n in D12] + [(n, b, torch.float32, F32_TOL) for n, b in F32_CASES])
def test_backward_per_order_matches_reference_autograd(Sampler, name, backend, dtype, tol):
    """test_derivatives.py:123,214-215,349-352: grads of each output wrt (means, values, conics)."""
    z, t = load_case(name, dtype)
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    s = Sampler(True, fuse="none", backend=backend)
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], t["samples"])
    assert (s._plan is not None) == (backend == "binned")
    getters = (s.sample_gaussians, s.sample_gaussians_derivative, s.sample_gaussians_laplacian,
               s.sample_gaussians_third_derivative)
    for o, get in enumerate(getters):
        out = get()
        r = dev(z[f"r{o}"], dtype)
        gm, gv, gc = torch.autograd.grad((out * r).sum(), (t["means"], t["values"], t["conics"]))
        if dtype == torch.float32:
            em, ec, ev = c_oracle.backward(*(t[k].detach().cpu().double().numpy()
                                             for k in ("means", "conics", "values", "samples")),
                                           {o: r.cpu().double().numpy()})
        else:
            em, ev, ec = z[f"gmeans{o}_f64"], z[f"gvalues{o}_f64"], z[f"gconics{o}_f64"]
        assert rel(gm, em) < tol, (name, o, "means")
        assert rel(gv, ev) < tol, (name, o, "values")
        assert rel(gc, ec) < tol, (name, o, "conics")
        if dtype == torch.float32:
            assert rel(gm, z[f"gmeans{o}_f64"]) < tol + input_rounding(em, z[f"gmeans{o}_f64"])
            assert rel(gc, z[f"gconics{o}_f64"]) < tol + input_rounding(ec, z[f"gconics{o}_f64"])


@pytest.mark.parametrize("name", ["random_d2_c2.npz", "ref_test_derivatives.npz", "random_d1_c2.npz"])
def test_backward_fused_after_second_preprocess(Sampler, name):
    """model_pn.py:766-788: outputs stay differentiable after a later preprocess() rebinds the
    sampler; one loss over orders 0..2 (test_no_mlp.py:127-146) gives one fused backward."""
    z, t = load_case(name, torch.float64)
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    s = Sampler(False, fuse="all")
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], t["samples"])
    u, ux, uxx = s.sample_gaussians(), s.sample_gaussians_derivative(), s.sample_gaussians_laplacian()
    # rebind to something else entirely before backward
    s.preprocess(t["means"][:3].detach() + 1, t["values"][:3].detach(), None, t["conics"][:3].detach(),
                 t["samples"][:5])
    _ = s.sample_gaussians()
    loss = sum((o * dev(z[f"r{k}"], torch.float64)).sum() for k, o in enumerate((u, ux, uxx)))
    loss.backward()
    for key, g in (("gmeans", t["means"].grad), ("gvalues", t["values"].grad), ("gconics", t["conics"].grad)):
        exp = sum(z[f"{key}{k}_f64"] for k in range(3))
        assert rel(g, exp) < F64_TOL, (name, key)


def test_no_grad_and_sample_api(Sampler):
    z, t = load_case("random_d2_c1.npz", torch.float32)
    s = Sampler(False)
    with torch.no_grad():
        s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], t["samples"])
        u, ux, uxx, uxxx = s.sample((0, 1, 2, 3))
    assert not u.requires_grad
    exp = c_oracle.forward(*(t[k].cpu().double().numpy() for k in ("means", "conics", "values", "samples")),
                           orders=(0, 1, 2, 3))
    for o, out in enumerate((u, ux, uxx, uxxx)):
        assert rel(out, exp[o]) < F32_TOL
        assert rel(out, z[f"out{o}_f64"]) < F32_TOL + input_rounding(exp[o], z[f"out{o}_f64"])
    # derivative outputs are symmetric in their derivative indices
    assert torch.equal(uxx[:, 0, 1], uxx[:, 1, 0])
    assert torch.equal(uxxx[:, 0, 0, 1], uxxx[:, 1, 0, 0])


def test_1d_call_shapes(Sampler):
    """test_1d.py:27-30: samples given as a 1-D tensor, conics as [N,1]."""
    z, t = load_case("ref_test_1d.npz", torch.float32)
    s = Sampler(True)
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], t["samples"].reshape(-1))
    assert tuple(s.sample_gaussians().shape) == (200, 1)
    assert tuple(s.sample_gaussians_derivative().shape) == (200, 1, 1)
    assert tuple(s.sample_gaussians_laplacian().shape) == (200, 1, 1, 1)
    exp = c_oracle.forward(*(t[k].cpu().double().numpy() for k in ("means", "conics", "values", "samples")), orders=(2,))
    assert rel(s.sample_gaussians_laplacian(), exp[2]) < F32_TOL
    assert rel(s.sample_gaussians_laplacian(), z["out2_f64"]) < F32_TOL + input_rounding(exp[2], z["out2_f64"])


def test_1d_training_call_conventions(Sampler):
    """The reference's 1-D training scripts hand over covariances / conics as [N, 1, 1] (test_initialize_1d.py:54-58:
    ``exp(scaling).reshape(-1, d, d)``, ``1.0 / covariances``) or [N, 1] (test_no_mlp_1d.py:109-113), two channels
    (``values = rand((n, d))`` is [N, 1] there; [N, 2] here), and differentiate a loss on ``sample_gaussians()``
    back to the raw parameters behind tanh / exp: outputs and the gradients that reach those leaves against the
    oracle's (float64 autograd through the same chain)."""
    rng = np.random.default_rng(3)
    N, M = 80, 128
    raw_means = torch.tensor(np.arctanh(np.linspace(-0.95, 0.95, N)).reshape(N, 1), dtype=torch.float32, device="cuda", requires_grad=True)
    scaling = torch.full((N, 1), -5.0, device="cuda", requires_grad=True)
    values = torch.tensor(rng.uniform(0, 1, (N, 2)), dtype=torch.float32, device="cuda", requires_grad=True)
    samples = torch.tensor(rng.uniform(-1, 1, (M, 1)), dtype=torch.float32, device="cuda")
    r = torch.tensor(rng.uniform(-1, 1, (M, 2)), dtype=torch.float32, device="cuda")
    for shape in ((-1, 1, 1), (-1, 1)):
        for t in (raw_means, scaling, values):
            t.grad = None
        means = torch.tanh(raw_means)
        covariances = torch.exp(scaling).reshape(*shape)
        conics = 1.0 / covariances
        s = Sampler(True)
        s.preprocess(means, values, covariances, conics, samples)
        img = s.sample_gaussians()
        assert tuple(img.shape) == (M, 2)
        (img * r).sum().backward()
        # the oracle on the same float32 inputs, its gradients chained by hand through tanh / exp / reciprocal
        args = [x.detach().cpu().double().numpy() for x in (means, conics.reshape(N, 1), values, samples)]
        exp = c_oracle.forward(*args, orders=(0,))
        assert rel(img, exp[0]) < F32_TOL
        gm, gc, gv = c_oracle.backward(*args, {0: r.cpu().double().numpy()})
        m64, c64 = args[0], args[1]
        assert rel(values.grad, gv) < 1e-5
        assert rel(raw_means.grad, gm * (1.0 - m64 ** 2)) < 1e-5                 # d tanh
        assert rel(scaling.grad, gc.reshape(N, 1) * (-c64)) < 1e-5             # conic = exp(-scaling)


def test_masked_noncontiguous_inputs(Sampler):
    """model_pn.py:769: inputs arrive as boolean-mask selections / non-contiguous views."""
    z, t = load_case("random_d2_c1.npz", torch.float64)
    N = t["means"].shape[0]
    mask = torch.ones(N, dtype=torch.bool, device="cuda")
    means_nc = torch.stack((t["means"], t["means"]), dim=2)[:, :, 0]     # non-contiguous view
    assert not means_nc.is_contiguous()
    s = Sampler(True)
    s.preprocess(means_nc[mask], t["values"][mask], t["covariances"][mask], t["conics"][mask], t["samples"])
    assert rel(s.sample_gaussians(), z["out0_f64"]) < F64_TOL


def test_empty_inputs(Sampler):
    s = Sampler(True)
    e = lambda *sh: torch.zeros(*sh, device="cuda")
    s.preprocess(e(0, 2), e(0, 1), e(0, 3), e(0, 3), torch.rand(7, 2, device="cuda"))
    assert tuple(s.sample_gaussians().shape) == (7, 1) and not s.sample_gaussians().any()
    assert not s.sample_gaussians_laplacian().any()
    s.preprocess(torch.rand(5, 2, device="cuda"), e(5, 1) + 1, e(5, 3) + 1, e(5, 3) + 1, e(0, 2))
    assert tuple(s.sample_gaussians_derivative().shape) == (0, 2, 1)
    means = torch.rand(5, 2, device="cuda", requires_grad=True)
    s.preprocess(means, e(5, 1) + 1, None, e(5, 3) + 1, e(0, 2))
    s.sample_gaussians().sum().backward()
    assert not means.grad.any()


@pytest.mark.parametrize("N,M,d,c", [(1, 1, 2, 1), (63, 65, 2, 1), (257, 1000, 2, 2), (1000, 1024, 2, 1),
                                      (5, 4097, 1, 1), (300, 129, 1, 3), (129, 70, 2, 4)])
def test_ragged_sizes_against_oracle(Sampler, N, M, d, c):
    rng = np.random.default_rng(N * 7 + M)
    means = rng.uniform(-1, 1, (N, d))
    nf = d * (d + 1) // 2
    if d == 2:
        s0 = np.exp(rng.normal(-3, 0.5, (N, 2)))
        tau = np.tanh(rng.normal(0, 0.7, N)) * np.sqrt(s0[:, 0] * s0[:, 1])
        det = s0[:, 0] * s0[:, 1] - tau ** 2
        con = np.stack((s0[:, 1] / det, -tau / det, s0[:, 0] / det), -1)
    else:
        con = 1.0 / np.exp(rng.normal(-4, 0.5, (N, 1)))
    values = rng.uniform(-1, 1, (N, c))
    samples = rng.uniform(-1.1, 1.1, (M, d))
    t = [dev(a, torch.float32) for a in (means, values, con, samples)]
    for x in t[:3]:
        x.requires_grad_(True)
    s = Sampler(True, fuse="all")
    s.preprocess(t[0], t[1], None, t[2], t[3])
    outs = s.sample((0, 1, 2, 3))
    args = [x.detach().cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]
    exp = c_oracle.forward(*args, orders=(0, 1, 2, 3))
    rs = {}
    loss = 0
    for o, out in enumerate(outs):
        assert rel(out, exp[o]) < F32_TOL, (o,)
        rs[o] = rng.uniform(-1, 1, exp[o].shape)
        loss = loss + (out * dev(rs[o], torch.float32)).sum()
    loss.backward()
    em, ec, ev = c_oracle.backward(*args, {o: dev(r, torch.float32).cpu().double().numpy() for o, r in rs.items()})
    assert rel(t[0].grad, em) < F32_TOL
    assert rel(t[1].grad, ev) < F32_TOL
    assert rel(t[2].grad, ec) < F32_TOL


def test_config1_1d_256x4096(Sampler):
    """BASELINE.json configs[0]: 1-D, 256 Gaussians x 4096 points."""
    from pigs_amd import synthetic
    gs, pts = synthetic.CONFIGS["c1"]()
    t = {k: v.float().cuda() for k, v in gs.items()}
    s = Sampler(True)
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts.float().cuda())
    outs = s.sample((0, 1, 2))
    exp = c_oracle.forward(t["means"].cpu().double().numpy(), t["conics"].cpu().double().numpy(),
                           t["values"].cpu().double().numpy(), pts.float().double().numpy(), orders=(0, 1, 2))
    for o, out in enumerate(outs):
        assert rel(out, exp[o]) < F32_TOL


@pytest.mark.parametrize("backend", ["dense", "binned"])
def test_reference_autograd_call_patterns(Sampler, backend):
    """The calls the reference's own scripts make on the sampler's outputs (test_derivatives.py:123,
    214-215, 349-352): ``autograd.grad`` with ``retain_graph=True`` and ``create_graph=True`` on the
    order-0 output, then one component of the derivative / Hessian outputs at a time on the same graph
    (the node runs its backward again and again), the last call without retaining."""
    rng = np.random.default_rng(77)
    N, M = 60, 900
    means = rng.uniform(-1, 1, (N, 2))
    s = np.exp(rng.normal(-3.0, 0.3, (N, 2)))
    con = np.stack((1 / s[:, 0], np.zeros(N), 1 / s[:, 1]), -1)
    values = rng.uniform(0, 1, (N, 1))
    pts = rng.uniform(-1, 1, (M, 2))
    t = [dev(a, torch.float32) for a in (means, values, con, pts)]
    for x in t[:3]:
        x.requires_grad_(True)
    smp = Sampler(True, backend=backend)
    smp.preprocess(t[0], t[1], None, t[2], t[3])
    args = [x.detach().cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]

    def expect(order, comp):
        shape = {0: (M, 1), 1: (M, 2, 1), 2: (M, 2, 2, 1)}[order]
        r = np.zeros(shape)
        r[(slice(None),) + comp] = 1.0
        return {order: r}

    def check(got, grads, what):
        # per entry: a few ulp of the sum of the absolute contributions (conftest.py); got = (means, values, conics)
        bad = grads_within_accumulation_bound((got[0], got[2], got[1]), args, grads)
        assert not bad, (what, bad)

    u = smp.sample_gaussians()
    check(torch.autograd.grad(u.sum(), t[:3], retain_graph=True, create_graph=True), expect(0, ()), "order 0")
    ux = smp.sample_gaussians_derivative().squeeze()
    check(torch.autograd.grad(ux[..., 0].sum(), t[:3], retain_graph=True), expect(1, (0,)), "d/dx")
    check(torch.autograd.grad(ux[..., 1].sum(), t[:3]), expect(1, (1,)), "d/dy")
    h = smp.sample_gaussians_laplacian().squeeze()
    check(torch.autograd.grad(h[..., 0, 0].sum(), t[:3], retain_graph=True), expect(2, (0, 0)), "xx")
    check(torch.autograd.grad(h[..., 0, 1].sum(), t[:3], retain_graph=True), expect(2, (0, 1)), "xy")
    check(torch.autograd.grad(h[..., 1, 0].sum(), t[:3], retain_graph=True), expect(2, (1, 0)), "yx")
    check(torch.autograd.grad(h[..., 1, 1].sum(), t[:3]), expect(2, (1, 1)), "yy")

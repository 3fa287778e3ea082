"""GPU parity of the binned (culled) path: preprocess plan + forward + backward against the dense
CPU oracle.  The cut-off (q_max = 36) drops terms below e^-18 of a term's scale, so the bar stays
1e-5 relative (north_star)."""
import numpy as np
import pytest
import torch

from conftest import grads_within_accumulation_bound
from oracle import c_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)   # float32 cannot hold less


def elementwise_ok(a, b, rtol=1e-5, floor=1e-6):
    """Every element on its own: |a - b| <= rtol |b| + floor max|b| (tests/test_parity_gpu.py)."""
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return bool((np.abs(a - b) <= rtol * np.abs(b) + floor * max(np.abs(b).max(), 1e-30)).all())


def dev32(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32, device="cuda")


@pytest.fixture(scope="module")
def Sampler(hip_lib):
    assert torch.cuda.is_available()
    from diff_gaussian_sampling import GaussianSampler
    return GaussianSampler


def random_gaussians(rng, N, c, log_sigma_mean=-3.5, log_sigma_std=0.6, lo=-1.0, hi=1.0):
    means = rng.uniform(lo, hi, (N, 2))
    s0 = np.exp(2 * rng.normal(log_sigma_mean, log_sigma_std, (N, 2)))      # variances
    tau = np.tanh(rng.normal(0, 0.7, N)) * np.sqrt(s0[:, 0] * s0[:, 1])
    det = s0[:, 0] * s0[:, 1] - tau ** 2
    con = np.stack((s0[:, 1] / det, -tau / det, s0[:, 0] / det), -1)
    values = rng.uniform(-1, 1, (N, c))
    return means, con, values


def check_case(Sampler, means, con, values, samples, orders=(0, 1, 2, 3), bwd=True, tol=TOL, gtol=None):
    gtol = tol if gtol is None else gtol
    t = [dev32(a) for a in (means, values, con, samples)]
    for x in t[:3]:
        x.requires_grad_(True)
    s = Sampler(True, backend="binned", fuse="none")
    s.preprocess(t[0], t[1], None, t[2], t[3])
    assert s._plan is not None
    outs = s.sample(orders)
    args = [x.detach().cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]
    exp = c_oracle.forward(*args, orders=orders)
    rng = np.random.default_rng(1)
    loss, rs = 0, {}
    for o, out in zip(orders, outs):
        assert out.shape == exp[o].shape
        assert rel(out, exp[o]) < tol, ("order", o, rel(out, exp[o]))
        rs[o] = dev32(rng.uniform(-1, 1, exp[o].shape))
        loss = loss + (out * rs[o]).sum()
    if not bwd:
        return s
    loss.backward()
    r64 = {o: r.cpu().double().numpy() for o, r in rs.items()}
    if gtol == "bound":     # per entry: a few ulp of the sum of the absolute contributions (conftest.py)
        bad = grads_within_accumulation_bound((t[0].grad, t[2].grad, t[1].grad), args, r64)
        assert not bad, bad
        return s
    em, ec, ev = c_oracle.backward(*args, r64)
    assert rel(t[0].grad, em) < gtol, ("means", rel(t[0].grad, em))
    assert rel(t[1].grad, ev) < gtol, ("values", rel(t[1].grad, ev))
    assert rel(t[2].grad, ec) < gtol, ("conics", rel(t[2].grad, ec))
    return s


@pytest.mark.parametrize("N,M,c", [(1, 1, 1), (7, 3, 2), (500, 2000, 1), (3000, 5000, 2), (2048, 4096, 2)])
def test_random_points_and_gaussians(Sampler, N, M, c):
    rng = np.random.default_rng(N + M)
    means, con, values = random_gaussians(rng, N, c)
    samples = rng.uniform(-1.2, 1.2, (M, 2))
    check_case(Sampler, means, con, values, samples)


def test_mixed_scales_fill_every_level(Sampler):
    """Gaussians from far below the finest cell to larger than the whole domain."""
    rng = np.random.default_rng(3)
    N = 1500
    means, con, values = random_gaussians(rng, N, 1, log_sigma_mean=-3.0, log_sigma_std=1.8)
    samples = rng.uniform(-1.5, 1.5, (3000, 2))
    check_case(Sampler, means, con, values, samples)


def test_regular_grid_samples(Sampler):
    """test_gaussian_sampling.py:42-46 shape: meshgrid(indexing='xy') grid, x fastest."""
    from pigs_amd import synthetic
    gs = synthetic.lattice_gaussians(32, 32, kappa=0.7, seed=2)
    pts = synthetic.grid_samples(96)
    check_case(Sampler, gs["means"].numpy(), gs["conics"].numpy(), gs["values"].numpy(), pts.numpy())


def test_clustered_points_multiple_passes(Sampler):
    """Hundreds of points in one tile neighbourhood + a few outliers: 700 near-identical points with
    random-sign weights, so many gradient entries are small differences of large sums.  Forward: the
    usual bar.  Gradients: for every entry the float32 ACCUMULATION bound -- the error of every gradient entry against the float64 oracle is at most
    4 ulp (2.4e-7) of the sum of the ABSOLUTE per-sample contributions to that entry, computed by
    the oracle sample by sample (plus 1e-7 of the largest entry for what the cut-off drops): what
    is left over the 1e-5 bar is summation order, not arithmetic."""
    rng = np.random.default_rng(4)
    means, con, values = random_gaussians(rng, 400, 1)
    cluster = rng.normal(0.2, 1e-3, (700, 2))
    samples = np.concatenate((cluster, rng.uniform(-1, 1, (50, 2)), np.array([[5.0, -7.0]])))
    orders = (0, 1, 2, 3)
    t = [dev32(a) for a in (means, values, con, samples)]
    for x in t[:3]:
        x.requires_grad_(True)
    s = Sampler(True, backend="binned", fuse="none")
    s.preprocess(t[0], t[1], None, t[2], t[3])
    outs = s.sample(orders)
    args = [x.detach().cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]
    exp = c_oracle.forward(*args, orders=orders)
    rs = {}
    loss = 0
    for o, out in zip(orders, outs):
        assert rel(out, exp[o]) < TOL, ("order", o, rel(out, exp[o]))
        rs[o] = dev32(rng.uniform(-1, 1, exp[o].shape))
        loss = loss + (out * rs[o]).sum()
    loss.backward()
    r64 = {o: r.cpu().double().numpy() for o, r in rs.items()}
    want = c_oracle.backward(*args, r64)                                  # (means, conics, values)
    mag = [np.zeros_like(w) for w in want]
    for m in range(samples.shape[0]):                                      # sum of |per-sample contribution|
        part = c_oracle.backward(args[0], args[1], args[2], args[3][m:m + 1], {o: r[m:m + 1] for o, r in r64.items()})
        for a, b in zip(mag, part):
            a += np.abs(b)
    for name, got, w, a in (("means", t[0].grad, want[0], mag[0]), ("conics", t[2].grad, want[1], mag[1]),
                            ("values", t[1].grad, want[2], mag[2])):
        err = np.abs(got.cpu().double().numpy() - w)
        # + 1e-6 of the largest entry: what the cut-off drops (700 points times a term of e^-22 with its q^2.5
        # prefactor), a tenth of the bar
        bound = 1e-6 * a + 1e-6 * np.abs(w).max()
        assert (err <= bound).all(), (name, float((err / bound).max()))


def test_degenerate_geometry(Sampler):
    rng = np.random.default_rng(5)
    means, con, values = random_gaussians(rng, 300, 1)
    # all points identical
    check_case(Sampler, means, con, values, np.tile([[0.1, 0.2]], (130, 1)))
    # all points on a vertical line
    line = np.stack((np.full(500, 0.3), np.linspace(-1, 1, 500)), -1)
    check_case(Sampler, means, con, values, line)
    # all Gaussians at one centre
    means0 = np.tile([[0.0, 0.0]], (300, 1))
    check_case(Sampler, means0, con, values, rng.uniform(-0.5, 0.5, (400, 2)))
    # samples far outside the Gaussians' domain (everything culled -> exact zeros expected)
    far = rng.uniform(50, 51, (100, 2))
    t = [dev32(a) for a in (means, values, con, far)]
    s = Sampler(True, backend="binned")
    s.preprocess(t[0], t[1], None, t[2], t[3])
    assert not s.sample_gaussians().any() and not s.sample_gaussians_laplacian().any()


def test_truncation_at_cutoff_border(Sampler):
    """Points placed exactly around the q = q_max contour of isolated Gaussians: the dropped
    term is ~e^-18 of the peak, far below the bar, for every order."""
    rng = np.random.default_rng(6)
    N = 64
    means = np.stack(np.meshgrid(np.linspace(-1, 1, 8), np.linspace(-1, 1, 8)), -1).reshape(N, 2)
    sig = 0.01
    con = np.tile([1 / sig ** 2, 0.0, 1 / sig ** 2], (N, 1))
    values = np.ones((N, 1))
    ang = rng.uniform(0, 2 * np.pi, (N, 40))
    rad = sig * np.sqrt(36.0) * rng.uniform(0.9, 1.1, (N, 40))
    ring = means[:, None, :] + np.stack((np.cos(ang), np.sin(ang)), -1) * rad[..., None]
    samples = np.concatenate((ring.reshape(-1, 2), means))    # the centres set the output scale
    check_case(Sampler, means, con, values, samples)


def test_single_orders_and_fused_agree(Sampler):
    rng = np.random.default_rng(7)
    means, con, values = random_gaussians(rng, 800, 2)
    samples = rng.uniform(-1, 1, (1500, 2))
    t = [dev32(a) for a in (means, values, con, samples)]
    a = Sampler(False, backend="binned", fuse="none")
    a.preprocess(t[0], t[1], None, t[2], t[3])
    singles = (a.sample_gaussians(), a.sample_gaussians_derivative(), a.sample_gaussians_laplacian(),
               a.sample_gaussians_third_derivative())
    b = Sampler(False, backend="binned")
    b.preprocess(t[0], t[1], None, t[2], t[3])
    fused = b.sample((0, 1, 2, 3))
    for x, y in zip(singles, fused):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-6 * float(x.abs().max()))


def test_binned_equals_dense_hip(Sampler):
    rng = np.random.default_rng(8)
    means, con, values = random_gaussians(rng, 1200, 1)
    samples = rng.uniform(-1, 1, (2500, 2))
    t = [dev32(a) for a in (means, values, con, samples)]
    outs = {}
    for backend in ("dense", "binned"):
        s = Sampler(False, backend=backend)
        s.preprocess(t[0], t[1], None, t[2], t[3])
        outs[backend] = s.sample((0, 1, 2))
    for x, y in zip(outs["dense"], outs["binned"]):
        assert float((x - y).abs().max() / x.abs().max()) < 2e-6


@pytest.mark.parametrize("kappa", [0.5, 1.3])
def test_config2_binned_8k_x_256sq(Sampler, kappa):
    from pigs_amd import synthetic
    gs, pts = synthetic.CONFIGS["c2"](kappa)
    t = {k: v.float().cuda() for k, v in gs.items()}
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    pts = pts.float().cuda()
    s = Sampler(False, fuse="all")
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
    assert s._plan is not None          # auto picks the binned path at this size
    u, ux, uxx = s.sample((0, 1, 2))
    idx = torch.arange(0, pts.shape[0], 16, device="cuda")
    args = [t[k].detach().cpu().double().numpy() for k in ("means", "conics", "values")]
    exp = c_oracle.forward(*args, pts[idx].cpu().double().numpy(), orders=(0, 1, 2))
    for o, out in enumerate((u, ux, uxx)):
        assert rel(out[idx], exp[o]) < TOL, o
        assert elementwise_ok(out[idx], exp[o]), o
    g = torch.Generator(device="cpu").manual_seed(5)
    rs = [torch.rand(e.shape, generator=g, dtype=torch.float64) * 2 - 1 for e in (exp[0], exp[1], exp[2])]
    loss = sum((out[idx] * r.float().cuda()).sum() for out, r in zip((u, ux, uxx), rs))
    loss.backward()
    em, ec, ev = c_oracle.backward(*args, pts[idx].cpu().double().numpy(),
                                   {o: r.float().double().numpy() for o, r in enumerate(rs)})
    assert rel(t["means"].grad, em) < TOL
    assert rel(t["values"].grad, ev) < TOL
    assert rel(t["conics"].grad, ec) < TOL


def test_config2_size_order3_two_channels(Sampler):
    """The Navier-Stokes call pattern (c = 2, orders 0..3: model_pn.py:650-656, 770-781) at the size of
    BASELINE.json configs[1] on the binned path: forward of every order against the oracle on sampled
    points, backward of a loss supported on them."""
    from pigs_amd import synthetic
    gs = synthetic.lattice_gaussians(128, 64, 0.7, seed=3, c=2)
    pts = synthetic.grid_samples(256).float().cuda()
    t = {k: v.float().cuda() for k, v in gs.items()}
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    s = Sampler(False, backend="binned")
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
    outs = (s.sample_gaussians(), s.sample_gaussians_derivative(), s.sample_gaussians_laplacian(),
            s.sample_gaussians_third_derivative())
    assert s._plan is not None and s._plan3 is not None and tuple(outs[3].shape) == (65536, 2, 2, 2, 2)
    idx = torch.arange(5, pts.shape[0], 16, device="cuda")
    args = [t[k].detach().cpu().double().numpy() for k in ("means", "conics", "values")]
    p64 = pts[idx].cpu().double().numpy()
    exp = c_oracle.forward(*args, p64, orders=(0, 1, 2, 3))
    g = torch.Generator(device="cpu").manual_seed(6)
    rs = [torch.rand(exp[o].shape, generator=g) * 2 - 1 for o in range(4)]
    for o, out in enumerate(outs):
        assert rel(out[idx], exp[o]) < TOL, o
        assert elementwise_ok(out[idx], exp[o]), o
    loss = sum((out[idx] * r.cuda()).sum() for out, r in zip(outs, rs))
    loss.backward()
    em, ec, ev = c_oracle.backward(*args, p64, {o: r.double().numpy() for o, r in enumerate(rs)})
    assert rel(t["means"].grad, em) < TOL and rel(t["values"].grad, ev) < TOL and rel(t["conics"].grad, ec) < TOL


@pytest.mark.parametrize("kappa", [0.5, 1.3])
def test_config3_65k_x_1024sq_forward(Sampler, kappa):
    """BASELINE.json configs[2] at full size: oracle on 2048 sampled points; plus the
    size-independent property that every output is invariant under a permutation of the
    sample points (the sort inside preprocess must not leak into the results)."""
    from pigs_amd import synthetic
    gs, pts = synthetic.CONFIGS["c3"](kappa)
    t = {k: v.float().cuda() for k, v in gs.items()}
    pts = pts.float().cuda()
    s = Sampler(False, fuse="all")
    with torch.no_grad():
        s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
        u, ux, uxx = s.sample((0, 1, 2))
    idx = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(3))[:2048].cuda()
    args = [t[k].cpu().double().numpy() for k in ("means", "conics", "values")]
    exp = c_oracle.forward(*args, pts[idx].cpu().double().numpy(), orders=(0, 1, 2))
    for o, out in enumerate((u, ux, uxx)):
        assert rel(out[idx], exp[o]) < TOL, o
        assert elementwise_ok(out[idx], exp[o]), o
        assert torch.isfinite(out).all()
    # permutation invariance on a 64k-point subset
    sub = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(4))[:65536].cuda()
    with torch.no_grad():
        s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts[sub].contiguous())
        v = s.sample_gaussians_laplacian()
    assert float((v - uxx[sub]).abs().max() / uxx.abs().max()) < 2e-6


@pytest.mark.parametrize("kappa", [0.5, 1.3])
def test_config3_65k_x_1024sq_backward(Sampler, kappa):
    """BASELINE.json configs[2] at full size, backward, sparse and reference-like widths: (i) a loss supported on 4096 sampled points
    against the oracle's backward on those points; (ii) with gradients on ALL 1M points, the binned
    backward against the dense HIP backward (no cut-off, no sort), and linearity in the incoming
    gradients -- size-independent properties where the oracle would take hours."""
    from pigs_amd import synthetic
    gs, pts = synthetic.CONFIGS["c3"](kappa)
    t = {k: v.float().cuda() for k, v in gs.items()}
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    pts = pts.float().cuda()
    params = (t["means"], t["values"], t["conics"])
    s = Sampler(False, fuse="all", backend="binned")
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
    outs = s.sample((0, 1, 2))
    # (i) oracle on a subsample-supported loss
    idx = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(8))[:4096].cuda()
    g = torch.Generator(device="cpu").manual_seed(9)
    rs = [torch.rand((4096,) + tuple(o.shape[1:]), generator=g) * 2 - 1 for o in outs]
    loss = sum((o[idx] * r.cuda()).sum() for o, r in zip(outs, rs))
    gm, gv, gc = torch.autograd.grad(loss, params, retain_graph=True)
    args = [t[k].detach().cpu().double().numpy() for k in ("means", "conics", "values")]
    em, ec, ev = c_oracle.backward(*args, pts[idx].cpu().double().numpy(),
                                   {o: r.double().numpy() for o, r in enumerate(rs)})
    assert rel(gm, em) < TOL and rel(gv, ev) < TOL and rel(gc, ec) < TOL
    # (ii) whole grid: binned == dense HIP, and linearity
    full = [torch.rand(o.shape, generator=g).cuda() * 2 - 1 for o in outs]
    half = [torch.rand(o.shape, generator=g).cuda() * 2 - 1 for o in outs]
    b1 = torch.autograd.grad(outs, params, grad_outputs=full, retain_graph=True)
    b2 = torch.autograd.grad(outs, params, grad_outputs=half, retain_graph=True)
    b12 = torch.autograd.grad(outs, params, grad_outputs=[a + b for a, b in zip(full, half)])
    sd = Sampler(False, fuse="all", backend="dense")
    sd.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
    d1 = torch.autograd.grad(sd.sample((0, 1, 2)), params, grad_outputs=full)
    for a, b, ab, d in zip(b1, b2, b12, d1):
        scale = float(d.abs().max())
        assert torch.isfinite(a).all()
        assert float((a - d).abs().max()) / scale < TOL          # fp32 sums over 1M points, different orders
        assert float((a + b - ab).abs().max()) / scale < TOL


def test_config4_shard_of_4096sq_grid(Sampler):
    """BASELINE.json configs[3]: one GPU's share of 65k Gaussians x 4096^2 points sharded over 8 GPUs
    (512 rows x 4096 = 2M points).  Forward against the oracle on sampled points; and the property
    the multi-GPU backward rests on: outputs of sub-shards concatenate, and their
    parameter gradients ADD up to the shard's -- what the all-reduce computes across GPUs."""
    from pigs_amd import synthetic
    gs = synthetic.lattice_gaussians(256, 256, 0.5, seed=0)
    t = {k: v.float().cuda() for k, v in gs.items()}
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    params = (t["means"], t["values"], t["conics"])
    rank, world, res = 3, 8, 4096
    rows = res // world
    pts = synthetic.grid_samples(res, res, row0=rank * rows, rows=rows).float().cuda()
    assert pts.shape[0] == 2 * 1024 * 1024
    s = Sampler(False, fuse="all")
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
    assert s._plan is not None
    outs = s.sample((0, 1, 2))
    idx = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(1))[:2048].cuda()
    args = [t[k].detach().cpu().double().numpy() for k in ("means", "conics", "values")]
    exp = c_oracle.forward(*args, pts[idx].cpu().double().numpy(), orders=(0, 1, 2))
    for o, out in enumerate(outs):
        assert rel(out[idx], exp[o]) < TOL, o
    g = torch.Generator(device="cpu").manual_seed(2)
    gouts = [torch.rand(o.shape, generator=g).cuda() * 2 - 1 for o in outs]
    whole = torch.autograd.grad(outs, params, grad_outputs=gouts)
    parts, half = [], pts.shape[0] // 2
    for lo in (0, half):
        sub = Sampler(False, fuse="all")
        sub.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts[lo:lo + half])
        o = sub.sample((0, 1, 2))
        for a, b in zip(o, outs):
            assert float((a - b[lo:lo + half]).detach().abs().max()) <= 2e-6 * float(b.detach().abs().max())
        parts.append(torch.autograd.grad(o, params, grad_outputs=[x[lo:lo + half] for x in gouts]))
    for w, a, b in zip(whole, parts[0], parts[1]):
        assert float((a + b - w).abs().max()) / float(w.abs().max()) < TOL


@pytest.mark.parametrize("c,orders", [(1, (0, 1, 2)), (2, (0, 1, 2, 3)), (1, (0, 1, "lap"))])
def test_crowded_cell_many_flushes(Sampler, c, orders):
    """Hundreds of points crowded into a few tiles AND hundreds of Gaussians reaching each of them: the
    group lists run to several hundred entries, so the forward walks them in many 32-record chunks per
    row and the backward in many 64-entry steps of the tile list, with the other rows padded by the
    all-zero record; points and Gaussians in shuffled order."""
    rng = np.random.default_rng(11)
    N = 700
    means, con, values = random_gaussians(rng, N, c, log_sigma_mean=-1.6, log_sigma_std=0.2, lo=-0.3, hi=0.3)
    crowd = rng.uniform(-0.004, 0.004, (450, 2)) + np.array([0.05, -0.02])
    samples = np.concatenate((crowd, rng.uniform(-1, 1, (2500, 2))))
    rng.shuffle(samples)
    if "lap" not in orders:
        check_case(Sampler, means, con, values, samples, orders=orders)
        return
    t = [dev32(a) for a in (means, values, con, samples)]
    for x in t[:3]:
        x.requires_grad_(True)
    s = Sampler(True, backend="binned", fuse="none")
    s.preprocess(t[0], t[1], None, t[2], t[3])
    u, ux, lap = s.sample(orders)
    args = [x.detach().cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]
    exp = c_oracle.forward(*args, orders=(0, 1, 2))
    elap = exp[2][:, 0, 0, :] + exp[2][:, 1, 1, :]
    assert rel(u, exp[0]) < TOL and rel(ux, exp[1]) < TOL and rel(lap, elap) < TOL
    r = rng.uniform(-1, 1, elap.shape)
    (lap * dev32(r)).sum().backward()
    gH = np.zeros_like(exp[2])
    gH[:, 0, 0, :] = r
    gH[:, 1, 1, :] = r
    em, ec, ev = c_oracle.backward(*args, {2: gH})
    assert rel(t[0].grad, em) < TOL and rel(t[1].grad, ev) < TOL and rel(t[2].grad, ec) < TOL


def test_sample_plans_are_remembered_across_alternating_point_sets(Sampler):
    """The reference's training step alternates preprocess(collocation points) and
    preprocess(boundary points) with new Gaussians every step (model_pn.py:766-785): the sorted
    sample structure of EACH recent tensor is reused, an in-place write invalidates it, and the
    outputs are those of a sampler that rebuilds everything."""
    rng = np.random.default_rng(11)
    pts_a = dev32(rng.uniform(-1, 1, (3000, 2)))
    pts_b = dev32(rng.uniform(-1.5, 1.5, (1000, 2)))
    s = Sampler(False, backend="binned")
    fresh = Sampler(False, backend="binned", reuse_samples=False)
    seen = {}
    for step in range(3):
        means, con, values = random_gaussians(rng, 400, 1)
        g = [dev32(x) for x in (means, values, con)]
        for name, pts in (("a", pts_a), ("b", pts_b)):
            s.preprocess(g[0], g[1], None, g[2], pts)
            sp = s._plan.samples
            if name in seen:
                assert sp is seen[name], "the samples half was rebuilt for a tensor seen one call earlier"
            seen[name] = sp
            fresh.preprocess(g[0], g[1], None, g[2], pts)
            assert fresh._plan.samples is not sp
            for o, f in zip(s.sample((0, 1, 2)), fresh.sample((0, 1, 2))):
                # the same sorted order of points and Gaussians up to the order atomics arrived in: last-bit differences
                assert rel(o, f.cpu().double().numpy()) < 1e-6
    assert fresh._sample_plans == []
    pts_a.mul_(1.0)                                   # an in-place write: the version counter moves
    means, con, values = random_gaussians(rng, 400, 1)
    s.preprocess(dev32(means), dev32(values), None, dev32(con), pts_a)
    assert s._plan.samples is not seen["a"]
    s.preprocess(dev32(means), dev32(values), None, dev32(con), pts_b)
    assert s._plan.samples is seen["b"]
    # the memory is bounded: at most ``reuse_samples`` structures are kept
    s2 = Sampler(False, backend="binned", reuse_samples=2)
    for k in range(5):
        s2.preprocess(dev32(means), dev32(values), None, dev32(con), dev32(rng.uniform(-1, 1, (500 + k, 2))))
    assert len(s2._sample_plans) == 2


def test_plan_workspaces_are_recycled_only_after_their_plan_died(Sampler):
    """A build into a workspace whose previous plan is gone skips the zeroing launch
    (PIGS_BUILD_PLAN_WS_CLEAN: every build leaves its counters zeroed).  Results must not depend on
    it, a plan that is still referenced must keep its workspace, and plans of other sizes must not
    be mixed up."""
    import gc
    rng = np.random.default_rng(23)
    pts = dev32(rng.uniform(-1, 1, (4000, 2)))
    s = Sampler(False, backend="binned")
    dense = Sampler(False, backend="dense")
    kept = []
    ptrs = set()
    for step in range(8):
        N = 300 if step % 3 else 500                      # two sizes alternate: two pool keys
        means, con, values = random_gaussians(rng, N, 1)
        g = [dev32(x) for x in (means, values, con)]
        s.preprocess(g[0], g[1], None, g[2], pts)
        plan = s._plan
        for k in kept:                                      # a live plan's workspace is never handed out again
            assert plan.workspace.data_ptr() != k.workspace.data_ptr()
        if step == 2:
            kept.append(plan)                               # as an autograd node would
        ptrs.add(plan.workspace.data_ptr())
        dense.preprocess(g[0], g[1], None, g[2], pts)
        for o, e in zip(s.sample((0, 1, 2)), dense.sample((0, 1, 2))):
            assert rel(o, e.cpu().double().numpy()) < TOL
        del plan
        gc.collect()
    assert len(ptrs) < 8, "no workspace was ever reused"
    # the kept plan still answers for ITS Gaussians after all the rebuilding around it
    assert kept[0].N == 300


@pytest.mark.parametrize("N,mode", [(8000, "groups"), (20000, "points")])
def test_sparse_scattered_points_keep_their_group_lists(Sampler, N, mode):
    """Points far apart from each other (the thin outskirts of a clustered cloud) share no Gaussians:
    a tile of 64 of them meets more than the tile list holds while its four group lists still fit.
    Such tiles keep their group lists (forward) and the backward walks those.  With more Gaussians per
    point even a group list overflows: those tiles keep no lists at all and every lane walks the Gaussian
    grid around its own point at sampling time (TILE_MODE_POINTS; record ranges remain for tiles of close
    points under very wide Gaussians: test_wide_gaussians_*).  Results as ever in both cases."""
    rng = np.random.default_rng(31)
    means, con, values = random_gaussians(rng, N, 1, log_sigma_mean=-4.6, log_sigma_std=0.2)
    core = rng.normal(0, 0.02, (60000, 2))                       # a dense core sets the cell size ...
    far = rng.uniform(-1, 1, (1500, 2))                          # ... and the rest is scattered thinly
    samples = np.clip(np.concatenate((core, far)), -1, 1)
    s = check_case(Sampler, means, con, values, samples, orders=(0, 1, 2), tol=TOL, gtol="bound")
    from tools.prof_step import list_stats
    st = list_stats(s._plan)
    # the case is what it claims to be: spread-out tiles with long or overflowing lists (the moderately sparse ones of
    # the first case keep group lists only, or walk per point when that is cheaper; the second case's cannot keep lists)
    assert (st["groups_only_tiles"] + st["points_tiles"] if mode == "groups" else st["points_tiles"]) > 0, st


def test_backward_tile_shuffle_with_a_partial_last_chunk(Sampler):
    """The backward deals the tiles of every full chunk of 1024 out in a shuffled order and leaves a
    trailing partial chunk in launch order: a size with one full chunk and a ragged rest (and a last
    tile that is not full) against the oracle."""
    rng = np.random.default_rng(41)
    M = 1024 * 64 + 70 * 64 + 37
    means, con, values = random_gaussians(rng, 1500, 1, log_sigma_mean=-3.2, log_sigma_std=0.4)
    samples = rng.uniform(-1, 1, (M, 2))
    check_case(Sampler, means, con, values, samples, orders=(0, 1, 2), gtol="bound")


def test_scan_recompute_path_gives_the_same_plan(Sampler):
    """The in-kernel scans hand totals from workgroup to workgroup; a workgroup that does not receive a
    predecessor's total within the bounded wait sums that predecessor's counters itself (ABI <= 5 gave up
    and left a wrong plan behind a flag nobody read outside debug mode).  PIGS_BUILD_DEBUG_NO_LOOKBACK
    forces that path in every workgroup of both scans (sample cells: 66 workgroups, Gaussian cells: 11):
    the diagnostic word is set, and forward + backward still meet the oracle."""
    import ctypes
    from pigs_amd import _lib, sampler as S
    lib = _lib.load()
    rng = np.random.default_rng(17)
    N, M = 4000, 1 << 20
    means, con, values = random_gaussians(rng, N, 1, log_sigma_mean=-4.0, log_sigma_std=0.3)
    pts = rng.uniform(-1, 1, (M, 2))
    t = [dev32(a) for a in (means, values, con, pts)]
    sws = torch.empty(lib.pigs_samples_workspace_bytes(M), dtype=torch.uint8, device="cuda")
    ws = torch.empty(lib.pigs_plan_workspace_bytes(N, M, 1), dtype=torch.uint8, device="cuda")
    vp = lambda x: ctypes.c_void_p(x.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for flags, expect_flag in ((1, False), (1 | 4, True)):
        rc = lib.pigs_plan_build(vp(ws), ws.numel(), vp(sws), sws.numel(), flags, N, M, 1, 36.0, 44.0,
                                 vp(t[0]), vp(t[2]), vp(t[1]), vp(t[3]), stream)
        assert rc == 0
        torch.cuda.synchronize()
        for w, off in ((sws, lib.pigs_samples_error_offset()), (ws, lib.pigs_plan_error_offset())):
            assert bool(int(w[off:off + 4].view(torch.int32).item())) == expect_flag
        u = torch.empty((M, 1), device="cuda"); du = torch.empty((M, 2, 1), device="cuda"); h = torch.empty((M, 2, 2, 1), device="cuda")
        rc = lib.pigs_plan_forward(vp(ws), ws.numel(), vp(sws), sws.numel(), N, M, 1, 36.0, 7, vp(u), vp(du), vp(h), None, stream)
        assert rc == 0
        gm, gc, gv = torch.empty_like(t[0]), torch.empty_like(t[2]), torch.empty_like(t[1])
        go = [torch.rand_like(o) for o in (u, du, h)]
        rc = lib.pigs_plan_backward(vp(ws), ws.numel(), vp(sws), sws.numel(), N, M, 1, 36.0, 7, vp(go[0]), vp(go[1]), vp(go[2]),
                                    None, vp(gm), vp(gc), vp(gv), stream)
        assert rc == 0
        torch.cuda.synchronize()
        sel = rng.choice(M, 2048, replace=False)
        args = [x.detach().cpu().double().numpy() for x in (t[0], t[2], t[1])]
        exp = c_oracle.forward(*args, pts[sel].astype(np.float32).astype(np.float64), orders=(0, 1, 2))
        for o, out in enumerate((u, du, h)):
            got = out[torch.as_tensor(sel, device="cuda")]
            assert np.abs(got.cpu().double().numpy() - exp[o]).max() / np.abs(exp[o]).max() < TOL, (flags, o)
        gsel = rng.choice(N, 256, replace=False)
        sub = (args[0][gsel], args[1][gsel], args[2][gsel], t[3].cpu().double().numpy())
        want, bound = c_oracle.accumulation_bound(*sub, {o: g.cpu().double().numpy() for o, g in enumerate(go)})
        for name, g, w, b in zip(("means", "conics", "values"), (gm, gc, gv), want, bound):
            err = np.abs(g.cpu().double().numpy()[gsel] - w)
            assert (err <= b).all(), (flags, name, float((err / b).max()))


def test_wide_gaussians_keep_record_ranges(Sampler):
    """More Gaussians reach a 16-point group than a list holds (very wide Gaussians over close points): such
    tiles keep the grid's record ranges around their box, and the sampling kernels test the ranges' records
    against the group boxes themselves (the row-cooperative fallback; scattered points take the per-point
    walk instead: test_sparse_scattered_points_keep_their_group_lists)."""
    rng = np.random.default_rng(8)
    means, con, values = random_gaussians(rng, 1500, 1, log_sigma_mean=-1.2, log_sigma_std=0.3, lo=-0.5, hi=0.5)
    samples = rng.uniform(-0.5, 0.5, (3000, 2))
    s = check_case(Sampler, means, con, values, samples, orders=(0, 1, 2), gtol="bound")
    from tools.prof_step import list_stats
    st = list_stats(s._plan)
    assert st["ranges_tiles"] > 0 and st["points_tiles"] == 0, st


def test_clustered_cloud_with_thin_outskirts(Sampler):
    """Clamped normal points (test_no_mlp.py:86 draws randn / 2 clamped; here sigma = 0.15 of the half-width at
    60 000 points): a dense core sets the sample cells' size, the outskirts are a few isolated points per
    cell block.  Core tiles use lists, outskirt tiles the per-point walk; every point and every gradient
    against the oracle."""
    rng = np.random.default_rng(12)
    means, con, values = random_gaussians(rng, 12000, 1, log_sigma_mean=-4.4, log_sigma_std=0.25)
    samples = np.clip(rng.normal(0, 0.15, (60000, 2)), -1, 1)
    s = check_case(Sampler, means, con, values, samples, orders=(0, 1, 2), gtol="bound")
    from tools.prof_step import list_stats
    st = list_stats(s._plan)
    assert st["points_tiles"] > 0, st

"""GPU: the two ways a samples build sorts the caller's points (include/pigs_amd.h, pigs_samples_build):
the one-pass build (points that arrive in runs sharing a cell) and the coarse-bin build (points in no
order: torch.rand collocation points, /root/reference/main_pn.py:103; clamped normals,
test_no_mlp.py:86).  Both must give the oracle's numbers on any input; the library's memory of how a
point set of a given size last arrived only chooses between them."""
import os

import numpy as np
import pytest
import torch

from test_binned_gpu import check_case, random_gaussians, rel, dev32, TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Sampler(hip_lib):
    assert torch.cuda.is_available()
    from diff_gaussian_sampling import GaussianSampler
    return GaussianSampler


class forced_order:
    """PIGS_SAMPLES_ORDER for the builds inside the block (the library reads it at every build)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = os.environ.get("PIGS_SAMPLES_ORDER")
        if self.mode is None:
            os.environ.pop("PIGS_SAMPLES_ORDER", None)
        else:
            os.environ["PIGS_SAMPLES_ORDER"] = self.mode

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("PIGS_SAMPLES_ORDER", None)
        else:
            os.environ["PIGS_SAMPLES_ORDER"] = self.old


def point_sets(rng):
    yield "one point", rng.uniform(-1, 1, (1, 2))
    yield "63 points", rng.uniform(-1, 1, (63, 2))
    yield "random 5k", rng.uniform(-1, 1, (5000, 2))
    for M in (2048, 2049, 6143):                 # whole chunks of the coarse-bin build, one point more, one point less
        yield f"random {M}", rng.uniform(-1, 1, (M, 2))
    yield "random 70k + ragged tile", rng.uniform(-1, 1, (70 * 1024 + 37, 2))
    g = np.linspace(-1, 1, 200)
    gx, gy = np.meshgrid(g, g, indexing="xy")
    grid = np.stack((gx, gy), -1).reshape(-1, 2)
    yield "grid in row order", grid
    yield "shuffled grid", grid[rng.permutation(grid.shape[0])]
    yield "clamped normal", np.clip(rng.normal(0, 0.3, (50000, 2)), -1, 1)
    # (thousands of IDENTICAL contributions round the same way: the per-entry accumulation bound assumes
    # independent roundings, so these two are held to the plain 1e-5 of the largest entry)
    yield "all points on one spot", np.full((3000, 2), 0.25)
    line = np.stack((np.linspace(-1, 1, 9000), np.zeros(9000)), -1)
    yield "a line (degenerate box)", line[rng.permutation(9000)]


@pytest.mark.parametrize("mode", ["unordered", "ordered"])
def test_both_builds_match_the_oracle_on_any_point_set(Sampler, mode):
    rng = np.random.default_rng(5)
    means, con, values = random_gaussians(rng, 1200, 2, log_sigma_mean=-3.0, log_sigma_std=0.5)
    with forced_order(mode):
        for name, pts in point_sets(rng):
            try:
                check_case(Sampler, means, con, values, pts, orders=(0, 1, 2),
                           gtol=TOL if name.startswith(("all points", "a line")) else "bound")
            except AssertionError as e:
                raise AssertionError(f"{mode} build, {name}: {e}") from e


def test_coarse_bin_build_at_bench_size_equals_the_one_pass_build(Sampler):
    """C3-sized random points (1 M, 65 536 lattice Gaussians, kappa 0.5): the two builds sort the same points
    into the same cells, so the outputs differ by summation order inside a tile only; a subset against the
    oracle."""
    from pigs_amd import synthetic
    from oracle import c_oracle
    gs = synthetic.lattice_gaussians(256, 256, 0.5, seed=1)
    t = {k: v.float().cuda() for k, v in gs.items()}
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand((1 << 20, 2), generator=g) * 2 - 1).cuda()
    outs = {}
    for mode in ("ordered", "unordered"):
        with forced_order(mode), torch.no_grad():
            s = Sampler(False, backend="binned", fuse="all", reuse_samples=False)
            s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
            outs[mode] = [o.clone() for o in s.sample((0, 1, 2))]
    for a, b in zip(outs["ordered"], outs["unordered"]):
        assert rel(a, b.cpu().double().numpy()) < 2e-6
    sub = torch.randperm(pts.shape[0], generator=g)[:1024]
    args = [t[k].detach().cpu().double().numpy() for k in ("means", "conics", "values")] + [pts[sub.cuda()].cpu().double().numpy()]
    exp = c_oracle.forward(*args, orders=(0, 1, 2))
    for o, out in enumerate(outs["unordered"]):
        assert rel(out[sub.cuda()], exp[o]) < TOL, (o, rel(out[sub.cuda()], exp[o]))


def test_library_remembers_how_a_point_set_of_a_size_arrived(Sampler, hip_lib):
    """No environment override: builds of M >= 131 072 points leave their run statistic behind (asked for after
    the first two builds of a size and after every 16th), and later builds of that M take the recommended
    path -- random points switch to coarse bins, a lattice of the same size switches back.  Results hold
    on either path."""
    M = 400 * 400
    rng = np.random.default_rng(9)
    means, con, values = random_gaussians(rng, 2000, 1, log_sigma_mean=-3.3, log_sigma_std=0.4)
    g = np.linspace(-1, 1, 400)
    gx, gy = np.meshgrid(g, g, indexing="xy")
    grid = np.stack((gx, gy), -1).reshape(-1, 2)
    rnd = rng.uniform(-1, 1, (M, 2))
    t = [dev32(a) for a in (means, values, con)]
    with forced_order(None):
        for pts, want in ((rnd, 1), (grid, 0), (rnd, 1)):
            s = Sampler(False, backend="binned", reuse_samples=False)
            p = dev32(pts)
            for _ in range(36):          # two statistic copies land in this many builds whatever the count so far
                with torch.no_grad():
                    s.preprocess(t[0], t[1], None, t[2], p)
                torch.cuda.synchronize()
            assert hip_lib.pigs_samples_order_hint(M) == want, (want, hip_lib.pigs_samples_order_hint(M))
            check_case(Sampler, means, con, values, pts, orders=(0, 1), bwd=False)
        assert hip_lib.pigs_samples_order_hint(12345) == -1       # small sets are never noted


def test_captured_step_on_the_coarse_bin_build(Sampler):
    """The coarse-bin build is five plain launches too: captured into a hipGraph, replayed on new points."""
    from pigs_amd.graphs import GraphedStep
    rng = np.random.default_rng(13)
    means, con, values = random_gaussians(rng, 900, 1, log_sigma_mean=-3.0, log_sigma_std=0.3)
    M = 40000
    with forced_order("unordered"):
        s = Sampler(False, backend="binned", fuse="all")

        def make_inputs():
            return tuple(dev32(a) for a in (means, values, con, rng.uniform(-1, 1, (M, 2))))

        def fn(m, v, c, p):
            with torch.no_grad():
                s.preprocess(m, v, None, c, p)
                return s.sample((0, 1, 2))

        step = GraphedStep(fn, make_inputs)
        new = dev32(rng.uniform(-0.7, 0.9, (M, 2)))
        step.inputs[3].copy_(new)
        outs = [o.clone() for o in step()]
        torch.cuda.synchronize()
        s2 = Sampler(False, backend="dense")
        with torch.no_grad():
            s2.preprocess(*step.inputs[:2], None, step.inputs[2], new)
            for a, b in zip(outs, s2.sample((0, 1, 2))):
                assert rel(a, b.cpu().double().numpy()) < TOL


def test_the_memory_does_not_run_out_of_slots(Sampler, hip_lib):
    """The library remembers a handful of sizes.  Sizes that were built once and never came back leave slots
    waiting for their statistic: a new size must still get a slot (the landed copies are noted, the least
    recently used slot is given away) and be recognised as unordered."""
    rng = np.random.default_rng(17)
    means, con, values = random_gaussians(rng, 300, 1, log_sigma_mean=-3.0, log_sigma_std=0.3)
    t = [dev32(a) for a in (means, values, con)]
    with forced_order(None), torch.no_grad():
        s = Sampler(False, backend="binned", reuse_samples=False)
        for k in range(24):                              # more one-off sizes than there are slots
            s.preprocess(t[0], t[1], None, t[2], dev32(rng.uniform(-1, 1, (140000 + 64 * k, 2))))
        torch.cuda.synchronize()
        M = 151111
        p = dev32(rng.uniform(-1, 1, (M, 2)))
        for _ in range(36):
            s.preprocess(t[0], t[1], None, t[2], p)
            torch.cuda.synchronize()
        assert hip_lib.pigs_samples_order_hint(M) == 1


class forced_stage:
    """PIGS_STAGE for the sampling launches inside the block (read at every launch)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("PIGS_STAGE")
        os.environ["PIGS_STAGE"] = "1" if self.on else "0"

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("PIGS_STAGE", None)
        else:
            os.environ["PIGS_STAGE"] = self.old


@pytest.mark.parametrize("orders", [(0, 1, 2), (0, 1, "lap"), (0, 1), (2,), (1, "lap")])
def test_staged_outputs_and_gradients_match_the_direct_path(Sampler, orders):
    """Points in no order send their outputs / fetch their incoming gradients through one 32-byte record per point
    (PlanView::stage, plan.h) instead of three scattered accesses.  Staging forced on a shuffled point set with
    ragged last tile, every combination of requested outputs (the launches cover the subset with the (0, 1, 2) /
    (0, 1, trace) kernels: unrequested outputs and absent gradients are null there): outputs and gradients equal
    to the direct path's up to float32 summation order, outputs against the oracle."""
    from oracle import c_oracle
    rng = np.random.default_rng(21)
    means, con, values = random_gaussians(rng, 1500, 1, log_sigma_mean=-3.1, log_sigma_std=0.4)
    pts = rng.uniform(-1, 1, (50 * 64 + 29, 2))
    res = {}
    for on in (False, True):
        with forced_stage(on):
            t = [dev32(a) for a in (means, values, con, pts)]
            for x in t[:3]:
                x.requires_grad_(True)
            s = Sampler(True, backend="binned")
            s.preprocess(t[0], t[1], None, t[2], t[3])
            outs = s.sample(orders)
            gen = torch.Generator(device="cpu").manual_seed(5)
            rs = [torch.randn(o.shape, generator=gen).cuda() for o in outs]
            sum((o * r).sum() for o, r in zip(outs, rs)).backward()
            res[on] = ([o.detach().clone() for o in outs], [x.grad.clone() for x in t[:3]], rs)
    for a, b in zip(res[False][0], res[True][0]):       # (two builds: the order inside a cell, hence the lists, may differ)
        assert rel(b, a.cpu().double().numpy()) < 2e-6
    for a, b in zip(res[False][1], res[True][1]):
        assert rel(b, a.cpu().double().numpy()) < 2e-6
    args = [np.asarray(x, dtype=np.float64) for x in (means, con, values, pts)]
    want = c_oracle.forward(*args, orders=tuple(o for o in orders if o != "lap") + ((2,) if "lap" in orders else ()))
    for o, out in zip(orders, res[True][0]):
        if o == "lap":
            exp = want[2][:, 0, 0, :] + want[2][:, 1, 1, :]
        else:
            exp = want[o]
        assert rel(out, exp) < TOL, (o, rel(out, exp))


def test_a_dense_patch_shares_its_bin_between_workgroups(Sampler):
    """A clamped-normal cloud with a tight core (test_no_mlp.py:86 at scale): its central coarse bins hold tens of
    thousands of points -- more than one workgroup's batch -- and are sorted by eight workgroups, each taking a slice of
    the bin's cells and offsetting itself by the points it sees below its slice (samples_binsort_kernel).  Every point
    and every gradient against the oracle; and the same cloud through the one-pass build."""
    rng = np.random.default_rng(29)
    means, con, values = random_gaussians(rng, 1500, 1, log_sigma_mean=-3.6, log_sigma_std=0.4)
    pts = np.clip(rng.normal(0, 0.08, (150000, 2)), -1, 1)
    for mode in ("unordered", "ordered"):
        with forced_order(mode):
            check_case(Sampler, means, con, values, pts, orders=(0, 1, 2), gtol="bound")

"""GPU: the INDEX-TILED order of a samples build (include/pigs_amd.h, pigs_samples_lattice_offset): a point set
that arrives as a lattice in row order -- meshgrid(indexing="xy").reshape(-1, 2), the reference's own grids
(/root/reference/test_gaussian_sampling.py:43-46, main_pn.py:317-324) -- is not sorted; its tiles are index
arithmetic.  Results must be the oracle's whichever order the build takes, and the build must take the sort
whenever the index tiles would not be compact."""
import os

import numpy as np
import pytest
import torch

from test_binned_gpu import check_case, random_gaussians, dev32, rel as _rel

pytestmark = pytest.mark.gpu


def rel(a, b):
    return _rel(a, b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else b)


@pytest.fixture(scope="module")
def Sampler(hip_lib):
    assert torch.cuda.is_available()
    from diff_gaussian_sampling import GaussianSampler
    return GaussianSampler


class lattice_env:
    """PIGS_LATTICE for the builds inside the block (read at every build)."""

    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self.old = os.environ.get("PIGS_LATTICE")
        if self.value is None:
            os.environ.pop("PIGS_LATTICE", None)
        else:
            os.environ["PIGS_LATTICE"] = self.value

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("PIGS_LATTICE", None)
        else:
            os.environ["PIGS_LATTICE"] = self.old


def lattice_of(sampler, hip_lib):
    off = hip_lib.pigs_samples_lattice_offset()
    ws = sampler._plan.samples.workspace
    rf, rs = ws[off:off + 8].view(torch.int32).cpu().tolist()
    return rf, rs


def grid(rx, ry, indexing="xy", lo=(-1.0, -0.7), hi=(1.0, 0.9)):
    gx, gy = np.meshgrid(np.linspace(lo[0], hi[0], rx), np.linspace(lo[1], hi[1], ry), indexing=indexing)
    return np.stack((gx, gy), -1).reshape(-1, 2)


CASES = [
    ("8 x 8: one tile", grid(8, 8), (8, 8)),
    ("64 x 64", grid(64, 64), (64, 64)),
    ("24 x 40: odd tile counts both ways", grid(24, 40), (24, 40)),
    ("40 x 24", grid(40, 24), (40, 24)),
    ("8 x 200: one tile column", grid(8, 200), (8, 200)),
    ("264 x 8: one tile row", grid(264, 8), (264, 8)),
    ("ij order (y fastest)", grid(48, 32, indexing="ij"), (32, 48)),
    ("3000 x 8: row longer than the first search window", grid(3000, 8), (3000, 8)),
    ("100 x 100: not multiples of 8", grid(100, 100), (0, 0)),
    ("60 x 64", grid(60, 64), (0, 0)),
    ("64 x 60", grid(64, 60), (0, 0)),
]


@pytest.mark.parametrize("name,pts,want", CASES, ids=[c[0] for c in CASES])
def test_lattices_are_index_tiled_and_match_the_oracle(Sampler, hip_lib, name, pts, want):
    rng = np.random.default_rng(7)
    means, con, values = random_gaussians(rng, 600, 1, log_sigma_mean=-2.6, log_sigma_std=0.5)
    with lattice_env("1"):           # (without the variable the index-tiled order starts at 2^12 points)
        s = check_case(Sampler, means, con, values, pts, orders=(0, 1, 2), gtol="bound")
        assert lattice_of(s, hip_lib) == want
    with lattice_env("0"):
        s = check_case(Sampler, means, con, values, pts, orders=(0, 1, 2), gtol="bound")
        assert lattice_of(s, hip_lib) == (0, 0)


def test_a_lattice_candidate_that_is_not_compact_is_sorted(Sampler, hip_lib):
    """The first descent says 64 points per row and 64 rows -- but the rows come in no order (a tile of 8 index
    rows spans the domain): the build must fall back to the sort, and be right either way."""
    rng = np.random.default_rng(11)
    means, con, values = random_gaussians(rng, 500, 1, log_sigma_mean=-2.6, log_sigma_std=0.4)
    g = grid(64, 64).reshape(64, 64, 2)
    rows = rng.permutation(64)
    rows = np.concatenate(([0], rows[rows != 0]))        # row 0 stays first: the candidate is found
    scrambled = g[rows].reshape(-1, 2)
    with lattice_env("1"):
        s = check_case(Sampler, means, con, values, scrambled, orders=(0, 1, 2), gtol="bound")
        assert lattice_of(s, hip_lib) == (0, 0)
        # columns in no order inside every row: the first descent comes early, no candidate at all
        cols = g[:, rng.permutation(64)].reshape(-1, 2)
        s = check_case(Sampler, means, con, values, cols, orders=(0, 1, 2), gtol="bound")
        assert lattice_of(s, hip_lib) == (0, 0)


def test_jittered_lattice_and_nonfinite_points(Sampler, hip_lib, monkeypatch):
    monkeypatch.setenv("PIGS_LATTICE", "1")
    rng = np.random.default_rng(13)
    means, con, values = random_gaussians(rng, 500, 1, log_sigma_mean=-2.6, log_sigma_std=0.4)
    g = grid(64, 48, lo=(-1, -1), hi=(1, 1))
    step = np.array([2 / 63, 2 / 47])
    jit = g + rng.uniform(-0.2, 0.2, g.shape) * step        # rows stay monotone in x: still a lattice in row order
    s = check_case(Sampler, means, con, values, jit, orders=(0, 1, 2), gtol="bound")
    assert lattice_of(s, hip_lib) == (64, 48)
    # a NaN point: the index tiles are "not compact", the sort takes over; the NaN point's outputs are NaN,
    # every other point's the oracle's
    bad = g.copy()
    bad[1000] = np.nan
    t = [dev32(a) for a in (means, values, con, bad)]
    s = Sampler(True, backend="binned")
    s.preprocess(t[0], t[1], None, t[2], t[3])
    u = s.sample_gaussians()
    assert lattice_of(s, hip_lib) == (0, 0)
    d = Sampler(True, backend="dense")
    d.preprocess(t[0], t[1], None, t[2], t[3])
    ud = d.sample_gaussians()
    keep = np.arange(g.shape[0]) != 1000
    assert rel(u[keep], ud[keep]) < 1e-5


def test_index_tiled_samples_are_reused_and_survive_new_gaussians(Sampler, hip_lib, monkeypatch):
    """The reference's roll-out (main_pn.py:317-324): new Gaussians on the same grid every step -- the index-tiled
    samples half is built once and shared."""
    from oracle import c_oracle
    monkeypatch.setenv("PIGS_LATTICE", "1")
    rng = np.random.default_rng(17)
    pts = dev32(grid(96, 64))
    s = Sampler(False, backend="binned", fuse="all")
    first = None
    for step in range(3):
        means, con, values = random_gaussians(rng, 700, 1, log_sigma_mean=-2.8, log_sigma_std=0.4)
        t = [dev32(a) for a in (means, values, con)]
        s.preprocess(t[0], t[1], None, t[2], pts)
        if first is None:
            first = s._plan.samples
        assert s._plan.samples is first and lattice_of(s, hip_lib) == (96, 64)
        outs = s.sample((0, 1, 2))
        exp = c_oracle.forward(*[x.cpu().double().numpy() for x in (t[0], t[2], t[1])], pts.cpu().double().numpy(), orders=(0, 1, 2))
        for o in range(3):
            assert rel(outs[o], exp[o]) < 1e-5


def test_bench_grid_is_index_tiled_and_equals_the_sorted_build(Sampler, hip_lib):
    """C3's own points (1024^2 grid, 65 536 lattice Gaussians, kappa 0.5; the library's own choice: index-tiled from
    2^12 points): index-tiled and sorted builds evaluate the same pairs up to the order inside a tile; a slice against
    the oracle.  BASELINE configs[1]'s 256^2 grid is index-tiled too, a 56 x 56 grid is sorted unless asked otherwise."""
    from oracle import c_oracle
    from pigs_amd import synthetic
    gs = synthetic.lattice_gaussians(256, 256, 0.5, seed=0)
    pts = synthetic.grid_samples(1024).float().cuda()
    t = {k: v.float().cuda() for k, v in gs.items()}
    outs = {}
    for mode in (None, "0"):
        with lattice_env(mode):
            s = Sampler(False, backend="binned", fuse="all")
            with torch.no_grad():
                s.preprocess(t["means"], t["values"], None, t["conics"], pts)
                outs[mode] = [o.clone() for o in s.sample((0, 1, 2))]
            assert lattice_of(s, hip_lib) == ((1024, 1024) if mode is None else (0, 0))
    for a, b in zip(outs[None], outs["0"]):
        assert rel(a, b) < 2e-6
    idx = torch.arange(0, pts.shape[0], 509, device="cuda")[:2048]
    args = [gs[k].float().double().numpy() for k in ("means", "conics", "values")]
    exp = c_oracle.forward(*args, pts[idx].cpu().double().numpy(), orders=(0, 1, 2))
    for o in range(3):
        assert rel(outs[None][o][idx], exp[o]) < 1e-5
    with lattice_env(None), torch.no_grad():
        s = Sampler(False, backend="binned")
        s.preprocess(t["means"], t["values"], None, t["conics"], synthetic.grid_samples(256).float().cuda())
        s.sample_gaussians()
        assert lattice_of(s, hip_lib) == (256, 256)
        s.preprocess(t["means"], t["values"], None, t["conics"], synthetic.grid_samples(56).float().cuda())
        s.sample_gaussians()
        assert lattice_of(s, hip_lib) == (0, 0)


def test_points_that_stop_being_a_lattice_after_the_library_expected_one(Sampler, hip_lib, monkeypatch):
    """The library remembers the row length of the last build of a size and, expecting a lattice, launches an eighth
    of the one-pass count's workgroups (they leave at once when the points are index-tiled).  When the next point set
    of that size is no lattice they stride over all the points: same results, and the memory turns around."""
    monkeypatch.setenv("PIGS_LATTICE", "1")
    rng = np.random.default_rng(23)
    means, con, values = random_gaussians(rng, 500, 1, log_sigma_mean=-2.6, log_sigma_std=0.4)
    g = grid(128, 64)                                     # 8 192 points: 8 blocks of the one-pass count
    t = [dev32(a) for a in (means, values, con)]
    s = Sampler(False, backend="binned", reuse_samples=False)
    for _ in range(4):                                    # the row length lands in the library's memory
        with torch.no_grad():
            s.preprocess(t[0], t[1], None, t[2], dev32(g))
            s.sample_gaussians()
        torch.cuda.synchronize()
    assert lattice_of(s, hip_lib) == (128, 64)
    for pts in (rng.uniform(-1, 1, g.shape), g[rng.permutation(g.shape[0])], g):
        s2 = check_case(Sampler, means, con, values, pts, orders=(0, 1, 2), gtol="bound")
        assert lattice_of(s2, hip_lib) == ((128, 64) if pts is g else (0, 0))


@pytest.mark.parametrize("world,rank", [(2, 1), (8, 3)])
def test_a_rank_s_rows_of_the_weak_scaling_grid_are_index_tiled(Sampler, hip_lib, world, rank):
    """bench.py --gpus N (weak scaling): rank r's block of rows of the side x side grid, side a multiple of 8 N -- a
    lattice with both sides multiples of 8, taken in index-tiled order by the library's own choice; a slice of it
    against the oracle."""
    import math
    from oracle import c_oracle
    from pigs_amd import synthetic
    side = max(1, int(round(1024 * math.sqrt(world) / (8 * world)))) * 8 * world
    rows = side // world
    pts = synthetic.grid_samples(side, side, row0=rank * rows, rows=rows).float().cuda()
    rng = np.random.default_rng(29)
    means, con, values = random_gaussians(rng, 3000, 1, log_sigma_mean=-3.4, log_sigma_std=0.4)
    t = [dev32(a) for a in (means, values, con)]
    with lattice_env(None), torch.no_grad():
        s = Sampler(False, backend="binned", fuse="all")
        s.preprocess(t[0], t[1], None, t[2], pts)
        outs = s.sample((0, 1, 2))
    assert lattice_of(s, hip_lib) == (side, rows)
    idx = torch.arange(0, pts.shape[0], 257, device="cuda")[:3000]
    exp = c_oracle.forward(*[x.cpu().double().numpy() for x in (t[0], t[2], t[1])], pts[idx].cpu().double().numpy(), orders=(0, 1, 2))
    for o in range(3):
        assert rel(outs[o][idx], exp[o]) < 1e-5


@pytest.mark.parametrize("strips", ["0", "1"])
def test_gaussians_one_launch_ahead_and_points_that_break_the_expectation(Sampler, hip_lib, monkeypatch, strips):
    """With a lattice expected and the same bounding box in the last two completed builds of a size, the Gaussians are
    binned on the REMEMBERED box in the launch that looks at the points (plan.hip, BuildArgs::ahead): three launches
    in front of the tile lists instead of four.  Results must not depend on it -- not when the Gaussians change, not
    when the next point set of that size lies elsewhere (the Gaussians' grid then covers the wrong domain: slower,
    never wrong), not when it is no lattice at all (its workgroups scan and scatter in the third launch)."""
    from oracle import c_oracle
    monkeypatch.setenv("PIGS_LATTICE", "1")
    # strips = "1": the Gaussians keep the caller's order (PIGS_GAUSS_STRIPS) -- nothing of them is left for the second and
    # third launch then, there is no third, and points that are no lattice after all are counted, scanned and scattered
    # inside the second, behind device-wide barriers (plan.hip, samples_sort_in_count)
    monkeypatch.setenv("PIGS_GAUSS_STRIPS", strips)
    rng = np.random.default_rng(31)
    g = grid(96, 64)
    s = Sampler(False, backend="binned", fuse="all", reuse_samples=False)

    def run(points, lo=-1.0, hi=1.0, n=700):
        means, con, values = random_gaussians(rng, n, 1, log_sigma_mean=-2.8, log_sigma_std=0.4, lo=lo, hi=hi)
        t = [dev32(a) for a in (means, values, con)]
        points = points.astype(np.float32).astype(np.float64)      # (the oracle sees the coordinates the device sees)
        p = dev32(points)
        with torch.no_grad():
            s.preprocess(t[0], t[1], None, t[2], p)
            outs = s.sample((0, 1, 2))
        torch.cuda.synchronize()
        exp = c_oracle.forward(*[x.cpu().double().numpy() for x in (t[0], t[2], t[1])], points, orders=(0, 1, 2))      # (the inputs the device saw)
        for o in range(3):
            assert rel(outs[o], exp[o]) < 1e-5, (o, rel(outs[o], exp[o]))
        return lattice_of(s, hip_lib)

    for _ in range(6):                      # the memory fills, the plan workspaces are recycled ones: ahead from here on
        assert run(g) == (96, 64)
    assert run(g * 0.5 + 3.0, lo=2.5, hi=3.5) == (96, 64)      # the same lattice somewhere else: the remembered box is wrong
    assert run(g) == (96, 64)
    assert run(rng.uniform(-1, 1, g.shape)) == (0, 0)           # no lattice: the fall-back of the third launch
    assert run(g[rng.permutation(g.shape[0])]) == (0, 0)
    for _ in range(5):
        run(rng.uniform(-1, 1, g.shape))                        # ... until the memory has turned around
    for _ in range(6):
        assert run(g) == (96, 64)                               # and back


def test_a_million_points_that_stop_being_a_lattice_are_sorted_inside_the_count_launch(Sampler, hip_lib, monkeypatch):
    """The same at C3's size (1024^2 points, 256 sample workgroups striding over 1 024 blocks, 65 scan blocks dealt out
    among them): lattice, lattice, ..., then a permutation of it and uniform random points -- every result against the
    dense HIP path on a slice, and against the sorted build of the same points."""
    monkeypatch.setenv("PIGS_GAUSS_STRIPS", "1")
    from pigs_amd import synthetic
    gs = synthetic.lattice_gaussians(64, 64, 0.5, seed=1)
    t = {k: v.float().cuda() for k, v in gs.items()}
    lat = synthetic.grid_samples(1024).float().cuda()
    gen = torch.Generator().manual_seed(3)
    others = [lat[torch.randperm(lat.shape[0], generator=gen).cuda()], (torch.rand(lat.shape, generator=gen) * 2 - 1).cuda()]
    s = Sampler(False, backend="binned", fuse="all", reuse_samples=False)

    def run(pts):
        with torch.no_grad():
            s.preprocess(t["means"], t["values"], None, t["conics"], pts)
            outs = [o.clone() for o in s.sample((0, 1, 2))]
        torch.cuda.synchronize()
        return outs, lattice_of(s, hip_lib)

    for _ in range(5):
        assert run(lat)[1] == (1024, 1024)
    idx = torch.arange(0, lat.shape[0], 997, device="cuda")[:1024]
    for pts in others:
        outs, kind = run(pts)                      # expected a lattice: sorted inside the count launch
        assert kind == (0, 0)
        d = Sampler(False, backend="dense")
        with torch.no_grad():
            d.preprocess(t["means"], t["values"], None, t["conics"], pts[idx].contiguous())
            ref = d.sample((0, 1, 2))
        for o in range(3):
            assert rel(outs[o][idx], ref[o].cpu().double().numpy()) < 2e-5
        assert bool(torch.isfinite(outs[2]).all())
    assert run(lat)[1] == (1024, 1024)


def test_a_lattice_of_another_shape_and_the_same_count(Sampler, hip_lib, monkeypatch):
    """With a row length remembered the first launch does not search for one: it verifies the remembered one.  A
    lattice of the same count and another shape fails that (its row ends are steps as wide as the domain), is sorted,
    the memory forgets the row length, and a later build finds the new one.  Every build against the oracle."""
    from oracle import c_oracle
    monkeypatch.setenv("PIGS_LATTICE", "1")
    rng = np.random.default_rng(37)
    means, con, values = random_gaussians(rng, 500, 1, log_sigma_mean=-2.6, log_sigma_std=0.4)
    t = [dev32(a) for a in (means, values, con)]
    args = [x.cpu().double().numpy() for x in (t[0], t[2], t[1])]
    s = Sampler(False, backend="binned", fuse="all", reuse_samples=False)

    def run(g):
        p = dev32(g)
        with torch.no_grad():
            s.preprocess(t[0], t[1], None, t[2], p)
            outs = s.sample((0, 1, 2))
        torch.cuda.synchronize()
        exp = c_oracle.forward(*args, p.cpu().double().numpy(), orders=(0, 1, 2))
        for o in range(3):
            assert rel(outs[o], exp[o]) < 1e-5
        return lattice_of(s, hip_lib)

    a, b = grid(128, 64), grid(64, 128)
    for _ in range(4):
        assert run(a) == (128, 64)
    kinds = [run(b) for _ in range(40)]
    assert kinds[0] in ((0, 0), (64, 128)) and kinds[-1] == (64, 128), kinds

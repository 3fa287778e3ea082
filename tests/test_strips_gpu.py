"""GPU: builds that keep the Gaussians in the CALLER's order (include/pigs_amd.h, pigs_plan_strips_offset; plan.h,
PlanParams::strips): every 16 consecutive Gaussians are a strip with a bounding box (16 strips a super-strip), the tile lists come from the strip
boxes -- no count, scan or scatter.  Results must be the oracle's whatever the order of the Gaussians (a lattice as the
reference lays it out, /root/reference/model_pn.py:338-342; a shuffled one: every strip then reaches everywhere), and
the library must take the strips only for Gaussians whose strips cover the domain a few times over."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from pigs_amd import synthetic
from test_binned_gpu import check_case, random_gaussians, dev32, rel as _rel

pytestmark = pytest.mark.gpu


def rel(a, b):
    return _rel(a, b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else b)


@pytest.fixture(scope="module")
def Sampler(hip_lib):
    assert torch.cuda.is_available()
    from diff_gaussian_sampling import GaussianSampler
    return GaussianSampler


class strips_env:
    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self.old = os.environ.get("PIGS_GAUSS_STRIPS")
        if self.value is None:
            os.environ.pop("PIGS_GAUSS_STRIPS", None)
        else:
            os.environ["PIGS_GAUSS_STRIPS"] = self.value

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("PIGS_GAUSS_STRIPS", None)
        else:
            os.environ["PIGS_GAUSS_STRIPS"] = self.old


def strips_of(sampler, hip_lib, plan=None):
    off = hip_lib.pigs_plan_strips_offset()
    ws = (plan or sampler._plan).workspace
    return int(ws[off:off + 4].view(torch.int32).cpu()[0])


def lattice(nx, ny, kappa, c=1, seed=0):
    gs = synthetic.lattice_gaussians(nx, ny, kappa, seed=seed, c=c)
    return [gs[k].float().double().numpy() for k in ("means", "conics", "values")]


CASES = [
    ("lattice 48 x 40", lambda rng: lattice(48, 40, 0.8), lambda rng: synthetic.grid_samples(96).numpy()),
    ("lattice, two channels", lambda rng: lattice(40, 40, 0.7, c=2), lambda rng: rng.uniform(-1, 1, (5000, 2))),
    ("lattice in no order", lambda rng: lattice(40, 40, 0.8), lambda rng: synthetic.grid_samples(64).numpy()),
    ("random Gaussians, 1000 (the last strip ragged)", lambda rng: list(random_gaussians(rng, 1000, 1, log_sigma_mean=-3.0, log_sigma_std=0.5)), lambda rng: rng.uniform(-1, 1, (7000, 2))),
    ("37 Gaussians: one ragged strip", lambda rng: list(random_gaussians(rng, 37, 1, log_sigma_mean=-2.0, log_sigma_std=0.5)), lambda rng: synthetic.grid_samples(72).numpy()),
    ("very wide Gaussians (record ranges)", lambda rng: list(random_gaussians(rng, 1500, 1, log_sigma_mean=-1.2, log_sigma_std=0.3, lo=-0.5, hi=0.5)), lambda rng: rng.uniform(-0.5, 0.5, (3000, 2))),
    ("thin outskirts", lambda rng: list(random_gaussians(rng, 6000, 1, log_sigma_mean=-4.2, log_sigma_std=0.25)), lambda rng: np.clip(rng.normal(0, 0.2, (20000, 2)), -1, 1)),
]


@pytest.mark.parametrize("name,gauss,points", CASES, ids=[c[0] for c in CASES])
def test_strip_builds_match_the_oracle(Sampler, hip_lib, name, gauss, points):
    rng = np.random.default_rng(41)
    means, con, values = gauss(rng)
    if name == "lattice in no order":
        perm = rng.permutation(means.shape[0])
        means, con, values = means[perm], con[perm], values[perm]
    pts = points(rng)
    orders = (0, 1, 2, 3) if values.shape[1] == 2 or "37" in name else (0, 1, 2)
    with strips_env("1"):
        s = check_case(Sampler, means, con, values, pts, orders=orders, gtol="bound")
        assert strips_of(s, hip_lib) == 1
    with strips_env("0"):
        s = check_case(Sampler, means, con, values, pts, orders=orders, gtol="bound")
        assert strips_of(s, hip_lib) == 0


def test_degenerate_and_nonfinite_gaussians_in_strips(Sampler, hip_lib, monkeypatch):
    """A NaN centre, a singular conic: such a strip reaches everywhere; the other Gaussians' sums are untouched by the
    order (compared with the cell order's results, which the existing tests hold to the oracle)."""
    monkeypatch.setenv("PIGS_GAUSS_STRIPS", "1")
    rng = np.random.default_rng(43)
    means, con, values = random_gaussians(rng, 900, 1, log_sigma_mean=-2.8, log_sigma_std=0.4)
    values[100] = 0.0
    con[100] = (1.0, 1.0, 1.0)          # singular conic with value 0: contributes nothing, must not hide others
    pts = synthetic.grid_samples(80).numpy()
    t = [dev32(a) for a in (means, values, con, pts)]
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PIGS_GAUSS_STRIPS", mode)
        s = Sampler(False, backend="binned")
        with torch.no_grad():
            s.preprocess(t[0], t[1], None, t[2], t[3])
            outs[mode] = [o.clone() for o in s.sample((0, 1, 2))]
        assert strips_of(s, hip_lib) == int(mode)
    for a, b in zip(outs["1"], outs["0"]):
        assert rel(a, b) < 2e-6


def test_the_library_takes_strips_for_a_lattice_and_cells_for_gaussians_in_no_order(Sampler, hip_lib):
    """The library's own choice: the first builds of a size go through the cells and measure how often the strips
    would cover the domain; from then on a lattice keeps the caller's order, shuffled Gaussians never do; Gaussians of
    the same count that lose their order fall back once the measurement has come home.  Every step against the
    oracle."""
    rng = np.random.default_rng(47)
    pts_np = synthetic.grid_samples(96).numpy()
    pts = dev32(pts_np)
    base = lattice(48, 48, 0.8, seed=3)

    def run(s, gauss):
        t = [dev32(a) for a in (gauss[0], gauss[2], gauss[1])]
        with torch.no_grad():
            s.preprocess(t[0], t[1], None, t[2], pts)
            outs = s.sample((0, 1, 2))
        torch.cuda.synchronize()
        exp = c_oracle.forward(*[x.cpu().double().numpy() for x in (t[0], t[2], t[1])], pts_np, orders=(0, 1, 2))
        for o in range(3):
            assert rel(outs[o], exp[o]) < 1e-5
        return strips_of(s, hip_lib)

    def moved(gauss, scale):
        m = gauss[0] + rng.normal(0, scale * 2.0 / 48, gauss[0].shape)      # a fraction of a spacing
        return [m, gauss[1], gauss[2]]

    with strips_env(None):
        s = Sampler(False, backend="binned", fuse="all")
        kinds = [run(s, moved(base, 0.1)) for _ in range(8)]
        assert kinds[0] == 0 and kinds[-1] == 1, kinds
        perm = rng.permutation(base[0].shape[0])
        shuffled = [a[perm] for a in base]
        kinds = [run(s, shuffled) for _ in range(12)]          # same count, no order any more
        assert kinds[-1] == 0, kinds
        s2 = Sampler(False, backend="binned", fuse="all")
        rnd = list(random_gaussians(rng, 2500, 1, log_sigma_mean=-3.2, log_sigma_std=0.4))
        kinds = [run(s2, [rnd[0], rnd[1], rnd[2]]) for _ in range(6)]
        assert kinds == [0] * 6, kinds


def test_strips_through_training_steps(Sampler, hip_lib, monkeypatch):
    """forward + backward + new Gaussians every step, strips forced: gradients against the oracle."""
    monkeypatch.setenv("PIGS_GAUSS_STRIPS", "1")
    rng = np.random.default_rng(53)
    pts_np = synthetic.grid_samples(64).numpy()
    pts = dev32(pts_np)
    s = Sampler(False, backend="binned", fuse="all")
    for step in range(3):
        g = lattice(40, 40, 0.8, seed=step)
        t = [dev32(a).requires_grad_(True) for a in (g[0], g[2], g[1])]
        s.preprocess(t[0], t[1], None, t[2], pts)
        outs = s.sample((0, 1, 2))
        assert strips_of(s, hip_lib) == 1
        r = [dev32(rng.uniform(-1, 1, tuple(o.shape))) for o in outs]
        grads = torch.autograd.grad(sum((o * w).sum() for o, w in zip(outs, r)), t)
        args = [x.detach().cpu().double().numpy() for x in (t[0], t[2], t[1])]
        em, ec, ev = c_oracle.backward(*args, pts_np, {k: w.cpu().double().numpy() for k, w in enumerate(r)})
        assert rel(grads[0], em) < 1e-5 and rel(grads[1], ev) < 1e-5 and rel(grads[2], ec) < 1e-5


def test_a_cloud_with_thin_outskirts_keeps_the_cells(Sampler, hip_lib):
    """Tiles of far-apart points are walked point by point through the grid at sampling time (TILE_MODE_POINTS): lattice
    Gaussians under a clamped-normal cloud must stay with the cells -- whether the library learns it from a build through
    the cells (its queue of such tiles) or, with the strips forced for a while, from what a strips build noted."""
    from tools.prof_step import list_stats
    rng = np.random.default_rng(59)
    g = lattice(64, 64, 0.5, seed=5)
    t = [dev32(a) for a in (g[0], g[2], g[1])]
    pts_np = np.clip(rng.normal(0, 0.15, (150000, 2)), -1, 1).astype(np.float32).astype(np.float64)
    pts = dev32(pts_np)
    idx = np.arange(0, pts_np.shape[0], 37)[:3000]
    exp = c_oracle.forward(*[x.cpu().double().numpy() for x in (t[0], t[2], t[1])], pts_np[idx], orders=(0, 1, 2))

    def run(s):
        with torch.no_grad():
            s.preprocess(t[0], t[1], None, t[2], pts)
            outs = s.sample((0, 1, 2))
        torch.cuda.synchronize()
        for o in range(3):
            assert rel(outs[o][torch.as_tensor(idx, device="cuda")], exp[o]) < 1e-5
        return strips_of(s, hip_lib)

    with strips_env(None):
        s = Sampler(False, backend="binned", fuse="all")
        kinds = [run(s) for _ in range(8)]
        assert list_stats(s._plan)["points_tiles"] > 0
        assert kinds == [0] * 8, kinds
    with strips_env("1"):                      # forced: right, and slower
        s1 = Sampler(False, backend="binned", fuse="all")
        assert [run(s1) for _ in range(6)][-1] == 1
    with strips_env(None):                     # the memory has what the strips builds noted: back to the cells
        kinds = [run(s1) for _ in range(3)]
        assert kinds[-1] == 0, kinds


def test_a_captured_step_with_strips_replays_on_new_gaussians(hip_lib):
    """preprocess + forward + backward captured into a hipGraph with the strips forced, replayed on moved Gaussians
    (tests/test_graph_gpu.py's recipe: every replay against the oracle)."""
    from test_graph_gpu import test_captured_step_replays_on_new_parameters as captured_step
    with strips_env("1"):
        captured_step(hip_lib, "binned", 48, 96)

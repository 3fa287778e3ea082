"""GPU: PIGS_BUILD_DEFER_LISTS (include/pigs_amd.h) -- preprocess stops in front of the tile lists and the plan's
first sampling call builds them, a forward in the same launch as its own evaluation (plan_lists_forward_kernel).
Whatever comes first -- a fused forward, a forward with no fused variant, a backward, a residual -- and whatever
mode the tiles end up in, the numbers are the oracle's and the lists are there for every later call."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from pigs_amd import synthetic
from test_binned_gpu import random_gaussians, dev32, rel as _rel

pytestmark = pytest.mark.gpu


def rel(a, b):
    return _rel(a, b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else b)


@pytest.fixture(scope="module")
def Sampler(hip_lib):
    assert torch.cuda.is_available()
    from diff_gaussian_sampling import GaussianSampler
    return GaussianSampler


class env:
    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = os.environ.get(self.name)
        if self.value is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.value

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.old


def problem(c=1, seed=3, res=72):
    gs = synthetic.lattice_gaussians(24, 24, 0.8, seed=seed, c=c)
    t = {k: v.float().cuda() for k, v in gs.items()}
    pts = synthetic.grid_samples(res).float().cuda()
    args = [gs[k].float().double().numpy() for k in ("means", "conics", "values")]
    return t, pts, args


@pytest.mark.parametrize("host", ["native", "ctypes"])
@pytest.mark.parametrize("c,first", [(1, (0, 1, 2)), (1, (0,)), (1, (0, 1, "lap")), (1, (1,)), (1, (0, 1, 2, 3)), (2, (0, 1, 2)), (2, (0,))])
def test_first_forward_builds_the_lists_and_later_calls_read_them(Sampler, host, c, first):
    t, pts, args = problem(c)
    p64 = pts.cpu().double().numpy()
    exp = c_oracle.forward(*args, p64, orders=(0, 1, 2, 3))
    trace = exp[2][:, 0, 0] + exp[2][:, 1, 1]
    want = {0: exp[0], 1: exp[1], 2: exp[2], 3: exp[3], "lap": trace}
    for fused in (True, False):
        with env("PIGS_NO_FUSED_FIRST", None if fused else "1"):
            s = Sampler(False, backend="binned", fuse="none", host=host, defer_lists=True)
            req = {k: t[k].clone().requires_grad_(True) for k in ("means", "values", "conics")}
            s.preprocess(req["means"], req["values"], None, req["conics"], pts)
            outs = s.sample(first)
            for o, out in zip(first, outs):
                assert rel(out, want[o]) < 1e-5, (fused, o, rel(out, want[o]))
            # every later call reads the lists the first one wrote: the other orders, then the backward of all
            later = s.sample((0, 1, 2))
            for o in range(3):
                assert rel(later[o], want[o]) < 1e-5, (fused, "later", o)
            gen = torch.Generator().manual_seed(5)
            r = [torch.rand(o.shape, generator=gen).cuda() for o in later]
            grads = torch.autograd.grad(sum((o * w).sum() for o, w in zip(later, r)), list(req.values()))
            gm, gc, gv = c_oracle.backward(*args, p64, {k: w.cpu().double().numpy() for k, w in enumerate(r)})
            assert rel(grads[0], gm) < 1e-5 and rel(grads[1], gv) < 1e-5 and rel(grads[2], gc) < 1e-5


def test_residual_and_backward_as_first_calls(Sampler, hip_lib):
    """The C ABI directly: a residual forward as the first call (fused), and a BACKWARD as the first call on a plan
    with deferred lists (the list launch runs in front of it)."""
    import ctypes
    from pigs_amd import sampler as S
    t, pts, args = problem(1, seed=9)
    p64 = pts.cpu().double().numpy()
    s = Sampler(False, backend="binned", defer_lists=True)
    s.preprocess(t["means"], t["values"], None, t["conics"], pts)
    r = s.residual(a0=0.7, a1=(0.2, -0.4), lap=-0.5)
    exp = c_oracle.forward(*args, p64, orders=(0, 1, 2))
    want = 0.7 * exp[0] + 0.2 * exp[1][:, 0] - 0.4 * exp[1][:, 1] - 0.5 * (exp[2][:, 0, 0] + exp[2][:, 1, 1])
    assert rel(r, want) < 1e-5
    # backward first: a fresh plan through the ctypes layer, no forward in between
    means, values, conics = (t[k].contiguous() for k in ("means", "values", "conics"))
    plan = S.Plan(means, values, conics, pts, 36.0, q_max_backward=40.0, defer_lists=True)
    gen = torch.Generator().manual_seed(2)
    gouts = [torch.rand((pts.shape[0],) + (2,) * k + (1,), generator=gen).cuda() for k in range(3)] + [None, None]
    gm, gv, gc = S.backward_raw(means, values, conics, pts, gouts, 7, plan)
    em, ec, ev = c_oracle.backward(*args, p64, {k: gouts[k].cpu().double().numpy() for k in range(3)})
    assert rel(gm, em) < 1e-5 and rel(gv, ev) < 1e-5 and rel(gc, ec) < 1e-5
    # and the forward afterwards finds the lists built
    outs = S.forward_raw(means, values, conics, pts, 7, plan)
    for o in range(3):
        assert rel(outs[o], exp[o]) < 1e-5


def test_every_tile_mode_through_the_fused_launch(Sampler):
    """Thin outskirts (per-point walk), very wide Gaussians (record ranges), scattered points (group lists only):
    the fused launch must sample them all itself."""
    from tools.prof_step import list_stats
    rng = np.random.default_rng(12)
    cases = []
    m, con, v = random_gaussians(rng, 12000, 1, log_sigma_mean=-4.4, log_sigma_std=0.25)
    cases.append(("points", m, con, v, np.clip(rng.normal(0, 0.15, (60000, 2)), -1, 1), "points_tiles"))
    m, con, v = random_gaussians(rng, 1500, 1, log_sigma_mean=-1.2, log_sigma_std=0.3, lo=-0.5, hi=0.5)
    cases.append(("ranges", m, con, v, rng.uniform(-0.5, 0.5, (3000, 2)), "ranges_tiles"))
    for name, m, con, v, pts, key in cases:
        t = [dev32(a) for a in (m, v, con, pts)]
        s = Sampler(True, backend="binned", defer_lists=True)
        s.preprocess(t[0], t[1], None, t[2], t[3])
        outs = s.sample((0, 1, 2))
        assert list_stats(s._plan)[key] > 0, name
        exp = c_oracle.forward(m, con, v, pts, orders=(0, 1, 2))
        for o in range(3):
            assert rel(outs[o], exp[o]) < 1e-5, (name, o, rel(outs[o], exp[o]))


def test_fused_first_at_bench_size_equals_two_launches(Sampler):
    gs = synthetic.lattice_gaussians(256, 256, 0.5, seed=0)
    pts = synthetic.grid_samples(1024).float().cuda()
    t = {k: v.float().cuda() for k, v in gs.items()}
    outs = {}
    for fused in (True, False):
        with env("PIGS_NO_FUSED_FIRST", None if fused else "1"), torch.no_grad():
            s = Sampler(False, backend="binned", fuse="all", defer_lists=True)
            s.preprocess(t["means"], t["values"], None, t["conics"], pts)
            outs[fused] = [o.clone() for o in s.sample((0, 1, 2))]
            again = s.sample_gaussians()                     # cached
            assert again is not None
    for a, b in zip(outs[True], outs[False]):
        assert rel(a, b) < 2e-6
    idx = torch.arange(0, pts.shape[0], 331, device="cuda")[:2048]
    args = [gs[k].float().double().numpy() for k in ("means", "conics", "values")]
    exp = c_oracle.forward(*args, pts[idx].cpu().double().numpy(), orders=(0, 1, 2))
    for o in range(3):
        assert rel(outs[True][o][idx], exp[o]) < 1e-5

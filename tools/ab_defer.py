#!/usr/bin/env python3
"""Same-process A/B of GaussianSampler(defer_lists=...) at C3: cold step, warm step, sampler-only fwd+bwd step
(wall clock over 200 steps each, alternating; the quieter of the repeats is what to compare)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

kappa = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
lat = int(sys.argv[2]) if len(sys.argv) > 2 else 256
res = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
gs = synthetic.lattice_gaussians(lat, lat, kappa, seed=0)
t = {k: v.float().cuda() for k, v in gs.items()}
pts = synthetic.grid_samples(res).float().cuda()
req = {k: t[k].clone().requires_grad_(True) for k in ("means", "values", "conics")}
gout = []


def timed(fn, n=200):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e6
    gc.enable()
    return dt


def make(defer):
    cold = GaussianSampler(False, fuse="all", backend="binned", reuse_samples=False, defer_lists=defer)
    warm = GaussianSampler(False, fuse="all", backend="binned", defer_lists=defer)

    def cold_step():
        with torch.no_grad():
            cold.preprocess(t["means"], t["values"], None, t["conics"], pts)
            return cold.sample((0, 1, 2))

    def warm_step():
        with torch.no_grad():
            warm.preprocess(t["means"], t["values"], None, t["conics"], pts)
            return warm.sample((0, 1, 2))

    def fb_step():
        warm.preprocess(req["means"], req["values"], None, req["conics"], pts)
        outs = warm.sample((0, 1, 2))
        if not gout:
            gout.extend(torch.randn_like(o) for o in outs)
        return torch.autograd.grad(outs, list(req.values()), grad_outputs=gout)
    return cold_step, warm_step, fb_step


steps = {d: make(d) for d in (False, True)}
for rep in range(3):
    for d in (False, True):
        c, w, f = (timed(fn) for fn in steps[d])
        print(f"defer_lists={d!s:5}: cold {c:7.2f} us  warm {w:7.2f} us  fwd+bwd {f:7.2f} us", flush=True)

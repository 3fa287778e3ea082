#!/usr/bin/env python3
"""Host-side cost of the eager call sequence at PINN-loop sizes (cProfile over 3000 iterations)."""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

gs = synthetic.lattice_gaussians(20, 20, 1.3, seed=1)
t = {k: v.float().cuda() for k, v in gs.items()}
pts = (torch.rand((1024, 2)) * 2 - 1).cuda()
s = GaussianSampler(False)


def call():
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
    return s.sample_gaussians(), s.sample_gaussians_derivative(), s.sample_gaussians_laplacian()


with torch.no_grad():
  if len(sys.argv) <= 1 or sys.argv[1] != "c3":
    for _ in range(200):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000):
        call()
    torch.cuda.synchronize()
    print("no_grad eager: %.1f us/call" % ((time.perf_counter() - t0) / 3000 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3000):
        call()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)


def c3_training_step(iters=300):
    """eager sampler-only training step at C3 (binned, fused orders 0..2, backward): wall vs host time"""
    g3 = synthetic.lattice_gaussians(256, 256, 0.5, seed=0)
    p3 = synthetic.grid_samples(1024, 1024).float().cuda()
    req = {k: g3[k].float().cuda().requires_grad_(True) for k in ("means", "values", "conics")}
    s3 = GaussianSampler(False, fuse="all", backend="binned")
    gout = None

    def step():
        nonlocal gout
        s3.preprocess(req["means"], req["values"], None, req["conics"], p3)
        outs = s3.sample((0, 1, 2))
        if gout is None:
            gout = [torch.randn_like(o) for o in outs]
        torch.autograd.backward(outs, gout)
        for v in req.values():
            v.grad = None

    for _ in range(50):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("C3 eager fwd+bwd: %.1f us/step wall, %.1f us/step host issue time" % (t_all / iters * 1e6, t_host / iters * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(iters):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if len(sys.argv) > 1 and sys.argv[1] == "c3":
    c3_training_step()

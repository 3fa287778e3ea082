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

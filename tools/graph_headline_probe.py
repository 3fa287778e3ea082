#!/usr/bin/env python3
"""Headline step (preprocess + fused forward, C3) eager vs replayed from a hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from pigs_amd.graphs import GraphedStep
from diff_gaussian_sampling import GaussianSampler

kappa = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
gs = synthetic.lattice_gaussians(256, 256, kappa, seed=0)
t = {k: v.float().cuda() for k, v in gs.items()}
pts = synthetic.grid_samples(1024).float().cuda()
s = GaussianSampler(False, fuse="all", backend="binned")


def step(means, values, conics, samples):
    with torch.no_grad():
        s.preprocess(means, values, None, conics, samples)
        return s.sample((0, 1, 2))


def timed(f, n=300):
    for _ in range(30):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("eager  %.1f us/step" % timed(lambda: step(t["means"], t["values"], t["conics"], pts)))
g = GraphedStep(step, lambda: (t["means"].clone(), t["values"].clone(), t["conics"].clone(), pts.clone()))
print("graph  %.1f us/step" % timed(g))

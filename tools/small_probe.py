#!/usr/bin/env python3
"""Where the host time of the reference's training-step pattern goes (N = 1 600, 1 024 + 1 024 points, native host):
per-call issue times (perf_counter, no synchronisation inside the step), averaged over 2 000 steps."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

dev = torch.device("cuda")
gs = synthetic.lattice_gaussians(40, 40, 1.3, seed=1)
t = {k: v.float().to(dev) for k, v in gs.items()}
for k in ("means", "values", "conics"):
    t[k].requires_grad_(True)
gen = torch.Generator().manual_seed(3)
pts = (torch.rand((1024, 2), generator=gen) * 2 - 1).to(dev)
bc = (torch.rand((1024, 2), generator=gen) * 2 - 1).to(dev)
smp = GaussianSampler(False)
gouts = None
names = ["preprocess", "u", "du", "hess", "preprocess bc", "u bc", "autograd.grad"]
acc = [0.0] * len(names)


def step(record):
    global gouts
    ts = [time.perf_counter()]
    smp.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts); ts.append(time.perf_counter())
    u = smp.sample_gaussians(); ts.append(time.perf_counter())
    du = smp.sample_gaussians_derivative(); ts.append(time.perf_counter())
    h = smp.sample_gaussians_laplacian(); ts.append(time.perf_counter())
    smp.preprocess(t["means"], t["values"], t["covariances"], t["conics"], bc); ts.append(time.perf_counter())
    ub = smp.sample_gaussians(); ts.append(time.perf_counter())
    outs = [u, du, h, ub]
    if gouts is None:
        gouts = [torch.randn_like(o) for o in outs]
    torch.autograd.grad(outs, (t["means"], t["values"], t["conics"]), grad_outputs=gouts); ts.append(time.perf_counter())
    if record:
        for i in range(len(names)):
            acc[i] += ts[i + 1] - ts[i]


for _ in range(200):
    step(False)
torch.cuda.synchronize()
gc.disable()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    step(True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("step: %.1f us issue, %.1f us wall" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
for nm, a in zip(names, acc):
    print("  %-14s %6.1f us" % (nm, a / n * 1e6))

#!/usr/bin/env python3
"""Preprocess + fused forward on C3-sized inputs whose ORDER / distribution stresses the build's
atomics: argv[1] = grid | random | shuffled (grid points in random order) | clustered; argv[2] = kappa.
Run under rocprofv3 --kernel-trace --stats to see the per-kernel cost of each case."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

case, kappa = sys.argv[1], float(sys.argv[2])
gs = synthetic.lattice_gaussians(256, 256, kappa, seed=1)
t = {k: v.float().cuda() for k, v in gs.items()}
g = torch.Generator().manual_seed(3)
if case == "grid":
    pts = synthetic.grid_samples(1024).float()
elif case == "random":
    pts = torch.rand((1 << 20, 2), generator=g) * 2 - 1
elif case == "shuffled":
    pts = synthetic.grid_samples(1024).float()
    pts = pts[torch.randperm(pts.shape[0], generator=g)]
elif case == "clustered":
    pts = (torch.randn((1 << 20, 2), generator=g) * 0.15).clamp(-1, 1)
else:
    raise SystemExit("unknown case")
pts = pts.cuda()
s = GaussianSampler(False, backend="binned")


def step():
    s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
    return s.sample((0, 1, 2))


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(f"{case} kappa={kappa}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us/step", flush=True)

#!/usr/bin/env python3
"""Tile-list statistics (rows per tile, entries) by point order / distribution at C3 size: argv[1] = grid | random | shuffled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler
from tools.prof_step import list_stats

kappa = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
gs = synthetic.lattice_gaussians(256, 256, kappa, seed=1)
t = {k: v.float().cuda() for k, v in gs.items()}
g = torch.Generator().manual_seed(3)
for case in sys.argv[1].split(","):
    if case == "grid":
        pts = synthetic.grid_samples(1024).float()
    elif case == "random":
        pts = torch.rand((1 << 20, 2), generator=g) * 2 - 1
    elif case == "shuffled":
        pts = synthetic.grid_samples(1024).float()
        pts = pts[torch.randperm(pts.shape[0], generator=g)]
    else:
        pts = (torch.randn((1 << 20, 2), generator=g) * float(case.split(":")[1])).clamp(-1, 1)
    s = GaussianSampler(False, fuse="all", backend="binned", host="ctypes", defer_lists=False)
    with torch.no_grad():
        s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts.cuda())
    print(case, list_stats(s._plan), flush=True)

#!/usr/bin/env python3
"""What a strips build costs when its expectation fails: C3's Gaussians in NO order (every strip's box spans the
domain), strips forced (PIGS_GAUSS_STRIPS=1) against the cells (=0): cold step by HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

gs = synthetic.lattice_gaussians(256, 256, 0.5, seed=0)
perm = torch.randperm(gs["means"].shape[0], generator=torch.Generator().manual_seed(1))
t = {k: v[perm].float().cuda() for k, v in gs.items()}
pts = synthetic.grid_samples(1024).float().cuda()
for mode in ("0", "1"):
    os.environ["PIGS_GAUSS_STRIPS"] = mode
    s = GaussianSampler(False, fuse="all", backend="binned", reuse_samples=False)
    with torch.no_grad():
        for _ in range(3):
            s.preprocess(t["means"], t["values"], None, t["conics"], pts); s.sample((0, 1, 2))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            s.preprocess(t["means"], t["values"], None, t["conics"], pts); out = s.sample((0, 1, 2))
        e1.record(); torch.cuda.synchronize()
    print(f"PIGS_GAUSS_STRIPS={mode}: cold step {e0.elapsed_time(e1) / 5 * 1e3:.1f} us with 65 536 Gaussians in no order", flush=True)

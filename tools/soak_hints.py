#!/usr/bin/env python3
"""30 000 cold steps over 20 grid sizes (more than the library's 16 memory slots per kind), alternating with runs of one
size: the time per 5 000 steps and the allocated memory must not drift (pinned buffers, events and slots are reused)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler
gs = synthetic.lattice_gaussians(128, 128, 0.5, seed=0)
t = {k: v.float().cuda() for k, v in gs.items()}
sizes = [256, 264, 272, 280, 288, 296, 304, 312, 320, 328, 336, 344, 352, 360, 368, 376, 384, 392, 400, 408]   # 20 sizes > 16 hint slots
pts = {r: synthetic.grid_samples(r).float().cuda() for r in sizes}
s = GaussianSampler(False, fuse="all", backend="binned", reuse_samples=False)
t0 = time.perf_counter(); marks = []
with torch.no_grad():
    for it in range(30000):
        r = sizes[it % len(sizes)] if (it // 500) % 2 else sizes[0]
        s.preprocess(t["means"], t["values"], None, t["conics"], pts[r]); out = s.sample((0, 1, 2))
        if it % 5000 == 4999:
            torch.cuda.synchronize(); marks.append(time.perf_counter() - t0); t0 = time.perf_counter()
print("seconds per 5000 steps:", [round(m, 2) for m in marks], "finite", bool(torch.isfinite(out[2]).all()), "mem MB", torch.cuda.memory_allocated() // 2**20)

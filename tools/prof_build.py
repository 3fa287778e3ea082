#!/usr/bin/env python3
"""Time the preprocess (plan build) launches in isolation: python tools/prof_build.py [N M]...
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import importlib
importlib.import_module("pigs_amd.build").ensure_built()      # before anything touches the GPU; never builds behind rocprofv3
from pigs_amd import synthetic
from pigs_amd.sampler import GaussianSampler

cases = [(256, 1024), (1, 1024), (256, 4)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
for n, res in cases:
    gs = synthetic.lattice_gaussians(n, n, 0.5)
    pts = synthetic.grid_samples(res).float().cuda()
    t = {k: v.float().cuda() for k, v in gs.items()}
    s = GaussianSampler(False, backend="binned")
    for rnd in ("grid", "random"):
        p = pts if rnd == "grid" else pts[torch.randperm(pts.shape[0], device="cuda")].contiguous()
        for _ in range(3):
            s.preprocess(t["means"], t["values"], None, t["conics"], p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            s.preprocess(t["means"], t["values"], None, t["conics"], p)
        torch.cuda.synchronize()
        print(f"N={n*n} M={res*res} {rnd}: preprocess {(time.perf_counter()-t0)/20*1e6:.1f} us", flush=True)

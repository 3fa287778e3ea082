#!/usr/bin/env python3
"""Why did profiles/r03_aggregate.txt show the float32 neighbour-list build at N = 65 536 at 1 238 us with the
counting pass and the float64 one at 340?  Times preprocess_aggregate (counting pass + read-back + lists) for both
dtypes in both orders, each with a fresh sampler, wall clock around synchronised calls (what a caller sees) and HIP
events (what tools/bench_aggregate.py reports), with and without a binned plan bound by the preceding preprocess."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import importlib
importlib.import_module("pigs_amd.build").ensure_built()
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

gs = synthetic.lattice_gaussians(256, 256, 0.5, seed=2)


def run(dtype, backend):
    means, conics, values = (gs[k].to(dtype).cuda() for k in ("means", "conics", "values"))
    s = GaussianSampler(False, unpinned_aggregate=True, backend=backend)
    s.preprocess(means, values, None, conics, means)
    for _ in range(5):
        s.preprocess_aggregate()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        s.preprocess_aggregate()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        s.preprocess_aggregate()
    e1.record()
    torch.cuda.synchronize()
    return wall, e0.elapsed_time(e1) / 20 * 1e3, s._neighbors.cap


for order in ((torch.float32, torch.float64), (torch.float64, torch.float32)):
    for dtype in order:
        for backend in ("auto", "dense"):
            w, e, cap = run(dtype, backend)
            print(f"{str(dtype)[6:]:8s} preprocess backend={backend:5s}: preprocess_aggregate wall {w:8.1f} us, events {e:8.1f} us (cap {cap})", flush=True)

#!/usr/bin/env python3
"""Per-rank step time of bench.py's weak-scaling shards, emulated on one GPU (no process group):
for world = 1, 2, 4, 8 the step of the first, a middle and the last rank."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

kappa = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
gs = synthetic.lattice_gaussians(256, 256, kappa, seed=0)
t = {k: v.float().cuda() for k, v in gs.items()}
s = GaussianSampler(False, fuse="all", backend="binned")


def timed(pts):
    def step():
        s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
        return s.sample((0, 1, 2))
    for _ in range(5):
        step()
    best = 1e30
    for rep in range(5):                 # min of 5 x 20 steps: allocator stalls out of the way
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
    return best


for world in (1, 2, 4, 8):
    for layout in ("band", "square"):
        if layout == "band":
            rx, ry, rows = 1024, 1024 * world, 1024
        else:
            rx = ry = int(round(1024 * math.sqrt(world)))
            rows = ry // world
        out = []
        for rank in range(world):
            pts = synthetic.grid_samples(rx, ry, row0=rank * rows, rows=rows).float().cuda()
            out.append(f"{timed(pts):5.1f}")
        print(f"world {world} {layout:6s} grid {rx}x{ry} rows/rank {rows}: us/step per rank: " + " ".join(out), flush=True)

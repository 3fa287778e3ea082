#!/usr/bin/env python3
"""One rank's step of bench.py's weak-scaling workload, alone on the GPU: rank r of `world` samples ITS rows of the
side x side grid over [-1, 1]^2 with all 65 536 Gaussians (most of which lie outside its strip of the domain).
argv: world [rank]."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else world // 2
side = max(1, int(round(1024 * math.sqrt(world) / (8 * world)))) * 8 * world
rows = side // world
gs = synthetic.lattice_gaussians(256, 256, 0.5, seed=0)
t = {k: v.float().cuda() for k, v in gs.items()}
pts = synthetic.grid_samples(side, side, row0=rank * rows, rows=rows).float().cuda()
for reuse, name in ((False, "cold"), (True, "warm")):
    s = GaussianSampler(False, fuse="all", backend="binned", reuse_samples=reuse)
    with torch.no_grad():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            s.preprocess(t["means"], t["values"], None, t["conics"], pts); s.sample((0, 1, 2))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            s.preprocess(t["means"], t["values"], None, t["conics"], pts); out = s.sample((0, 1, 2))
        torch.cuda.synchronize()
    print(f"world {world} rank {rank}: {side} x {rows} points, {name} step {(time.perf_counter() - t0) / 200 * 1e6:.1f} us, finite {bool(torch.isfinite(out[2]).all())}", flush=True)

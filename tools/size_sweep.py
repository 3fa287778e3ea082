#!/usr/bin/env python3
"""Throughput of the fused forward (orders 0-2) over grid sizes, 65 536 Gaussians, kappa 0.5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from pigs_amd import sampler as S
from diff_gaussian_sampling import GaussianSampler

gs = synthetic.lattice_gaussians(256, 256, 0.5)
t = {k: v.float().cuda() for k, v in gs.items()}
NSTEP = 50
for res in ([int(x) for x in sys.argv[1:]] or (512, 1024, 2048, 4096)):
    pts = synthetic.grid_samples(res).float().cuda()
    s = GaussianSampler(False, fuse="all", reuse_samples=False)        # cold steps, as the bench's headline
    with torch.no_grad():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:      # untimed pre-heat, as bench.py: a short run must not time a ramping chip
            s.preprocess(t["means"], t["values"], None, t["conics"], pts); out = s.sample((0, 1, 2))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(NSTEP):
            s.preprocess(t["means"], t["values"], None, t["conics"], pts); out = s.sample((0, 1, 2))
        torch.cuda.synchronize(); step = (time.perf_counter() - t0) / NSTEP
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m, v, c, sm = s._inputs
        e0.record()
        for _ in range(NSTEP):
            S.forward_raw(m, v, c, sm, 7, s._plan)
        e1.record(); torch.cuda.synchronize()
    k = e0.elapsed_time(e1) / NSTEP * 1e-3
    M = res * res
    print(f"{res}x{res}: step {step*1e6:8.1f} us  {M/step:.3e} pts/s | kernel {k*1e6:8.1f} us  {(24*65536+36*M)/k/8e12*100:5.1f} % of 8 TB/s | finite {bool(torch.isfinite(out[2]).all())}", flush=True)
    del pts, s, out

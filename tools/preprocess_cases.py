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
LAT = int(os.environ.get("PIGS_CASE_LAT", "256"))          # Gaussians: LAT x LAT lattice (C3: 256)
RES = int(os.environ.get("PIGS_CASE_RES", "1024"))         # points: RES x RES (C3: 1024)
gs = synthetic.lattice_gaussians(LAT, LAT, kappa, seed=1)
t = {k: v.float().cuda() for k, v in gs.items()}
g = torch.Generator().manual_seed(3)
if case == "grid":
    pts = synthetic.grid_samples(RES).float()
elif case == "random":
    pts = torch.rand((RES * RES, 2), generator=g) * 2 - 1
elif case == "shuffled":
    pts = synthetic.grid_samples(RES).float()
    pts = pts[torch.randperm(pts.shape[0], generator=g)]
elif case.startswith("clustered"):       # clustered, or clustered:<sigma> (test_no_mlp.py:86 draws randn / 2, clamped)
    sigma = float(case.split(":")[1]) if ":" in case else 0.15
    pts = (torch.randn((RES * RES, 2), generator=g) * sigma).clamp(-1, 1)
else:
    raise SystemExit("unknown case")
pts = pts.cuda()
res = {}
with torch.no_grad():
    only = sys.argv[3] if len(sys.argv) > 3 else None          # "cold" / "warm": that loop only (per-kernel profiles)
    for name, reuse in (("cold", False), ("warm", True)):
        if only and name != only:
            res[name] = float("nan")
            continue
        s = GaussianSampler(False, fuse="all", backend="binned", reuse_samples=reuse)

        def step():
            s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
            return s.sample((0, 1, 2))

        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:          # untimed pre-heat, as bench.py
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 100 * 1e6
    from pigs_amd import sampler as S
    m, v, c, sm = s._inputs
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        S.forward_raw(m, v, c, sm, 7, s._plan)
    e1.record(); torch.cuda.synchronize()
    M = pts.shape[0]
    go = [torch.randn((M,) + (2,) * k + (1,), device="cuda") for k in range(3)] + [None, None]
    for _ in range(3):
        S.backward_raw(m, v, c, sm, go, 7, s._plan)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record()
    for _ in range(20):
        S.backward_raw(m, v, c, sm, go, 7, s._plan)
    b1.record(); torch.cuda.synchronize()
print(f"{case:9s} kappa={kappa}: cold {res['cold']:7.1f} us/step  warm {res['warm']:7.1f} us/step  forward kernel {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us"
      f"  backward kernels {b0.elapsed_time(b1) / 20 * 1e3:7.1f} us", flush=True)

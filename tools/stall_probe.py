"""Looks for isolated long steps: K cold steps, each timed on its own (synchronised), at one kappa."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler
kappa = float(sys.argv[1]) if len(sys.argv) > 1 else 1.3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
if len(sys.argv) > 3 and sys.argv[3] == "nogc":
    import gc
    gc.collect(); gc.freeze(); gc.disable()
gs = synthetic.lattice_gaussians(256, 256, kappa, seed=0)
t = {k: v.float().cuda() for k, v in gs.items()}
pts = synthetic.grid_samples(1024).float().cuda()
s = GaussianSampler(False, fuse="all", backend="binned", reuse_samples=False)
ts = np.zeros(K)
with torch.no_grad():
    for k in range(K):
        t0 = time.perf_counter()
        s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
        s.sample((0, 1, 2))
        torch.cuda.synchronize()
        ts[k] = time.perf_counter() - t0
us = ts * 1e6
print(f"kappa={kappa}: median {np.median(us):.1f} us, p99 {np.percentile(us, 99):.1f}, max {us.max():.1f} at step {us.argmax()}, steps over 1 ms: {np.nonzero(us > 1000)[0].tolist()[:20]}, over 3x median: {[(int(k), round(float(us[k]))) for k in np.nonzero(us > 3 * np.median(us))[0][:20]]}")

#!/usr/bin/env python3
"""Kernel-only times (HIP events, back-to-back launches on one plan) of the binned forward / backward at C3:
argv: [kappa] [q_max_backward].  With PIGS_AMD_LIB=<variant> PIGS_AMD_HOST=ctypes it times a variant build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic, sampler as S
from diff_gaussian_sampling import GaussianSampler

kappa = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
kw = {"q_max_backward": float(sys.argv[2])} if len(sys.argv) > 2 else {}
gs = synthetic.lattice_gaussians(256, 256, kappa, seed=0)
pts = synthetic.grid_samples(1024).float().cuda()
t = {k: v.float().cuda() for k, v in gs.items()}
s = GaussianSampler(False, fuse="all", backend="binned", **kw)
with torch.no_grad():
    s.preprocess(t["means"], t["values"], None, t["conics"], pts)
    m, v, c, sm = s._inputs
    plan = s._plan
    M = pts.shape[0]
    go = [torch.randn((M,) + (2,) * k + (1,), device="cuda") for k in range(3)] + [None, None]

    def timed(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    f = timed(lambda: S.forward_raw(m, v, c, sm, 7, plan))
    b = timed(lambda: S.backward_raw(m, v, c, sm, go, 7, plan))
    b1 = timed(lambda: S.backward_raw(m, v, c, sm, [go[0], None, None, None, None], 1, plan))
print(f"kappa={kappa} q_b={s.q_max_backward} lib={os.environ.get('PIGS_AMD_LIB', 'default')}: forward {f:.1f} us, backward(0..2)+unpermute {b:.1f} us, backward(order 0) {b1:.1f} us")

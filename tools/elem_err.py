"""Element-wise relative error of the forward (HIP float32 vs float64 oracle on the same inputs) on the elements
above a fraction of the tensor's maximum: what an element-wise bar can hold (tests/test_binned_gpu.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler
for cfg, kappa in (("c2", 0.5), ("c2", 1.3), ("c3", 0.5)):
    gs, pts = synthetic.CONFIGS[cfg](kappa)
    t = {k: v.float().cuda() for k, v in gs.items()}
    pts = pts.float().cuda()
    for backend in ("binned", "dense"):
        if cfg == "c3" and backend == "dense": continue
        s = GaussianSampler(False, fuse="all", backend=backend)
        with torch.no_grad():
            s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
            outs = s.sample((0, 1, 2))
        idx = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(3))[:4096].cuda()
        args = [t[k].cpu().double().numpy() for k in ("means", "conics", "values")]
        exp = c_oracle.forward(*args, pts[idx].cpu().double().numpy(), orders=(0, 1, 2))
        for o, out in enumerate(outs):
            got = out[idx].cpu().double().numpy(); want = exp[o]
            mx = np.abs(want).max()
            for frac in (1e-1, 1e-2, 1e-3):
                sel = np.abs(want) > frac * mx
                e = np.abs(got - want)[sel] / np.abs(want)[sel]
                print(cfg, kappa, backend, "order", o, "frac", frac, "n", int(sel.sum()), "max elem rel %.2e" % e.max(), "p99 %.2e" % np.quantile(e, 0.99), "global %.2e" % (np.abs(got - want).max() / mx))

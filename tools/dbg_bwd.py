import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pigs_amd.sampler import GaussianSampler
rng = np.random.default_rng(0)
N, M = 40, 256
means = rng.uniform(-1, 1, (N, 2)); s0 = np.exp(2*rng.normal(-2.0, 0.3, (N, 2)))
con = np.stack((1/s0[:,0], np.zeros(N), 1/s0[:,1]), -1); values = rng.uniform(0.5, 1, (N, 1))
samples = rng.uniform(-1, 1, (M, 2))
res = {}
for backend in ("dense", "binned"):
    t = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=(i < 3)) for i, a in enumerate((means, values, con, samples))]
    s = GaussianSampler(True, backend=backend, fuse="none")
    s.preprocess(t[0], t[1], None, t[2], t[3])
    u = s.sample_gaussians()
    u.sum().backward()
    res[backend] = [x.grad.cpu().numpy() for x in t[:3]]
for k, name in enumerate(("means", "values", "conics")):
    d, b = res["dense"][k], res["binned"][k]
    print(name, "max|dense|", np.abs(d).max(), "max err", np.abs(d - b).max())
    print("  ratio (first 6 rows)", (b[:6] / d[:6]).round(3).tolist())

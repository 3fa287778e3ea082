#!/usr/bin/env python3
"""aggregate_neighbors forward + backward under rocprofv3 (--kernel-trace --stats): argv = side kappa [f64]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import importlib
importlib.import_module("pigs_amd.build").ensure_built()      # before anything touches the GPU; never builds behind rocprofv3
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

side = int(sys.argv[1]) if len(sys.argv) > 1 else 40
kappa = float(sys.argv[2]) if len(sys.argv) > 2 else 1.3
dtype = torch.float64 if "f64" in sys.argv else torch.float32
N, L, K, F = side * side, 16, 16, 6
E = 4 * F + 1
gs = synthetic.lattice_gaussians(side, side, kappa, seed=2)
means, conics, values = (gs[k].to(dtype).cuda() for k in ("means", "conics", "values"))
g = torch.Generator(device="cpu").manual_seed(5)
mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(dtype).cuda().requires_grad_(True)
args = [mk(N, L), mk(L, L), mk(N, K), mk(N, K), mk(F), mk(L, 2 * E)]
s = GaussianSampler(False, unpinned_aggregate=True)
s.preprocess(means, values, None, conics, means)
gout = torch.randn((N, L), dtype=dtype, device="cuda")
for _ in range(30):
    s.preprocess_aggregate()
    out = s.aggregate_neighbors(*args)
    torch.autograd.grad(out, args, grad_outputs=gout)
torch.cuda.synchronize()

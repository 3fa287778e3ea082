#!/usr/bin/env python3
"""Latency of the sampler at the sizes the reference's PINN loops use (model_pn.py:768-772:
N ~ 1e3 Gaussians, 1024 collocation points): preprocess + u, grad u, Hessian (+ backward)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

for n, M in ((20, 1024), (40, 1024), (40, 4096), (90, 65536)):
    gs = synthetic.lattice_gaussians(n, n, 1.3, seed=1)
    t = {k: v.float().cuda() for k, v in gs.items()}
    for k in ("means", "values", "conics"):
        t[k].requires_grad_(True)
    pts = ((torch.rand((M, 2)) * 2 - 1)).cuda()
    s = GaussianSampler(False)

    def fwd():
        s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
        return s.sample_gaussians(), s.sample_gaussians_derivative(), s.sample_gaussians_laplacian()

    def step():
        u, ux, uxx = fwd()
        loss = (u ** 2).mean() + (ux ** 2).mean() + (uxx ** 2).mean()
        return torch.autograd.grad(loss, [t["means"], t["values"], t["conics"]])

    for name, f in (("fwd(0..2)", fwd), ("fwd+loss+bwd", step)):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            f()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 50 * 1e6
        # the same call captured once into a hipGraph (no entry point allocates or synchronises)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            f()
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            graph.replay()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / 50 * 1e6
        print(f"N={n*n:5d} M={M:6d} {name:>14}: eager {eager:8.1f} us/call   hipGraph replay {replay:8.1f} us/call"
              f"  (plan: {s._plan is not None})", flush=True)

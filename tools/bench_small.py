#!/usr/bin/env python3
"""Latency of the sampler at the sizes the reference's PINN loops use (model_pn.py:768-772:
N ~ 1e3 Gaussians, 1024 collocation points): preprocess + u, grad u, Hessian (+ backward),
launched eagerly and replayed from a hipGraph captured once (no entry point allocates or
synchronises, so the whole step is capturable)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler
import gc
gc.collect(); gc.freeze()      # a full collection of a torch process takes 35-45 ms: keep it out of the short timed loops


def make_case(n, M):
    """Fresh leaves + sampler, created under the CURRENT stream (autograd remembers the stream a
    leaf's accumulation node was first used on; a capture must not meet one from another stream)."""
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

    return s, {"fwd(0..2)": fwd, "fwd+loss+bwd": step}


def timed(f, reps=50):
    for _ in range(20):      # the caching allocator needs a few calls to settle after a graph pool is freed
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def compare(n, M, name, modes=("eager", "graph")):
    out = {}
    if "eager" in modes:
        s, fs = make_case(n, M)
        out["eager"] = timed(fs[name])
    if "graph" in modes:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            s2, fs2 = make_case(n, M)
            for _ in range(3):
                fs2[name]()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            keep = fs2[name]()
        out["graph"] = timed(graph.replay)
        out["plan"] = s2._plan is not None
        del graph, keep
    return out


if __name__ == "__main__":
    for n, M in ((20, 1024), (40, 1024), (40, 4096), (90, 65536)):
        for name in ("fwd(0..2)", "fwd+loss+bwd"):
            r = compare(n, M, name)
            print(f"N={n*n:5d} M={M:6d} {name:>14}: eager {r['eager']:8.1f} us/call   hipGraph replay "
                  f"{r['graph']:8.1f} us/call  (plan: {r['plan']})", flush=True)

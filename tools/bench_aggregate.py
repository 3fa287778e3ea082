#!/usr/bin/env python3
"""Timing of preprocess_aggregate / aggregate_neighbors (this repo's definition: DESIGN.md 9) at the
model's sizes (model_pn.py:44-49: L = K = 16, F = 6, E = 25) and beyond: HIP events around 50 calls."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler


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


for dtype in (torch.float32, torch.float64):
    for side, kappa in ((40, 1.3), (128, 1.3), (256, 0.5)):
        N, L, K, F = side * side, 16, 16, 6
        E = 4 * F + 1
        gs = synthetic.lattice_gaussians(side, side, kappa, seed=2)
        means, conics = gs["means"].to(dtype).cuda(), gs["conics"].to(dtype).cuda()
        values = gs["values"].to(dtype).cuda()
        g = torch.Generator(device="cpu").manual_seed(5)
        mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(dtype).cuda().requires_grad_(True)
        args = [mk(N, L), mk(L, L), mk(N, K), mk(N, K), mk(F), mk(L, 2 * E)]
        s = GaussianSampler(False, unpinned_aggregate=True)
        s.preprocess(means, values, None, conics, means)
        t_lists = timed(s.preprocess_aggregate)
        nb = s._neighbors
        s2 = GaussianSampler(False, unpinned_aggregate=True, aggregate_cap=nb.cap)      # slab size given: one pass, no read-back
        s2.preprocess(means, values, None, conics, means)
        t_lists_cap = timed(s2.preprocess_aggregate)
        pairs = int(nb.row_counts.sum())
        with torch.no_grad():
            t_fwd = timed(lambda: s.aggregate_neighbors(*args))
        gout = torch.randn((N, L), dtype=dtype, device="cuda")

        def fb():
            out = s.aggregate_neighbors(*args)
            torch.autograd.grad(out, args, grad_outputs=gout)
        t_fb = timed(fb)
        print(f"{str(dtype)[6:]:8s} N={N:6d} kappa={kappa}: {pairs / N:6.1f} neighbours per Gaussian (cap {nb.cap}) | lists {t_lists:7.1f} us (cap given: {t_lists_cap:7.1f}) | "
              f"forward {t_fwd:7.1f} us | forward + backward (all six gradients) {t_fb:7.1f} us", flush=True)

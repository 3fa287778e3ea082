#!/usr/bin/env python3
"""Fused covariance builder (pigs_amd.covariances, one HIP launch each way) against the same
function written as the chain of stock torch ops the reference issues (gaussians.py:163-189:
tanh, prod, sqrt, diag_embed, two index writes, batched inverse, two gathers), forward + backward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import covariances


def torch_chain(s, t):
    tau = torch.tanh(t) * s.prod(-1).sqrt().unsqueeze(-1)
    S = torch.diag_embed(s)
    S[..., 1, 0] = tau[..., 0]
    S[..., 0, 1] = tau[..., 0]
    C = torch.inverse(S)
    pick = [0, 1, 3]
    return S.reshape(-1, 4)[:, pick], C.reshape(-1, 4)[:, pick]


def timed(f, reps=50):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for N in (400, 1600, 65536, 1 << 20):
    g = torch.Generator().manual_seed(0)
    s = torch.exp(torch.randn((N, 2), generator=g) - 4).cuda().requires_grad_(True)
    t = torch.randn((N, 1), generator=g).cuda().requires_grad_(True)
    r1, r2 = torch.rand((N, 3), device="cuda"), torch.rand((N, 3), device="cuda")
    res = {}
    for name, fn in (("fused", covariances.build_covariances), ("torch chain", torch_chain)):
        def fwd():
            with torch.no_grad():
                return fn(s, t)

        def fwd_bwd():
            cov, con = fn(s, t)
            return torch.autograd.grad((cov * r1).sum() + (con * r2).sum(), (s, t))
        res[name] = (timed(fwd), timed(fwd_bwd))
    a, b = fn(s, t), covariances.build_covariances(s, t)
    err = max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(a, b))
    print(f"N={N:8d}: fused fwd {res['fused'][0]:7.1f} us, fwd+bwd {res['fused'][1]:7.1f} us | torch chain fwd "
          f"{res['torch chain'][0]:7.1f} us, fwd+bwd {res['torch chain'][1]:7.1f} us | max rel diff {err:.1e}", flush=True)

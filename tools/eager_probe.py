"""Where the eager sampler-only training step spends its host time: variants of the call sequence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler
g3 = synthetic.lattice_gaussians(256, 256, 0.5, seed=0)
p3 = synthetic.grid_samples(1024, 1024).float().cuda()
t = {k: v.float().cuda() for k, v in g3.items()}
req = {k: t[k].clone().requires_grad_(True) for k in ("means", "values", "conics")}
gouts = None


def make(cov, use_grad, fresh_sampler=False):
    s3 = GaussianSampler(False, fuse="all", backend="binned")

    def step():
        global gouts
        s3.preprocess(req["means"], req["values"], t["covariances"] if cov else None, req["conics"], p3)
        outs = s3.sample((0, 1, 2))
        if gouts is None:
            gouts = tuple(torch.randn_like(o) for o in outs)
        if use_grad:
            return torch.autograd.grad(outs, list(req.values()), grad_outputs=gouts)
        torch.autograd.backward(outs, gouts)
        for v in req.values():
            v.grad = None
    return step


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); ta = time.perf_counter() - t0
    return "wall %.1f us, host %.1f us" % (ta / n * 1e6, th / n * 1e6)


for cov in (False, True):
    for use_grad in (False, True):
        print("covariances" if cov else "no covariances", "| autograd.grad" if use_grad else "| autograd.backward", "|", timeit(make(cov, use_grad)), flush=True)

# the same step after what bench.py does in front of it: 0.25 s of cold steps, 200 cold + 200 warm steps
print("--- after bench.py's preamble", flush=True)
cold = GaussianSampler(False, fuse="all", backend="binned", reuse_samples=False)
warm = GaussianSampler(False, fuse="all", backend="binned", reuse_samples=True)
with torch.no_grad():
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        cold.preprocess(t["means"], t["values"], t["covariances"], t["conics"], p3); cold.sample((0, 1, 2))
    for smp in (cold, warm):
        for _ in range(210):
            smp.preprocess(t["means"], t["values"], t["covariances"], t["conics"], p3); smp.sample((0, 1, 2))
    torch.cuda.synchronize()


def step_w():
    global gouts
    warm.preprocess(req["means"], req["values"], t["covariances"], req["conics"], p3)
    outs = warm.sample((0, 1, 2))
    return torch.autograd.grad(outs, list(req.values()), grad_outputs=gouts)


print("warm sampler of the preamble | autograd.grad |", timeit(step_w, 100), flush=True)
print("fresh sampler               | autograd.grad |", timeit(make(True, True), 100), flush=True)
import gc
gc.disable()
print("fresh sampler, gc disabled   | autograd.grad |", timeit(make(True, True), 100), flush=True)

#!/usr/bin/env python3
"""Does the latency-bound plan build of step k+1 hide under the VALU-bound forward of step k?
Steps (preprocess + fused forward, C3) issued round-robin on S streams, one sampler per stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigs_amd import synthetic
from diff_gaussian_sampling import GaussianSampler

kappa = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
gs = synthetic.lattice_gaussians(256, 256, kappa, seed=0)
t = {k: v.float().cuda() for k, v in gs.items()}
pts = synthetic.grid_samples(1024).float().cuda()
for S in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(S)]
    samplers = [GaussianSampler(False, fuse="all", backend="binned") for _ in range(S)]
    outs = [None] * S

    def run(n):
        for i in range(n):
            k = i % S
            with torch.cuda.stream(streams[k]):
                samplers[k].preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
                outs[k] = samplers[k].sample((0, 1, 2))

    with torch.no_grad():
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        run(6)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            run(30)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 30 * 1e6)
    print(f"kappa {kappa} streams {S}: {best:.1f} us/step", flush=True)

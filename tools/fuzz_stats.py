#!/usr/bin/env python3
"""Error of the binned path against the dense HIP path over many fuzz seeds, per output order and
per gradient, for several cut-offs q_max (argv: n_seeds q_max...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_fuzz_gpu import make_case
from diff_gaussian_sampling import GaussianSampler

n = int(sys.argv[1])
qs = [float(x) for x in sys.argv[2:]]
worst = {q: np.zeros(7) for q in qs}
count = {q: np.zeros(7) for q in qs}
for seed in range(n):
    rng = np.random.default_rng(1000 + seed)
    means, values, con, pts = make_case(rng)
    res = {}
    for key in ["dense"] + qs:
        t = [torch.tensor(a, dtype=torch.float32, device="cuda") for a in (means, values, con, pts)]
        for x in t[:3]:
            x.requires_grad_(True)
        s = GaussianSampler(False, backend="dense" if key == "dense" else "binned", fuse="all",
                            q_max=36.0 if key == "dense" else key)
        s.preprocess(t[0], t[1], None, t[2], t[3])
        o = s.sample((0, 1, 2, 3))
        torch.manual_seed(seed)
        loss = sum((x * torch.randn_like(x)).sum() for x in o)
        loss.backward()
        res[key] = [x.detach() for x in o] + [x.grad for x in t[:3]]
    for q in qs:
        for k, (a, b) in enumerate(zip(res["dense"], res[q])):
            e = float((a - b).abs().max() / a.abs().max())
            worst[q][k] = max(worst[q][k], e)
            count[q][k] += e > (1e-5 if k < 4 else 5e-5)
for q in qs:
    print("q_max %5.1f worst rel err: out0..3 %s | grads m,v,c %s | seeds over the bar: %s" % (
        q, " ".join("%.1e" % x for x in worst[q][:4]), " ".join("%.1e" % x for x in worst[q][4:]), count[q].astype(int)), flush=True)

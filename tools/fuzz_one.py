#!/usr/bin/env python3
"""Re-run one seed of tests/test_fuzz_gpu.py and print per-output errors (argv: seeds...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_fuzz_gpu import make_case
from diff_gaussian_sampling import GaussianSampler

for seed in map(int, [a for a in sys.argv[1:] if not a.startswith("--")]):
    rng = np.random.default_rng(1000 + seed)
    means, values, con, pts = make_case(rng)
    orders = (0, 1, "lap") if seed % 3 == 2 else (0, 1, 2, 3)
    print("seed", seed, "N", means.shape[0], "M", pts.shape[0], "c", values.shape[1], "orders", orders,
          "pts span", np.ptp(pts, axis=0), "means span", np.ptp(means, axis=0))
    outs, grads = {}, {}
    for backend in ("dense", "binned"):
        t = [torch.tensor(a, dtype=torch.float32, device="cuda") for a in (means, values, con, pts)]
        for x in t[:3]:
            x.requires_grad_(True)
        s = GaussianSampler(True, backend=backend, fuse="all")
        s.preprocess(t[0], t[1], None, t[2], t[3])
        o = s.sample(orders)
        torch.manual_seed(seed)
        loss = sum((x * torch.randn_like(x)).sum() for x in o)
        loss.backward()
        outs[backend] = [x.detach() for x in o]
        grads[backend] = [x.grad for x in t[:3]]
    for k, (a, b) in enumerate(zip(outs["dense"], outs["binned"])):
        d = (a - b).abs()
        i = int(d.reshape(d.shape[0], -1).max(1).values.argmax())
        print("  out", k, "scale %.3e maxdiff %.3e rel %.2e at point %d" % (float(a.abs().max()), float(d.max()), float(d.max() / a.abs().max()), i),
              "finite", bool(torch.isfinite(b).all()), "pt", pts[i])
    for k, (a, b) in enumerate(zip(grads["dense"], grads["binned"])):
        print("  grad", k, "rel %.2e" % float((a - b).abs().max() / a.abs().max()), "dense finite", bool(torch.isfinite(a).all()),
              "binned finite", bool(torch.isfinite(b).all()), "dense max %.3e" % float(a[torch.isfinite(a)].abs().max()))
    # which of the two float32 paths is nearer the float64 oracle (accumulation order vs something real)?
    if "--oracle" in sys.argv:
        from oracle import c_oracle          # diagnostic only: this tool is not part of the product path
        import torch as _t
        _t.manual_seed(seed)
        gouts = {}
        for o, x in zip(orders, outs["dense"]):
            gouts[o if o != "lap" else "lap"] = _t.randn_like(x).cpu().double().numpy()
        if "lap" not in orders:
            r32 = [np.asarray(a, dtype=np.float32).astype(np.float64) for a in (means, con, values, pts)]     # what the GPU paths were given
            gm, gc, gv = c_oracle.backward(*r32, {int(o): gouts[o] for o in orders})
            for name, e, k in (("means", gm, 0), ("values", gv, 1), ("conics", gc, 2)):
                for backend in ("dense", "binned"):
                    g = grads[backend][k].cpu().double().numpy()
                    print("  grad", name, backend, "vs oracle f64: rel %.2e" % (np.abs(g - e).max() / np.abs(e).max()))

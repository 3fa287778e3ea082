#!/usr/bin/env python3
"""Campaign: the dense HIP path (float32 and float64) against the fp64 C oracle on the adversarial
cases of tests/test_fuzz_gpu.py, shrunk so that the oracle stays fast (argv: n_seeds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_fuzz_gpu import make_case
from oracle import c_oracle
from diff_gaussian_sampling import GaussianSampler

n = int(sys.argv[1])
worst = {torch.float32: np.zeros(7), torch.float64: np.zeros(7)}
bad = []
for seed in range(n):
    rng = np.random.default_rng(5000 + seed)
    means, values, con, pts = make_case(rng)
    means, values, con, pts = means[:300], values[:300], con[:300], pts[:500]
    lam = (con[:, 0] + con[:, 2]) / 2 + np.sqrt(((con[:, 0] - con[:, 2]) / 2) ** 2 + con[:, 1] ** 2)
    term = [float((np.abs(values).max(1) * lam ** (k / 2)).max()) for k in range(4)]
    for dtype in (torch.float32, torch.float64):
        t = [torch.tensor(a, dtype=dtype, device="cuda") for a in (means, values, con, pts)]
        args = [x.cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]     # the rounded inputs
        for x in t[:3]:
            x.requires_grad_(True)
        s = GaussianSampler(False, backend="dense", fuse="all")
        s.preprocess(t[0], t[1], None, t[2], t[3])
        o = s.sample((0, 1, 2, 3))
        exp = c_oracle.forward(*args, orders=(0, 1, 2, 3))
        g = np.random.default_rng(seed)
        rs = [g.uniform(-1, 1, e.shape) for e in (exp[0], exp[1], exp[2], exp[3])]
        loss = sum((x * torch.tensor(r, dtype=dtype, device="cuda")).sum() for x, r in zip(o, rs))
        loss.backward()
        em, ec, ev = c_oracle.backward(*args, {k: torch.tensor(r, dtype=dtype).double().numpy() for k, r in enumerate(rs)})
        errs = []
        under = 1.0
        for k in range(4):
            a = o[k].detach().cpu().double().numpy()
            top = np.abs(exp[k]).max()
            errs.append(np.abs(a - exp[k]).max() / max(top, term[k], 1e-300))
            under = max(under, term[k] / max(top, 1e-300))
        for a, e in ((t[0].grad, em), (t[1].grad, ev), (t[2].grad, ec)):
            a = a.cpu().double().numpy()
            errs.append(np.abs(a - e).max() / max(np.abs(e).max() * under, 1e-300) if np.isfinite(a).all() else np.inf)
        errs = np.array(errs)
        worst[dtype] = np.maximum(worst[dtype], errs)
        bar = 1e-5 if dtype == torch.float32 else 1e-11
        if (errs > np.array([bar] * 4 + [5 * bar] * 3)).any():
            bad.append((seed, str(dtype), ["%.1e" % e for e in errs]))
for dtype, w in worst.items():
    print(dtype, "worst: out0..3", " ".join("%.1e" % x for x in w[:4]), "| grads m,v,c", " ".join("%.1e" % x for x in w[4:]))
print("over the bar:", len(bad), bad[:8])
import json
print("BAD_SEEDS_F32", json.dumps([b[0] for b in bad if "float32" in b[1]]))
print("BAD_F32_WORST", json.dumps({str(b[0]): b[2] for b in bad if "float32" in b[1]}))

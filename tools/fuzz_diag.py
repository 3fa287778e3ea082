#!/usr/bin/env python3
"""Which mechanism is behind the binned path's larger conic-gradient error in the fuzz_big cases:
the same case through dense, binned(q_max = 36; backward cut-off 36 / 40 / 44) and binned(q_max = 60) against the float64 oracle on the
Gaussians where binned and dense differ most.  argv: case index (seed 1 of tools/fuzz_big.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diff_gaussian_sampling import GaussianSampler
from oracle import c_oracle
from tools.fuzz_big import gen_cases

k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
_, kind, means, con, values, pts, orders = next(cs for cs in gen_cases(32, 1) if cs[0] == k)
N, M, c = means.shape[0], pts.shape[0], values.shape[1]
rng = np.random.default_rng(100 + k)
f32 = [a.astype(np.float32) for a in (means, values, con, pts)]
shapes = {0: (M, c), 1: (M, 2, c), 2: (M, 2, 2, c), "lap": (M, c)}
rs = {o: rng.uniform(0, 1, shapes[o]).astype(np.float32) for o in orders}
res = {}
for name, kw in (("dense", dict(backend="dense")), ("binned36", dict(backend="binned", q_max_backward=36.0)),
                 ("bwd40", dict(backend="binned", q_max_backward=40.0)), ("bwd44", dict(backend="binned", q_max_backward=44.0)),
                 ("binned60", dict(backend="binned", q_max=60.0))):
    t = [torch.tensor(a, device="cuda") for a in f32]
    for x in t[:3]:
        x.requires_grad_(True)
    smp = GaussianSampler(False, **kw)
    smp.preprocess(t[0], t[1], None, t[2], t[3])
    outs = smp.sample(orders)
    loss = sum((o * torch.tensor(rs[n], device="cuda")).sum() for n, o in zip(orders, outs))
    loss.backward()
    res[name] = [t[0].grad.cpu().double().numpy(), t[2].grad.cpu().double().numpy(), t[1].grad.cpu().double().numpy()]
a64 = [a.astype(np.float64) for a in (f32[0], f32[2], f32[1], f32[3])]
gdiff = sum(np.abs(a - b).reshape(N, -1).max(1) / np.abs(a).max() for a, b in zip(res["dense"], res["binned36"]))
gsel = np.argsort(gdiff)[-256:]
g64 = {}
for n in orders:
    if n == "lap":
        g2 = np.zeros((M, 2, 2, c)); g2[:, 0, 0] = rs[n]; g2[:, 1, 1] = rs[n]; g64[2] = g2
    else:
        g64[n] = rs[n].astype(np.float64)
sub = (a64[0][gsel], a64[1][gsel], a64[2][gsel], a64[3])
want = c_oracle.backward(*sub, g64)
mag = c_oracle.backward(*sub, g64, absolute=True)
print(f"case {k} {kind} N={N} M={M} c={c} orders={orders}")
for name in res:
    for gi, gname in enumerate(("means", "conics", "values")):
        g = res[name][gi][gsel]
        fs = np.abs(res["dense"][gi]).max()
        err = np.abs(g - want[gi])
        print(f"  {name:9s} {gname:7s}: max err / max entry {err.max() / fs:.2e}   max err / (1e-6 mag) {(err / (1e-6 * mag[gi] + 1e-300)).max():.2f}"
              f"   max err / entry-abs-sum {(err / (mag[gi] + 1e-300)).max():.2e}")
# the worst Gaussian of binned36's conic gradient
e = np.abs(res["binned36"][1][gsel] - want[1]).max(1)
w = gsel[np.argmax(e)]
C = np.array([[a64[1][w, 0], a64[1][w, 1]], [a64[1][w, 1], a64[1][w, 2]]])
ev = np.linalg.eigvalsh(np.linalg.inv(C))
x = a64[3] - a64[0][w]
q = np.einsum("mi,ij,mj->m", x, C, x)
print(f"  worst Gaussian {w}: mean {a64[0][w]}, sigma axes {np.sqrt(ev)}, points with q<36: {(q < 36).sum()}, 36<=q<60: {((q >= 36) & (q < 60)).sum()}")
print(f"    conic grad oracle {want[1][np.argmax(e)]}, dense {res['dense'][1][w]}, binned36 {res['binned36'][1][w]}, binned60 {res['binned60'][1][w]}")

#!/usr/bin/env python3
"""Tile modes and list lengths of a plan at C3 size for a point distribution: argv = case (as preprocess_cases.py)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pigs_amd import synthetic, _lib
from diff_gaussian_sampling import GaussianSampler

case = sys.argv[1] if len(sys.argv) > 1 else "clustered"
gs = synthetic.lattice_gaussians(256, 256, 0.5, seed=1)
t = {k: v.float().cuda() for k, v in gs.items()}
g = torch.Generator().manual_seed(3)
if case == "grid":
    pts = synthetic.grid_samples(1024).float()
elif case == "random":
    pts = torch.rand((1 << 20, 2), generator=g) * 2 - 1
else:
    sigma = float(case.split(":")[1]) if ":" in case else 0.15
    pts = (torch.randn((1 << 20, 2), generator=g) * sigma).clamp(-1, 1)
s = GaussianSampler(False, backend="binned", defer_lists=False)
with torch.no_grad():
    s.preprocess(t["means"], t["values"], None, t["conics"], pts.cuda())
plan = s._plan
info = (ctypes.c_int64 * 6)()
_lib.load().pigs_plan_layout_info(plan.N, plan.M, plan.c, info)
nt, off = info[0], info[2]
hdr = plan.workspace[off:off + 32 * nt].view(torch.int32).cpu().numpy().astype(np.uint32).reshape(nt, 8)
mode = hdr[:, 0] >> 30
ng = hdr[:, 1:5].astype(np.int64)
rows = ng.max(1)
print(case, "tiles", nt, "modes LIST/RANGES/GROUPS/POINTS", [(mode == k).sum() for k in range(4)])
lst = (mode == 0) | (mode == 2)
print("  rows (longest group list) of list tiles: mean %.1f  p50 %d p90 %d p99 %d max %d  sum %d" % (
    rows[lst].mean(), *np.percentile(rows[lst], [50, 90, 99]).astype(int), rows[lst].max(), rows[lst].sum()))
for lo, hi in ((0, 48), (48, 96), (96, 200), (200, 513)):
    sel = lst & (rows >= lo) & (rows < hi)
    print(f"  rows in [{lo},{hi}): {sel.sum()} tiles, {rows[sel].sum()} rows")

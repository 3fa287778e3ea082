"""One clean loop of the hot path for rocprofv3 (kernel stats without bench.py's side measurements):

    python tools/prof_step.py [cold|warm|fwdbwd|fwd] [--kappa 0.5] [--res 1024] [--steps 200] [--stats]

cold: preprocess (samples rebuilt every step) + fused forward; warm: the samples half reused;
fwd: the forward launch alone on one plan; fwdbwd: warm preprocess + forward + backward (incoming
gradients supplied).  --stats prints the tile-list statistics of the plan (entries per tile, rows
per step after the group split)."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def list_stats(plan):
    from pigs_amd import _lib
    lib = _lib.load()
    info = (ctypes.c_int64 * 6)()
    assert lib.pigs_plan_layout_info(plan.N, plan.M, plan.c, info) == 0
    ntiles, cap, off_hdr = info[0], info[1], info[2]
    ws = plan.workspace
    hdr = ws[off_hdr:off_hdr + 32 * ntiles].view(torch.int32).cpu().numpy().astype(np.uint32).reshape(ntiles, 8)
    mode, count = hdr[:, 0] >> 30, hdr[:, 0] & ((1 << 30) - 1)
    ng = hdr[:, 1:5].astype(np.float64)
    lst = (mode == 0) | (mode == 2)     # 0: tile list + group lists, 2: group lists only, 1: record ranges, 3: every point walks the grid
    return {"tiles": int(ntiles), "list_cap": int(cap), "ranges_tiles": int((mode == 1).sum()), "points_tiles": int((mode == 3).sum()),
            "groups_only_tiles": int((mode == 2).sum()),
            "entries_mean": float(count[mode == 0].mean()) if (mode == 0).any() else 0.0,
            "entries_max": int(count[mode == 0].max()) if (mode == 0).any() else 0,
            "rows_mean": float(ng[lst].max(1).mean()) if lst.any() else 0.0, "per_group_mean": float(ng[lst].mean()) if lst.any() else 0.0,
            "per_group_max": int(ng[lst].max()) if lst.any() else 0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", nargs="?", default="cold", choices=["cold", "warm", "fwd", "bwd", "fwdbwd"])
    ap.add_argument("--kappa", type=float, default=0.5)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--lat", type=int, default=256)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--defer", action="store_true", help="PIGS_BUILD_DEFER_LISTS: the first forward builds the tile lists in its own launch")
    a = ap.parse_args()
    import importlib
    importlib.import_module("pigs_amd.build").ensure_built()      # before anything touches the GPU; never builds behind rocprofv3
    from diff_gaussian_sampling import GaussianSampler
    from pigs_amd import synthetic, sampler as S
    dev = torch.device("cuda", 0)
    gs = synthetic.lattice_gaussians(a.lat, a.lat, a.kappa, seed=0)
    pts = synthetic.grid_samples(a.res, a.res).float().to(dev)
    t = {k: v.float().to(dev) for k, v in gs.items()}
    s = GaussianSampler(False, fuse="all", backend="binned", reuse_samples=a.mode != "cold", defer_lists=a.defer)
    req = {k: t[k].clone().requires_grad_(True) for k in ("means", "values", "conics")}
    gouts = None

    def step():
        nonlocal gouts
        if a.mode == "fwdbwd":
            s.preprocess(req["means"], req["values"], t["covariances"], req["conics"], pts)
            outs = s.sample((0, 1, 2))
            if gouts is None:
                gouts = tuple(torch.randn_like(o) for o in outs)
            torch.autograd.grad(outs, list(req.values()), grad_outputs=gouts)
        elif a.mode == "fwd":
            S.forward_raw(*s._inputs, 7, s._plan)
        elif a.mode == "bwd":
            S.backward_raw(*s._inputs, bw_gouts, 7, s._plan)
        else:
            with torch.no_grad():
                s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
                s.sample((0, 1, 2))

    bw_gouts = None
    if a.mode in ("fwd", "bwd"):
        with torch.no_grad():
            s.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
            M = pts.shape[0]
            bw_gouts = [torch.randn((M,) + (2,) * k + (1,), device=dev) for k in range(3)] + [None, None]
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    print(f"{a.mode} kappa={a.kappa} res={a.res}: {(time.perf_counter() - t0) / a.steps * 1e6:.1f} us/step")
    if a.stats:
        print(list_stats(s._plan))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Diagnostic: per-wave phase times of the forward kernel from a PIGS_STAMPS=1 build
(PIGS_AMD_LIB=.../libpigs_amd_stamps.so python tools/stamps.py [kappa])."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pigs_amd import synthetic, _lib
from pigs_amd.sampler import GaussianSampler

kappa = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
gs, pts = synthetic.CONFIGS["c3"](kappa)
t = {k: v.float().cuda() for k, v in gs.items()}
pts = pts.float().cuda()
s = GaussianSampler(False, backend="binned")
with torch.no_grad():
    for _ in range(3):
        s.preprocess(t["means"], t["values"], None, t["conics"], pts)
        s.sample((0, 1, 2))
    torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros((32768, 6), dtype=np.uint64)
assert lib.pigs_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
b = buf[buf[:, 0] > 0].astype(np.int64)
t0 = b[:, 0].min()
tick = 10.0  # ns per s_memrealtime tick (100 MHz)
print("waves", len(b), "kernel span us", (b[:, 4].max() - t0) * tick / 1e3)
names = ["load+bbox", "traverse+test+compact", "evaluate", "store(+wait)"]
for k in range(4):
    d = (b[:, k + 1] - b[:, k]) * tick / 1e3
    print(f"{names[k]:>24}: mean {d.mean():7.2f} us  p10 {np.percentile(d,10):6.2f}  p50 {np.percentile(d,50):6.2f}  p90 {np.percentile(d,90):6.2f}")
life = (b[:, 4] - b[:, 0]) * tick / 1e3
print("wave lifetime mean", life.mean(), "accepted mean", b[:, 5].mean())
start = (b[:, 0] - t0) * tick / 1e3
print("wave start times: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(start, [10, 50, 90, 100])))
end = (b[:, 4] - t0) * tick / 1e3
print("wave end times:   p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(end, [10, 50, 90, 100])))

#!/usr/bin/env python3
"""Binned vs dense HIP at sizes where every plan mechanism is in play (more than 1024 tiles: the
backward's tile shuffle; clustered and scattered points: group-only and record-range tiles; repeated
preprocess on the same points: recycled workspaces, remembered sample structures).  argv: cases, seed.
``gen_cases`` is also what tests/test_fuzz_gpu.py::test_fuzz_big_worst_cases_against_the_oracle draws
its inputs from (the three cases with the largest binned-vs-dense gradient difference of the
32-case run of seed 0, adjudicated there against the float64 oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np


def gen_cases(cases, seed=0):
    """Yields (k, kind, means, con, values, pts, orders); one sequential random stream, as the tool ran."""
    rng = np.random.default_rng(seed)
    for k in range(cases):
        N = int(rng.integers(500, 20000))
        M = int(rng.integers(70000, 250000))
        c = int(rng.integers(1, 3))
        kind = ("uniform", "clustered", "core+halo", "grid")[k % 4]
        if kind == "uniform":
            pts = rng.uniform(-1, 1, (M, 2))
        elif kind == "clustered":
            pts = np.clip(rng.normal(0, rng.uniform(0.2, 0.6), (M, 2)), -1, 1)
        elif kind == "core+halo":
            pts = np.clip(np.concatenate((rng.normal(0, 0.05, (M - M // 40, 2)), rng.uniform(-1, 1, (M // 40, 2)))), -1, 1)
        else:
            r = int(np.sqrt(M)); M = r * r
            gx, gy = np.meshgrid(np.linspace(-1, 1, r), np.linspace(-1, 1, r), indexing="xy")
            pts = np.stack((gx, gy), -1).reshape(M, 2)
        spacing = np.sqrt(4.0 / N)
        sig = np.exp(rng.normal(np.log(rng.uniform(0.4, 1.2) * spacing), 0.3, (N, 2)))
        tau = np.tanh(rng.normal(0, 0.5, N)) * sig[:, 0] * sig[:, 1]
        s0, s1 = sig[:, 0] ** 2, sig[:, 1] ** 2
        det = s0 * s1 - tau ** 2
        con = np.stack((s1 / det, -tau / det, s0 / det), -1)
        means = rng.uniform(-1, 1, (N, 2))
        values = rng.uniform(-1, 1, (N, c))
        orders = (0, 1, 2) if k % 3 else (0, 1, "lap")
        yield k, kind, means, con, values, pts, orders


def main():
    import torch
    from diff_gaussian_sampling import GaussianSampler
    from prof_step import list_stats
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    binned = GaussianSampler(False, backend="binned")
    dense = GaussianSampler(False, backend="dense")
    worst = {"out": 0.0, "grad": 0.0}
    for k, kind, means, con, values, pts, orders in gen_cases(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0):
        N, M, c = means.shape[0], pts.shape[0], values.shape[1]
        res = {}
        for name, smp in (("dense", dense), ("binned", binned)):
            t = [torch.tensor(a, dtype=torch.float32, device="cuda") for a in (means, values, con, pts)]
            for x in t[:3]:
                x.requires_grad_(True)
            smp.preprocess(t[0], t[1], None, t[2], t[3])
            outs = smp.sample(orders)
            torch.manual_seed(k)
            loss = sum((o * torch.rand_like(o)).sum() for o in outs)           # positive weights: no cancellation in the sums
            loss.backward()
            res[name] = ([o.detach() for o in outs], [x.grad for x in t[:3]])
        st = list_stats(binned._plan)
        eo = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(*[res[n][0] for n in ("dense", "binned")]))
        eg = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(*[res[n][1] for n in ("dense", "binned")]))
        worst["out"], worst["grad"] = max(worst["out"], eo), max(worst["grad"], eg)
        print(f"{k:3d} {kind:10s} N={N:6d} M={M:7d} c={c} orders={orders}: tiles {st['tiles']} groups-only {st['groups_only_tiles']} "
              f"ranges {st['ranges_tiles']} | out {eo:.2e} grad {eg:.2e}", flush=True)
        assert eo < 1e-5 and eg < 5e-5, "binned and dense disagree"
    print("worst:", worst)


if __name__ == "__main__":
    main()

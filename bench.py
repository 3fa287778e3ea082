#!/usr/bin/env python3
"""Benchmark of the sampler hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2] [--kappa 0.5]

One step = one pass of the hot path over one batch of synthetic input, through the reference's
operator surface: ``GaussianSampler.preprocess(...)`` + one fused launch producing u, grad u and
the Hessian (orders 0..2) for every sample point.  Inputs are resident in HBM before the timed
region.  N > 1: launched by torch.distributed.run, one rank per GPU; the sample grid is sharded
by rows (weak scaling: every rank owns a res x res block of a res x (N res) grid over the same
65k Gaussians); the forward needs no collective, the backward all-reduces the parameter
gradients (reported in the extra ``fwd_bwd`` field, not part of the timed K steps).

Prints ONE JSON line on rank 0 (contract in the task description): metric, value (whole-job
sample-points/s), roofline of the dominant kernel (HBM-bound designation of BASELINE.json),
cpu_baseline (the reference's dense PyTorch algorithm on the host cores, bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
VALU_PAIR_PEAK = 3.6e12    # pairs/s: 157.3 TFLOP/s fp32 / 2 / ~22 issue slots per pair


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2"])
    ap.add_argument("--kappa", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bwd", action="store_true")
    ap.add_argument("--backend", default="auto", choices=["auto", "dense", "binned"])
    # rehearsal of the N > 1 path on a box with fewer GPUs than ranks (gloo, ranks share devices)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    return ap.parse_args()


def measured_traffic(workload, kappa, binned):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    same command, corrected as MI355X_MICROARCH.md 'HBM' prescribes); None when no profile of
    this exact configuration is committed."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        key = f"{workload}:kappa={kappa}:{'binned' if binned else 'dense'}"
        return table.get(key, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def graph_replay(GaussianSampler, t, pts_d, backend, n):
    """ms per replay of the captured training step / sampler-only step (fresh leaves and sampler made
    under the capture stream: autograd remembers the stream a leaf was first used on)."""
    dev = pts_d.device
    out = {}
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        req = {k: t[k].clone().requires_grad_(True) for k in ("means", "values", "conics")}
        sampler = GaussianSampler(False, fuse="all", backend=backend)
        gouts = [None]

        def train_step():
            sampler.preprocess(req["means"], req["values"], t["covariances"], req["conics"], pts_d)
            u, ux, uxx = sampler.sample((0, 1, 2))
            loss = ((u[:, 0] - (uxx[:, 0, 0, 0] + uxx[:, 1, 1, 0])) ** 2).mean() + (ux ** 2).mean()
            return torch.autograd.grad(loss, list(req.values()))

        def sampler_step():
            sampler.preprocess(req["means"], req["values"], t["covariances"], req["conics"], pts_d)
            outs = sampler.sample((0, 1, 2))
            if gouts[0] is None:
                gouts[0] = tuple(torch.randn_like(o) for o in outs)
            return torch.autograd.grad(outs, list(req.values()), grad_outputs=gouts[0])

        def trace_step():
            sampler.preprocess(req["means"], req["values"], t["covariances"], req["conics"], pts_d)
            u, ux, lap = sampler.sample((0, 1, "lap"))
            loss = ((u - lap) ** 2).mean() + (ux ** 2).mean()
            return torch.autograd.grad(loss, list(req.values()))

        for name, fn in (("ms_per_step", train_step), ("sampler_only_ms_per_step", sampler_step),
                         ("trace_residual_ms_per_step", trace_step)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                keep = fn()
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n):
                graph.replay()
            torch.cuda.synchronize(dev)
            out[name] = (time.perf_counter() - t0) / n * 1e3
            del graph, keep
    torch.cuda.current_stream(dev).wait_stream(side)
    return out


def host_cores():
    """Cores this process may actually use: affinity, capped by the cgroup CPU quota and by the
    GPU box's per-GPU share (16)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, 16))


def cpu_baseline(gs, pts, budget_pairs=6e8):
    """The reference's dense PyTorch algorithm (oracle/dense_torch.py) on the host cores, on a
    bounded slice of the same workload; linear in M, so points/s extrapolates."""
    from oracle import dense_torch
    cores = host_cores()
    torch.set_num_threads(cores)
    N = gs["means"].shape[0]
    m = int(max(256, min(pts.shape[0], budget_pairs // N)))
    m -= m % 256
    sl = pts[:: max(1, pts.shape[0] // m)][:m].float().contiguous()
    args = (gs["means"].float(), gs["conics"].float(), gs["values"].float(), sl)
    dense_torch.forward(*[a[:256] if i == 3 else a for i, a in enumerate(args)])  # warm-up
    t0 = time.perf_counter()
    dense_torch.forward(*args, orders=(0, 1, 2), chunk=256)
    dt = time.perf_counter() - t0
    return {
        "value": m / dt, "unit": "sample-points/s", "cores": cores, "kind": "port",
        "sample": f"{m} of {pts.shape[0]} points x {N} Gaussians, orders 0-2, fp32, "
                  f"torch dense (reference algorithm, oracle/dense_torch.py), {dt:.1f} s, linear in M",
    }


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    dev_index = local_rank if a.dist_backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    import pigs_amd
    if local_rank == 0:
        pigs_amd.build()                 # no-op when the in-tree library matches the sources
    if dist is not None:
        dist.barrier()                   # the other ranks wait for the library instead of rebuilding it
    from diff_gaussian_sampling import GaussianSampler
    from pigs_amd import synthetic, sampler as S

    # ---------------- workload (seeded, generated on the CPU, then moved) ----------------
    if a.workload == "c3":
        nx = ny = 256
        res = 1024
    else:
        nx, ny, res = 128, 64, 256
    gs = synthetic.lattice_gaussians(nx, ny, a.kappa, seed=0)
    # weak scaling (the grid shards by blocks of rows, the Gaussians are replicated): the
    # global grid is the square side x side grid over [-1,1]^2 with ~res*res points per GPU
    # (side = round(res * sqrt(world)): 1024, 1448, 2048, 2896 for 1, 2, 4, 8 GPUs -- isotropic
    # spacing at every N); rank r owns rows [r*rows, (r+1)*rows), rows = side // world
    side = int(round(res * math.sqrt(world)))
    rows = side // world
    pts = synthetic.grid_samples(side, side, row0=rank * rows, rows=rows)
    N, M = gs["means"].shape[0], pts.shape[0]
    t = {k: v.float().to(dev) for k, v in gs.items()}
    pts_d = pts.float().to(dev)
    sampler = GaussianSampler(False, fuse="all", backend=a.backend)

    def step():
        sampler.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts_d)
        return sampler.sample((0, 1, 2))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(a.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        barrier()
        dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / a.steps * 1e3
    value = M * world / (dt / a.steps)

    # ---------------- the same steps dealt round-robin to two HIP streams ----------------
    # (extra figure, not `value`: independent evaluation steps -- frames of a roll-out -- can overlap;
    # the latency-bound plan build of step k+1 then hides under the VALU-bound forward of step k)
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    samplers2 = [GaussianSampler(False, fuse="all", backend=a.backend) for _ in range(2)]
    keep = [None, None]

    def run2(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % 2]):
                samplers2[i % 2].preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts_d)
                keep[i % 2] = samplers2[i % 2].sample((0, 1, 2))

    with torch.no_grad():
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        run2(max(2, a.warmup))
        barrier()
        t0 = time.perf_counter()
        run2(a.steps)
        barrier()
        dt2 = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt2], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt2 = float(tmax.item())
    two_streams = {"ms_per_step": dt2 / a.steps * 1e3, "value": M * world / (dt2 / a.steps),
                   "what": "the same K steps issued round-robin on two HIP streams (one sampler per stream)"}
    del keep, samplers2

    # ---------------- dominant kernel: HIP events on the launch stream ----------------
    means, values, conics, samples = sampler._inputs
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        plan = sampler._plan
        S.forward_raw(means, values, conics, samples, 7, plan)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(a.steps):
            S.forward_raw(means, values, conics, samples, 7, plan)
        e1.record()
        torch.cuda.synchronize(dev)
    kernel_s = e0.elapsed_time(e1) / a.steps * 1e-3
    algo_bytes = 24 * N + 36 * M       # fp32, d=2, c=1: 6 floats/Gaussian + (2 in + 7 out) floats/point
    achieved = algo_bytes / kernel_s
    roofline = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": measured_traffic(a.workload, a.kappa, plan is not None),
                "kernel": ("binned_forward_kernel<1,7>" if plan is not None else "dense_forward_kernel<float,2,1,7,4>"),
                "kernel_ms": kernel_s * 1e3, "algorithmic_bytes": algo_bytes}
    if plan is None:
        roofline["valu_frac_dense_pairs"] = N * M / kernel_s / VALU_PAIR_PEAK

    # ---------------- fwd + bwd step (second half of BASELINE.json's metric) ----------------
    fwd_bwd = None
    if not a.no_bwd:
        req = {k: t[k].clone().requires_grad_(True) for k in ("means", "values", "conics")}

        from pigs_amd.distributed import replicated

        def train_step():
            # parameter grads are summed over the ranks by ONE all-reduce of a packed [N,6] buffer
            m_r, v_r, c_r = replicated(req["means"], req["values"], req["conics"])
            sampler.preprocess(m_r, v_r, t["covariances"], c_r, pts_d)
            u, ux, uxx = sampler.sample((0, 1, 2))
            # diffusion residual shape of test_no_mlp.py:144 (u_t replaced by u: same data flow)
            loss = ((u[:, 0] - (uxx[:, 0, 0, 0] + uxx[:, 1, 1, 0])) ** 2).mean() + (ux ** 2).mean()
            return torch.autograd.grad(loss, list(req.values()))

        gouts = None

        def sampler_step():
            # the sampler's own share of a training step: incoming gradients supplied, no loss kernels
            nonlocal gouts
            m_r, v_r, c_r = replicated(req["means"], req["values"], req["conics"])
            sampler.preprocess(m_r, v_r, t["covariances"], c_r, pts_d)
            outs = sampler.sample((0, 1, 2))
            if gouts is None:
                gouts = tuple(torch.randn_like(o) for o in outs)
            return torch.autograd.grad(outs, list(req.values()), grad_outputs=gouts)

        def timed(fn, n):
            fn()
            barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            barrier()
            return (time.perf_counter() - t0) / n * 1e3

        def trace_step():
            # the same residual through the fused Hessian-trace output (4 floats per point instead of 7,
            # no slicing of [M,2,2,1] in the loss): extension of the reference API, SURVEY.md 8f-4
            m_r, v_r, c_r = replicated(req["means"], req["values"], req["conics"])
            sampler.preprocess(m_r, v_r, t["covariances"], c_r, pts_d)
            u, ux, lap = sampler.sample((0, 1, "lap"))
            loss = ((u - lap) ** 2).mean() + (ux ** 2).mean()
            return torch.autograd.grad(loss, list(req.values()))

        nb = max(1, min(a.steps, 10))
        fwd_bwd = {"ms_per_step": timed(train_step, nb), "sampler_only_ms_per_step": timed(sampler_step, nb),
                   "trace_residual_ms_per_step": timed(trace_step, nb), "steps": nb,
                   "what": "ms_per_step: preprocess + fused fwd(0..2) + torch residual loss + fused bwd; "
                           "sampler_only: the same without the loss (grad_outputs supplied); "
                           "trace_residual: the training step with the fused u, grad u, u_xx+u_yy outputs "
                           "(sample((0, 1, 'lap'))) instead of the full Hessian; "
                           "hipgraph_replay (1 GPU): the same steps captured once and replayed"
                           + ("; parameter grads all-reduced as one [N,6] buffer" if dist is not None else "")}

        if dist is None:
            # the same two steps captured ONCE into a hipGraph and replayed (no entry point allocates or
            # synchronises): what a training loop pays when the host is taken out of the way
            try:
                fwd_bwd["hipgraph_replay"] = graph_replay(GaussianSampler, t, pts_d, a.backend, nb)
            except Exception as e:            # report, never lose the bench line over the extra figure
                fwd_bwd["hipgraph_replay"] = {"error": f"{type(e).__name__}: {e}"[:200]}

    line = {
        "metric": "sample-points/sec (fwd + 1st + 2nd derivatives, fused)", "value": value,
        "unit": "sample-points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (seeded lattice Gaussians, regular sample grid)",
        "config": {"workload": f"{a.workload}: {N} Gaussians x {side}x{side} grid, {rows} rows x {side} points per "
                               f"GPU, d=2, c=1, kappa={a.kappa}, orders 0-2", "gaussians": N, "points_per_gpu": M,
                   "kappa": a.kappa, "path": "binned" if sampler._plan is not None else "dense", "step": "preprocess + fused forward (orders 0..2)"},
        "roofline": roofline, "fwd_bwd": fwd_bwd, "two_streams": two_streams,
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(gs, pts)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

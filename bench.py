#!/usr/bin/env python3
"""Benchmark of the sampler hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c4] [--kappa 0.5]

One step = one pass of the hot path over one batch of synthetic input, through the reference's
operator surface: ``GaussianSampler.preprocess(...)`` + one fused launch producing u, grad u and
the Hessian (orders 0..2) for every sample point.  Inputs are resident in HBM before the timed
region.  ``value`` times the COLD step: nothing is reused between steps (samples re-sorted, Gaussians
re-binned, tile lists rebuilt every time); ``value_warm_plan`` is the same step when ``preprocess``
is handed the same, unmodified samples tensor as before and reuses its sorted sample structure --
the reference's roll-out pattern (main_pn.py:317-324).

N > 1: launched by torch.distributed.run, one rank per GPU.  Default (``c3``) is weak scaling: every
rank owns ~1 M points of a square grid of side round(1024 sqrt(N)) over the same 65k Gaussians.
``--workload c4`` is BASELINE configs[3]: the 4096^2 grid sharded by rows over the ranks (strong
scaling; one rank = all 16.8 M points).  The forward needs no collective; the backward all-reduces
the parameter gradients as ONE packed [N,6] buffer (RCCL over xGMI): for N > 1 the line also carries
``fwd_bwd.value`` = points/s of the whole preprocess + forward + backward + all-reduce step.

Prints ONE JSON line on rank 0 (contract in the task description): metric, value (whole-job
sample-points/s), roofline of the dominant kernel (HBM-bound designation of BASELINE.json) with the
VALU-issue figure beside it, roofline_bwd, cpu_baseline (the reference's dense PyTorch algorithm on
the host cores, bounded sample).
"""
import argparse
import gc
import ctypes
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
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2       # wave-instructions/s: 1 024 SIMDs, one wave64 VALU op per 2 cycles
ROW_VALU_INSTR = 21.25     # VALU instructions per evaluated row (16 points x 1 Gaussian x 4 rows of a wave), ISA count
ROW_NS_UBENCH = 30.8       # ns per wave-row per SIMD of the bare row loop at 8 waves/SIMD (tools/ubench/rowloop.hip)
PREHEAT_S = 0.25


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c4"])
    ap.add_argument("--kappa", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bwd", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip kappa_1_3, two_streams, hipgraph_replay")
    ap.add_argument("--backend", default="auto", choices=["auto", "dense", "binned"])
    # rehearsal of the N > 1 path on a box with fewer GPUs than ranks (gloo, ranks share devices)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    return ap.parse_args()


def measured_traffic(workload, kappa, binned, which="forward"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, corrected
    as MI355X_MICROARCH.md 'HBM' prescribes).  The record names the source hash of the library it was
    measured on: (None, reason) when no profile of this exact configuration AND this exact source
    is committed -- a stale figure is never reported."""
    try:
        import importlib
        B = importlib.import_module("pigs_amd.build")      # the module (the package re-exports its build())
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        key = f"{workload}:kappa={kappa}:{'binned' if binned else 'dense'}"
        rec = table.get(key)
        if not rec:
            return None, f"no PMC profile committed for {key}"
        if rec.get("source_hash") != B.source_hash():
            return None, f"profiles/pmc_traffic.json[{key}] was measured on other kernel sources (hash mismatch)"
        if which == "backward":
            if "backward" not in rec:
                return None, f"profiles/pmc_traffic.json[{key}] holds no backward record"
            return rec["backward"].get("hbm_bytes_per_launch"), f"profiles/pmc_traffic.json[{key}].backward @ source {rec['source_hash'][:12]}"
        return rec.get("hbm_bytes_per_launch"), f"profiles/pmc_traffic.json[{key}] @ source {rec['source_hash'][:12]}"
    except (OSError, ValueError, KeyError) as e:
        return None, f"{type(e).__name__}: {e}"


def plan_rows(plan):
    """Evaluated rows (16 points x 1 Gaussian, four per wave instruction) of one forward launch on this
    plan, read from its tile headers: sum over the tiles of the longest group list, padded to 2."""
    from pigs_amd import _lib
    lib = _lib.load()
    info = (ctypes.c_int64 * 6)()
    if lib.pigs_plan_layout_info(plan.N, plan.M, plan.c, info) != 0:
        return None
    ntiles, off_hdr = info[0], info[2]
    hdr = plan.workspace[off_hdr:off_hdr + 32 * ntiles].view(torch.int32).reshape(ntiles, 8)
    if int(((hdr[:, 0] >> 30) & 1).sum()) != 0:
        return None                       # range-mode / point-mode tiles: the headers do not hold the row count
    ng = hdr[:, 1:5].to(torch.int64)
    wave_rows = int(((ng.max(dim=1).values + 1) // 2 * 2).sum())
    return {"wave_rows": wave_rows, "pairs": int(ng.sum()) * 16}


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
        # the bench's points never change: the captures reuse the sorted sample structure the warm-up runs built
        # (GaussianSampler.static_samples), so a replay is a WARM step like the eager steps it is compared with
        sampler.static_samples = True
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

        def residual_step():
            sampler.preprocess(req["means"], req["values"], t["covariances"], req["conics"], pts_d)
            loss = sampler.residual(a0=1.0, lap=-1.0).pow(2).mean()
            return torch.autograd.grad(loss, list(req.values()))

        for name, fn in (("ms_per_step", train_step), ("sampler_only_ms_per_step", sampler_step),
                         ("trace_residual_ms_per_step", trace_step), ("fused_residual_ms_per_step", residual_step)):
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


def small_record(GaussianSampler, dev, reps=200):
    """The reference's own training sizes (main_pn.py:57,103: N ~ 1e3 Gaussians, 1 024 collocation points)
    in the call pattern of Model.sample (model_pn.py:766-788): preprocess(samples) + sample_gaussians /
    _derivative / _laplacian, preprocess(bc_samples) + sample_gaussians, then ONE backward of all four
    outputs (incoming gradients supplied: the sampler's own share of the step).  At these sizes the GPU
    work is a few launches of a few microseconds: the figure is the host's.  Eager through the native host
    extension, eager through the ctypes host, and the same step replayed from a hipGraph."""
    from pigs_amd import synthetic
    gs = synthetic.lattice_gaussians(40, 40, 1.3, seed=1)
    gen = torch.Generator().manual_seed(3)
    out = {"gaussians": 1600, "points": 1024, "bc_points": 1024,
           "pattern": "model_pn.py:766-788: 2 preprocess, orders 0-2 + boundary u, 1 backward (sampler only)"}

    def make(host):
        t = {k: v.float().to(dev) for k, v in gs.items()}
        for k in ("means", "values", "conics"):
            t[k].requires_grad_(True)
        pts = (torch.rand((1024, 2), generator=gen) * 2 - 1).to(dev)
        bc = (torch.rand((1024, 2), generator=gen) * 2 - 1).to(dev)
        smp = GaussianSampler(False, host=host)
        gouts = []

        def step():
            smp.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts)
            outs = [smp.sample_gaussians(), smp.sample_gaussians_derivative(), smp.sample_gaussians_laplacian()]
            smp.preprocess(t["means"], t["values"], t["covariances"], t["conics"], bc)
            outs.append(smp.sample_gaussians())
            if not gouts:
                gouts.extend(torch.randn_like(o) for o in outs)
            return torch.autograd.grad(outs, (t["means"], t["values"], t["conics"]), grad_outputs=gouts)
        return step

    def timed(fn):
        for _ in range(20):
            fn()
        torch.cuda.synchronize(dev)
        gc.disable()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        gc.enable()
        return (time.perf_counter() - t0) / reps * 1e3

    for host in ("native", "ctypes"):
        out[f"eager_{host}_ms_per_step"] = timed(make(host))
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        step = make("native")
        for _ in range(3):
            step()
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        keep = step()
    out["graph_replay_ms_per_step"] = timed(graph.replay)
    del graph, keep
    torch.cuda.current_stream(dev).wait_stream(side)
    out["eager_over_replay"] = out["eager_native_ms_per_step"] / out["graph_replay_ms_per_step"]
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
    # build (or wait for local rank 0's build) BEFORE anything touches the GPU: compiling forks hipcc / g++,
    # and a compiler exec chain behind a GPU-initialised or profiled process is forbidden on the GPU pool
    import importlib
    importlib.import_module("pigs_amd.build").ensure_built(wait_for_rank0=local_rank != 0)
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

    from diff_gaussian_sampling import GaussianSampler
    from pigs_amd import synthetic, sampler as S

    # ---------------- workload (seeded, generated on the CPU, then moved) ----------------
    def workload(kappa):
        if a.workload == "c2":
            nx, ny, res = 128, 64, 256
        else:
            nx = ny = 256
            res = 1024
        gs_ = synthetic.lattice_gaussians(nx, ny, kappa, seed=0)
        if a.workload == "c4":
            # BASELINE configs[3]: the 4096^2 grid sharded by rows over the ranks (strong scaling)
            side_ = 4096
            from pigs_amd.distributed import shard_bounds
            r0, r1 = shard_bounds(side_, world, rank)
        else:
            # weak scaling (the grid shards by blocks of rows, the Gaussians are replicated):
            # the global grid is the square side x side grid over [-1,1]^2 with ~res*res points per GPU
            # (side = res * sqrt(world) rounded to a multiple of 8 * world: 1024, 1456, 2048, 2880 for 1, 2, 4, 8 GPUs --
            # isotropic spacing at every N, and every rank's block of rows is a lattice whose sides are multiples of 8,
            # which the samples build takes in index-tiled order like the 1024 x 1024 grid of N = 1: per-GPU work stays
            # the same KIND of work; points per GPU within 1.2 % of 1024^2, the exact count is in `config`);
            # rank r owns rows [r*rows, (r+1)*rows), rows = side // world
            side_ = max(1, int(round(res * math.sqrt(world) / (8 * world)))) * 8 * world
            r0, r1 = rank * (side_ // world), (rank + 1) * (side_ // world)
        pts_ = synthetic.grid_samples(side_, side_, row0=r0, rows=r1 - r0)
        return gs_, pts_, side_, r1 - r0

    gs, pts, side, rows = workload(a.kappa)
    N, M = gs["means"].shape[0], pts.shape[0]
    t = {k: v.float().to(dev) for k, v in gs.items()}
    pts_d = pts.float().to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if dist is None:
            return x
        tmax = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    host_issue = [0.0]

    def timed_steps(step, warmup, steps):
        """W untimed + exactly K timed steps between barrier + synchronize pairs; max over the ranks."""
        for _ in range(warmup):
            step()
        barrier()
        # Python's cyclic collector stays out of the timed region: a full collection of a process that has
        # imported torch walks ~10^6 objects and takes 35-45 ms (tools/stall_probe.py: one step in ~900),
        # more than the K steps together.  Reference counting still frees every tensor at once.
        gc_was_on = gc.isenabled()
        gc.disable()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        host_issue[0] = (time.perf_counter() - t0) / max(steps, 1)      # this rank's host: the loop before the device has drained
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        if gc_was_on:
            gc.enable()
        return dt

    def forward_only(tt, pp, reuse):
        smp = GaussianSampler(False, fuse="all", backend=a.backend, reuse_samples=reuse)

        def step():
            smp.preprocess(tt["means"], tt["values"], tt["covariances"], tt["conics"], pp)
            return smp.sample((0, 1, 2))
        return smp, step

    def settle(fn, seconds=0.05):
        """Untimed steps of a NEW sampler before its timed ones: its first steps allocate and first-touch
        fresh workspaces; a few hundred untimed steps keep that out of the measurement (profiles/README.md)."""
        t0_ = time.perf_counter()
        while time.perf_counter() - t0_ < seconds:
            for _ in range(10):
                fn()
            torch.cuda.synchronize(dev)

    # ---------------- untimed pre-heat: a short --steps run must not time a chip that is still ramping ----------------
    sampler, step = forward_only(t, pts_d, False)
    with torch.no_grad():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < PREHEAT_S:
            for _ in range(10):
                step()
            torch.cuda.synchronize(dev)
        preheat_ms = (time.perf_counter() - t0) * 1e3
        # ---------------- the headline: K cold steps ----------------
        dt = timed_steps(step, a.warmup, a.steps)
        host_cold = host_issue[0]
        ms_per_step = dt / a.steps * 1e3
        value = M * world / (dt / a.steps)
        # ---------------- the same with the samples half of the plan reused ----------------
        sampler_w, step_w = forward_only(t, pts_d, True)
        settle(step_w)
        dtw = timed_steps(step_w, a.warmup, a.steps)
        warm = {"value": M * world / (dtw / a.steps), "ms_per_step": dtw / a.steps * 1e3,
                "what": "the same K steps with preprocess() handed the same unmodified samples tensor every time: the "
                        "sorted sample structure is reused, the Gaussians are re-binned and the tile lists rebuilt"}

    # ---------------- dominant kernel: HIP events on the launch stream ----------------
    def kernel_ms(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize(dev)
        gc.disable()                    # a collector pause on the host would leave the device idle between the events
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        gc.enable()
        return e0.elapsed_time(e1) / n

    def kernel_in_step_ms(pre, sample, n):
        """The forward launch WHERE IT RUNS: n cold steps, HIP events around the step's sample() -- the device is
        never idle in such a loop (the host issues a step in a third of the time the device takes), so the events
        bracket the launch; the slowest quarter (a host hiccup between two records is not the kernel) is dropped.
        Back-to-back launches of the forward alone on one plan -- round 3's figure -- run 25-33 us on boxes where
        the kernel inside a step takes 27.5-28.6 under rocprofv3: each starts on its predecessor's write-back."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        pre(); sample()
        torch.cuda.synchronize(dev)
        gc.disable()
        for e0, e1 in evs:
            pre()
            e0.record()
            sample()
            e1.record()
        torch.cuda.synchronize(dev)
        gc.enable()
        ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        keep = max(1, len(ms) * 3 // 4)
        return sum(ms[:keep]) / keep

    def roofline_of(smp, kappa, tt=None, pp=None):
        means, values, conics, samples = smp._inputs
        plan = smp._plan
        with torch.no_grad():
            if plan is not None and tt is not None:
                k_ms = kernel_in_step_ms(lambda: smp.preprocess(tt["means"], tt["values"], tt["covariances"], tt["conics"], pp),
                                         lambda: smp.sample((0, 1, 2)), max(5, min(a.steps, 200)))
                plan = smp._plan
            else:
                k_ms = kernel_ms(lambda: S.forward_raw(means, values, conics, samples, 7, plan), a.steps)
        algo_bytes = 24 * N + 36 * M       # fp32, d=2, c=1: 6 floats/Gaussian + (2 in + 7 out) floats/point
        achieved = algo_bytes / (k_ms * 1e-3)
        traffic, source = measured_traffic(a.workload, kappa, plan is not None)
        r = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
             "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": source,
             "kernel": ("tile_forward_kernel<1,7>" if plan is not None else "dense_forward_kernel<float,2,1,7,4>"),
             "kernel_ms": k_ms, "algorithmic_bytes": algo_bytes}
        if plan is None:
            r["valu"] = {"pairs": N * M, "frac_of_pair_peak": N * M / (k_ms * 1e-3) / VALU_PAIR_PEAK}
        else:
            pr = plan_rows(plan)
            if pr is not None:
                instr = pr["wave_rows"] * ROW_VALU_INSTR
                r["valu"] = {
                    "pairs": pr["pairs"], "wave_rows": pr["wave_rows"], "pairs_per_point": pr["pairs"] / M,
                    "frac_of_pair_peak": pr["pairs"] / (k_ms * 1e-3) / VALU_PAIR_PEAK,
                    "issue_frac": instr / (k_ms * 1e-3) / VALU_ISSUE_PEAK,
                    "row_loop_floor_ms": pr["wave_rows"] * ROW_NS_UBENCH * 1e-6 / 1024,
                    "ceiling_frac": algo_bytes / (instr / VALU_ISSUE_PEAK) / HBM_PEAK,
                    "what": "pairs = (point, Gaussian) evaluations of one launch, from the tile headers; issue_frac = "
                            f"wave_rows x {ROW_VALU_INSTR} VALU instructions (ISA count of the row loop) / kernel time / "
                            "(1024 SIMDs x 1 instruction per 2 cycles x 2.4 GHz); row_loop_floor_ms = wave_rows x the "
                            f"{ROW_NS_UBENCH} ns a row costs in the bare row loop at 8 waves/SIMD (tools/ubench/rowloop.hip) "
                            "/ 1024 SIMDs: what the launch would take if it were only that loop; ceiling_frac = the HBM-roofline "
                            "fraction this float32-VALU formulation could reach at this density if the launch were nothing "
                            "but the row loop's instructions at the 2-cycle issue peak (read it beside `frac`)"}
        return r

    roofline = roofline_of(sampler, a.kappa, t, pts_d)
    if sampler._plan is not None:
        roofline["kernel_ms_how"] = ("mean over the timed kind of step (cold) of HIP events around the step's forward launch on its stream, "
                                     "slowest quarter dropped (bench.py, kernel_in_step_ms)")
    binned = sampler._plan is not None

    # ---------------- A/B: the tile lists deferred into the first forward's launch (round 4, an option) ----------------
    # GaussianSampler(defer_lists=True): preprocess() stops in front of the tile lists (PIGS_BUILD_DEFER_LISTS) and the
    # step's sample() builds them in the SAME launch as its evaluation (plan_lists_forward_kernel<1,7>).  Measured, not
    # the default: the cold step with it, and HIP events around that one launch.
    deferred = None
    if binned and not a.no_extras:
        try:
            with torch.no_grad():
                smp_d = GaussianSampler(False, fuse="all", backend=a.backend, reuse_samples=False, defer_lists=True)

                def step_d():
                    smp_d.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts_d)
                    return smp_d.sample((0, 1, 2))
                settle(step_d)
                n1 = max(5, min(a.steps, 100))
                dtd = timed_steps(step_d, min(a.warmup, 10), n1)
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n1)]
                gc.disable()
                for e0, e1 in evs:
                    smp_d.preprocess(t["means"], t["values"], t["covariances"], t["conics"], pts_d)
                    e0.record()
                    smp_d.sample((0, 1, 2))
                    e1.record()
                torch.cuda.synchronize(dev)
                gc.enable()
                f_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
                f_ms = sum(f_ms[: max(1, len(f_ms) * 3 // 4)]) / max(1, len(f_ms) * 3 // 4)      # a host hiccup between the two records is not the kernel
            deferred = {"cold_ms_per_step": dtd / n1 * 1e3, "value": M * world / (dtd / n1), "kernel": "plan_lists_forward_kernel<1,7>",
                        "kernel_ms": f_ms, "steps": n1,
                        "what": "the cold step with the tile-list build deferred into the first forward's launch (one launch instead of "
                                "plan_lists_kernel + tile_forward_kernel<1,7>); kernel_ms: HIP events around that launch"}
            del smp_d
        except Exception as e:
            deferred = {"error": f"{type(e).__name__}: {e}"[:200]}

    # ---------------- fwd + bwd step (second half of BASELINE.json's metric) ----------------
    fwd_bwd = roofline_bwd = None
    if not a.no_bwd:
        req = {k: t[k].clone().requires_grad_(True) for k in ("means", "values", "conics")}
        from pigs_amd.distributed import replicated
        gouts = None

        def train_step():
            # parameter grads are summed over the ranks by ONE all-reduce of a packed [N,6] buffer
            m_r, v_r, c_r = replicated(req["means"], req["values"], req["conics"])
            sampler_w.preprocess(m_r, v_r, t["covariances"], c_r, pts_d)
            u, ux, uxx = sampler_w.sample((0, 1, 2))
            # diffusion residual shape of test_no_mlp.py:144 (u_t replaced by u: same data flow)
            loss = ((u[:, 0] - (uxx[:, 0, 0, 0] + uxx[:, 1, 1, 0])) ** 2).mean() + (ux ** 2).mean()
            return torch.autograd.grad(loss, list(req.values()))

        def sampler_step():
            # the sampler's own share of a training step: incoming gradients supplied, no loss kernels
            nonlocal gouts
            m_r, v_r, c_r = replicated(req["means"], req["values"], req["conics"])
            sampler_w.preprocess(m_r, v_r, t["covariances"], c_r, pts_d)
            outs = sampler_w.sample((0, 1, 2))
            if gouts is None:
                gouts = tuple(torch.randn_like(o) for o in outs)
            return torch.autograd.grad(outs, list(req.values()), grad_outputs=gouts)

        def trace_step():
            # the same residual through the fused Hessian-trace output (4 floats per point instead of 7,
            # no slicing of [M,2,2,1] in the loss): extension of the reference API, SURVEY.md 8f-4
            m_r, v_r, c_r = replicated(req["means"], req["values"], req["conics"])
            sampler_w.preprocess(m_r, v_r, t["covariances"], c_r, pts_d)
            u, ux, lap = sampler_w.sample((0, 1, "lap"))
            loss = ((u - lap) ** 2).mean() + (ux ** 2).mean()
            return torch.autograd.grad(loss, list(req.values()))

        def residual_step():
            # the diffusion residual of test_no_mlp.py:144 (u_t replaced by u) in ONE forward launch (4 B per point
            # out) and one backward launch: sampler.residual() -- extension of the reference API, SURVEY.md 8f-4
            m_r, v_r, c_r = replicated(req["means"], req["values"], req["conics"])
            sampler_w.preprocess(m_r, v_r, t["covariances"], c_r, pts_d)
            loss = sampler_w.residual(a0=1.0, lap=-1.0).pow(2).mean()
            return torch.autograd.grad(loss, list(req.values()))

        nb = max(1, min(a.steps, 100))
        wb = min(10, nb)        # eager steps settle after a few iterations (caching allocator, autograd graph reuse)
        sampler_only_s = timed_steps(sampler_step, wb, nb) / nb
        host_sampler_only = host_issue[0]
        fwd_bwd = {"ms_per_step": timed_steps(train_step, wb, nb) / nb * 1e3,
                   "sampler_only_ms_per_step": sampler_only_s * 1e3,
                   "sampler_only_host_issue_ms_per_step": host_sampler_only * 1e3,
                   "trace_residual_ms_per_step": timed_steps(trace_step, wb, nb) / nb * 1e3,
                   "fused_residual_ms_per_step": timed_steps(residual_step, wb, nb) / nb * 1e3, "steps": nb,
                   "value": M * world / sampler_only_s,
                   "dist_backend": (dist.get_backend() if dist is not None else None), "world_size": world,
                   "what": "value: points/s of the sampler-only step = preprocess (samples half reused) + fused fwd(0..2) "
                           "+ fused bwd with the incoming gradients supplied"
                           + (" + ONE all-reduce of the packed [N,6] parameter gradients over the ranks"
                              if dist is not None else "")
                           + "; ms_per_step: the same with a torch residual loss; trace_residual: the training step "
                           "with the fused u, grad u, u_xx+u_yy outputs (sample((0, 1, 'lap'))) instead of the full "
                           "Hessian; fused_residual: the training step of the residual u - lap u through "
                           "sampler.residual() (one forward launch writing 4 B per point, one backward launch, the loss "
                           "one pow + mean); hipgraph_replay (1 GPU): the same steps captured once (static_samples: the sorted "
                           "sample structure is reused, as in the eager steps) and replayed"}
        # backward kernel alone
        means, values, conics, samples = sampler_w._inputs
        plan = sampler_w._plan
        with torch.no_grad():
            go = [torch.randn((M,) + (2,) * k + (1,), device=dev) for k in range(3)] + [None, None]
            kb_ms = kernel_ms(lambda: S.backward_raw(means, values, conics, samples, go, 7, plan), max(5, nb))
        bwd_bytes = 48 * N + 36 * M
        btraffic, bsource = measured_traffic(a.workload, a.kappa, plan is not None, "backward")
        roofline_bwd = {"bound": "hbm", "achieved": bwd_bytes / (kb_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": bwd_bytes / (kb_ms * 1e-3) / HBM_PEAK, "traffic": btraffic,
                        "traffic_source": bsource, "q_max_backward": sampler_w.q_max_backward,
                        "kernel": ("tile_backward_kernel<1,7> + plan_unpermute_kernel<1>" if plan is not None
                                   else "dense_backward_kernel"),
                        "kernel_ms": kb_ms, "algorithmic_bytes": bwd_bytes}
        if dist is None and not a.no_extras:
            # the same two steps captured ONCE into a hipGraph and replayed (no entry point allocates or
            # synchronises): what a training loop pays when the host is taken out of the way
            try:
                fwd_bwd["hipgraph_replay"] = graph_replay(GaussianSampler, t, pts_d, a.backend, nb)
            except Exception as e:            # report, never lose the bench line over the extra figure
                fwd_bwd["hipgraph_replay"] = {"error": f"{type(e).__name__}: {e}"[:200]}

    # ---------------- extras ----------------
    two_streams = kappa13 = small = None
    if not a.no_extras and dist is None:
        try:
            small = small_record(GaussianSampler, dev)
        except Exception as e:                # report, never lose the bench line over the extra figure
            small = {"error": f"{type(e).__name__}: {e}"[:200]}
    if not a.no_extras:
        # the same steps dealt round-robin to two HIP streams (extra figure, not `value`: independent evaluation
        # steps -- frames of a roll-out -- can overlap the latency-bound plan build of one with the forward of another)
        streams = [torch.cuda.Stream(dev) for _ in range(2)]
        samplers2 = [GaussianSampler(False, fuse="all", backend=a.backend, reuse_samples=False) for _ in range(2)]
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
            settle(lambda: run2(2))
            barrier()
            gc.disable()
            t0 = time.perf_counter()
            run2(a.steps)
            barrier()
            dt2 = max_over_ranks(time.perf_counter() - t0)
            gc.enable()
        two_streams = {"ms_per_step": dt2 / a.steps * 1e3, "value": M * world / (dt2 / a.steps),
                       "what": "the same K cold steps issued round-robin on two HIP streams (one sampler per stream)"}
        del keep, samplers2
        if abs(a.kappa - 1.3) > 1e-9 and a.workload != "c4":
            # the reference-like width (report both): kappa = 1.3, ~190 Gaussians within the cut-off
            gs13, pts13, _, _ = workload(1.3)
            t13 = {k: v.float().to(dev) for k, v in gs13.items()}
            s13, step13 = forward_only(t13, pts_d, False)
            with torch.no_grad():
                n13 = max(5, a.steps // 4)
                settle(step13)
                d13 = timed_steps(step13, 3, n13)
            r13 = roofline_of(s13, 1.3, t13, pts_d)
            kappa13 = {"value": M * world / (d13 / n13), "ms_per_step": d13 / n13 * 1e3, "steps": n13,
                       "kernel_ms": r13["kernel_ms"], "frac": r13["frac"], "valu": r13.get("valu")}
            if not a.no_bwd:
                m13, v13, c13, sm13 = s13._inputs
                with torch.no_grad():
                    go13 = [torch.randn((M,) + (2,) * k + (1,), device=dev) for k in range(3)] + [None, None]
                    kappa13["bwd_kernel_ms"] = kernel_ms(lambda: S.backward_raw(m13, v13, c13, sm13, go13, 7, s13._plan), 5)
                kappa13["bwd_frac"] = (48 * N + 36 * M) / (kappa13["bwd_kernel_ms"] * 1e-3) / HBM_PEAK

    c2 = None
    if not a.no_extras and dist is None and a.workload == "c3":
        # BASELINE configs[1]: 8k Gaussians x 256^2 grid, fwd + deriv + bwd on one GPU (same kappa)
        try:
            gs2 = synthetic.lattice_gaussians(128, 64, a.kappa, seed=0)
            t2 = {k: v.float().to(dev) for k, v in gs2.items()}
            p2 = synthetic.grid_samples(256, 256).float().to(dev)
            s2, step2 = forward_only(t2, p2, False)
            with torch.no_grad():
                settle(step2)
                d2 = min(timed_steps(step2, 10, 100) for _ in range(2)) / 100       # (a step this short is bound by the host's issue rate: the quieter of two runs)
            req2 = {k: t2[k].clone().requires_grad_(True) for k in ("means", "values", "conics")}
            sw2 = GaussianSampler(False, fuse="all", backend=a.backend)
            g2 = []

            def fb2():
                sw2.preprocess(req2["means"], req2["values"], t2["covariances"], req2["conics"], p2)
                outs = sw2.sample((0, 1, 2))
                if not g2:
                    g2.extend(torch.randn_like(o) for o in outs)
                return torch.autograd.grad(outs, list(req2.values()), grad_outputs=g2)
            settle(fb2)
            dfb = min(timed_steps(fb2, 10, 100) for _ in range(2)) / 100
            c2 = {"workload": "c2: 8192 Gaussians x 256x256 grid, d=2, c=1, orders 0-2", "path": "binned" if s2._plan is not None else "dense",
                  "value": p2.shape[0] / d2, "ms_per_step": d2 * 1e3, "fwd_bwd_sampler_only_ms_per_step": dfb * 1e3,
                  "what": "value: cold preprocess + fused forward (orders 0..2); fwd_bwd: preprocess (samples half reused) + "
                          "fused forward + fused backward, incoming gradients supplied, eager; each the quieter of two runs of 100 steps"}
        except Exception as e:
            c2 = {"error": f"{type(e).__name__}: {e}"[:200]}

    unordered = None
    if not a.no_extras and dist is None and a.workload == "c3":
        # the same Gaussians sampled at M uniform random points (torch.rand collocation points, main_pn.py:103): the
        # library switches to the coarse-bin samples build and to staged outputs / gradients once it has seen how a
        # point set of this size arrives -- extra figures, not `value`
        try:
            gen = torch.Generator().manual_seed(3)
            pr = (torch.rand((M, 2), generator=gen) * 2 - 1).to(dev)
            sr, stepr = forward_only(t, pr, False)
            srw, steprw = forward_only(t, pr, True)
            with torch.no_grad():
                settle(stepr)
                dc = timed_steps(stepr, 10, 100) / 100
                settle(steprw)
                dw = timed_steps(steprw, 10, 100) / 100
                mr, vr, cr, smr = srw._inputs
                gor = [torch.randn((M,) + (2,) * k + (1,), device=dev) for k in range(3)] + [None, None]
                kf = kernel_ms(lambda: S.forward_raw(mr, vr, cr, smr, 7, srw._plan), 10)
                kb = kernel_ms(lambda: S.backward_raw(mr, vr, cr, smr, gor, 7, srw._plan), 5)
            from pigs_amd import _lib
            unordered = {"points": "uniform random in [-1, 1]^2, M = %d" % M, "cold_ms_per_step": dc * 1e3, "warm_ms_per_step": dw * 1e3,
                         "forward_launches_ms": kf, "backward_launches_ms": kb, "value_cold": M / dc,
                         "library_took": {1: "coarse-bin samples build, staged outputs and gradients", 0: "one-pass samples build",
                                          -1: "no record"}[_lib.load().pigs_samples_order_hint(M)],
                         "what": "cold / warm step as `value` / `value_warm_plan`; forward / backward: all launches of one "
                                 "sampler.sample((0, 1, 2)) / its backward on a built plan (tile kernel + the staging launch)"}
        except Exception as e:
            unordered = {"error": f"{type(e).__name__}: {e}"[:200]}

    line = {
        "metric": "sample-points/sec (fwd + 1st + 2nd derivatives, fused)", "value": value,
        "unit": "sample-points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if a.workload == "c4" else "weak",
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (seeded lattice Gaussians, regular sample grid)",
        "config": {"workload": f"{a.workload}: {N} Gaussians x {side}x{side} grid, {rows} rows x {side} points per "
                               f"GPU, d=2, c=1, kappa={a.kappa}, orders 0-2", "gaussians": N, "points_per_gpu": M,
                   "kappa": a.kappa, "path": "binned" if binned else "dense",
                   "step": "preprocess (cold: nothing reused) + fused forward (orders 0..2)"},
        "host_issue_ms_per_step": host_cold * 1e3,
        "preheat_ms": preheat_ms, "value_warm_plan": warm["value"], "warm_plan": warm,
        "roofline": roofline, "roofline_bwd": roofline_bwd, "deferred_lists": deferred, "fwd_bwd": fwd_bwd, "two_streams": two_streams,
        "kappa_1_3": kappa13, "small": small, "c2": c2, "unordered_points": unordered, "host": sampler.host,
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

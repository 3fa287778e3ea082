"""Worker of tests/test_distributed_gpu.py: one of two gloo ranks that share the one GPU of the test box.  Each
rank samples ITS shard of the points through the HIP sampler with the Gaussians wrapped by
pigs_amd.distributed.replicated(); the all-reduced gradients must equal the unsharded HIP gradients.
Started by torch.distributed.run (never imported by pytest)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from diff_gaussian_sampling import GaussianSampler
    from pigs_amd import synthetic, distributed as D

    seen = []
    orig = D._single_buffer
    D._single_buffer = lambda grads: seen.append(orig(grads) is not None) or orig(grads)

    gs = synthetic.lattice_gaussians(32, 32, 0.7, seed=4)
    pts = synthetic.grid_samples(96).float().to(dev)
    M = pts.shape[0]
    gen = torch.Generator().manual_seed(21)
    weights = [torch.rand((M,) + (2,) * k + (1,), generator=gen).to(dev) for k in range(3)]
    report = {"rank": rank, "world": world}
    for backend in ("binned", "dense"):
        for host in ("native", "ctypes"):
            req = {k: gs[k].float().to(dev).requires_grad_(True) for k in ("means", "values", "conics")}
            m_r, v_r, c_r = D.replicated(req["means"], req["values"], req["conics"])
            sl = D.shard_rows(M)
            s = GaussianSampler(False, backend=backend, fuse="all", host=host)
            s.preprocess(m_r, v_r, None, c_r, pts[sl].contiguous())
            outs = s.sample((0, 1, 2))
            sum((o * w[sl]).sum() for o, w in zip(outs, weights)).backward()
            ref = {k: gs[k].float().to(dev).requires_grad_(True) for k in ("means", "values", "conics")}
            s2 = GaussianSampler(False, backend=backend, fuse="all", host=host)
            s2.preprocess(ref["means"], ref["values"], None, ref["conics"], pts)
            outs2 = s2.sample((0, 1, 2))
            sum((o * w).sum() for o, w in zip(outs2, weights)).backward()
            key = f"{backend}/{host}"
            report[key] = {
                "out": max(rel(o, o2[sl]) for o, o2 in zip(outs, outs2)),
                "grad": max(rel(req[k].grad, ref[k].grad) for k in req),
            }
    report["single_buffer"] = seen
    # every rank holds the same reduced gradients: compare a checksum across the ranks
    chk = torch.stack([req[k].grad.double().sum() for k in req]).cpu()
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    report["ranks_agree"] = bool(all(torch.equal(both[0], b) for b in both))
    gathered = [None] * world
    dist.all_gather_object(gathered, report)
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(gathered, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

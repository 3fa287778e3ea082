"""CPU, world_size 2, gloo: the multi-GPU path of the sampler (sample sharding + ONE all-reduce of
the packed parameter gradients).  The GPU kernels cannot run here, so the per-rank sampling is
stood in for by the differentiable torch oracle -- this test is about the sharding and the
collective, which are backend independent."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, path, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import dense_torch
        from pigs_amd import distributed as D
        z = np.load(path)
        means, values, conics, samples = (torch.from_numpy(z[k]) for k in ("means", "values", "conics", "samples"))
        for t in (means, values, conics):
            t.requires_grad_(True)
        M = samples.shape[0]
        sl = D.shard_rows(M)
        m_r, v_r, c_r = D.replicated(means, values, conics)
        outs = dense_torch.forward(m_r, c_r, v_r, samples[sl], orders=(0, 1, 2))
        loss = sum((outs[o] * torch.from_numpy(z[f"r{o}"])[sl]).sum() for o in range(3))
        loss.backward()
        out_q.put((rank, sl.start, sl.stop, means.grad.numpy(), values.grad.numpy(), conics.grad.numpy(),
                   {o: outs[o].detach().numpy() for o in range(3)}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["random_d2_c2.npz", "ref_test_1d.npz"])
def test_sharded_forward_and_allreduced_grads(name):
    path = os.path.join(GOLDEN, name)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    z = np.load(path)
    # shards tile the points exactly
    assert results[0][1] == 0 and results[0][2] == results[1][1] and results[1][2] == z["samples"].shape[0]
    # forward: concatenated shards == unsharded reference outputs
    for o in range(3):
        got = np.concatenate([r[6][o] for r in results], axis=0)
        assert np.abs(got - z[f"out{o}_f64"]).max() <= 1e-12 * np.abs(z[f"out{o}_f64"]).max()
    # backward: every rank holds the SAME, fully reduced gradient == unsharded reference gradient
    for key, idx in (("gmeans", 3), ("gvalues", 4), ("gconics", 5)):
        exp = sum(z[f"{key}{o}_f64"] for o in range(3))
        for r in results:
            assert np.abs(r[idx] - exp).max() <= 1e-11 * np.abs(exp).max(), (key, r[0])
        assert np.array_equal(results[0][idx], results[1][idx])


def _worker_views(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pigs_amd import distributed as D
        g = torch.Generator().manual_seed(7)
        means = torch.randn((9, 2), generator=g, dtype=torch.float64).requires_grad_(True)
        values = torch.randn((9, 2), generator=g, dtype=torch.float64).requires_grad_(True)
        conics = torch.randn((9, 3), generator=g, dtype=torch.float64).requires_grad_(True)
        w = torch.randn((3, 9), generator=g, dtype=torch.float64)
        m_r, v_r, c_r = D.replicated(means, values, conics)
        # gradients that reach the packing as non-contiguous views: an expanded scalar (stride 0) for the
        # means, a transposed matrix for the conics; the values get none at all on rank 1
        loss = (rank + 1) * m_r.sum() + (c_r.t() * w).sum() * (2 - rank)
        if rank == 0:
            loss = loss + (v_r[:, 1] ** 2).sum()
        loss.backward()
        out_q.put((rank, means.grad.numpy(), values.grad.numpy(), conics.grad.numpy(), values.detach().numpy(), w.numpy()))
    finally:
        dist.destroy_process_group()


def test_packed_allreduce_with_noncontiguous_and_missing_grads():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_views, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    values, w = results[0][4], results[0][5]
    exp_means = np.full((9, 2), 1.0 + 2.0)
    exp_values = np.zeros((9, 2))
    exp_values[:, 1] = 2 * values[:, 1]
    exp_conics = w.T * (2 + 1)
    for r in results:
        assert np.allclose(r[1], exp_means, rtol=0, atol=1e-14)
        assert np.allclose(r[2], exp_values, rtol=0, atol=1e-14)
        assert np.allclose(r[3], exp_conics, rtol=0, atol=1e-14)


def test_shard_bounds_cover_everything():
    from pigs_amd.distributed import shard_bounds
    for n in (0, 1, 7, 1024, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_replicated_is_identity_without_process_group():
    from pigs_amd.distributed import replicated
    a = torch.randn(5, 2, requires_grad=True)
    b = torch.randn(5, 1, requires_grad=True)
    c = torch.randn(5, 3, requires_grad=True)
    ar, br, cr = replicated(a, b, c)
    (ar.sum() * 2 + (br * 3).sum() + (cr * cr).sum()).backward()
    assert torch.allclose(a.grad, torch.full_like(a, 2.0))
    assert torch.allclose(b.grad, torch.full_like(b, 3.0))
    assert torch.allclose(c.grad, 2 * c.detach())


class _FlatGrads(torch.autograd.Function):
    """Stands in for the sampler's backward: the three gradients are views of ONE flat allocation
    [means | values | conics] (pigs_amd/sampler.py::_gradient_views, csrc_host/pigs_host.cpp::gradient_views)."""

    @staticmethod
    def forward(ctx, means, values, conics, w):
        ctx.save_for_backward(means, values, conics, w)
        return (means.sum(1) * w).sum() + (values.sum(1) * w).sum() * 2 + (conics.sum(1) * w).sum() * 3

    @staticmethod
    def backward(ctx, g):
        means, values, conics, w = ctx.saved_tensors
        nm, nv, nc = means.numel(), values.numel(), conics.numel()
        flat = torch.empty(nm + nv + nc, dtype=means.dtype)
        gm, gv, gc = flat[:nm].view(means.shape), flat[nm:nm + nv].view(values.shape), flat[nm + nv:].view(conics.shape)
        gm.copy_((w * g)[:, None].expand_as(means))
        gv.copy_((2 * w * g)[:, None].expand_as(values))
        gc.copy_((3 * w * g)[:, None].expand_as(conics))
        return gm, gv, gc, None


def _worker_flat(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pigs_amd import distributed as D
        g = torch.Generator().manual_seed(3)
        means = torch.randn((11, 2), generator=g, dtype=torch.float64).requires_grad_(True)
        values = torch.randn((11, 1), generator=g, dtype=torch.float64).requires_grad_(True)
        conics = torch.randn((11, 3), generator=g, dtype=torch.float64).requires_grad_(True)
        w = torch.randn((11,), generator=g, dtype=torch.float64)
        seen = []
        orig = D._single_buffer
        D._single_buffer = lambda grads: seen.append(orig(grads)) or seen[-1]
        m_r, v_r, c_r = D.replicated(means, values, conics)
        loss = _FlatGrads.apply(m_r, v_r, c_r, w) * (rank + 1)
        if rank == 1:
            loss = loss + (m_r ** 2).sum()          # a second consumer: autograd sums before the collective -> the packed path
        loss.backward()
        out_q.put((rank, seen[0] is not None, means.grad.numpy(), values.grad.numpy(), conics.grad.numpy(),
                   means.detach().numpy(), w.numpy()))
    finally:
        dist.destroy_process_group()


def test_single_buffer_gradients_are_reduced_in_place_and_mix_with_the_packed_path():
    """The sampler's backward hands out views of one flat buffer: rank 0 all-reduces it in place; rank 1's
    gradients were summed by autograd on the way (two consumers) and take the packing path -- same layout, same
    collective, same result on both."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_flat, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1] and not results[1][1]          # in place on rank 0, packed on rank 1
    means, w = results[0][5], results[0][6]
    exp_m = np.repeat(w[:, None], 2, 1) * 3 + 2 * means
    exp_v = np.repeat(w[:, None], 1, 1) * 2 * 3
    exp_c = np.repeat(w[:, None], 3, 1) * 3 * 3
    for r in results:
        assert np.allclose(r[2], exp_m, rtol=0, atol=1e-13)
        assert np.allclose(r[3], exp_v, rtol=0, atol=1e-13)
        assert np.allclose(r[4], exp_c, rtol=0, atol=1e-13)

"""Multi-GPU execution of the sampler: one process per GPU, sample points sharded, Gaussians
replicated (the reference itself is single-GPU).

The path shards by sample points: forward values and derivatives of different points are
independent, so each rank samples its own block of points with no communication.  Only the
backward has an exchange step: every rank holds a partial gradient of the shared Gaussian
parameters, summed with ONE all-reduce of a packed [N, d + d(d+1)/2 + c] buffer (1.5 MB at 65k
Gaussians) over RCCL (``torch.distributed`` backend "nccl" on ROCm) -- or gloo on CPU in tests.

    means_r, values_r, conics_r = replicated(means, values, conics)      # identity in forward
    sampler.preprocess(means_r, values_r, covariances, conics_r, samples[shard_rows(M, world, rank)])
    loss = f(sampler.sample_gaussians(), ...)                              # local points only
    loss.backward()                      # parameter grads = sum over ranks (one all-reduce)
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world_size, rank):
    """[begin, end) of the contiguous block of items owned by ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(n_items, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_rows(n_items, world_size=None, rank=None):
    """``slice`` of this rank's contiguous block of ``n_items`` sample points (or grid rows)."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    return slice(*shard_bounds(n_items, world_size, rank))


def _numel(shape):
    n = 1
    for k in shape:
        n *= k
    return n


class _Replicated(torch.autograd.Function):
    """Identity on the replicated parameters; the backward sums their gradients over the ranks
    with a single all-reduce of one packed buffer."""

    @staticmethod
    def forward(ctx, group, *params):
        ctx.group = group
        ctx.shapes = [tuple(p.shape) for p in params]
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    def backward(ctx, *grads):
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(ctx.group) > 1):
            return (None, *grads)          # single process: nothing to sum, no packing copy either
        if all(g is None for g in grads):
            return (None, *grads)
        # one packed [N, d + d(d+1)/2 + c] buffer: whatever arrives (expanded, transposed, sliced views, or
        # nothing at all for a parameter the loss did not touch) is laid out contiguously before the ONE
        # collective; every rank packs the same layout, so the ranks may differ in which gradients exist
        shapes = ctx.shapes
        n = shapes[0][0]
        ref = next(g for g in grads if g is not None)
        cols = [g.reshape(n, -1) if g is not None else ref.new_zeros((n, max(1, _numel(shp) // max(n, 1))))
                for g, shp in zip(grads, shapes)]
        packed = torch.cat(cols, dim=1).contiguous()
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=ctx.group)
        out, col = [], 0
        for shp in shapes:
            w = _numel(shp) // max(n, 1)
            out.append(packed[:, col:col + w].reshape(shp))
            col += w
        return (None, *out)


def replicated(means, values, conics, group=None):
    """Mark the Gaussian parameters as replicated over the process group: returns views whose
    gradients are all-reduced (summed) over the ranks during backward.  All three must have the
    same leading dimension N, dtype and device."""
    if not (means.shape[0] == values.shape[0] == conics.shape[0]):
        raise ValueError("means, values and conics must share their leading dimension N")
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        # a single process: nothing to sum, and no autograd node in the eager step's host path either
        return means, values.reshape(values.shape[0], -1), conics.reshape(conics.shape[0], -1)
    return _Replicated.apply(group, means, values.reshape(values.shape[0], -1), conics.reshape(conics.shape[0], -1))

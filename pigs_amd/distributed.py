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


def _single_buffer(grads):
    """The flat 1-D buffer that the gradients tile exactly, in order -- what the sampler's backward hands out (one
    allocation [d means | c values | d(d+1)/2 conics per Gaussian, block after block], the three gradients views of
    it) -- or None when they arrive any other way (accumulated by autograd, produced by torch ops, missing)."""
    if any(g is None for g in grads):
        return None
    base = grads[0]._base
    if base is None or base.dim() != 1 or not base.is_contiguous():
        return None
    off = base.storage_offset()
    for g in grads:
        if g._base is not base or not g.is_contiguous() or g.storage_offset() != off:
            return None
        off += g.numel()
    return base if off == base.storage_offset() + base.numel() else None


class _Replicated(torch.autograd.Function):
    """Identity on the replicated parameters; the backward sums their gradients over the ranks
    with a single all-reduce of one flat buffer."""

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
        # ONE collective over one flat buffer laid out block after block: [means | values | conics].  The sampler's
        # backward writes its three gradients into exactly that buffer (one allocation, three views): it is reduced
        # in place -- no copy in, no copy out.  Whatever arrives any other way (expanded, transposed or sliced views,
        # sums autograd accumulated, or nothing at all for a parameter the loss did not touch) is packed into the
        # same layout first; the layout being the same, the ranks may differ in which path they take.
        flat = _single_buffer(grads)
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
            return (None, *grads)
        shapes = ctx.shapes
        ref = next(g for g in grads if g is not None)
        flat = torch.cat([g.reshape(-1) if g is not None else ref.new_zeros(_numel(shp)) for g, shp in zip(grads, shapes)])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
        out, off = [], 0
        for shp in shapes:
            n = _numel(shp)
            out.append(flat[off:off + n].view(shp))
            off += n
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

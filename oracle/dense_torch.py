"""Dense PyTorch restatement of the reference's CPU-runnable path (TEST INFRASTRUCTURE, see
oracle/__init__.py).  This is the "reference PyTorch-CPU path" timed as ``cpu_baseline`` by
bench.py: same algorithm as /root/reference/gaussians.py:48-58 (order 0), :89-101 (order 1) and
:103-116 (order 2) -- broadcast the [m, N, d] point-minus-mean differences, batched conic
mat-vec, exp, reduce over the Gaussians -- written independently and chunked over the points so
the [m, N, ...] intermediates stay bounded.  Checked against the fixtures in tests/test_oracle.py.
"""
import torch


def full_conics(flat, d):
    if d == 1:
        return flat.reshape(-1, 1, 1)
    a, b, c = flat[:, 0], flat[:, 1], flat[:, 2]
    return torch.stack((torch.stack((a, b), -1), torch.stack((b, c), -1)), -2)


def forward(means, conics_flat, values, samples, orders=(0, 1, 2), chunk=256):
    """Returns {order: tensor} for orders within 0..2; dtype/device follow the inputs."""
    N, d = means.shape
    C = full_conics(conics_flat.reshape(N, -1), d)           # [N, d, d]
    values = values.reshape(N, -1)
    samples = samples.reshape(-1, d)
    outs = {o: [] for o in orders}
    for m0 in range(0, samples.shape[0], chunk):
        s = samples[m0:m0 + chunk]
        x = s[:, None, :] - means[None, :, :]                # [m, N, d]
        p = torch.einsum("nij,mnj->mni", C, x)
        dens = torch.exp(-0.5 * (x * p).sum(-1))            # [m, N]
        if 0 in outs:
            outs[0].append(dens @ values)
        if 1 in outs:
            outs[1].append(-torch.einsum("mn,mni,nc->mic", dens, p, values))
        if 2 in outs:
            outer = p[..., :, None] * p[..., None, :] - C[None]
            outs[2].append(torch.einsum("mn,mnij,nc->mijc", dens, outer, values))
    return {o: torch.cat(v, 0) for o, v in outs.items()}

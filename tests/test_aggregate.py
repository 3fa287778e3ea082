"""CPU: host logic of aggregate_neighbors.  Its arithmetic is this repo's own
definition (parity unpinned; oracle/aggregate_torch.py is the dense torch checker of pigs_amd/csrc/aggregate.hip); what the reference fixes -- shapes, float64,
differentiability wrt all six arguments (test_neighbor_aggregation.py:75-98) -- is checked here
with the same torch.autograd.gradcheck call and the same sizes."""
import math

import torch

from oracle import aggregate_torch as aggregate


def setup(nx=5, d=2, L=2, K=4, E=21):
    torch.manual_seed(0)
    t = torch.linspace(-1, 1, nx, dtype=torch.float64)
    gx, gy = torch.meshgrid((t, t), indexing="ij")
    means = torch.stack((gx, gy), dim=-1).reshape(nx * nx, d)
    var = math.exp(-1.5)
    conics = torch.tensor([1 / var, 0.0, 1 / var], dtype=torch.float64).repeat(nx * nx, 1)
    F = (E - 1) // d // 2
    N = nx * nx
    args = [torch.rand((N, L)), torch.rand((L, L)), torch.rand((N, K)), torch.rand((N, K)), torch.randn(F) * 10,
            torch.rand((L, 2 * E))]
    return means, conics, [a.double().requires_grad_(True) for a in args]


def test_gradcheck_all_six_arguments():
    means, conics, args = setup()
    nb = aggregate.neighbor_structure(means, conics, 36.0)
    assert torch.autograd.gradcheck(lambda *a: aggregate.aggregate(*nb, *a), args)


def test_shapes_self_neighbour_and_locality():
    means, conics, args = setup(nx=6, L=3)
    mask, delta, g = aggregate.neighbor_structure(means, conics, 4.0)
    assert mask.diagonal().all()                      # every Gaussian reaches its own centre
    assert not mask.all()                             # a tight cut-off leaves a sparse relation
    assert torch.equal(delta, -delta.transpose(0, 1))
    args[0] = torch.rand((36, 3), dtype=torch.float64)
    args[1] = torch.rand((3, 3), dtype=torch.float64)
    args[5] = torch.rand((3, 42), dtype=torch.float64)
    out = aggregate.aggregate(mask, delta, g, *args)
    assert out.shape == (36, 3)
    # a Gaussian whose only neighbour is itself gets exactly its own message
    mask1 = torch.eye(36, dtype=torch.bool)
    out1 = aggregate.aggregate(mask1, delta, g, *args)
    emb_self = torch.cat((torch.tensor([0.0, 1.0] * 10 + [1.0]),) * 2).double()   # sin 0, cos 0, ..., bias; g_ii = 1
    expect = args[0] @ args[1].t() + args[5] @ emb_self
    assert torch.allclose(out1, expect)

"""``preprocess_aggregate`` / ``aggregate_neighbors`` of the reference's sampler surface (SURVEY.md 8f-2;
call sites /root/reference/model_pn.py:257-264, test_neighbor_aggregation.py:75-98), on the HIP
library (pigs_amd/csrc/aggregate.hip behind pigs_aggregate_* of include/pigs_amd.h).

PARITY UNPINNED.  The arithmetic of these two methods exists only in the reference's absent CUDA
source; the call sites fix the signature, the shapes (features [N,L], transform [L,L], queries / keys
[N,K], frequencies [F], distance_transform [L,2E] with E = 2 d F + 1 -> [N,L]), the dtype (float64 in
the reference's gradcheck) and that the result is differentiable wrt all six arguments -- nothing
else.  The definition is this repository's own (DESIGN.md "aggregate_neighbors"; the checker is
oracle/aggregate_torch.py):

* neighbours of Gaussian i = the Gaussians j whose q <= q_max ellipse reaches the centre of i;
* weight a_ij = softmax over the neighbours j of <queries_i, keys_j> / sqrt(K);
* message m_ij = transform @ features_j + distance_transform @ [e_ij ; g_ij e_ij], with e_ij the
  Fourier embedding of mu_j - mu_i and g_ij = exp(-q_ij / 2) the density of Gaussian j at the centre of i;
* out_i = sum_j a_ij m_ij.

The neighbour relation lives in index lists (``NeighborLists``); no [N, N, ...] tensor exists at any
point.  The three small GEMMs of the backward (gout @ [transform | distance_transform], gout^T @ acc)
are torch.matmul; everything per (i, j) pair runs in the kernels.  d = 2, float32 / float64.
"""
import ctypes

import torch

from . import _lib

_DTYPES = {torch.float32: _lib.PIGS_F32, torch.float64: _lib.PIGS_F64}
MAX_NEIGHBORS = {torch.float32: 8192, torch.float64: 4096}      # the backward parks two values per neighbour in LDS


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class NeighborLists:
    """Index lists of the neighbour relation: by rows (the j of an i) and by columns (the i of a j)."""

    def __init__(self, means, conics, q_max):
        lib = _lib.load()
        if means.dim() != 2 or means.shape[1] != 2:
            raise NotImplementedError("aggregate_neighbors is implemented for d = 2")
        if means.dtype not in _DTYPES:
            raise TypeError(f"dtype {means.dtype} is not supported (float32 / float64)")
        self.means = means.detach().contiguous()
        self.conics = conics.detach().reshape(means.shape[0], 3).contiguous()
        self.N = means.shape[0]
        self.cap = max(1, min(self.N, MAX_NEIGHBORS[means.dtype]))
        dev = means.device
        self.row_counts = torch.empty(self.N, dtype=torch.int32, device=dev)
        self.col_counts = torch.empty(self.N, dtype=torch.int32, device=dev)
        self.row_lists = torch.empty((self.N, self.cap), dtype=torch.int32, device=dev)
        self.col_lists = torch.empty((self.N, self.cap), dtype=torch.int32, device=dev)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.pigs_aggregate_lists(_DTYPES[means.dtype], self.N, self.cap, _ptr(self.means), _ptr(self.conics),
                                          float(q_max), _ptr(self.row_counts), _ptr(self.row_lists),
                                          _ptr(self.col_counts), _ptr(self.col_lists), _ptr(self.overflow),
                                          _stream(dev))
        _lib.check(rc, "pigs_aggregate_lists")

    def check(self):
        """Synchronising check (debug mode): a neighbour list longer than its slab was truncated."""
        if int(self.overflow.item()):
            raise _lib.PigsError(f"aggregate: a Gaussian has more than {self.cap} neighbours (list truncated)")


class _Aggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nb, features, transform, queries, keys, frequencies, distance_transform):
        lib = _lib.load()
        N, L = features.shape
        K, F = queries.shape[1], frequencies.shape[0]
        E = 4 * F + 1
        dt = nb.means.dtype
        args = [a.detach().to(dt).contiguous() for a in (features, transform, queries, keys, frequencies, distance_transform)]
        f, tr, q, k, fr, dist = args
        out = torch.empty((N, L), dtype=dt, device=f.device)
        lse = torch.empty(N, dtype=dt, device=f.device)
        acc = torch.empty((N, L + 2 * E), dtype=dt, device=f.device)
        with torch.cuda.device(f.device):
            rc = lib.pigs_aggregate_forward(_DTYPES[dt], N, nb.cap, L, K, F, _ptr(nb.means), _ptr(nb.conics),
                                            _ptr(nb.row_counts), _ptr(nb.row_lists), _ptr(f), _ptr(tr), _ptr(q), _ptr(k),
                                            _ptr(fr), _ptr(dist), _ptr(out), _ptr(lse), _ptr(acc), _stream(f.device))
        _lib.check(rc, "pigs_aggregate_forward")
        ctx.nb = nb
        ctx.save_for_backward(f, tr, q, k, fr, dist, lse, acc)
        ctx.dims = (N, L, K, F, E)
        ctx.in_dtypes = tuple(a.dtype for a in (features, transform, queries, keys, frequencies, distance_transform))
        return out.to(features.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        lib = _lib.load()
        nb = ctx.nb
        f, tr, q, k, fr, dist, lse, acc = ctx.saved_tensors
        N, L, K, F, E = ctx.dims
        dt = f.dtype
        gout = gout.to(dt).contiguous()
        dacc = gout @ torch.cat((tr, dist), dim=1)                 # [N, L + 2E]
        D = (dacc * acc).sum(dim=1).contiguous()
        g_f = torch.empty_like(f)
        g_q = torch.empty_like(q)
        g_k = torch.empty_like(k)
        g_fr_rows = torch.empty((N, F), dtype=dt, device=f.device)
        with torch.cuda.device(f.device):
            rc = lib.pigs_aggregate_backward(_DTYPES[dt], N, nb.cap, L, K, F, _ptr(nb.means), _ptr(nb.conics),
                                             _ptr(nb.row_counts), _ptr(nb.row_lists), _ptr(nb.col_counts),
                                             _ptr(nb.col_lists), _ptr(f), _ptr(q), _ptr(k), _ptr(fr), _ptr(lse),
                                             _ptr(dacc), _ptr(D), _ptr(g_f), _ptr(g_q), _ptr(g_k), _ptr(g_fr_rows),
                                             _stream(f.device))
        _lib.check(rc, "pigs_aggregate_backward")
        g_tr = gout.t() @ acc[:, :L]
        g_dist = gout.t() @ acc[:, L:]
        grads = (g_f, g_tr, g_q, g_k, g_fr_rows.sum(dim=0), g_dist)
        return (None,) + tuple(g.to(d) for g, d in zip(grads, ctx.in_dtypes))


def aggregate(nb, features, transform, queries, keys, frequencies, distance_transform):
    N = nb.N
    if features.dim() != 2 or features.shape[0] != N:
        raise ValueError(f"features must be [N={N}, L], got {tuple(features.shape)}")
    L, K, F = features.shape[1], queries.shape[1], frequencies.shape[0]
    E = 4 * F + 1
    if (transform.shape != (L, L) or queries.shape != (N, K) or keys.shape != (N, K)
            or distance_transform.shape != (L, 2 * E)):
        raise ValueError(f"aggregate_neighbors: expected transform [{L},{L}], queries/keys [{N},{K}], "
                         f"distance_transform [{L},{2 * E}] (E = 2*d*F + 1 = {E})")
    for name, t in (("features", features), ("transform", transform), ("queries", queries), ("keys", keys),
                    ("frequencies", frequencies), ("distance_transform", distance_transform)):
        if not t.is_cuda:
            raise RuntimeError(f"{name} is on {t.device}: aggregate_neighbors runs on the GPU only (no CPU fallback)")
    if L + 2 * E > 128:
        raise NotImplementedError(f"L + 2E = {L + 2 * E} > 128 is not supported")
    return _Aggregate.apply(nb, features, transform, queries, keys, frequencies, distance_transform)

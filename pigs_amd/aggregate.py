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

The neighbour relation lives in index lists (``NeighborLists``: [N, cap] with cap = the longest list,
found by a counting pass).  Forward = one launch, backward = four (the three small GEMMs
gout @ [transform | distance_transform], gout^T @ acc included).  d = 2, float32 / float64.
"""
import ctypes

import torch

from . import _lib

_DTYPES = {torch.float32: _lib.PIGS_F32, torch.float64: _lib.PIGS_F64}
BRUTE_MAX = 2048     # pigs_amd/csrc/aggregate.hip AGG_BRUTE_MAX: up to here every pair is tested, cap = N
MAX_NEIGHBORS = {torch.float32: 8192, torch.float64: 4096}      # slab size when the counting pass cannot be read back (capture)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class NeighborLists:
    """Index lists of the neighbour relation: by rows (the j of an i) and by columns (the i of a j),
    ``cap`` int32 slots per Gaussian.  Up to N = 2048 (the model's sizes) every pair is tested and cap = N:
    two launches, nothing read back.  Beyond: built through the sampler's multi-level Gaussian grid (one wave per
    Gaussian walks the cells around its centre / its ellipse; pigs_amd/csrc/aggregate.hip), in two passes:
    a counting pass, then -- with ``cap`` = the longest list rounded up to 64, read back ONCE (the only
    host synchronisation of ``preprocess_aggregate``) -- the lists themselves.  ``cap`` given (or a hipGraph
    being captured, where nothing may be read back): one pass into slabs of that size, and a list that does
    not fit sets ``overflow`` (checked by :meth:`check`; debug mode calls it)."""

    def __init__(self, means, conics, q_max, cap=None):
        lib = _lib.load()
        if means.dim() != 2 or means.shape[1] != 2:
            raise NotImplementedError("aggregate_neighbors is implemented for d = 2")
        if means.dtype not in _DTYPES:
            raise TypeError(f"dtype {means.dtype} is not supported (float32 / float64)")
        self.means = means.detach().contiguous()
        self.conics = conics.detach().reshape(means.shape[0], 3).contiguous()
        self.N = N = means.shape[0]
        dev = means.device
        dt = _DTYPES[means.dtype]
        nbytes = lib.pigs_aggregate_workspace_bytes(dt, N)
        if nbytes == 0:
            raise _lib.PigsError(f"aggregate_neighbors does not support N={N}")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.row_counts = torch.empty(N, dtype=torch.int32, device=dev)
        self.col_counts = torch.empty(N, dtype=torch.int32, device=dev)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        self.row_lists = self.col_lists = None

        def run(flags, cap_, with_lists):
            with torch.cuda.device(dev):
                rc = lib.pigs_aggregate_lists(dt, N, cap_, _ptr(self.means), _ptr(self.conics), float(q_max),
                                              _ptr(self.workspace), nbytes, flags, _ptr(self.row_counts),
                                              _ptr(self.row_lists) if with_lists else ctypes.c_void_p(0),
                                              _ptr(self.col_counts),
                                              _ptr(self.col_lists) if with_lists else ctypes.c_void_p(0),
                                              _ptr(self.overflow), _stream(dev))
            _lib.check(rc, "pigs_aggregate_lists")

        flags = 1                                                  # PIGS_AGGREGATE_BUILD_GRID
        if cap is None and N <= BRUTE_MAX:
            cap = max(1, N)                                        # every pair is tested; a slab of N cannot overflow
        if cap is None and N > 0 and torch.cuda.is_current_stream_capturing():
            cap = min(N, MAX_NEIGHBORS[means.dtype])               # nothing can be read back inside a capture
        if cap is None and N > 0:
            run(flags, 1, False)                                   # counting pass (full lengths)
            flags = 0
            longest = int(torch.maximum(self.row_counts.max(), self.col_counts.max()).item())
            cap = max(64, (longest + 63) // 64 * 64)
        self.cap = max(1, int(cap if cap is not None else 1))
        self.row_lists = torch.empty((N, self.cap), dtype=torch.int32, device=dev)
        self.col_lists = torch.empty((N, self.cap), dtype=torch.int32, device=dev)
        if N > 0:
            run(flags, self.cap, True)

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
        g_f, g_tr, g_q, g_k, g_fr, g_dist = (torch.empty_like(t) for t in (f, tr, q, k, fr, dist))
        nbytes = lib.pigs_aggregate_backward_scratch_bytes(_DTYPES[dt], N, L, F)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=f.device)
        with torch.cuda.device(f.device):
            rc = lib.pigs_aggregate_backward(_DTYPES[dt], N, nb.cap, L, K, F, _ptr(nb.means), _ptr(nb.conics),
                                             _ptr(nb.row_counts), _ptr(nb.row_lists), _ptr(nb.col_counts),
                                             _ptr(nb.col_lists), _ptr(f), _ptr(tr), _ptr(q), _ptr(k), _ptr(fr), _ptr(dist),
                                             _ptr(lse), _ptr(acc), _ptr(gout), _ptr(scratch), nbytes,
                                             _ptr(g_f), _ptr(g_tr), _ptr(g_q), _ptr(g_k), _ptr(g_fr), _ptr(g_dist),
                                             _stream(f.device))
        _lib.check(rc, "pigs_aggregate_backward")
        if N == 0:
            for g in (g_tr, g_fr, g_dist):
                g.zero_()
        grads = (g_f, g_tr, g_q, g_k, g_fr, g_dist)
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

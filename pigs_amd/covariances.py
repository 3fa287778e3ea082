"""Fused covariance builder: drop-in for the three helpers of /root/reference/gaussians.py

    build_full_covariances(s, t) -> (covariances [N,2,2], conics [N,2,2])      (:163-183)
    flatten_covariances(covariances, conics) -> ([N,3], [N,3])                 (:185-189)
    build_covariances(s, t) -> flat (covariances [N,3], conics [N,3])          (:191-193)

with the same argument meaning: ``s`` [N,2] variances (> 0), ``t`` [N,1] raw correlation
(squashed by tanh).  One HIP launch forward and one backward (C ABI: pigs_build_covariances*)
instead of the reference's chain of small torch kernels; d = 2, float32 / float64, GPU only.
"""
import ctypes

import torch

from . import _lib

_DTYPES = {torch.float32: 0, torch.float64: 1}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _check(s, t):
    if not (isinstance(s, torch.Tensor) and isinstance(t, torch.Tensor)):
        raise TypeError("scaling and transform must be torch.Tensors")
    if s.dim() != 2 or s.shape[1] != 2:
        raise NotImplementedError(f"scaling must be [N, 2] (d = 2), got {tuple(s.shape)}")
    if t.numel() != s.shape[0]:
        raise ValueError(f"transform must hold N = {s.shape[0]} elements, got {tuple(t.shape)}")
    if not s.is_cuda or t.device != s.device:
        raise RuntimeError(f"scaling is on {s.device}, transform on {t.device}: the builder runs on one GPU "
                           "(no CPU fallback)")
    if s.dtype not in _DTYPES or t.dtype != s.dtype:
        raise TypeError(f"dtypes {s.dtype} / {t.dtype}: float32 or float64, both alike")


class _BuildCovariances(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, t):
        lib = _lib.load()
        sc, tc = s.contiguous(), t.contiguous()
        N = sc.shape[0]
        cov = torch.empty((N, 3), dtype=sc.dtype, device=sc.device)
        con = torch.empty((N, 3), dtype=sc.dtype, device=sc.device)
        with torch.cuda.device(sc.device):
            rc = lib.pigs_build_covariances(_DTYPES[sc.dtype], N, _ptr(sc), _ptr(tc), _ptr(cov), _ptr(con),
                                            ctypes.c_void_p(torch.cuda.current_stream(sc.device).cuda_stream))
        _lib.check(rc, "pigs_build_covariances")
        ctx.save_for_backward(sc, tc)
        ctx.t_shape = t.shape
        ctx.set_materialize_grads(False)
        return cov, con

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_cov, g_con):
        sc, tc = ctx.saved_tensors
        if g_cov is None and g_con is None:
            return None, None
        lib = _lib.load()
        N = sc.shape[0]
        g_cov = g_cov.contiguous() if g_cov is not None else None
        g_con = g_con.contiguous() if g_con is not None else None
        g_s = torch.empty_like(sc)
        g_t = torch.empty_like(tc)
        with torch.cuda.device(sc.device):
            rc = lib.pigs_build_covariances_backward(
                _DTYPES[sc.dtype], N, _ptr(sc), _ptr(tc), _ptr(g_cov), _ptr(g_con), _ptr(g_s), _ptr(g_t),
                ctypes.c_void_p(torch.cuda.current_stream(sc.device).cuda_stream))
        _lib.check(rc, "pigs_build_covariances_backward")
        return g_s, g_t.reshape(ctx.t_shape)


def build_covariances(s, t):
    """Flat (covariances [N,3], conics [N,3]) = (xx, xy, yy); gaussians.py:191-193."""
    _check(s, t)
    return _BuildCovariances.apply(s, t)


def build_full_covariances(s, t):
    """(covariances [N,2,2], conics [N,2,2]); gaussians.py:163-183."""
    cov, con = build_covariances(s, t)
    # no index tensor: torch.tensor(..., device=...) is a pageable host-to-device copy on every call
    # and cannot be captured into a hipGraph
    def full(f):
        return torch.stack((f[:, 0], f[:, 1], f[:, 1], f[:, 2]), dim=-1).reshape(-1, 2, 2)
    return full(cov), full(con)


def _upper(m):
    return torch.stack((m[..., 0, 0], m[..., 0, 1], m[..., 1, 1]), dim=-1)


def flatten_covariances(covariances, conics):
    """[..., 2, 2] symmetric matrices -> their (xx, xy, yy) triples; gaussians.py:185-189."""
    return _upper(covariances), _upper(conics)

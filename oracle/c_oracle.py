"""ctypes binding of oracle/pigs_oracle.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpigs_oracle.so")
_lib = None

_P = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    src = os.path.join(_HERE, "pigs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libpigs_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.pigs_oracle_forward.restype = ctypes.c_int
        _lib.pigs_oracle_forward.argtypes = [ctypes.c_int] * 3 + [ctypes.c_long] * 2 + [_P] * 8
        _lib.pigs_oracle_backward.restype = ctypes.c_int
        _lib.pigs_oracle_backward.argtypes = [ctypes.c_int] * 3 + [ctypes.c_long] * 2 + [_P] * 11
        _lib.pigs_oracle_backward_abs.argtypes = [ctypes.c_int] * 3 + [ctypes.c_long] * 2 + [_P] * 11
        _lib.pigs_oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return lib().pigs_oracle_num_threads()


def _ptr(a):
    return a.ctypes.data_as(_P) if a is not None else None


def _f64(a, shape):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))


def _shapes(M, d, c):
    return {0: (M, c), 1: (M, d, c), 2: (M, d, d, c), 3: (M, d, d, d, c)}


def forward(means, conics_flat, values, samples, orders=(0, 1, 2)):
    """Dense forward in float64.  conics_flat: [N, d(d+1)/2].  Returns {order: ndarray}."""
    means = np.asarray(means)
    N, d = means.shape
    means = _f64(means, (N, d))
    conics = _f64(conics_flat, (N, d * (d + 1) // 2))
    values = np.asarray(values)
    c = values.shape[1] if values.ndim == 2 else 1
    values = _f64(values, (N, c))
    samples = np.asarray(samples)
    M = samples.size // d
    samples = _f64(samples, (M, d))
    mask = sum(1 << o for o in orders)
    sh = _shapes(M, d, c)
    out = {o: np.zeros(sh[o]) for o in orders}
    rc = lib().pigs_oracle_forward(d, c, mask, N, M, _ptr(means), _ptr(conics), _ptr(values), _ptr(samples),
                                   *[_ptr(out.get(o)) for o in range(4)])
    if rc:
        raise ValueError(f"pigs_oracle_forward: unsupported d={d} c={c}")
    return out


def backward(means, conics_flat, values, samples, grads, absolute=False):
    """Dense VJP in float64.  grads: {order: grad_output}.  Returns (g_means, g_conics_flat, g_values).
    ``absolute``: every (sample, Gaussian) pair's contribution enters in absolute value -- the magnitude
    float32 accumulation errors are measured against (see :func:`accumulation_bound`)."""
    means = np.asarray(means)
    N, d = means.shape
    means = _f64(means, (N, d))
    conics = _f64(conics_flat, (N, d * (d + 1) // 2))
    values = np.asarray(values)
    c = values.shape[1] if values.ndim == 2 else 1
    values = _f64(values, (N, c))
    samples = np.asarray(samples)
    M = samples.size // d
    samples = _f64(samples, (M, d))
    sh = _shapes(M, d, c)
    gs = {o: _f64(g, sh[o]) for o, g in grads.items() if g is not None}
    mask = sum(1 << o for o in gs)
    gm, gc, gv = np.zeros_like(means), np.zeros_like(conics), np.zeros_like(values)
    fn = lib().pigs_oracle_backward_abs if absolute else lib().pigs_oracle_backward
    rc = fn(d, c, mask, N, M, _ptr(means), _ptr(conics), _ptr(values), _ptr(samples),
            *[_ptr(gs.get(o)) for o in range(4)], _ptr(gm), _ptr(gc), _ptr(gv))
    if rc:
        raise ValueError(f"pigs_oracle_backward: unsupported d={d} c={c}")
    return gm, gc, gv


def accumulation_bound(means, conics_flat, values, samples, grads, ulps=1e-6, floor=1e-6):
    """Per-entry error bound of a float32 backward against :func:`backward`: ``ulps`` (1e-6 = ~8 float32
    ulp: per-pair arithmetic + the order of summation) of the sum of the ABSOLUTE per-pair contributions to
    the entry, plus ``floor`` of the largest entry (what a cut-off at exp(-q_max/2) may drop; a tenth of
    the 1e-5 bar).  Returns (want, bound): two triples (g_means, g_conics_flat, g_values)."""
    want = backward(means, conics_flat, values, samples, grads)
    mag = backward(means, conics_flat, values, samples, grads, absolute=True)
    return want, tuple(ulps * a + floor * np.abs(w).max() for a, w in zip(mag, want))

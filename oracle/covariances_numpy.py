"""TEST INFRASTRUCTURE -- CPU restatement (numpy, float64) of the reference's covariance builder.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (pigs_amd/) never does.  Pinned against tests/golden/ref_build_covariances.npz, which
tools/gen_golden.py produced by running the reference's own functions.

Follows /root/reference/gaussians.py:
  :163-165  t <- tanh(t) * sqrt(s0 * s1); S = diag(s) with both off-diagonals = t
  :181-183  covariances = S; conics = inverse(S)
  :186-189  flat layout = entries [0, 1, 3] of the row-major 2x2 = (xx, xy, yy)
(d = 2 only, like the reference's flattening.)
"""
import numpy as np


def build_covariances(scaling, transform):
    """scaling [N,2] > 0, transform [N,1] or [N] -> (cov_flat [N,3], conic_flat [N,3])."""
    s = np.asarray(scaling, dtype=np.float64)
    t = np.asarray(transform, dtype=np.float64).reshape(-1)
    tau = np.tanh(t) * np.sqrt(s[:, 0] * s[:, 1])
    det = s[:, 0] * s[:, 1] - tau * tau
    cov = np.stack((s[:, 0], tau, s[:, 1]), axis=-1)
    conic = np.stack((s[:, 1] / det, -tau / det, s[:, 0] / det), axis=-1)
    return cov, conic


def build_covariances_backward(scaling, transform, g_cov, g_conic):
    """Gradients of L = <g_cov, cov> + <g_conic, conic> wrt scaling [N,2] and transform [N,1].

    With h = tanh(t), r = sqrt(s0 s1), k = 1 / (1 - h^2):
      cov = (s0, h r, s1),  conic = (k / s0, -h k / r, k / s1).
    """
    s = np.asarray(scaling, dtype=np.float64)
    t = np.asarray(transform, dtype=np.float64).reshape(-1)
    gc = np.zeros((s.shape[0], 3)) if g_cov is None else np.asarray(g_cov, dtype=np.float64)
    gq = np.zeros((s.shape[0], 3)) if g_conic is None else np.asarray(g_conic, dtype=np.float64)
    s0, s1 = s[:, 0], s[:, 1]
    h = np.tanh(t)
    r = np.sqrt(s0 * s1)
    k = 1.0 / (1.0 - h * h)
    tau = h * r
    g_s0 = gc[:, 0] + gc[:, 1] * tau / (2 * s0) - gq[:, 0] * k / (s0 * s0) + gq[:, 1] * h * k / (2 * r * s0)
    g_s1 = gc[:, 2] + gc[:, 1] * tau / (2 * s1) - gq[:, 2] * k / (s1 * s1) + gq[:, 1] * h * k / (2 * r * s1)
    g_h = gc[:, 1] * r + 2 * h * k * k * (gq[:, 0] / s0 + gq[:, 2] / s1) - gq[:, 1] * k * k * (1 + h * h) / r
    g_t = g_h / k
    return np.stack((g_s0, g_s1), axis=-1), g_t.reshape(-1, 1)

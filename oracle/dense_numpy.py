"""Dense numpy restatement of the reference sampler (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows, term by term, the reference's pure-PyTorch twin of the CUDA sampler:

* order 0  ``gaussians.sample_gaussians``        /root/reference/gaussians.py:48-58
* order 1  ``gaussians.gaussian_derivative``     /root/reference/gaussians.py:89-101
* order 2  ``gaussians.gaussian_derivative2``    /root/reference/gaussians.py:103-116
  (the "laplacian" of the sampler API is this full Hessian: model_pn.py:652)
* order 3  has no in-tree function; it is the analytic derivative of order 2
  (``model_pn.py:654`` gives the shape ``n, d, d, d, c``); pinned by triple autograd
  through ``sample_gaussians`` in tools/gen_golden.py.
* backward: closed form of ``torch.autograd`` through the functions above wrt
  (means, values, full conics), as exercised by test_derivatives.py:122-124, 208-220, 340-356.

Notation: x = s - mu, p = C x, q = x^T p, g = exp(-q/2).

Any d, any c; dtype follows the inputs (use float64 for the checker).  Pinned against
tests/golden/*.npz by tests/test_oracle.py.
"""
import numpy as np

ORDER_SHAPES = {
    0: lambda M, d, c: (M, c),
    1: lambda M, d, c: (M, d, c),
    2: lambda M, d, c: (M, d, d, c),
    3: lambda M, d, c: (M, d, d, d, c),
}


def n_flat(d):
    return d * (d + 1) // 2


def _triu(d):
    return np.triu_indices(d)


def full_from_flat(flat, d):
    """[N, d(d+1)/2] row-major upper triangle -> symmetric [N, d, d].

    For d = 2 this is the inverse of the reference's ``[..., [0, 1, 3]]`` flatten
    (/root/reference/gaussians.py:186-189); d = 1 callers pass [N, 1] (test_1d.py:24-30).
    """
    flat = np.asarray(flat).reshape(-1, n_flat(d))
    iu = _triu(d)
    full = np.zeros((flat.shape[0], d, d), dtype=flat.dtype)
    full[:, iu[0], iu[1]] = flat
    full[:, iu[1], iu[0]] = flat
    return full


def flat_grad_from_full(G, d):
    """Gradient wrt the flat conic from the gradient wrt the full (unconstrained) matrix:
    off-diagonal flat entries feed two matrix entries, so their gradients add."""
    iu = _triu(d)
    out = G[:, iu[0], iu[1]].copy()
    off = iu[0] != iu[1]
    out[:, off] += G[:, iu[1][off], iu[0][off]]
    return out


def _pair_terms(means, conics, samples):
    x = samples[:, None, :] - means[None, :, :]                # [m, N, d]
    p = np.einsum("nij,mnj->mni", conics, x)                   # conics @ x
    q = np.einsum("mni,mni->mn", x, p)
    g = np.exp(-0.5 * q)
    return x, p, g


def forward(means, conics, values, samples, orders=(0, 1, 2, 3), chunk=None):
    """Returns {order: array}.  ``conics`` is the full [N, d, d] matrix."""
    means = np.asarray(means)
    N, d = means.shape
    conics = np.asarray(conics).reshape(N, d, d)
    values = np.asarray(values).reshape(N, -1)
    samples = np.asarray(samples).reshape(-1, d)
    c = values.shape[1]
    M = samples.shape[0]
    dt = np.result_type(means, conics, values, samples)
    out = {o: np.zeros(ORDER_SHAPES[o](M, d, c), dtype=dt) for o in orders}
    if chunk is None:
        chunk = max(1, int(2**24 // max(1, N * d * d)))
    for m0 in range(0, M, chunk):
        s = samples[m0:m0 + chunk]
        x, p, g = _pair_terms(means, conics, s)
        sl = slice(m0, m0 + s.shape[0])
        if 0 in out:
            out[0][sl] = np.einsum("mn,nc->mc", g, values)
        if 1 in out:
            out[1][sl] = -np.einsum("mn,mni,nc->mic", g, p, values)
        if 2 in out:
            out[2][sl] = (np.einsum("mn,mni,mnj,nc->mijc", g, p, p, values)
                          - np.einsum("mn,nij,nc->mijc", g, conics, values))
        if 3 in out:
            out[3][sl] = (np.einsum("mn,nij,mnk,nc->mijkc", g, conics, p, values)
                          + np.einsum("mn,nik,mnj,nc->mijkc", g, conics, p, values)
                          + np.einsum("mn,njk,mni,nc->mijkc", g, conics, p, values)
                          - np.einsum("mn,mni,mnj,mnk,nc->mijkc", g, p, p, p, values))
    return out


def backward(means, conics, values, samples, grads, chunk=None):
    """Closed-form VJP.  ``grads`` = {order: grad_output}.  Returns
    (g_means [N,d], g_conics_full [N,d,d], g_values [N,c]); the conic gradient treats the
    d*d matrix entries as independent (what autograd through the reference's
    ``full_conics`` gives); use :func:`flat_grad_from_full` for the sampler's flat layout.
    """
    means = np.asarray(means)
    N, d = means.shape
    conics = np.asarray(conics).reshape(N, d, d)
    values = np.asarray(values).reshape(N, -1)
    samples = np.asarray(samples).reshape(-1, d)
    c = values.shape[1]
    M = samples.shape[0]
    dt = np.result_type(means, conics, values, samples)
    g_means = np.zeros((N, d), dtype=dt)
    g_conics = np.zeros((N, d, d), dtype=dt)
    g_values = np.zeros((N, c), dtype=dt)
    if chunk is None:
        chunk = max(1, int(2**23 // max(1, N * d * d * c)))
    gr = {o: np.asarray(g).reshape(ORDER_SHAPES[o](M, d, c)) for o, g in grads.items() if g is not None}
    for m0 in range(0, M, chunk):
        s = samples[m0:m0 + chunk]
        m = s.shape[0]
        sl = slice(m0, m0 + m)
        x, p, g = _pair_terms(means, conics, s)
        F = np.zeros((m, N, c), dtype=dt)          # per-channel polynomial factor
        dA = np.zeros((m, N, d), dtype=dt)         # sum_c v_c dF_c/dp
        E = np.zeros((m, N, d, d), dtype=dt)       # sum_c v_c dF_c/dC (explicit C terms)
        if 0 in gr:
            F += gr[0][sl][:, None, :]
        if 1 in gr:
            g1 = gr[1][sl]
            F -= np.einsum("mic,mni->mnc", g1, p)
            dA -= np.einsum("mlc,nc->mnl", g1, values)
        if 2 in gr:
            g2 = gr[2][sl]
            F += np.einsum("mijc,mni,mnj->mnc", g2, p, p) - np.einsum("mijc,nij->mnc", g2, conics)
            dA += np.einsum("mljc,mnj,nc->mnl", g2, p, values) + np.einsum("milc,mni,nc->mnl", g2, p, values)
            E -= np.einsum("mklc,nc->mnkl", g2, values)
        if 3 in gr:
            g3 = gr[3][sl]
            F += (np.einsum("mijkc,nij,mnk->mnc", g3, conics, p)
                  + np.einsum("mijkc,nik,mnj->mnc", g3, conics, p)
                  + np.einsum("mijkc,njk,mni->mnc", g3, conics, p)
                  - np.einsum("mijkc,mni,mnj,mnk->mnc", g3, p, p, p))
            dA += (np.einsum("mijlc,nij,nc->mnl", g3, conics, values)
                   + np.einsum("milkc,nik,nc->mnl", g3, conics, values)
                   + np.einsum("mljkc,njk,nc->mnl", g3, conics, values)
                   - np.einsum("mljkc,mnj,mnk,nc->mnl", g3, p, p, values)
                   - np.einsum("milkc,mni,mnk,nc->mnl", g3, p, p, values)
                   - np.einsum("mijlc,mni,mnj,nc->mnl", g3, p, p, values))
            E += (np.einsum("mklrc,mnr,nc->mnkl", g3, p, values)
                  + np.einsum("mkrlc,mnr,nc->mnkl", g3, p, values)
                  + np.einsum("mrklc,mnr,nc->mnkl", g3, p, values))
        g_values += np.einsum("mn,mnc->nc", g, F)
        A = np.einsum("mnc,nc->mn", F, values)
        # d g / d mu = g p ;  d p_i / d mu_l = -C_il
        g_means += np.einsum("mn,mnl->nl", g * A, p) - np.einsum("mn,mni,nil->nl", g, dA, conics)
        # d(-q/2)/dC_kl = -x_k x_l / 2 ;  d p_i / d C_kl = delta_ik x_l
        g_conics += (-0.5 * np.einsum("mn,mnk,mnl->nkl", g * A, x, x)
                     + np.einsum("mn,mnk,mnl->nkl", g, dA, x)
                     + np.einsum("mn,mnkl->nkl", g, E))
    return g_means, g_conics, g_values

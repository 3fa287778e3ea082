"""The Hessian's trace (mask bit 16, `sample((..., "lap"))`): u_xx + u_yy in its own accumulator
instead of the full Hessian.  Expected values come from the oracle's full Hessian (trace taken
here) and, for the backward, from the oracle's backward fed with gl * identity."""
import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().cpu().double().numpy()
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def make(rng, N, M, d, c):
    means = rng.uniform(-1, 1, (N, d))
    if d == 2:
        s = np.exp(2 * rng.normal(-2.5, 0.5, (N, 2)))
        tau = np.tanh(rng.normal(0, 0.7, N)) * np.sqrt(s[:, 0] * s[:, 1])
        det = s[:, 0] * s[:, 1] - tau ** 2
        con = np.stack((s[:, 1] / det, -tau / det, s[:, 0] / det), -1)
    else:
        con = 1.0 / np.exp(2 * rng.normal(-2.5, 0.5, (N, 1)))
    values = rng.uniform(-1, 1, (N, c))
    samples = rng.uniform(-1.1, 1.1, (M, d))
    return means, con, values, samples


def expected(args, d, orders_np, r):
    """Oracle outputs for (0, 1, trace) and parameter gradients of sum_k <out_k, r_k>."""
    exp = c_oracle.forward(*args, orders=(0, 1, 2))
    lap = sum(exp[2][:, i, i, :] for i in range(d))
    gH = np.zeros_like(exp[2])
    for i in range(d):
        gH[:, i, i, :] = r["lap"]
    grads = {2: gH}
    if 0 in orders_np:
        grads[0] = r[0]
    if 1 in orders_np:
        grads[1] = r[1]
    em, ec, ev = c_oracle.backward(*args, grads)
    return exp[0], exp[1], lap, (em, ev, ec)


@pytest.mark.parametrize("backend,N,M,d,c,dtype,tol", [
    ("dense", 40, 300, 2, 1, torch.float64, 1e-11),
    ("dense", 64, 500, 2, 2, torch.float32, 1e-5),
    ("dense", 30, 200, 1, 2, torch.float32, 1e-5),
    ("binned", 900, 4000, 2, 1, torch.float32, 1e-5),
    ("binned", 700, 3000, 2, 2, torch.float32, 1e-5),
])
@pytest.mark.parametrize("orders", [("lap",), (0, 1, "lap"), (0, "lap")])
def test_trace_forward_backward(hip_lib, backend, N, M, d, c, dtype, tol, orders):
    from diff_gaussian_sampling import GaussianSampler
    rng = np.random.default_rng(N + M + c)
    means, con, values, samples = make(rng, N, M, d, c)
    t = [torch.as_tensor(a, dtype=dtype, device="cuda") for a in (means, values, con, samples)]
    args = [x.cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]
    for x in t[:3]:
        x.requires_grad_(True)
    s = GaussianSampler(True, backend=backend, fuse="none")
    s.preprocess(t[0], t[1], None, t[2], t[3])
    assert (s._plan is not None) == (backend == "binned")
    outs = s.sample(orders)
    r = {0: rng.uniform(-1, 1, (M, c)), 1: rng.uniform(-1, 1, (M, d, c)), "lap": rng.uniform(-1, 1, (M, c))}
    e0, e1, elap, (em, ev, ec) = expected(args, d, orders, r)
    exp = {0: e0, 1: e1, "lap": elap}
    loss = 0
    for o, out in zip(orders, outs):
        assert out.shape == exp[o].shape
        assert rel(out, exp[o]) < tol, (o, rel(out, exp[o]))
        loss = loss + (out * torch.as_tensor(r[o], dtype=dtype, device="cuda")).sum()
    loss.backward()
    assert rel(t[0].grad, em) < tol and rel(t[1].grad, ev) < tol and rel(t[2].grad, ec) < tol


def test_trace_beside_hessian_and_third_order(hip_lib):
    from diff_gaussian_sampling import GaussianSampler
    rng = np.random.default_rng(5)
    means, con, values, samples = make(rng, 50, 400, 2, 2)
    t = [torch.as_tensor(a, dtype=torch.float32, device="cuda") for a in (means, values, con, samples)]
    args = [x.cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]
    exp = c_oracle.forward(*args, orders=(2, 3))
    lap = exp[2][:, 0, 0, :] + exp[2][:, 1, 1, :]
    s = GaussianSampler(True)
    s.preprocess(t[0], t[1], None, t[2], t[3])
    H, L = s.sample((2, "lap"))                       # one launch: the trace is the Hessian's diagonal
    assert rel(H, exp[2]) < 1e-5 and rel(L, lap) < 1e-5
    s.preprocess(t[0], t[1], None, t[2], t[3])
    T3, L = s.sample((3, "lap"))                      # no fused kernel: two launches
    assert rel(T3, exp[3]) < 1e-5 and rel(L, lap) < 1e-5
    s.preprocess(t[0], t[1], None, t[2], t[3])
    assert rel(s.sample_gaussians_laplacian_trace(), lap) < 1e-5


def test_abi_rejects_hessian_and_trace_together(hip_lib):
    z = torch.zeros(8, device="cuda")
    p = z.data_ptr()
    rc = hip_lib.pigs_sample_forward(0, 2, 1, 4 | 16, 1, 1, p, p, p, p, p, p, p, p, None)
    assert rc == 1                                    # PIGS_ERR_INVALID
    rc = hip_lib.pigs_sample_forward(0, 2, 1, 8 | 16, 1, 1, p, p, p, p, p, p, p, p, None)
    assert rc == 2                                    # PIGS_ERR_UNSUPPORTED: no fused kernel

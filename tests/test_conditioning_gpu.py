"""GPU: where float32 cannot hold 1e-5 -- ill-conditioned conics.  The reference evaluates q = x^T C x as
a dx^2 + 2 b dx dy + c dy^2 (gaussians.py:48-58, through C x), and so do the kernels; for a strongly correlated,
strongly anisotropic Gaussian (|rho| -> 0.99, axis ratio in the hundreds) the three terms are each thousands of times
larger than their sum, and a float32 evaluation of ANY ordering of that formula loses those digits: tools/fuzz_dense.py
found 50 of 1 500 adversarial cases outside 1e-5 (up to 2e-3) for exactly this reason.  This test pins the limit:
float64 is exact to 1e-11 on the same inputs, the float32 error of every point stays inside the bound that the
cancellation of the quadratic form predicts (and well-conditioned conics of the same sizes stay inside 1e-5), so a
float32 error here is the arithmetic of the formula, not a defect of a kernel."""
import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu


def conics(rng, n, rho, ratio):
    s1 = np.exp(rng.normal(-3.0, 0.3, n))                     # variances: long axis sigma ~ 0.2, short = long / ratio
    s0 = s1 / ratio ** 2
    tau = rho * np.sqrt(s0 * s1) * rng.choice([-1.0, 1.0], n)
    det = s0 * s1 - tau ** 2
    return np.stack((s1 / det, -tau / det, s0 / det), -1), np.stack((s0, tau, s1), -1)


@pytest.mark.parametrize("backend", ["dense", "binned"])
def test_float32_error_of_ill_conditioned_conics_is_the_quadratic_forms_cancellation(hip_lib, backend):
    from diff_gaussian_sampling import GaussianSampler
    rng = np.random.default_rng(41)
    n, m = 300, 4096
    means = rng.uniform(-0.6, 0.6, (n, 2))
    values = rng.uniform(0.2, 1.0, (n, 1))
    res = {}
    for name, rho, ratio in (("well", 0.3, 3.0), ("ill", 0.99, 300.0)):
        con, cov = conics(rng, n, rho, ratio)
        # points along the Gaussians' long axes, where they overlap the most and the quadratic form cancels the most
        k = rng.integers(0, n, m)
        L = np.linalg.cholesky(np.stack((np.stack((cov[k, 0], cov[k, 1]), -1), np.stack((cov[k, 1], cov[k, 2]), -1)), -2))
        pts = means[k] + np.einsum("mij,mj->mi", L, rng.normal(0, 1.0, (m, 2)))
        out = {}
        for dtype in (torch.float32, torch.float64):
            if backend == "binned" and dtype == torch.float64:
                continue
            t = [torch.tensor(a, dtype=dtype, device="cuda") for a in (means, values, con, pts)]
            args = [x.cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]        # the rounded inputs
            s = GaussianSampler(False, backend=backend)
            s.preprocess(t[0], t[1], None, t[2], t[3])
            u = s.sample_gaussians().cpu().double().numpy()
            exp = c_oracle.forward(*args, orders=(0,))[0]
            out[dtype] = (np.abs(u - exp)[:, 0], np.abs(exp).max(), args)
        res[name] = out
    for name in res:
        if torch.float64 in res[name]:
            err, top, _ = res[name][torch.float64]
            assert err.max() <= 1e-11 * top, (name, "float64", err.max() / top)
    err, top, _ = res["well"][torch.float32]
    assert err.max() <= 1e-5 * top, ("well-conditioned float32", err.max() / top)
    # ill-conditioned, float32: per point, the bound the cancellation predicts -- every pair's weight v g times half
    # the float32 rounding of the three terms of q (a few ulp of their absolute sum), summed over the Gaussians
    err, top, (mu, con, val, pts) = res["ill"][torch.float32]
    dx = pts[:, None, 0] - mu[None, :, 0]
    dy = pts[:, None, 1] - mu[None, :, 1]
    t1, t2, t3 = con[None, :, 0] * dx * dx, 2 * con[None, :, 1] * dx * dy, con[None, :, 2] * dy * dy
    q = t1 + t2 + t3
    g = np.exp(-0.5 * np.maximum(q, 0))
    # the coordinates' own rounding enters too: dx, dy are differences of float32 numbers (exact), but C x is formed
    # from rounded products; 8 ulp of the absolute sum covers every evaluation order seen (fma or not)
    bound = (np.abs(val[None, :, 0]) * g * 0.5 * (np.abs(t1) + np.abs(t2) + np.abs(t3)) * 8 * 2.0 ** -24).sum(1)
    assert (err <= bound + 1e-5 * top).all(), float((err / (bound + 1e-5 * top)).max())
    # and the limit is real: the documented level (up to 2e-3 of the largest value), not 1e-5
    assert err.max() <= 5e-3 * top
    print(f"{backend}: ill-conditioned float32 max error {err.max() / top:.2e} of the largest value "
          f"(bound {bound.max() / top:.2e}); well-conditioned {res['well'][torch.float32][0].max() / res['well'][torch.float32][1]:.2e}")


# The 50 of the first 1 500 cases of tools/fuzz_dense.py (rng seed 5000 + k, the adversarial generator of
# tests/test_fuzz_gpu.py cut to 300 Gaussians x 500 points) whose float32 dense results leave the 1e-5 bar; measured
# on MI355X, round 4 (gpurun_out/r4_fuzz_dense.txt): worst 1.1e-3 / 1.8e-3 / 5.8e-4 / 2.2e-3 in the outputs of orders
# 0..3, 5.9e-4 in the gradients; float64 on the same cases: 5e-13 / 2e-12.
OVER_THE_BAR = [2, 48, 78, 130, 143, 158, 216, 223, 224, 241, 264, 296, 366, 372, 395, 423, 430, 449, 487, 489, 523, 534, 627, 630,
                664, 711, 736, 751, 759, 768, 780, 793, 832, 848, 852, 925, 1005, 1085, 1106, 1118, 1184, 1216, 1218, 1220, 1234,
                1287, 1358, 1373, 1384, 1484]


def fuzz_errors(seed, dtype):
    """The measure of tools/fuzz_dense.py: outputs of orders 0..3 against max(largest value, the term's own scale
    |v| lambda^(k/2)), gradients against their largest entry times the same understatement factor."""
    from diff_gaussian_sampling import GaussianSampler
    from test_fuzz_gpu import make_case
    rng = np.random.default_rng(5000 + seed)
    means, values, con, pts = make_case(rng)
    means, values, con, pts = means[:300], values[:300], con[:300], pts[:500]
    lam = (con[:, 0] + con[:, 2]) / 2 + np.sqrt(((con[:, 0] - con[:, 2]) / 2) ** 2 + con[:, 1] ** 2)
    term = [float((np.abs(values).max(1) * lam ** (k / 2)).max()) for k in range(4)]
    t = [torch.tensor(a, dtype=dtype, device="cuda") for a in (means, values, con, pts)]
    args = [x.cpu().double().numpy() for x in (t[0], t[2], t[1], t[3])]     # the rounded inputs
    for x in t[:3]:
        x.requires_grad_(True)
    s = GaussianSampler(False, backend="dense", fuse="all")
    s.preprocess(t[0], t[1], None, t[2], t[3])
    o = s.sample((0, 1, 2, 3))
    exp = c_oracle.forward(*args, orders=(0, 1, 2, 3))
    g = np.random.default_rng(seed)
    rs = [g.uniform(-1, 1, exp[k].shape) for k in range(4)]
    sum((x * torch.tensor(r, dtype=dtype, device="cuda")).sum() for x, r in zip(o, rs)).backward()
    em, ec, ev = c_oracle.backward(*args, {k: torch.tensor(r, dtype=dtype).double().numpy() for k, r in enumerate(rs)})
    errs, under = [], 1.0
    for k in range(4):
        a = o[k].detach().cpu().double().numpy()
        top = np.abs(exp[k]).max()
        errs.append(np.abs(a - exp[k]).max() / max(top, term[k], 1e-300))
        under = max(under, term[k] / max(top, 1e-300))
    for a, e in ((t[0].grad, em), (t[1].grad, ev), (t[2].grad, ec)):
        a = a.cpu().double().numpy()
        errs.append(np.abs(a - e).max() / max(np.abs(e).max() * under, 1e-300) if np.isfinite(a).all() else np.inf)
    rho2 = con[:, 1] ** 2 / (con[:, 0] * con[:, 2])          # squared correlation of every Gaussian
    return np.array(errs), float(rho2.max())


def test_the_fifty_adversarial_cases_outside_1e5_stay_at_their_measured_level(hip_lib):
    """float32 dense against the fp64 oracle on the cases tools/fuzz_dense.py found outside the bar: float64 is exact
    on every one of them (the kernels' arithmetic is right), float32 stays inside the measured level (5e-3 asserted;
    2.2e-3 measured), every such case holds a Gaussian with rho^2 > 0.999 (the three terms of its quadratic form are
    more than 1 000 times their sum: necessary, not sufficient), and the other cases of the first 60 are inside the bar."""
    worst32 = np.zeros(7)
    for seed in OVER_THE_BAR:
        e64, _ = fuzz_errors(seed, torch.float64)
        assert (e64[:4] <= 1e-11).all() and (e64[4:] <= 5e-11).all(), (seed, e64)
        e32, rho2 = fuzz_errors(seed, torch.float32)
        assert np.isfinite(e32).all() and (e32 <= 5e-3).all(), (seed, e32)
        assert rho2 > 0.999, (seed, rho2)
        worst32 = np.maximum(worst32, e32)
    assert worst32.max() > 1e-5          # the limit is real: if this fails the kernels have become better than documented
    inside = [k for k in range(60) if k not in OVER_THE_BAR]
    for seed in inside:
        e32, _ = fuzz_errors(seed, torch.float32)
        assert (e32[:4] <= 1e-5).all() and (e32[4:] <= 5e-5).all(), (seed, e32)
    print("float32 worst over the 50 cases: out0..3 " + " ".join("%.1e" % x for x in worst32[:4]) + " | grads " + " ".join("%.1e" % x for x in worst32[4:]))

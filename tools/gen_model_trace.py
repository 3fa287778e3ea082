#!/usr/bin/env python3
"""Call-trace fixtures of the reference's PINN model for BASELINE config 5 (model_pn.py end to end).

TEST INFRASTRUCTURE, build container only: imports /root/reference/model_pn.py read-only (never
copied; the file cannot travel to the GPU box) and runs a few seeded training steps of
``model_pn.Model`` on the CPU -- the file hard-codes device="cuda" at ~60 sites, so a
TorchFunctionMode rewrites device arguments and ``.cuda()`` becomes the identity -- with a stand-in
``diff_gaussian_sampling.GaussianSampler`` that evaluates every sampler call with a dense float64
torch oracle (orders 0..3, differentiable) and RECORDS, per ``preprocess``:

  * the tensors handed to it (means, values, conics, samples: float32, as the model passes them),
  * every ``sample_*`` output, in call order (float64),
  * the gradient that arrived at each output and the gradients that left towards means / values /
    conics during ``loss.backward()`` (autograd through the float64 oracle),
  * the phase it belongs to: the calls whose outputs were alive together when one backward ran.

The drive loop follows the reference's training loop (main_pn.py:99-232: random collocation and
boundary points, ``randomize`` / ``set_initial_params``, ``sample``, then per timestep ``forward`` ->
``compute_loss`` -> ``backward`` -> ``optim.step`` -> ``clear`` / ``sample`` / ``detach``); main_pn.py
itself cannot be imported (it needs a missing ``model`` module and data files).
``preprocess_aggregate`` / ``aggregate_neighbors`` have no visible reference semantics (parity
unpinned); the stand-in serves them with the repo's dense torch checker (oracle/aggregate_torch.py) so that the model runs.

    python tools/gen_model_trace.py            # writes tests/golden/model_pn_trace_*.npz

tests/test_model_trace_gpu.py replays every record through the HIP sampler.
"""
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)


class CudaToCpu(TorchFunctionMode):
    """device="cuda" -> "cpu" for every torch call made while the mode is active."""

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        dev = kwargs.get("device")
        if dev is not None and torch.device(dev).type == "cuda":
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


# ---------------------------------------------------------------------------------------------
# dense float64 oracle, differentiable (torch autograd): orders 0..3 of
#   u = sum_n v_n exp(-x^T C_n x / 2)   (gaussians.py:48-58, 89-116; order 3: model_pn.py:654)
# ---------------------------------------------------------------------------------------------
def dense_orders(means, values, conics, samples, orders):
    N, d = means.shape
    if d == 1:
        C = conics.reshape(N, 1, 1)
    else:
        a, b, c = conics[:, 0], conics[:, 1], conics[:, 2]
        C = torch.stack((torch.stack((a, b), -1), torch.stack((b, c), -1)), -2)      # [N, d, d]
    x = samples[:, None, :] - means[None, :, :]                                      # [M, N, d]
    p = torch.einsum("nij,mnj->mni", C, x)
    g = torch.exp(-0.5 * (x * p).sum(-1))                                            # [M, N]
    out = {}
    if 0 in orders:
        out[0] = g @ values
    if 1 in orders:
        out[1] = -torch.einsum("mn,mni,nc->mic", g, p, values)
    if 2 in orders:
        t = p[..., :, None] * p[..., None, :] - C[None]
        out[2] = torch.einsum("mn,mnij,nc->mijc", g, t, values)
    if 3 in orders:
        ppp = p[..., :, None, None] * p[..., None, :, None] * p[..., None, None, :]
        cp = (C[None, :, :, :, None] * p[:, :, None, None, :] + C[None, :, :, None, :] * p[:, :, None, :, None]
              + C[None, :, None, :, :] * p[:, :, :, None, None])
        out[3] = torch.einsum("mn,mnijk,nc->mijkc", g, cp - ppp, values)
    return out


TRACE = []          # records of the run in progress
PHASE = [0]         # current phase id (bumped by the drive loop after every backward)


class RecordingSampler:
    """Stand-in for diff_gaussian_sampling.GaussianSampler (interface: SURVEY.md 8b)."""

    def __init__(self, flag=False):
        self.rec = None

    def preprocess(self, means, values, covariances, conics, samples):
        if values.dim() == 1:
            values = values.reshape(-1, 1)
        if samples.dim() == 1:
            samples = samples.reshape(-1, 1)
        rec = {"means": means.detach().clone(), "values": values.detach().clone(),
               "conics": conics.detach().reshape(means.shape[0], -1).clone(), "samples": samples.detach().clone(),
               "calls": [], "out": {}, "gout": {}, "phase": PHASE[0], "grad_mode": torch.is_grad_enabled()}
        TRACE.append(rec)
        self.rec = rec
        # identity nodes: whatever gradient reaches them came through THIS call's outputs only
        self.m, self.v, self.c = means.view_as(means), values.view_as(values), conics.reshape(means.shape[0], -1)
        self.c = self.c.view_as(self.c)
        self.s = samples.detach()
        if torch.is_grad_enabled():
            for name, t in (("gmeans", self.m), ("gvalues", self.v), ("gconics", self.c)):
                if t.requires_grad:
                    t.register_hook(lambda g, r=rec, k=name: r.__setitem__(k, g.detach().clone()))
        self._agg = None

    def _order(self, o):
        rec = self.rec
        out64 = dense_orders(self.m.double(), self.v.double(), self.c.double(), self.s.double(), (o,))[o]
        rec["calls"].append(o)
        rec["out"][o] = out64.detach().clone()
        out = out64.to(self.m.dtype)
        if out.requires_grad:
            out.register_hook(lambda g, r=rec, k=o: r["gout"].__setitem__(k, g.detach().clone()))
        return out

    def sample_gaussians(self):
        return self._order(0)

    def sample_gaussians_derivative(self):
        return self._order(1)

    def sample_gaussians_laplacian(self):
        return self._order(2)

    def sample_gaussians_third_derivative(self):
        return self._order(3)

    # parity unpinned: served by this repo's own definition so that the model runs
    def preprocess_aggregate(self):
        from oracle import aggregate_torch as aggregate
        self._agg = aggregate.neighbor_structure(self.m.detach(), self.c.detach(), 36.0)

    def aggregate_neighbors(self, features, transform, queries, keys, frequencies, distance_transform):
        from oracle import aggregate_torch as aggregate
        mask, delta, g = self._agg
        return aggregate.aggregate(mask, delta.to(features.dtype), g.to(features.dtype), features, transform, queries,
                                   keys, frequencies, distance_transform)


def install():
    mod = types.ModuleType("diff_gaussian_sampling")
    mod.GaussianSampler = RecordingSampler
    sys.modules["diff_gaussian_sampling"] = mod
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def boundary_points(n, scale):
    """n points in the frame around [-scale, scale]^2 (the reference draws its boundary samples the
    same way: two sides per axis, 0..0.5 outside, the other coordinate over 1.5 x the domain)."""
    q = n // 4
    side = torch.cat((-torch.ones(q) - torch.rand(q) * 0.5, torch.ones(q) + torch.rand(q) * 0.5)) * scale
    pts = torch.zeros((n, 2))
    pts[n // 2:, 0] = (torch.rand(n // 2) * 2.0 - 1.0) * 1.5 * scale
    pts[n // 2:, 1] = side
    pts[:n // 2, 1] = (torch.rand(n // 2) * 2.0 - 1.0) * 1.5 * scale
    pts[:n // 2, 0] = side
    return pts


def run(problem_name, n_lattice, n_samples, epochs, timesteps, seed):
    import model_pn
    TRACE.clear()
    PHASE[0] = 0
    torch.manual_seed(seed)
    np.random.seed(seed)
    problem = getattr(model_pn.Problem, problem_name)
    scale, d, dt = 1.0, 2, 1.0
    model = model_pn.Model(problem, model_pn.IntegrationRule.TRAPEZOID, n_lattice, n_lattice, d, scale)
    optim = torch.optim.Adam(model.parameters())
    model.train()
    losses_log = []
    for epoch in range(epochs):
        time_samples = torch.rand(n_samples)
        samples = (torch.rand((n_samples, d)) * 2.0 - 1.0) * scale
        bc_samples = boundary_points(n_samples, scale)
        if problem in (model_pn.Problem.NAVIER_STOKES, model_pn.Problem.WAVE):
            # NAVIER_STOKES: the reference loads fitted Gaussians from files that are not in the tree (main_pn.py:39);
            # WAVE: its randomize() path fails in the reference itself (reset() concatenates a 2-channel boundary
            # state with a 1-channel random one, model_pn.py:530).  A seeded random 2-channel cloud of the same kind
            # stands in for both through set_initial_params (POISSON cannot be constructed at all: Model.__init__
            # has no channel count for it, model_pn.py:417)
            n = n_lattice * n_lattice
            means = (torch.rand((n, d)) * 2.0 - 1.0) * scale
            values = torch.randn((n, 2)) * 0.2
            scaling = torch.exp(torch.randn((n, d)) * 0.3 - 4.0) * scale
            transforms = torch.tanh(torch.randn((n, 1)) * 0.3)
            model.set_initial_params(means, values, scaling, transforms)
        else:
            model.randomize(n_lattice)
        model.sample(samples, bc_samples)
        PHASE[0] += 1
        for i in range(timesteps):
            model.forward(i * dt, dt, False)
            parts = model.compute_loss(i * dt, dt, samples, time_samples, bc_samples)
            loss = sum(p for p in parts[:4] if torch.isfinite(p).all())
            loss.backward()
            optim.step()
            optim.zero_grad()
            losses_log.append(float(loss))
            PHASE[0] += 1
            model.clear()
            model.sample(samples, bc_samples)
            model.detach()
            PHASE[0] += 1
    return losses_log


def save(path, losses):
    arrays = {"n_records": np.array(len(TRACE)), "losses": np.array(losses)}
    for k, r in enumerate(TRACE):
        for name in ("means", "values", "conics", "samples"):
            arrays[f"r{k}_{name}"] = r[name].numpy().astype(np.float32)
        arrays[f"r{k}_calls"] = np.array(r["calls"], dtype=np.int64)
        arrays[f"r{k}_phase"] = np.array(r["phase"])
        arrays[f"r{k}_grad_mode"] = np.array(int(r["grad_mode"]))
        for o, t in r["out"].items():
            arrays[f"r{k}_out{o}"] = t.numpy().astype(np.float64)
        for o, t in r["gout"].items():
            arrays[f"r{k}_gout{o}"] = t.numpy().astype(np.float32)
        for name in ("gmeans", "gvalues", "gconics"):
            if name in r:
                arrays[f"r{k}_{name}"] = r[name].numpy().astype(np.float64)
    np.savez_compressed(path, **arrays)
    size = os.path.getsize(path)
    nb = sum(1 for r in TRACE if "gmeans" in r)
    print(f"{path}: {len(TRACE)} sampler calls ({nb} with gradients), {size / 1e6:.2f} MB, losses {losses}")


def main():
    install()
    out = os.path.join(ROOT, "tests", "golden")
    with CudaToCpu():
        for name, n_lattice, n_samples, epochs, timesteps, seed in (
                ("DIFFUSION", 12, 256, 2, 2, 1),
                ("NAVIER_STOKES", 10, 192, 1, 2, 2),
                ("BURGERS", 9, 128, 1, 2, 3),
                ("WAVE", 9, 128, 1, 2, 4),
                ("TEST", 9, 128, 1, 2, 6)):
            losses = run(name, n_lattice, n_samples, epochs, timesteps, seed)
            save(os.path.join(out, f"model_pn_trace_{name.lower()}.npz"), losses)


if __name__ == "__main__":
    main()

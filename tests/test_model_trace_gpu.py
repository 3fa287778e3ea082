"""BASELINE config 5 (model_pn.py end to end): replay of the reference model's own sampler traffic.

tests/golden/model_pn_trace_*.npz were recorded by tools/gen_model_trace.py in the build container:
``model_pn.Model`` (imported from the reference, run on the CPU) was driven through seeded training
steps of its DIFFUSION, BURGERS, NAVIER_STOKES, WAVE and TEST problems (every problem type the reference's
Model can be constructed for: POISSON has no channel count in Model.__init__, model_pn.py:417) with a
recording stand-in sampler backed by a float64 dense oracle.  Every record holds what one ``preprocess`` was handed, each ``sample_*``
output in call order, the gradients that arrived at the outputs during ``loss.backward()`` and the
gradients that left towards means / values / conics.  Here every record goes through the HIP
sampler exactly as the model drove it (model_pn.py:644-664 sampling at the Gaussian means under
no_grad, NS third derivatives with c = 2; :766-788 two preprocess calls whose outputs stay alive in
lists and are back-propagated after the later preprocess; main_pn.py:171-232 one backward per
timestep) and must reproduce outputs and gradients to 1e-5 (float32 against the float64 record,
relative to the largest magnitude of the tensor).
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "model_pn_trace_*.npz")))
METHODS = ("sample_gaussians", "sample_gaussians_derivative", "sample_gaussians_laplacian",
           "sample_gaussians_third_derivative")
TOL = 1e-5


def rel(got, want):
    got = got.detach().cpu().double().numpy()
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def records(z):
    out = []
    for k in range(int(z["n_records"])):
        r = {name: z[f"r{k}_{name}"] for name in ("means", "values", "conics", "samples", "calls")}
        r["phase"] = int(z[f"r{k}_phase"])
        r["grad_mode"] = bool(z[f"r{k}_grad_mode"])
        r["out"] = {int(o): z[f"r{k}_out{o}"] for o in r["calls"]}
        r["gout"] = {o: z[f"r{k}_gout{o}"] for o in range(4) if f"r{k}_gout{o}" in z.files}
        r["grads"] = {n: z[f"r{k}_g{n}"] for n in ("means", "values", "conics") if f"r{k}_g{n}" in z.files}
        out.append(r)
    return out


def test_fixtures_present():
    assert len(FILES) >= 5, "tests/golden/model_pn_trace_*.npz missing (tools/gen_model_trace.py)"


@pytest.mark.parametrize("backend", ["auto", "binned"])
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[15:-4] for f in FILES])
def test_replay(hip_lib, path, backend):
    from diff_gaussian_sampling import GaussianSampler
    dev = torch.device("cuda", 0)
    recs = records(np.load(path))
    assert any(3 in r["calls"] for r in recs) == ("navier" in path)        # NS asks for third derivatives
    sampler = GaussianSampler(False, backend=backend)                        # ONE sampler, as in the model
    n_fwd = n_bwd = 0
    for phase in sorted(set(r["phase"] for r in recs)):
        alive = []                                                           # (record, leaves, outputs)
        for r in (x for x in recs if x["phase"] == phase):
            leaves = {n: torch.tensor(r[n], device=dev) for n in ("means", "values", "conics")}
            for t in leaves.values():
                t.requires_grad_(bool(r["grads"]))
            samples = torch.tensor(r["samples"], device=dev)
            with torch.set_grad_enabled(r["grad_mode"]):
                sampler.preprocess(leaves["means"], leaves["values"], None, leaves["conics"], samples)
                outs = {}
                for o in r["calls"]:
                    out = getattr(sampler, METHODS[o])()
                    assert tuple(out.shape) == r["out"][o].shape, (o, out.shape)
                    e = rel(out, r["out"][o])
                    assert e < TOL, (os.path.basename(path), "phase", phase, "order", o, e)
                    outs[o] = out
                    n_fwd += 1
            alive.append((r, leaves, outs))
        # the backward of the phase: after every preprocess of it, as loss.backward() comes after Model.sample()
        tensors, grads = [], []
        for r, leaves, outs in alive:
            for o, g in r["gout"].items():
                tensors.append(outs[o])
                grads.append(torch.tensor(g, device=dev))
        if tensors:
            torch.autograd.backward(tensors, grads)
        for r, leaves, outs in alive:
            for n, want in r["grads"].items():
                assert leaves[n].grad is not None, (phase, n)
                e = rel(leaves[n].grad, want.reshape(leaves[n].shape))
                assert e < TOL, (os.path.basename(path), "phase", phase, "grad", n, e)
                n_bwd += 1
    # (the TEST problem's loss does not reach the sampler: its trace is forward calls only)
    assert n_fwd > 0 and (n_bwd > 0 or not any(r["grads"] for r in recs))

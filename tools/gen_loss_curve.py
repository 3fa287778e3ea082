#!/usr/bin/env python3
"""Loss-curve fixture for BASELINE config 5 from the REFERENCE's own sampling functions.

TEST INFRASTRUCTURE, build container only (imports /root/reference/gaussians.py read-only; nothing is copied).
The MLP-free PINN loop of tests/test_training_gpu.py (loop shape of /root/reference/test_no_mlp.py:84-186: the
Gaussians are the parameters, Adam lr 1e-2, 1 024 random collocation points per step, 10 steps fitting an initial
condition, then the diffusion residual against the frozen previous state) is driven through a sampler whose
``sample_gaussians`` / ``_derivative`` / ``_laplacian`` are the reference's ``gaussians.sample_gaussians``,
``gaussian_derivative`` and ``gaussian_derivative2`` (float32, CPU, autograd through them).  The loss per step goes
to tests/golden/ref_loss_curve_no_mlp.npz; tests/test_training_gpu.py holds the HIP sampler's curve to it.

    MPLBACKEND=Agg python tools/gen_loss_curve.py
"""
import os
import sys
import unittest.mock

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import gaussians as ref          # the reference's PyTorch twin of its native sampler


def full(conics):
    """flat [N, 3] (xx, xy, yy: gaussians.py:186-189) -> [N, 2, 2]"""
    a, b, c = conics[:, 0], conics[:, 1], conics[:, 2]
    return torch.stack((torch.stack((a, b), -1), torch.stack((b, c), -1)), -2)


class ReferenceSampler:
    """The sampler's surface over the reference's functions (argument order: means, full conics, values, samples)."""

    def __init__(self, *_):
        pass

    def preprocess(self, means, values, covariances, conics, samples):
        self.args = (means, full(conics), values, samples)

    def sample_gaussians(self):
        return ref.sample_gaussians(*self.args)

    def sample_gaussians_derivative(self):
        return ref.gaussian_derivative(*self.args)

    def sample_gaussians_laplacian(self):
        real_ones = torch.ones      # gaussians.py:110 asks for a CUDA tensor of ones: redirected to the CPU at call time
        with unittest.mock.patch.object(torch, "ones", lambda *a, **k: real_ones(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})):
            return ref.gaussian_derivative2(*self.args)


def main():
    from test_training_gpu import run_loop
    torch.set_num_threads(8)
    steps = 30
    losses = run_loop(ReferenceSampler(), torch.device("cpu"), steps=steps)
    path = os.path.join(ROOT, "tests", "golden", "ref_loss_curve_no_mlp.npz")
    np.savez_compressed(path, losses=losses, steps=np.array(steps))
    print(path, losses[:3], "...", losses[-3:])


if __name__ == "__main__":
    main()

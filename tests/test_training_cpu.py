"""CPU: the torch oracle (oracle/dense_torch.py) drives the MLP-free PINN loop of tests/test_training_gpu.py to the
loss curve the REFERENCE's own sampling functions produce (tests/golden/ref_loss_curve_no_mlp.npz, recorded by
tools/gen_loss_curve.py in the build container): the checker the GPU loss-curve test leans on is pinned by the
reference at the level of a training run, not only call by call."""
import os

import numpy as np
import torch

from conftest import GOLDEN


def test_oracle_loop_reproduces_the_reference_loss_curve():
    import test_training_gpu as T
    ref = np.load(os.path.join(GOLDEN, "ref_loss_curve_no_mlp.npz"))["losses"]
    torch.set_num_threads(8)
    cpu = T.run_loop(T.OracleSampler(), torch.device("cpu"), steps=len(ref))
    rel = np.abs(cpu - ref) / np.maximum(np.abs(ref), 1e-12)
    assert rel.max() < 1e-3, (rel.max(), cpu, ref)

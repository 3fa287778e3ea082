#!/usr/bin/env python3
"""BASELINE.json configs[4] in the form that needs only the sampler (the reference's main_pn.py
imports a module and data that are not in its tree): the MLP-free PINN loop of
tests/test_training_gpu.py (shape of /root/reference/test_no_mlp.py:84-186: Adam lr 1e-2, 1 024
random collocation points per step, 10 steps fitting the initial condition, then the diffusion
residual) driven through the HIP sampler on the GPU and through the torch oracle on the CPU.
Writes both loss curves to profiles/r01_loss_curve_c5.json (argv[1] = steps, default 120)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_training_gpu import run_loop, OracleSampler
from diff_gaussian_sampling import GaussianSampler

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
t0 = time.perf_counter()
gpu = run_loop(GaussianSampler(False), torch.device("cuda"), steps=steps)
torch.cuda.synchronize()
t1 = time.perf_counter()
torch.set_num_threads(min(16, os.cpu_count() or 1))
cpu = run_loop(OracleSampler(), torch.device("cpu"), steps=steps)
t2 = time.perf_counter()
rel = np.abs(gpu - cpu) / np.maximum(np.abs(cpu), 1e-12)
out = {"what": "MLP-free PINN loop (test_no_mlp.py shape), 256 Gaussians, 1024 random collocation points per step, "
               "Adam lr 1e-2; loss per step through the HIP sampler (GPU, float32) and the torch oracle (CPU, float32)",
       "steps": steps, "loss_hip": gpu.tolist(), "loss_oracle": cpu.tolist(),
       "max_rel_deviation": float(rel.max()), "max_rel_deviation_first_30": float(rel[:30].max()),
       "seconds_hip_eager": t1 - t0, "seconds_oracle_cpu": t2 - t1}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "loss_curve_c5.json"), "w"), indent=1)
print("steps %d: loss %.3e -> %.3e (fit) -> %.3e (residual, step 10) -> %.3e (last); max rel deviation HIP vs oracle %.2e "
      "(first 30 steps %.2e); %.2f s HIP eager, %.2f s oracle CPU" % (
          steps, gpu[0], gpu[9], gpu[10], gpu[-1], rel.max(), rel[:30].max(), t1 - t0, t2 - t1))

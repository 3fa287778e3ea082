"""GPU: the C ABI from a plain C++ host (tests/abi_example/host_example.cpp: hipMalloc + hipStream_t,
no torch, no Python in the process), compiled here with hipcc against include/pigs_amd.h and linked to
pigs_amd/libpigs_amd.so; its results against the C oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import grads_within_accumulation_bound
from oracle import c_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_cpp_host_links_and_matches_oracle(hip_lib, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "host_example"
    libdir = os.path.join(ROOT, "pigs_amd")
    cmd = [hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_example", "host_example.cpp"), "-o", str(exe),
           "-L", libdir, "-lpigs_amd", f"-Wl,-rpath,{libdir}"]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)

    rng = np.random.default_rng(5)
    N, M, c = 700, 9000, 1
    means = rng.uniform(-1, 1, (N, 2))
    s = np.exp(2 * rng.normal(-3.3, 0.4, (N, 2)))
    tau = np.tanh(rng.normal(0, 0.6, N)) * np.sqrt(s[:, 0] * s[:, 1])
    det = s[:, 0] * s[:, 1] - tau ** 2
    conics = np.stack((s[:, 1] / det, -tau / det, s[:, 0] / det), -1)
    values = rng.uniform(-1, 1, (N, c))
    samples = rng.uniform(-1, 1, (M, 2))
    gouts = [rng.uniform(-1, 1, sh) for sh in ((M, c), (M, 2, c), (M, 2, 2, c))]
    f32 = [a.astype(np.float32) for a in (means, conics, values, samples, *gouts)]
    with open(tmp_path / "case.bin", "wb") as f:
        np.array([N, M, c], dtype=np.int64).tofile(f)
        for a in f32:
            a.tofile(f)
    out = subprocess.run([str(exe), str(tmp_path / "case.bin"), str(tmp_path / "res.bin")], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    res = np.fromfile(tmp_path / "res.bin", dtype=np.float32).astype(np.float64)
    sizes = [M * c, M * 2 * c, M * 4 * c] * 2 + [N * 2, N * 3, N * c]
    parts = np.split(res, np.cumsum(sizes)[:-1])
    args = [a.astype(np.float64) for a in f32[:4]]
    exp = c_oracle.forward(*args, orders=(0, 1, 2))
    for k in range(3):
        assert rel(parts[k], exp[k].ravel()) < 1e-5, ("dense", k)
        assert rel(parts[3 + k], exp[k].ravel()) < 1e-5, ("binned", k)
    # gradients: per entry within a few ulp of the sum of the absolute contributions (conftest.py)
    bad = grads_within_accumulation_bound((parts[6], parts[7], parts[8]), args,
                                          {k: f32[4 + k].astype(np.float64) for k in range(3)})
    assert not bad, bad

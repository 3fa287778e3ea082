"""GPU: the multi-GPU path (points sharded, Gaussians replicated, ONE all-reduce of the parameter
gradients) with the HIP sampler on device tensors -- two gloo ranks on the test box's one GPU, started as a
fresh child process (RCCL needs one GPU per rank; the collective's backend is not what is under test here: the
sharding, the in-place reduction of the sampler's own gradient buffer and the kernels on a shard are)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_sharded_points_equal_the_unsharded_gradients(hip_lib, tmp_path):
    out = tmp_path / "report.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(out)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    reports = json.load(open(out))
    assert len(reports) == 2
    for rep in reports:
        assert rep["world"] == 2 and rep["ranks_agree"]
        for key in ("binned/native", "binned/ctypes", "dense/native", "dense/ctypes"):
            # outputs of a shard against the same points of the unsharded launch; gradients after the all-reduce
            # against the unsharded backward (float32 sums in another order: 1e-5 of the largest entry)
            assert rep[key]["out"] < 2e-6, (key, rep[key])
            assert rep[key]["grad"] < 1e-5, (key, rep[key])
        # the sampler's three gradients arrive as one flat allocation: reduced in place, no packing copy
        assert rep["single_buffer"] and all(rep["single_buffer"]), rep["single_buffer"]

"""CPU: the C-ABI library builds, loads and exports every symbol include/pigs_amd.h declares
(no compute call is made here: there is no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "pigs_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pigs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = header_functions()
    for required in ("pigs_abi_version", "pigs_sample_forward", "pigs_sample_backward"):
        assert required in names


def test_library_exports_every_declared_symbol(hip_lib):
    from pigs_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(raw, name), f"{name} declared in include/pigs_amd.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in pigs_amd/_lib.py"
    for name in _lib.SIGNATURES:
        assert name in header_functions(), f"{name} bound in _lib.py but not declared in the header"


def test_abi_version_and_status_strings(hip_lib):
    from pigs_amd import _lib
    assert hip_lib.pigs_abi_version() == _lib.ABI_VERSION
    assert hip_lib.pigs_status_string(0) == b"ok"
    assert b"unsupported" in hip_lib.pigs_status_string(2)


def test_argument_validation_needs_no_gpu(hip_lib):
    """Bad arguments are rejected before any HIP call."""
    null = ctypes.c_void_p(0)
    f = hip_lib.pigs_sample_forward
    assert f(0, 3, 1, 1, 4, 4, *([null] * 9)) == 2          # d = 3 unsupported
    assert f(0, 2, 9, 1, 4, 4, *([null] * 9)) == 2          # c = 9 unsupported
    assert f(7, 2, 1, 1, 4, 4, *([null] * 9)) == 2          # dtype
    assert f(0, 2, 1, 0, 4, 4, *([null] * 9)) == 1          # empty mask
    assert f(0, 2, 1, 1, -1, 4, *([null] * 9)) == 1         # negative size
    assert f(0, 2, 1, 1, 4, 4, *([null] * 9)) == 1          # null inputs with N, M > 0


def test_product_path_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under pigs_amd/ or diff_gaussian_sampling/
    may import it."""
    for pkg in ("pigs_amd", "diff_gaussian_sampling"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".h")):
                    text = open(os.path.join(dirpath, fn)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), (dirpath, fn)
                    assert "pigs_oracle" not in text, (dirpath, fn)


def test_sampler_rejects_cpu_tensors(hip_lib):
    import torch
    from diff_gaussian_sampling import GaussianSampler
    s = GaussianSampler(True)
    means = torch.zeros(4, 2); values = torch.ones(4, 1); con = torch.ones(4, 3); pts = torch.zeros(8, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.preprocess(means, values, con, con, pts)
    with pytest.raises(RuntimeError, match="preprocess"):
        s.sample_gaussians()


def test_workspace_sizes_cover_the_coarse_bin_build_and_the_staging(hip_lib):
    """Sizes are pure functions of (N, M, c): the samples workspace holds the coarse-bin build's count matrix,
    its scan and the 16-byte temporary records (include/pigs_amd.h, pigs_samples_build), the plan workspace one
    32-byte staging record per sample point; both grow monotonically with M."""
    last_s = last_p = 0
    for M in (1, 63, 64, 1000, 32768, 1 << 20, (1 << 20) + 5):
        sb = hip_lib.pigs_samples_workspace_bytes(M)
        pb = hip_lib.pigs_plan_workspace_bytes(4096, M, 1)
        assert sb >= (8 + 12 + 16) * M and pb >= (160 + 32) * M
        assert sb >= last_s and pb >= last_p
        last_s, last_p = sb, pb
    assert hip_lib.pigs_samples_workspace_bytes(0) == 0          # unsupported sizes report 0
    assert hip_lib.pigs_samples_order_hint(12345) == -1          # nothing remembered (and no GPU needed to ask)

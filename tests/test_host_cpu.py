"""CPU: host-side bookkeeping of the sampler that needs no device."""


def test_plan_pool_is_bounded_and_keyed():
    from pigs_amd.sampler import _PlanPool
    pool = _PlanPool()
    for k in range(10):                                  # ten problem sizes, one dead plan each
        pool.give(("sizes", k), object())
    assert sum(len(v) for v in pool.free.values()) == pool.KEEP_TOTAL
    assert pool.take(("sizes", 0)) is None               # the oldest went back to the allocator
    a, b, c = object(), object(), object()
    for ws in (a, b, c):
        pool.give(("sizes", 9), ws)
    assert len(pool.free[("sizes", 9)]) == pool.KEEP     # per key
    assert sum(len(v) for v in pool.free.values()) <= pool.KEEP_TOTAL
    got = pool.take(("sizes", 9))
    assert got is not None and pool.take(("other", 9)) is None


def test_native_host_extension_loads_and_rejects_cpu_tensors(hip_lib):
    """The native host side (pigs_amd/_pigs_host.so) is built by build(), imports without a GPU, was
    compiled against the ABI the library reports, and has no CPU path: same errors as the ctypes host."""
    import pytest
    import torch
    from pigs_amd import _lib, _pigs_host
    from pigs_amd.sampler import GaussianSampler
    assert _pigs_host.ABI_VERSION == _lib.ABI_VERSION == _lib.load().pigs_abi_version()
    for name in ("SamplerCore", "Plan", "SamplePlan", "forward_raw", "backward_raw"):
        assert hasattr(_pigs_host, name)
    means = torch.zeros(4, 2); values = torch.ones(4, 1); con = torch.ones(4, 3); pts = torch.zeros(8, 2)
    for host in ("native", "ctypes"):
        s = GaussianSampler(True, host=host)
        assert (s._core is not None) == (host == "native")
        with pytest.raises(RuntimeError, match="preprocess"):
            s.sample_gaussians()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            s.preprocess(means, values, con, con, pts)
        with pytest.raises(NotImplementedError):
            s.preprocess(torch.zeros(4, 3), values, con, con, pts)
        with pytest.raises(ValueError):
            s.preprocess(torch.zeros(4), values, con, con, pts)
        with pytest.raises(TypeError):
            s.preprocess(means, values, con, con, [0.0, 1.0])
        assert s._inputs is None and s._plan is None
    with pytest.raises(ValueError):
        GaussianSampler(True, host="jit")


def test_host_selection_by_environment(hip_lib, monkeypatch):
    from pigs_amd.sampler import GaussianSampler
    monkeypatch.setenv("PIGS_AMD_HOST", "ctypes")
    assert GaussianSampler(False).host == "ctypes"
    monkeypatch.delenv("PIGS_AMD_HOST")
    assert GaussianSampler(False).host == "native"

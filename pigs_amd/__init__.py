"""pigs_amd -- MI355X-native differentiable Gaussian sampler (the hot path of kr4b/pigs).

Public surface:
    GaussianSampler         drop-in for ``diff_gaussian_sampling.GaussianSampler``
    covariances             fused ``build_covariances`` / ``build_full_covariances`` (gaussians.py:163-193)
    build()                 compile the HIP library in-tree (hipcc, gfx950)
"""
from .build import build  # noqa: F401


def __getattr__(name):
    # lazy: importing the package must not require the built library (build() creates it)
    if name == "GaussianSampler":
        from .sampler import GaussianSampler
        return GaussianSampler
    raise AttributeError(name)

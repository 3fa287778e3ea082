"""pigs_amd -- MI355X-native differentiable Gaussian sampler (the hot path of kr4b/pigs).

Public surface:
    GaussianSampler         drop-in for ``diff_gaussian_sampling.GaussianSampler``
    covariances             fused ``build_covariances`` / ``build_full_covariances`` (gaussians.py:163-193)
    build()                 compile the HIP library (hipcc, gfx950) and the native host extension in-tree
"""
from .build import build_all as build  # noqa: F401  (libpigs_amd.so + the native host extension)


def __getattr__(name):
    # lazy: importing the package must not require the built library (build() creates it)
    if name == "GaussianSampler":
        from .sampler import GaussianSampler
        return GaussianSampler
    raise AttributeError(name)

"""Drop-in import name of the reference's native extension.

Every reference script does ``from diff_gaussian_sampling import GaussianSampler``
(/root/reference/model_pn.py:11, test_gaussian_sampling.py:11, test_derivatives.py:9, ...).
This package re-exports the MI355X-native implementation from :mod:`pigs_amd`.
"""
from pigs_amd.sampler import GaussianSampler

__all__ = ["GaussianSampler"]

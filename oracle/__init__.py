"""CPU oracle for the differentiable Gaussian sampler -- TEST INFRASTRUCTURE ONLY.

Nothing in the product path (``pigs_amd/``, ``diff_gaussian_sampling/``) may import
this package.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / reported CPU baseline.

Parity pinning: the restatements here are checked against outputs of the reference's
own PyTorch functions (``/root/reference/gaussians.py:48-58, 89-116``) and against
``torch.autograd`` through them, captured as fixtures under ``tests/golden/`` by
``tools/gen_golden.py`` (run in the build container, where the reference is mounted).
The reference's CUDA extension is an un-vendored submodule and cannot be built;
see DESIGN.md "Oracle".
"""

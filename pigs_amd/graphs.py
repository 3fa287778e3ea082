"""Capture a whole sampler step -- preprocess, sample_*, the loss, autograd -- into one hipGraph.

The C ABI never allocates, frees or synchronises (include/pigs_amd.h), so everything a PINN step
does on the GPU can be recorded once with ``torch.cuda.CUDAGraph`` and replayed; at the sizes the
reference trains with (N ~ 1e3 Gaussians, 1 024 collocation points, main_pn.py:57,103) the step is
bound by the host, and a replay is 3-4x faster than issuing it eagerly.

Two things a capture must get right, which this helper does:
  * the step runs a few times on the capture stream first (library load, allocator, per-stream
    state), and
  * the leaf tensors whose gradients the step asks for are created under that same stream:
    autograd remembers the stream a leaf's accumulation node was first used on, and meeting one
    from another stream inside a capture aborts it.
"""
import torch


class GraphedStep:
    """``step = GraphedStep(fn, make_inputs)``; then ``outputs = step()`` replays the captured graph.

    ``make_inputs()`` is called once, under the capture stream, and returns the tensors ``fn``
    reads (a tuple; leaves that need gradients get ``requires_grad_()`` there).  They are the
    graph's static inputs: write new values into ``step.inputs[i]`` in place (``copy_``, ``add_``
    under ``torch.no_grad()``) between replays.  ``fn(*inputs)`` returns a tensor or a tuple of
    tensors; the same objects are returned by every replay, refreshed in place.
    """

    def __init__(self, fn, make_inputs, warmup=3, device=None, samplers=(), static_samples=False):
        """``samplers`` / ``static_samples``: a capture records the samples build of every ``preprocess`` (a replay
        after an in-place update of a static samples input re-sorts the points), so a replay is a cold step.  When
        the sample points never change between replays, pass the ``GaussianSampler`` objects ``fn`` uses and
        ``static_samples=True``: the capture then reuses the sorted sample structures the warm-up runs built, the
        graph records the Gaussian half only (a warm step per replay), and this object keeps those structures
        alive.  Writing to such a samples input afterwards is NOT seen by the replays."""
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.inputs = tuple(make_inputs())
            for _ in range(max(1, warmup)):
                fn(*self.inputs)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        was = [s.static_samples for s in samplers]
        for s in samplers:
            s.static_samples = bool(static_samples)
        try:
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.outputs = fn(*self.inputs)
        finally:
            for s, w in zip(samplers, was):
                s.static_samples = w
        # the graph's launches read the sample structures the capture reused: they must outlive it
        self._keep = [list(s._sample_plans) for s in samplers] if static_samples else []
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def __call__(self):
        self.graph.replay()
        return self.outputs

"""Host side of the sampler: the reference's ``GaussianSampler`` operator surface on MI355X.

Mirrors the Python-visible interface of the reference's native extension
(``from diff_gaussian_sampling import GaussianSampler``), known from its call sites:

* ``GaussianSampler(flag)``                       model_pn.py:423 (False), tests (True)
* ``preprocess(means, values, covariances, conics, samples)``
                                                  model_pn.py:648,768,784; test_gaussian_sampling.py:56;
                                                  test_derivatives.py:82; test_1d.py:30
* ``sample_gaussians()            -> [M, c]``      model_pn.py:650
* ``sample_gaussians_derivative() -> [M, d, c]``   model_pn.py:651
* ``sample_gaussians_laplacian()  -> [M, d, d, c]`` (the full Hessian) model_pn.py:652
* ``sample_gaussians_third_derivative() -> [M, d, d, d, c]``  model_pn.py:654

Outputs are differentiable wrt the ``means``, ``values`` and ``conics`` passed to the preceding
``preprocess`` (test_derivatives.py:123, 214-215, 349-352), also after later ``preprocess`` calls
(model_pn.py:766-788 then main_pn.py:220): every autograd node owns the tensors it needs.

All arithmetic runs in the HIP library behind the C ABI of include/pigs_amd.h; there is no CPU
or PyTorch fallback -- CPU tensors are rejected.
"""
import ctypes
import os
import threading

import torch

from . import _lib

_ORDER_NAMES = ("sample_gaussians", "sample_gaussians_derivative", "sample_gaussians_laplacian",
                "sample_gaussians_third_derivative")
_DTYPES = {torch.float32: _lib.PIGS_F32, torch.float64: _lib.PIGS_F64}


TRACE = 4       # order index of the Hessian's trace (mask bit 16); it travels in pointer slot 2


def _out_shape(order, M, d, c):
    return (M, c) if order == TRACE else (M,) + (d,) * order + (c,)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    # the current stream's handle: the raw getter skips building a torch.cuda.Stream object (5 us a call)
    if _raw_stream is not None and device.index is not None:
        return ctypes.c_void_p(_raw_stream(device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device:
    """``torch.cuda.device(dev)`` only when ``dev`` is not already current (the context manager
    costs microseconds, which matters for the launch-bound problem sizes of the PINN loops)."""

    def __init__(self, device):
        self.ctx = None if torch.cuda.current_device() == device.index else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


_WORKSPACE_BYTES = {}


def _mask_orders(mask):
    return [k for k in range(5) if mask >> k & 1]


def _slot2(ts):
    """Pointer slot 2 of the C ABI: the full Hessian, or its trace when that is what the mask asks for."""
    return ts[2] if ts[2] is not None else ts[TRACE]


def _error_flag(workspace, offset):
    return int(workspace[offset:offset + 4].view(torch.int32).item())


class SamplePlan:
    """The samples half of what ``preprocess`` builds (C ABI: pigs_samples_*): the sample points
    sorted into 16-point cells.  A function of ``samples`` alone and immutable once built, so one
    object serves every ``preprocess`` that is handed the same, unmodified samples tensor again --
    the reference's roll-out (main_pn.py:317-324) and any fixed collocation grid."""

    __slots__ = ("workspace", "M", "source", "points", "version", "built")

    def __init__(self, samples, source=None):
        lib = _lib.load()
        self.M = samples.shape[0]
        nbytes = _WORKSPACE_BYTES.get(("s", self.M))
        if nbytes is None:
            nbytes = _WORKSPACE_BYTES[("s", self.M)] = lib.pigs_samples_workspace_bytes(self.M)
        if nbytes == 0:
            raise _lib.PigsError(f"binned path does not support M={self.M}")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=samples.device)
        # what this plan was built from: the caller's tensor (kept alive, so its address cannot be
        # handed to another tensor) and its version counter at build time
        self.source = source if source is not None else samples
        # the contiguous array the build reads: in index-tiled order (include/pigs_amd.h, ABI 7) the workspace holds no
        # copy of the points -- the sampling kernels read THIS array, so it lives as long as the workspace
        self.points = samples
        self.version = self.source._version
        self.built = False

    def matches(self, source):
        return (source is self.source or (
            source.data_ptr() == self.source.data_ptr() and source.shape == self.source.shape
            and source.stride() == self.source.stride() and source.dtype == self.source.dtype
            and source.device == self.source.device)) and source._version == self.version

    def scan_took_slow_path(self):
        """Diagnostic (synchronises): a workgroup of the build's scan recomputed a predecessor's total
        itself instead of receiving it (include/pigs_amd.h: the result is valid either way)."""
        return bool(_error_flag(self.workspace, _lib.load().pigs_samples_error_offset()))


class _PlanPool:
    """Plan workspaces whose plan has died, kept for the next ``preprocess`` of the same sizes on the
    same stream: a build leaves its workspace's counters zeroed, so a build INTO such a workspace
    skips the zeroing launch (PIGS_BUILD_PLAN_WS_CLEAN).  A workspace is only ever handed out again
    after its plan object is gone (no autograd node can still read it) and only to the stream its
    last build ran on (stream order then covers kernels that may still be in flight)."""

    KEEP = 2           # per key (sizes, device, stream)
    KEEP_TOTAL = 4     # over all keys: problem sizes that change from step to step must not pile up workspaces

    def __init__(self):
        self.free = {}             # key -> [workspace, ...]; dict order = least recently given first
        # plans die wherever their last reference is dropped -- the autograd engine's worker threads and
        # the garbage collector included -- while the main thread may be inside take()
        # (re-entrant: give() allocates while it holds the lock, a collector pass triggered there can finalise a
        # Plan in a reference cycle, whose __del__ comes back into give() on the same thread)
        self.lock = threading.RLock()

    def take(self, key):
        with self.lock:
            lst = self.free.get(key)
            if not lst:
                return None
            ws = lst.pop()
            if not lst:
                self.free.pop(key, None)
            return ws

    def give(self, key, workspace):
        with self.lock:
            lst = self.free.pop(key, [])
            if len(lst) < self.KEEP:
                lst.append(workspace)
            self.free[key] = lst       # most recently given last
            while sum(len(v) for v in self.free.values()) > self.KEEP_TOTAL:
                oldest = next(iter(self.free))
                self.free[oldest].pop(0)
                if not self.free[oldest]:
                    del self.free[oldest]


class Plan:
    """The Gaussian half of what ``preprocess`` builds (C ABI: pigs_plan_*): the Gaussians binned
    into the multi-level grid and the per-tile lists, on top of a :class:`SamplePlan`.  Immutable
    once built; autograd nodes keep a reference, so later ``preprocess`` calls never disturb a
    pending backward."""

    __slots__ = ("workspace", "samples", "N", "M", "c", "q_max", "q_max_backward", "_pool", "_pool_key",
                 "build_stream", "other_stream_used", "recorded_only")

    BUILD_SAMPLES, WS_CLEAN, DEFER_LISTS = 1, 2, 32      # pigs_amd.h: PIGS_BUILD_SAMPLES, _PLAN_WS_CLEAN, _DEFER_LISTS

    def __init__(self, means, values, conics, samples, q_max, sample_plan=None, source=None, pool=None,
                 recorded_only=False, q_max_backward=None, defer_lists=False):
        lib = _lib.load()
        self.N, self.M, self.c, self.q_max = means.shape[0], samples.shape[0], values.shape[1], float(q_max)
        self.q_max_backward = max(self.q_max, float(q_max_backward if q_max_backward is not None else q_max))
        self._pool = None
        self.other_stream_used = False
        self.recorded_only = bool(recorded_only)      # built inside a capture: has run only if that graph was replayed
        key = (self.N, self.M, self.c)
        nbytes = _WORKSPACE_BYTES.get(key)
        if nbytes is None:
            nbytes = _WORKSPACE_BYTES[key] = lib.pigs_plan_workspace_bytes(self.N, self.M, self.c)
        if nbytes == 0:
            raise _lib.PigsError(f"binned path does not support N={self.N} M={self.M} c={self.c}")
        if sample_plan is None:
            sample_plan = SamplePlan(samples, source)
        self.samples = sample_plan
        with _on_device(means.device):
            stream = _stream(means.device)
            self.build_stream = stream.value
            self._pool_key = (self.N, self.M, self.c, means.device, stream.value)
            self.workspace = pool.take(self._pool_key) if pool is not None else None
            flags = 0 if sample_plan.built else self.BUILD_SAMPLES
            if defer_lists:       # the tile lists are built by the plan's first sampling call, a forward in the same launch
                flags |= self.DEFER_LISTS
            if self.workspace is not None:
                flags |= self.WS_CLEAN
            else:
                self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=means.device)
            sws = sample_plan.workspace
            rc = lib.pigs_plan_build(_ptr(self.workspace), nbytes, _ptr(sws), sws.numel(),
                                     flags, self.N, self.M, self.c, self.q_max, self.q_max_backward,
                                     _ptr(means), _ptr(conics), _ptr(values), _ptr(samples), stream)
        _lib.check(rc, "pigs_plan_build")
        # a build that was only RECORDED into a hipGraph has not run: nothing eager may rely on it -- and a
        # SamplePlan that WAS built eagerly stays built whatever is recorded on top of it (lowering the flag
        # would make the next eager preprocess re-sort a workspace older plans still hold tile lists for)
        sample_plan.built = sample_plan.built or not recorded_only
        self._pool = pool            # only a workspace whose build was launched completely goes back

    def note_stream(self, stream_value):
        """A launch on another stream than the build's reads the workspace: stream order no longer
        covers its reuse, so it does not go back to the pool -- and the caching allocator, which hands a freed
        block out again in the order of its allocation stream, is told about the other stream."""
        if stream_value != self.build_stream:
            self.other_stream_used = True
            st = torch.cuda.current_stream(self.workspace.device)
            if st.cuda_stream == stream_value:
                self.workspace.record_stream(st)
                self.samples.workspace.record_stream(st)

    def __del__(self):
        try:
            if self._pool is not None and not self.other_stream_used:
                self._pool.give(self._pool_key, self.workspace)
        except Exception:            # interpreter shutdown: nothing to keep
            pass

    def scan_took_slow_path(self):
        """Diagnostic (synchronises): see :meth:`SamplePlan.scan_took_slow_path`."""
        return self.samples.scan_took_slow_path() or bool(
            _error_flag(self.workspace, _lib.load().pigs_plan_error_offset()))

    @staticmethod
    def supported(means, values, samples):
        return (means.dtype == torch.float32 and means.shape[1] == 2 and 1 <= values.shape[1] <= 2
                and means.shape[0] >= 1 and samples.shape[0] >= 1)


def forward_raw(means, values, conics, samples, mask, plan=None):
    """Launch the forward for the orders in ``mask`` on contiguous device tensors (through the
    plan when given, else dense).  Returns a list of 5 entries (tensor or None): orders 0..3 and the
    Hessian's trace."""
    lib = _lib.load()
    N, d = means.shape
    c = values.shape[1]
    M = samples.shape[0]
    outs = [None] * 5
    for k in _mask_orders(mask):
        outs[k] = torch.empty(_out_shape(k, M, d, c), dtype=means.dtype, device=means.device)
    if M > 0:
        with _on_device(means.device):
            if plan is not None:
                sws = plan.samples.workspace
                if hasattr(plan, "note_stream"):
                    plan.note_stream(_stream(means.device).value)
                rc = lib.pigs_plan_forward(_ptr(plan.workspace), plan.workspace.numel(), _ptr(sws), sws.numel(),
                                           N, M, c, plan.q_max, mask,
                                           _ptr(outs[0]), _ptr(outs[1]), _ptr(_slot2(outs)), _ptr(outs[3]),
                                           _stream(means.device))
                _lib.check(rc, "pigs_plan_forward")
            else:
                rc = lib.pigs_sample_forward(_DTYPES[means.dtype], d, c, mask, N, M, _ptr(means), _ptr(conics),
                                             _ptr(values), _ptr(samples), _ptr(outs[0]), _ptr(outs[1]),
                                             _ptr(_slot2(outs)), _ptr(outs[3]), _stream(means.device))
                _lib.check(rc, "pigs_sample_forward")
    return outs


def _gradient_views(means, values, conics):
    """The three parameter gradients as views of ONE flat allocation [means | values | conics]: the multi-GPU path
    (pigs_amd/distributed.py) all-reduces that buffer in place -- no packing copy in front of the collective, no
    slicing behind it."""
    nm, nv, nc = means.numel(), values.numel(), conics.numel()
    flat = torch.empty(nm + nv + nc, dtype=means.dtype, device=means.device)
    return flat[:nm].view(means.shape), flat[nm:nm + nv].view(values.shape), flat[nm + nv:].view(conics.shape)


def backward_raw(means, values, conics, samples, gouts, mask, plan=None):
    """Launch the backward; ``gouts`` has 5 entries (contiguous tensor or None: orders 0..3 and the
    trace), ``mask`` marks the non-None ones.  Returns (g_means, g_values, g_conics)."""
    lib = _lib.load()
    N, d = means.shape
    c = values.shape[1]
    M = samples.shape[0]
    g_means, g_values, g_conics = _gradient_views(means, values, conics)
    if N > 0:
        with _on_device(means.device):
            if plan is not None and M > 0:
                sws = plan.samples.workspace
                if hasattr(plan, "note_stream"):
                    plan.note_stream(_stream(means.device).value)
                rc = lib.pigs_plan_backward(_ptr(plan.workspace), plan.workspace.numel(), _ptr(sws), sws.numel(),
                                            N, M, c, plan.q_max, mask,
                                            _ptr(gouts[0]), _ptr(gouts[1]), _ptr(_slot2(gouts)),
                                            _ptr(gouts[3]), _ptr(g_means), _ptr(g_conics), _ptr(g_values),
                                            _stream(means.device))
                _lib.check(rc, "pigs_plan_backward")
            else:
                rc = lib.pigs_sample_backward(_DTYPES[means.dtype], d, c, mask, N, M, _ptr(means), _ptr(conics),
                                              _ptr(values), _ptr(samples), _ptr(gouts[0]), _ptr(gouts[1]),
                                              _ptr(_slot2(gouts)), _ptr(gouts[3]), _ptr(g_means), _ptr(g_conics),
                                              _ptr(g_values), _stream(means.device))
                _lib.check(rc, "pigs_sample_backward")
    return g_means, g_values, g_conics


def _residual_call(backward, means, values, conics, samples, coeffs, plan, target=None, gout=None):
    """pigs_residual_forward / _backward on contiguous device tensors (through the plan when given)."""
    lib = _lib.load()
    N, d = means.shape
    c = values.shape[1]
    M = samples.shape[0]
    cf = (ctypes.c_double * 4)(*coeffs)
    pw = (_ptr(plan.workspace), plan.workspace.numel(), _ptr(plan.samples.workspace), plan.samples.workspace.numel()) \
        if plan is not None else (ctypes.c_void_p(0), 0, ctypes.c_void_p(0), 0)
    with _on_device(means.device):
        stream = _stream(means.device)
        if plan is not None and hasattr(plan, "note_stream"):
            plan.note_stream(stream.value)
        if not backward:
            out = torch.empty((M, c), dtype=means.dtype, device=means.device)
            if M > 0:
                rc = lib.pigs_residual_forward(_DTYPES[means.dtype], d, c, N, M, _ptr(means), _ptr(conics), _ptr(values),
                                               _ptr(samples), cf, _ptr(target), _ptr(out), *pw, stream)
                _lib.check(rc, "pigs_residual_forward")
            return out
        g_means, g_values, g_conics = _gradient_views(means, values, conics)
        if N > 0:
            if M > 0:
                rc = lib.pigs_residual_backward(_DTYPES[means.dtype], d, c, N, M, _ptr(means), _ptr(conics), _ptr(values),
                                                _ptr(samples), cf, _ptr(gout), _ptr(g_means), _ptr(g_conics), _ptr(g_values),
                                                *pw, stream)
                _lib.check(rc, "pigs_residual_backward")
            else:
                for g in (g_means, g_values, g_conics):
                    g.zero_()
        return g_means, g_values, g_conics


class _ResidualFunction(torch.autograd.Function):
    """r = a0 u + a1 . grad u + aL lap u - target in one launch; its backward is one launch too.  The node
    owns its inputs and plan like :class:`_SampleFunction`."""

    @staticmethod
    def forward(ctx, means, values, conics, samples, target, coeffs, debug, plan):
        tgt = None if target is None else target.detach().to(means.dtype).contiguous()
        out = _residual_call(False, means, values, conics, samples, coeffs, plan, target=tgt)
        if debug:
            torch.cuda.synchronize(means.device)
        ctx.inputs = (means, values, conics, samples)
        ctx.versions = (means._version, values._version, conics._version, samples._version)
        ctx.coeffs, ctx.debug, ctx.plan = coeffs, debug, plan
        ctx.target_dtype = None if target is None else target.dtype
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        means, values, conics, samples = ctx.inputs
        if (means._version, values._version, conics._version, samples._version) != ctx.versions:
            raise RuntimeError("one of the tensors handed to GaussianSampler.preprocess() has been modified in place "
                               "before the backward of a residual() output that was computed from it")
        gout = gout.contiguous()
        g_means, g_values, g_conics = _residual_call(True, means, values, conics, samples, ctx.coeffs, ctx.plan, gout=gout)
        if ctx.debug:
            torch.cuda.synchronize(means.device)
        g_target = None if ctx.target_dtype is None or not ctx.needs_input_grad[4] else (-gout).to(ctx.target_dtype)
        return g_means, g_values, g_conics, None, g_target, None, None, None


class _SampleFunction(torch.autograd.Function):
    """One fused launch producing the outputs of every order in ``mask``; its backward is one
    fused launch over the outputs that received a gradient.

    The reference's scripts treat the outputs of the separate ``sample_*`` calls as independent graphs:
    they differentiate one component after the other, some calls with ``retain_graph=True`` and the
    last one on an output without (test_derivatives.py:214-215), and come back to another output of
    the same ``preprocess`` later (test_derivatives.py:349-352).  Outputs that were computed by ONE
    fused launch share this node, so the node must survive a backward that does not retain the graph:
    the inputs are therefore kept on the node itself rather than through ``save_for_backward`` (whose
    storage autograd releases after the first such backward), with the version check that
    ``save_for_backward`` would have done repeated by hand."""

    @staticmethod
    def forward(ctx, means, values, conics, samples, mask, debug, plan):
        outs = forward_raw(means, values, conics, samples, mask, plan)
        if debug:
            torch.cuda.synchronize(means.device)
        ctx.inputs = (means, values, conics, samples)
        ctx.versions = (means._version, values._version, conics._version, samples._version)
        ctx.mask = mask
        ctx.debug = debug
        ctx.plan = plan
        ctx.set_materialize_grads(False)
        return tuple(outs[k] for k in _mask_orders(mask))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grad_outputs):
        means, values, conics, samples = ctx.inputs
        if (means._version, values._version, conics._version, samples._version) != ctx.versions:
            raise RuntimeError("one of the tensors handed to GaussianSampler.preprocess() has been modified in place "
                               "before the backward of a sample_*() output that was computed from it")
        gouts = [None] * 5
        mask = 0
        for k, g in zip(_mask_orders(ctx.mask), grad_outputs):
            if g is not None:
                gouts[k] = g.contiguous()
                mask |= 1 << k
        if mask == 0:
            return None, None, None, None, None, None, None
        g_means, g_values, g_conics = backward_raw(means, values, conics, samples, gouts, mask, ctx.plan)
        if ctx.debug:
            torch.cuda.synchronize(means.device)
        return g_means, g_values, g_conics, None, None, None, None


_FUSE_CODES = {"auto": 0, "all": 1, "none": 2}
_BACKEND_CODES = {"auto": 0, "dense": 1, "binned": 2}


def _load_native_host():
    """The native host extension (pigs_amd/_pigs_host.so, csrc_host/pigs_host.cpp); there is no silent
    fallback to the ctypes host: a missing extension raises."""
    _lib.load()
    variant = os.environ.get("PIGS_AMD_LIB")
    if variant and os.path.realpath(variant) != os.path.realpath(os.path.join(_lib.HERE, "libpigs_amd.so")):
        # _pigs_host.so resolves libpigs_amd.so through its $ORIGIN rpath: the launches would run the in-tree
        # library while the ctypes calls ran the variant -- two instances with separate state, an A/B run
        # timing the wrong kernels without noticing
        raise ImportError(f"PIGS_AMD_LIB={variant} selects a variant library, which only the ctypes host can drive: "
                          "construct GaussianSampler(..., host='ctypes') or set PIGS_AMD_HOST=ctypes")
    try:
        from . import _pigs_host
    except ImportError as e:
        raise ImportError(
            "pigs_amd/_pigs_host.so (the native host side of GaussianSampler) is missing or does not load: "
            f"{e}.  Build it first (python -m pigs_amd.build), or ask for the ctypes host explicitly "
            "(GaussianSampler(..., host='ctypes') / PIGS_AMD_HOST=ctypes).") from e
    if _pigs_host.ABI_VERSION != _lib.ABI_VERSION:
        raise ImportError(f"_pigs_host.so was built against ABI {_pigs_host.ABI_VERSION}, expected {_lib.ABI_VERSION}; rebuild")
    return _pigs_host


class GaussianSampler:
    """MI355X-native replacement of ``diff_gaussian_sampling.GaussianSampler``.

    ``flag`` is the single positional bool of the reference constructor (True in its tests,
    False in model_pn.py:423; meaning not visible in the reference).  Here it is a debug switch:
    when set, every launch is followed by a device synchronise so that errors surface at the call.

    ``backend`` (extension, keyword only): ``"dense"`` evaluates every (point, Gaussian) pair --
    the reference's dense semantics exactly; ``"binned"`` builds the culling plan in ``preprocess``
    and drops pairs with q > ``q_max`` (relative truncation below exp(-q_max/2); launches that
    compute third derivatives use the wider ``q_max_order3``, default q_max + 8, through a second
    plan built on first use: dropped terms carry a q^1.5 prefactor there, q^2.5 in its conic
    gradient -- tools/fuzz_stats.py: 2 of 300 adversarial cases left the 1e-5 bar at 36, none at
    40; the backward of gradients that arrive at second derivatives (or the trace) uses
    ``q_max_backward``, default q_max + 4, through wider group masks kept in the SAME plan: the conic
    gradient of such a term carries q^2 and its sum nearly cancels -- tools/fuzz_diag.py, worst of the four
    worst fuzz cases: 2.5e-5 of the largest entry at 36, 3.5e-6 at 40 -- the dense kernel's own float32
    error on those cases is 3.3e-6 -- 3.4e-6 at 44); ``"auto"`` picks
    binned for float32, d = 2, c <= 2 once N*M >= 2**26 pairs, where the plan pays for itself.

    ``reuse_samples`` (extension, keyword only; binned path): when ``preprocess`` is called again with a
    samples tensor it has seen recently (same storage, shape and version counter, i.e. not written
    to in between) the sorted sample structure built for it is reused and only the Gaussian half
    of the plan is rebuilt -- the reference's roll-out binds new Gaussians to a fixed grid every
    step (main_pn.py:317-324), its training step alternates between the collocation points and
    the boundary points (model_pn.py:766-785).  ``True`` remembers the last 4 sample tensors, an
    integer that many, ``False`` rebuilds everything every time.  While a hipGraph is being captured
    nothing is looked up or remembered: the capture records the samples build itself, so a replay
    after an in-place update of the samples re-sorts them.

    ``unpinned_aggregate`` (extension, keyword only): ``preprocess_aggregate`` / ``aggregate_neighbors``
    follow this repository's own definition (the reference's is not visible: parity unpinned) and
    warn once per process unless this is set.

    ``aggregate_cap`` (extension, keyword only): slots per Gaussian of the neighbour lists of
    ``preprocess_aggregate``.  Default ``None``: sized by a counting pass whose result is read back once per
    call (a host synchronisation); an integer skips that (lists that do not fit are truncated and flagged:
    debug mode raises).

    ``fuse`` (extension, keyword only) controls how many derivative orders one launch computes:
    ``"auto"`` -- the first ``sample_*`` call after a ``preprocess`` computes orders 0..2 in one
    launch when the problem is small enough to be launch-bound (M <= 65536), otherwise only the
    order asked for; ``"all"`` / ``"none"`` force either behaviour.  :meth:`sample` is the
    explicit fused entry point.

    ``defer_lists`` (extension, keyword only; binned path; default False): with True ``preprocess`` stops in front
    of the tile lists and the first ``sample_*`` call builds them in the same launch as its own evaluation
    (PIGS_BUILD_DEFER_LISTS, include/pigs_amd.h).  Built in round 4 to hide the latency-bound list build behind the
    forward's arithmetic; measured at C3 it does not (49.8 us against 21.8 + 26.5 in two launches: every wave builds
    first and evaluates afterwards, in step with all the others -- DESIGN.md section 3.3), so it is an option, not
    the default.

    ``host`` (extension, keyword only): ``"native"`` (default; environment override PIGS_AMD_HOST) keeps
    the sampler's state and its autograd node in the C++ torch extension ``pigs_amd/_pigs_host.so``
    (csrc_host/pigs_host.cpp) -- what the reference's own boundary is (a compiled torch extension,
    model_pn.py:11); ``"ctypes"`` is the same logic in Python over ``ctypes`` (this file).  Both call
    the same C ABI; neither has a CPU fallback.

    Memory: the autograd node of a launch owns the tensors bound by ``preprocess`` and the plan (160 B
    per sample point at N > 512), and keeps them until every output of that launch is gone -- not just
    until the first backward (the reference's scripts come back to another output of the same launch
    after a non-retaining backward, test_derivatives.py:214-215, 349-352); saved-tensor hooks
    (``save_on_cpu``, checkpointing) do not see them.  Roll-outs that keep every step's outputs keep
    every step's plan.
    """

    FUSE_AUTO_MAX_POINTS = 1 << 16
    BINNED_AUTO_MIN_PAIRS = 1 << 26     # dense: ~1.2e12 pairs/s; the plan costs ~32 us to build

    _warned_samples_grad = False
    _warned_aggregate = False

    def __init__(self, flag=False, *, fuse="auto", backend="auto", q_max=36.0, q_max_order3=None,
                 q_max_backward=None, reuse_samples=True, unpinned_aggregate=False, aggregate_cap=None, host=None,
                 defer_lists=False):
        if fuse not in ("auto", "all", "none"):
            raise ValueError("fuse must be 'auto', 'all' or 'none'")
        if backend not in ("auto", "dense", "binned"):
            raise ValueError("backend must be 'auto', 'dense' or 'binned'")
        if not q_max > 0:
            raise ValueError("q_max must be positive")
        if host is None:
            host = os.environ.get("PIGS_AMD_HOST") or "native"
        if host not in ("native", "ctypes"):
            raise ValueError("host must be 'native' or 'ctypes'")
        self.debug = bool(flag)
        self.fuse = fuse
        self.backend = backend
        self.host = host
        self.q_max = float(q_max)
        self.q_max_order3 = float(q_max_order3) if q_max_order3 is not None else self.q_max + 8.0
        if self.q_max_order3 < self.q_max:
            raise ValueError("q_max_order3 must not be below q_max")
        self.q_max_backward = float(q_max_backward) if q_max_backward is not None else self.q_max + 4.0
        if self.q_max_backward < self.q_max:
            raise ValueError("q_max_backward must not be below q_max")
        self.reuse_samples = 4 if reuse_samples is True else max(0, int(reuse_samples))
        self.defer_lists = bool(defer_lists)
        self._static_samples = False
        self.unpinned_aggregate = bool(unpinned_aggregate)
        self.aggregate_cap = None if aggregate_cap is None else int(aggregate_cap)
        self._neighbors = None
        self._st_plan3 = None
        self._st_inputs = None
        self._st_plan = None
        self._st_sample_plans = []       # most recently used first, at most ``reuse_samples``
        self._plan_pool = _PlanPool()
        self._samples_source = None
        self._cache = {}
        _lib.load()  # fail at construction, not at first use, if the HIP library is missing
        self._core = None
        if host == "native":
            self._core = _load_native_host().SamplerCore(self.debug, _FUSE_CODES[fuse], _BACKEND_CODES[backend],
                                                         self.q_max, self.q_max_order3, self.q_max_backward,
                                                         self.reuse_samples)
            self._core.defer_lists = self.defer_lists

    @property
    def static_samples(self):
        """Settable.  While a hipGraph is being captured ``preprocess`` normally records the samples build too (a
        replay after an in-place update of the static samples input re-sorts them), so a replay is a COLD step.
        With ``static_samples = True`` a capture reuses the sorted sample structure that an eager ``preprocess``
        (the capture's warm-up runs) built for the same, unmodified samples tensor: the graph holds the Gaussian
        half only and a replay is a warm step.  The caller promises not to write to that samples tensor between
        replays (``pigs_amd.graphs.GraphedStep(..., samplers=[...], static_samples=True)`` sets this and keeps the
        reused structure alive)."""
        return self._static_samples

    @static_samples.setter
    def static_samples(self, value):
        self._static_samples = bool(value)
        if self._core is not None:
            self._core.static_samples = self._static_samples

    # state lives in the native core when there is one
    @property
    def _plan(self):
        return self._core.plan if self._core is not None else self._st_plan

    @property
    def _plan3(self):
        return self._core.plan3 if self._core is not None else self._st_plan3

    @property
    def _inputs(self):
        return self._core.inputs() if self._core is not None else self._st_inputs

    @property
    def _sample_plans(self):
        return self._core.sample_plans if self._core is not None else self._st_sample_plans

    # ------------------------------------------------------------------ preprocess
    def preprocess(self, means, values, covariances, conics, samples):
        """Bind the Gaussians and the sample points for the following ``sample_*`` calls.

        means [N,d]; values [N,c] (or [N]); covariances and conics flat [N, d(d+1)/2]
        (d=1: anything with N elements, e.g. [N,1] or [N,1,1]); samples [M,d] (d=1: also [M]).
        ``covariances`` does not enter the sampled values (the reference's dense twin,
        gaussians.py:48-58, never reads it); it is accepted for interface parity.
        """
        for name, t in (("means", means), ("values", values), ("conics", conics), ("samples", samples)):
            if not isinstance(t, torch.Tensor):
                raise TypeError(f"{name} must be a torch.Tensor")
        if self._core is not None:
            self._neighbors = None
            self._core.preprocess(means, values, covariances, conics, samples)
            return
        if means.dim() != 2:
            raise ValueError(f"means must be [N, d], got {tuple(means.shape)}")
        N, d = means.shape
        if d not in (1, 2):
            raise NotImplementedError(f"d = {d} is not supported (d in {{1, 2}})")
        nf = d * (d + 1) // 2
        for name, t in (("means", means), ("values", values), ("conics", conics), ("samples", samples)):
            if not t.is_cuda:
                raise RuntimeError(f"{name} is on {t.device}: GaussianSampler runs on the GPU only "
                                   "(no CPU fallback)")
            if t.device != means.device:
                raise RuntimeError(f"{name} is on {t.device}, means on {means.device}")
            if t.dtype != means.dtype:
                raise TypeError(f"{name} has dtype {t.dtype}, means {means.dtype}")
        if means.dtype not in _DTYPES:
            raise TypeError(f"dtype {means.dtype} is not supported (float32 / float64)")
        if values.dim() == 1:
            values = values.reshape(N, 1)
        if values.dim() != 2 or values.shape[0] != N:
            raise ValueError(f"values must be [N, c], got {tuple(values.shape)}")
        c = values.shape[1]
        if not 1 <= c <= 4:
            raise NotImplementedError(f"c = {c} channels is not supported (1..4)")
        if conics.numel() != N * nf:
            raise ValueError(f"conics must hold N*{nf} elements (flat upper triangle), got {tuple(conics.shape)}")
        conics = conics.reshape(N, nf)
        if covariances is not None and isinstance(covariances, torch.Tensor) and covariances.numel() != N * nf:
            raise ValueError(f"covariances must hold N*{nf} elements, got {tuple(covariances.shape)}")
        if samples.dim() == 1 and d == 1:
            samples = samples.reshape(-1, 1)
        if samples.dim() != 2 or samples.shape[1] != d:
            raise ValueError(f"samples must be [M, {d}], got {tuple(samples.shape)}")
        # no gradient flows to the sample points (the reference requests none from the sampler:
        # test_derivatives.py:123 asks for (means, values, conics) only)
        if samples.requires_grad and torch.is_grad_enabled() and not GaussianSampler._warned_samples_grad:
            GaussianSampler._warned_samples_grad = True
            import warnings
            warnings.warn("GaussianSampler: samples.requires_grad is set, but the sampler returns no gradient "
                          "with respect to the sample points (as the reference, whose tests ask for the gradients "
                          "of means, values and conics only); use the derivative outputs instead", stacklevel=2)
        self._st_inputs = (means.contiguous(), values.contiguous(), conics.contiguous(),
                           samples.detach().contiguous())
        self._samples_source = samples
        self._cache = {}
        self._st_plan = None
        self._st_plan3 = None
        self._neighbors = None
        mc, vc, cc, sc = self._st_inputs
        use_plan = self.backend == "binned" or (
            self.backend == "auto" and N * sc.shape[0] >= self.BINNED_AUTO_MIN_PAIRS)
        if use_plan and Plan.supported(mc, vc, sc):
            self._st_plan = self._build_plan(self.q_max)
        elif self.backend == "binned" and N > 0 and sc.shape[0] > 0:
            raise NotImplementedError("backend='binned' needs float32, d = 2, c <= 2")

    def _build_plan(self, q_max, sample_plan=None):
        """A plan for the bound inputs; the samples half is reused when ``preprocess`` was handed an
        unmodified samples tensor it remembers (``reuse_samples``).  While a hipGraph is being captured
        nothing is looked up and nothing is remembered: the capture has to record the samples build
        itself (a replay after an in-place update of the static samples input must re-sort them), its
        workspaces belong to the graph (they neither come from the pool nor go back to it), and a
        SamplePlan that was only recorded has not been built as far as later eager calls go."""
        mc, vc, cc, sc = self._st_inputs
        capturing = torch.cuda.is_current_stream_capturing()
        sp = sample_plan
        if sp is None and (not capturing or self.static_samples):
            sp = next((p for p in self._st_sample_plans if p.built and p.matches(self._samples_source)), None)
        pool = None if capturing else self._plan_pool
        with torch.no_grad():
            plan = Plan(mc.detach(), vc.detach(), cc.detach(), sc, q_max, sp, self._samples_source, pool,
                        recorded_only=capturing, q_max_backward=max(q_max, self.q_max_backward),
                        defer_lists=self.defer_lists)
        if self.reuse_samples and not capturing:
            self._st_sample_plans = [plan.samples] + [p for p in self._st_sample_plans if p is not plan.samples]
            del self._st_sample_plans[self.reuse_samples:]
        if self.debug and not capturing:
            torch.cuda.synchronize(mc.device)
        return plan

    # ------------------------------------------------------------------ sampling
    def _require_inputs(self):
        inputs = self._inputs
        if inputs is None:
            raise RuntimeError("preprocess() must be called before sampling")
        return inputs

    def _plan_for(self, mask):
        """The plan a launch with this order mask runs on: third derivatives get the wider cut-off."""
        if self._st_plan is None:
            return None
        capturing = torch.cuda.is_current_stream_capturing()
        # an eager call behind a capture (no preprocess in between) must not sample what the capture only
        # RECORDED: the plan is rebuilt eagerly, on a samples half of its own (the recorded one belongs to the
        # graph, whose replays re-sort it)
        if self._st_plan.recorded_only and not capturing:
            self._st_plan = self._build_plan(self.q_max)
            self._st_plan3 = None
        if not mask & 8 or self.q_max_order3 == self.q_max:
            return self._st_plan
        if self._st_plan3 is not None and self._st_plan3.recorded_only and not capturing:
            self._st_plan3 = None
        if self._st_plan3 is None:
            self._st_plan3 = self._build_plan(self.q_max_order3, self._st_plan.samples)     # same points: the sorted samples are shared
        return self._st_plan3

    def _compute(self, mask):
        means, values, conics, samples = self._require_inputs()
        outs = _SampleFunction.apply(means, values, conics, samples, mask, self.debug, self._plan_for(mask))
        for k, o in zip(_mask_orders(mask), outs):
            self._cache[k] = o

    def _get(self, order):
        if self._core is not None:
            return self._core.get(order)
        if order not in self._cache:
            means, _, _, samples = self._require_inputs()
            mask = 1 << order
            if order <= 2 and (self.fuse == "all" or (
                    self.fuse == "auto" and samples.shape[0] <= self.FUSE_AUTO_MAX_POINTS)):
                mask = 7 & ~sum(1 << k for k in self._cache)
                mask |= 1 << order
            self._compute(mask)
        return self._cache[order]

    def sample(self, orders=(0, 1, 2)):
        """Fused entry point: one launch for all ``orders`` (extension of the reference API).
        ``orders`` holds derivative orders 0..3 and / or ``"lap"`` -- the trace of the Hessian
        u_xx + u_yy as [M, c], what the PDE residuals consume (model_pn.py:614-617) -- e.g.
        ``sample((0, 1, "lap"))``.  Returns a tuple of outputs in the order given."""
        orders = tuple(TRACE if o == "lap" else int(o) for o in orders)
        if any(o < 0 or o > TRACE for o in orders):
            raise ValueError('orders must be in 0..3 or "lap"')
        if self._core is not None:
            return self._core.sample(list(orders))
        self._require_inputs()
        want = set(o for o in orders if o not in self._cache)
        if TRACE in want and (2 in want or 2 in self._cache):
            want.discard(TRACE)                    # the Hessian is (being) computed: take its diagonal
        if TRACE in want and 3 in want:            # no fused kernel for trace + order 3: two launches
            self._compute(1 << 3)
            want.discard(3)
        mask = sum(1 << o for o in want)
        if mask:
            self._compute(mask)
        if TRACE in orders and TRACE not in self._cache:
            self._cache[TRACE] = self._cache[2].diagonal(dim1=1, dim2=2).sum(-1)
        return tuple(self._cache[o] for o in orders)

    def residual(self, a0=0.0, a1=None, lap=0.0, target=None):
        """Extension of the reference API: the linear residual
        ``r = a0 u + a1 . grad u + lap (u_xx + u_yy) - target`` as [M, c] in ONE launch (4 bytes per point and
        channel instead of the 28 of u, grad u and the Hessian), differentiable wrt means, values, conics (one
        launch) and ``target``.  ``a0``, ``lap``: floats; ``a1``: d floats (default zero); ``target``: [M, c]
        (or [M] for c = 1) or None.  The reference's diffusion loss (model_pn.py:612-617, 834-849;
        test_no_mlp.py:127-144: ``(u - u_prev) / dt - D lap u``) is
        ``sampler.residual(a0=1 / dt, lap=-D, target=u_prev / dt).pow(2).mean()``.  Binned plans evaluate the
        backward with the wide cut-off ``q_max_backward``."""
        means, values, conics, samples = self._require_inputs()
        d, c, M = means.shape[1], values.shape[1], samples.shape[0]
        a1 = (0.0,) * d if a1 is None else tuple(float(x) for x in (a1 if hasattr(a1, "__len__") else (a1,)))
        if len(a1) != d:
            raise ValueError(f"a1 must hold d = {d} coefficients")
        coeffs = (float(a0), a1[0], a1[1] if d == 2 else 0.0, float(lap))
        if target is not None:
            if not isinstance(target, torch.Tensor) or not target.is_cuda:
                raise RuntimeError("target must be a tensor on the GPU (no CPU fallback)")
            if target.numel() != M * c:
                raise ValueError(f"target must hold M*c = {M * c} elements, got {tuple(target.shape)}")
            target = target.reshape(M, c)
        if self._core is not None:
            return self._core.residual(coeffs, target)
        return _ResidualFunction.apply(means, values, conics, samples, target, coeffs, self.debug, self._plan_for(0))

    def sample_gaussians(self):
        """u [M, c]"""
        return self._get(0)

    def sample_gaussians_derivative(self):
        """du/dx [M, d, c]"""
        return self._get(1)

    def sample_gaussians_laplacian(self):
        """full Hessian [M, d, d, c] (the reference's name is a misnomer: model_pn.py:614,652)"""
        return self._get(2)

    def sample_gaussians_laplacian_trace(self):
        """u_xx + u_yy [M, c]: the trace of the Hessian in its own launch (extension; 4 floats per
        point with u and grad u instead of 7, and no slicing of [M, d, d, c] in the residual)"""
        return self.sample(("lap",))[0]

    def sample_gaussians_third_derivative(self):
        """third derivatives [M, d, d, d, c]"""
        return self._get(3)

    # ------------------------------------------------------------------ neighbour aggregation
    def preprocess_aggregate(self):
        """Build the Gaussian <-> Gaussian neighbour structure for :meth:`aggregate_neighbors`
        (model_pn.py:257).  Semantics are this repo's own (parity unpinned): pigs_amd/aggregate.py."""
        from . import aggregate
        if not self.unpinned_aggregate and not GaussianSampler._warned_aggregate:
            GaussianSampler._warned_aggregate = True
            import warnings
            warnings.warn("GaussianSampler.preprocess_aggregate / aggregate_neighbors: the arithmetic of these two "
                          "methods exists only in the reference's absent CUDA source; what runs here is this "
                          "repository's own definition (pigs_amd/aggregate.py, DESIGN.md) -- a model trained with "
                          "the reference will not reproduce through it.  Pass unpinned_aggregate=True to "
                          "GaussianSampler to acknowledge.", stacklevel=2)
        if self._core is not None:
            self._core.preprocess_aggregate(-1 if self.aggregate_cap is None else self.aggregate_cap)
            self._neighbors = self._core.neighbors
            return
        means, _, conics, _ = self._require_inputs()
        if means.shape[1] != 2:
            raise NotImplementedError("aggregate_neighbors is implemented for d = 2")
        self._neighbors = aggregate.NeighborLists(means, conics, self.q_max, cap=self.aggregate_cap)
        if self.debug:
            self._neighbors.check()

    def aggregate_neighbors(self, features, transform, queries, keys, frequencies, distance_transform):
        """[N, L] attention-weighted neighbour messages (model_pn.py:262-264); differentiable wrt all
        six arguments (test_neighbor_aggregation.py:89-98).  Parity unpinned: pigs_amd/aggregate.py."""
        if self._core is not None:
            return self._core.aggregate_neighbors(features, transform, queries, keys, frequencies, distance_transform)
        from . import aggregate
        if self._neighbors is None:
            raise RuntimeError("preprocess_aggregate() must be called before aggregate_neighbors()")
        return aggregate.aggregate(self._neighbors, features, transform, queries, keys, frequencies, distance_transform)

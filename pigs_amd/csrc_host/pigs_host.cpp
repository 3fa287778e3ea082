// Native host side of the sampler: a torch C++ extension over the C ABI of include/pigs_amd.h.
//
// The reference's boundary is a compiled torch extension (`from diff_gaussian_sampling import
// GaussianSampler`, /root/reference/model_pn.py:11, .gitmodules:1-3); SURVEY.md section 8b asks for
// `extern "C"` launchers plus a thin torch shim.  This file is that shim: the state behind
// `GaussianSampler` (bound inputs, the plan, the remembered sample plans, the workspace pool, the
// per-order output cache) and the autograd node of the sample_*() outputs, in C++.  It calls the same
// C ABI as the ctypes host (pigs_amd/_lib.py) and torch-free hosts (tests/abi_example); none of the
// arithmetic lives here.  What it removes is the interpreter: a sampler-only training step at the
// reference's sizes (model_pn.py:766-788: two preprocess calls, orders 0-2, one backward) costs ~100 us of
// Python, ctypes and torch.autograd.Function plumbing per step, several times its GPU time.
//
// Semantics mirrored from pigs_amd/sampler.py (the ctypes host, kept for comparison and for the
// CPU-side tests): argument checks and error types of preprocess(); fuse / backend / q_max_order3 /
// reuse_samples; one autograd node per fused launch that OWNS its inputs and plan and survives a
// backward without retain_graph (test_derivatives.py:214-215, 349-352); in-place modification check;
// no gradient for samples / covariances.
#include <torch/extension.h>
#include <torch/csrc/autograd/function.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/core/DeviceGuard.h>
#include <hip/hip_runtime_api.h>

#include <array>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/pigs_amd.h"

namespace py = pybind11;

namespace {

constexpr int TRACE = 4;      // order index of the Hessian's trace (mask bit 16); pointer slot 2

struct PigsFailure : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void check(int rc, const char* what) {
    if (rc == PIGS_OK) return;
    std::string msg = std::string(what) + ": " + pigs_status_string(rc);
    if (rc == PIGS_ERR_LAUNCH) msg += std::string(": ") + pigs_last_hip_error();
    msg += " (status " + std::to_string(rc) + ")";
    throw PigsFailure(msg);
}

[[noreturn]] void raise_py(PyObject* type, const std::string& msg) {
    PyErr_SetString(type, msg.c_str());
    throw py::error_already_set();
}

std::string shape_str(const at::Tensor& t) {
    std::string s = "(";
    for (int64_t i = 0; i < t.dim(); ++i) s += (i ? ", " : "") + std::to_string(t.size(i));
    return s + (t.dim() == 1 ? ",)" : ")");
}

int dtype_code(const at::Tensor& t) { return t.scalar_type() == at::kDouble ? PIGS_F64 : PIGS_F32; }

void* ptr(const at::Tensor& t) { return t.defined() && t.numel() > 0 ? t.data_ptr() : nullptr; }

hipStream_t current_stream(const at::Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

void device_sync(const at::Tensor& t) {
    c10::DeviceGuard g(t.device());
    if (hipDeviceSynchronize() != hipSuccess) throw PigsFailure(std::string("hipDeviceSynchronize: ") + hipGetErrorString(hipGetLastError()));
}

std::vector<int64_t> out_shape(int order, int64_t M, int64_t d, int64_t c) {
    std::vector<int64_t> s{M};
    if (order != TRACE)
        for (int k = 0; k < order; ++k) s.push_back(d);
    s.push_back(c);
    return s;
}

// ---------------------------------------------------------------------------------------------
// the samples half of a plan (pigs_samples_*): a function of `samples` alone, immutable once built
// ---------------------------------------------------------------------------------------------
struct SamplePlan {
    at::Tensor workspace;
    int64_t M = 0;
    at::Tensor source;        // the caller's tensor (kept alive: its address cannot be handed to another tensor)
    at::Tensor points;        // the contiguous array the build read: in index-tiled order (pigs_amd.h, ABI 7) the workspace
                              // holds no copy of the points, the sampling kernels read THIS array
    uint32_t version = 0;     // its version counter at build time
    bool built = false;

    SamplePlan(const at::Tensor& samples, const at::Tensor& src) : M(samples.size(0)), source(src), points(samples) {
        const size_t nbytes = pigs_samples_workspace_bytes(M);
        if (nbytes == 0) throw PigsFailure("binned path does not support M=" + std::to_string(M));
        workspace = at::empty({(int64_t)nbytes}, samples.options().dtype(at::kByte));
        version = source._version();
    }
    bool matches(const at::Tensor& s) const {
        const bool same = s.unsafeGetTensorImpl() == source.unsafeGetTensorImpl() ||
                          (s.data_ptr() == source.data_ptr() && s.sizes() == source.sizes() &&
                           s.strides() == source.strides() && s.scalar_type() == source.scalar_type() &&
                           s.device() == source.device());
        return same && s._version() == version;
    }
};

// Plan workspaces whose plan has died, kept for the next preprocess of the same sizes on the same
// stream (a build leaves its workspace's counters zeroed: PIGS_BUILD_PLAN_WS_CLEAN saves the zeroing
// launch).  Plans die on whichever thread drops the last reference (the autograd engine's workers
// included): every access is under the mutex.
struct PlanPool {
    using Key = std::tuple<int64_t, int64_t, int, int, void*>;      // N, M, c, device, stream
    static constexpr size_t KEEP = 2, KEEP_TOTAL = 4;
    std::mutex mu;
    std::list<std::pair<Key, at::Tensor>> free;      // least recently given first

    at::Tensor take(const Key& key) {
        std::lock_guard<std::mutex> lock(mu);
        for (auto it = free.end(); it != free.begin();) {
            --it;
            if (it->first == key) {
                at::Tensor ws = std::move(it->second);
                free.erase(it);
                return ws;
            }
        }
        return at::Tensor();
    }
    void give(const Key& key, at::Tensor ws) {
        std::lock_guard<std::mutex> lock(mu);
        size_t same = 0;
        for (auto& e : free) same += e.first == key;
        if (same >= KEEP) return;
        free.emplace_back(key, std::move(ws));
        while (free.size() > KEEP_TOTAL) free.pop_front();
    }
    size_t size() {
        std::lock_guard<std::mutex> lock(mu);
        return free.size();
    }
};

// The Gaussian half (pigs_plan_*), on top of a SamplePlan.  Immutable once built; autograd nodes hold
// a reference, so later preprocess calls never disturb a pending backward.
struct Plan {
    at::Tensor workspace;
    std::shared_ptr<SamplePlan> samples;
    int64_t N = 0, M = 0;
    int c = 1;
    float q_max = 36.f, q_max_backward = 36.f;
    std::shared_ptr<PlanPool> pool;      // null: the workspace is not recycled (hipGraph capture)
    PlanPool::Key key;
    hipStream_t build_stream = nullptr;
    bool other_stream_used = false;      // a launch on another stream read the workspace: stream order no longer covers a reuse
    bool recorded_only = false;          // built inside a hipGraph capture: it has run only if (and when) that graph was replayed

    ~Plan() {
        if (pool && !other_stream_used && workspace.defined()) pool->give(key, std::move(workspace));
    }
    void note_stream(hipStream_t s) {
        if (s == build_stream) return;
        other_stream_used = true;
        // the caching allocator hands a freed block out again in the order of the stream it was allocated on:
        // kernels still queued on THIS stream must be known to it (both workspaces; idempotent per stream)
        c10::hip::HIPStream hs = c10::hip::getCurrentHIPStream(workspace.device().index());      // the caller's: what every call site passes
        if (hs.stream() != s) hs = c10::hip::getStreamFromExternal(s, workspace.device().index());
        c10::hip::HIPCachingAllocator::recordStream(workspace.storage().data_ptr(), hs);
        if (samples && samples->workspace.defined())
            c10::hip::HIPCachingAllocator::recordStream(samples->workspace.storage().data_ptr(), hs);
    }
};

std::shared_ptr<Plan> build_plan(const at::Tensor& means, const at::Tensor& values, const at::Tensor& conics,
                                 const at::Tensor& samples, float q_max, float q_max_backward,
                                 std::shared_ptr<SamplePlan> sp, const at::Tensor& source,
                                 const std::shared_ptr<PlanPool>& pool_or_null, bool defer_lists) {
    auto plan = std::make_shared<Plan>();
    plan->N = means.size(0); plan->M = samples.size(0); plan->c = (int)values.size(1); plan->q_max = q_max;
    plan->q_max_backward = q_max_backward > q_max ? q_max_backward : q_max;
    const size_t nbytes = pigs_plan_workspace_bytes(plan->N, plan->M, plan->c);
    if (nbytes == 0)
        throw PigsFailure("binned path does not support N=" + std::to_string(plan->N) + " M=" + std::to_string(plan->M) +
                          " c=" + std::to_string(plan->c));
    if (!sp) sp = std::make_shared<SamplePlan>(samples, source);
    plan->samples = sp;
    c10::DeviceGuard guard(means.device());
    const hipStream_t stream = current_stream(means);
    plan->build_stream = stream;
    plan->key = PlanPool::Key(plan->N, plan->M, plan->c, (int)means.device().index(), (void*)stream);
    int flags = sp->built ? 0 : PIGS_BUILD_SAMPLES;
    // the tile lists are built by the plan's first sampling call, a forward in the same launch (pigs_amd.h)
    if (defer_lists) flags |= PIGS_BUILD_DEFER_LISTS;
    if (pool_or_null) plan->workspace = pool_or_null->take(plan->key);
    if (plan->workspace.defined()) flags |= PIGS_BUILD_PLAN_WS_CLEAN;
    else plan->workspace = at::empty({(int64_t)nbytes}, means.options().dtype(at::kByte));
    check(pigs_plan_build(plan->workspace.data_ptr(), nbytes, sp->workspace.data_ptr(), (size_t)sp->workspace.numel(),
                          flags, plan->N, plan->M, plan->c, q_max, plan->q_max_backward, ptr(means), ptr(conics), ptr(values), ptr(samples),
                          stream),
          "pigs_plan_build");
    // a build that was only RECORDED into a hipGraph capture has not run: it never marks the samples half built
    // (an eager call that finds `built` set skips PIGS_BUILD_SAMPLES and would sample a workspace that never
    // executed); a SamplePlan that WAS built eagerly stays built whatever is recorded on top of it
    plan->recorded_only = capturing(stream);
    sp->built = sp->built || !plan->recorded_only;
    plan->pool = pool_or_null;      // only a workspace whose build was launched completely goes back
    return plan;
}

bool plan_supported(const at::Tensor& means, const at::Tensor& values, const at::Tensor& samples) {
    return means.scalar_type() == at::kFloat && means.size(1) == 2 && values.size(1) >= 1 && values.size(1) <= 2 &&
           means.size(0) >= 1 && samples.size(0) >= 1;
}

// ---------------------------------------------------------------------------------------------
// raw launches
// ---------------------------------------------------------------------------------------------
using Outs = std::array<at::Tensor, 5>;      // orders 0..3 and the trace

const at::Tensor& slot2(const Outs& t) { return t[2].defined() ? t[2] : t[TRACE]; }

Outs forward_raw(const at::Tensor& means, const at::Tensor& values, const at::Tensor& conics, const at::Tensor& samples,
                 int mask, Plan* plan) {
    const int64_t N = means.size(0), d = means.size(1), c = values.size(1), M = samples.size(0);
    Outs outs;
    for (int k = 0; k < 5; ++k)
        if (mask >> k & 1) outs[k] = at::empty(out_shape(k, M, d, c), means.options());
    if (M > 0) {
        c10::DeviceGuard guard(means.device());
        const hipStream_t stream = current_stream(means);
        if (plan) {
            plan->note_stream(stream);
            const at::Tensor& sws = plan->samples->workspace;
            check(pigs_plan_forward(plan->workspace.data_ptr(), (size_t)plan->workspace.numel(), sws.data_ptr(),
                                    (size_t)sws.numel(), N, M, (int)c, plan->q_max, mask, ptr(outs[0]), ptr(outs[1]),
                                    ptr(slot2(outs)), ptr(outs[3]), stream),
                  "pigs_plan_forward");
        } else {
            check(pigs_sample_forward(dtype_code(means), (int)d, (int)c, mask, N, M, ptr(means), ptr(conics), ptr(values),
                                      ptr(samples), ptr(outs[0]), ptr(outs[1]), ptr(slot2(outs)), ptr(outs[3]), stream),
                  "pigs_sample_forward");
        }
    }
    return outs;
}

// The three parameter gradients as views of ONE flat allocation [means | values | conics]: the multi-GPU path
// (pigs_amd/distributed.py) all-reduces that buffer in place -- no packing copy in front of the collective, no
// slicing behind it.
std::array<at::Tensor, 3> gradient_views(const at::Tensor& means, const at::Tensor& values, const at::Tensor& conics) {
    const int64_t nm = means.numel(), nv = values.numel(), nc = conics.numel();
    at::Tensor flat = at::empty({nm + nv + nc}, means.options());
    return {flat.narrow(0, 0, nm).view(means.sizes()), flat.narrow(0, nm, nv).view(values.sizes()),
            flat.narrow(0, nm + nv, nc).view(conics.sizes())};
}

std::array<at::Tensor, 3> backward_raw(const at::Tensor& means, const at::Tensor& values, const at::Tensor& conics,
                                       const at::Tensor& samples, const Outs& gouts, int mask, Plan* plan) {
    const int64_t N = means.size(0), d = means.size(1), c = values.size(1), M = samples.size(0);
    auto gv3 = gradient_views(means, values, conics);
    at::Tensor g_means = gv3[0], g_values = gv3[1], g_conics = gv3[2];
    if (N > 0) {
        c10::DeviceGuard guard(means.device());
        const hipStream_t stream = current_stream(means);
        if (plan && M > 0) {
            plan->note_stream(stream);
            const at::Tensor& sws = plan->samples->workspace;
            check(pigs_plan_backward(plan->workspace.data_ptr(), (size_t)plan->workspace.numel(), sws.data_ptr(),
                                     (size_t)sws.numel(), N, M, (int)c, plan->q_max, mask, ptr(gouts[0]), ptr(gouts[1]),
                                     ptr(slot2(gouts)), ptr(gouts[3]), ptr(g_means), ptr(g_conics), ptr(g_values), stream),
                  "pigs_plan_backward");
        } else {
            check(pigs_sample_backward(dtype_code(means), (int)d, (int)c, mask, N, M, ptr(means), ptr(conics), ptr(values),
                                       ptr(samples), ptr(gouts[0]), ptr(gouts[1]), ptr(slot2(gouts)), ptr(gouts[3]),
                                       ptr(g_means), ptr(g_conics), ptr(g_values), stream),
                  "pigs_sample_backward");
        }
    }
    return {g_means, g_values, g_conics};
}

// ---------------------------------------------------------------------------------------------
// The autograd node of one fused launch.  Outputs computed by ONE launch share it; the reference's
// scripts differentiate one output after the other, the last call without retain_graph, and come
// back to another output of the same preprocess later (test_derivatives.py:214-215, 349-352): the
// node keeps its inputs as plain members and ignores release_variables(), so it survives a
// non-retaining backward.  The inputs and the plan therefore live as long as any output does.
// ---------------------------------------------------------------------------------------------
struct SampleBackward : public torch::autograd::Node {
    at::Tensor means, values, conics, samples;
    uint32_t versions[4] = {0, 0, 0, 0};
    int mask = 0;
    bool debug = false;
    std::shared_ptr<Plan> plan;
    std::vector<int> orders;      // order index of every output, in output order

    std::string name() const override { return "PigsSampleBackward"; }
    void release_variables() override {}

    torch::autograd::variable_list apply(torch::autograd::variable_list&& grads) override {
        if (means._version() != versions[0] || values._version() != versions[1] || conics._version() != versions[2] ||
            samples._version() != versions[3])
            throw std::runtime_error(
                "one of the tensors handed to GaussianSampler.preprocess() has been modified in place before the "
                "backward of a sample_*() output that was computed from it");
        at::AutoGradMode no_grad(false);
        Outs gouts;
        int gmask = 0;
        for (size_t i = 0; i < orders.size() && i < grads.size(); ++i) {
            if (!grads[i].defined()) continue;
            if (grads[i].requires_grad())
                throw std::runtime_error("GaussianSampler: the sample_*() outputs are differentiable once "
                                         "(double backward through the sampler is not implemented)");
            gouts[orders[i]] = grads[i].contiguous();
            gmask |= 1 << orders[i];
        }
        if (gmask == 0) return {at::Tensor(), at::Tensor(), at::Tensor()};
        auto g = backward_raw(means, values, conics, samples, gouts, gmask, plan.get());
        if (debug) device_sync(means);
        return {g[0], g[1], g[2]};
    }
};

Outs sample_apply(const at::Tensor& means, const at::Tensor& values, const at::Tensor& conics, const at::Tensor& samples,
                  int mask, bool debug, const std::shared_ptr<Plan>& plan) {
    Outs outs = forward_raw(means, values, conics, samples, mask, plan.get());
    if (debug) device_sync(means);
    if (at::GradMode::is_enabled() && (means.requires_grad() || values.requires_grad() || conics.requires_grad())) {
        std::shared_ptr<SampleBackward> node(new SampleBackward(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(means, values, conics));
        node->means = means; node->values = values; node->conics = conics; node->samples = samples;
        node->versions[0] = means._version(); node->versions[1] = values._version();
        node->versions[2] = conics._version(); node->versions[3] = samples._version();
        node->mask = mask; node->debug = debug; node->plan = plan;
        for (int k = 0; k < 5; ++k)
            if (mask >> k & 1) {
                node->orders.push_back(k);
                torch::autograd::create_gradient_edge(outs[k], node);
            }
    }
    return outs;
}

// ---------------------------------------------------------------------------------------------
// residual(): r = a0 u + a1 . grad u + aL lap u - target in one launch (pigs_residual_*); the node owns its
// inputs and plan like SampleBackward.
// ---------------------------------------------------------------------------------------------
struct PlanPtrs {
    void *pw, *sw;
    size_t pb, sb;
};
PlanPtrs plan_ptrs(Plan* plan, hipStream_t stream) {
    if (!plan) return {nullptr, nullptr, 0, 0};
    plan->note_stream(stream);
    return {plan->workspace.data_ptr(), plan->samples->workspace.data_ptr(), (size_t)plan->workspace.numel(),
            (size_t)plan->samples->workspace.numel()};
}

struct ResidualBackward : public torch::autograd::Node {
    at::Tensor means, values, conics, samples;
    uint32_t versions[4] = {0, 0, 0, 0};
    double coeffs[4] = {0, 0, 0, 0};
    bool debug = false, has_target = false;
    at::ScalarType target_dtype = at::kFloat;
    std::shared_ptr<Plan> plan;

    std::string name() const override { return "PigsResidualBackward"; }
    void release_variables() override {}

    torch::autograd::variable_list apply(torch::autograd::variable_list&& grads) override {
        if (means._version() != versions[0] || values._version() != versions[1] || conics._version() != versions[2] ||
            samples._version() != versions[3])
            throw std::runtime_error(
                "one of the tensors handed to GaussianSampler.preprocess() has been modified in place before the "
                "backward of a residual() output that was computed from it");
        torch::autograd::variable_list res(has_target ? 4 : 3);
        if (grads.empty() || !grads[0].defined()) return res;
        at::AutoGradMode no_grad(false);
        if (grads[0].requires_grad()) throw std::runtime_error("GaussianSampler.residual() is differentiable once");
        const at::Tensor gout = grads[0].contiguous();
        const int64_t N = means.size(0), d = means.size(1), c = values.size(1), M = samples.size(0);
        auto gv3 = gradient_views(means, values, conics);
        at::Tensor g_means = gv3[0], g_values = gv3[1], g_conics = gv3[2];
        if (N > 0 && M > 0) {
            c10::DeviceGuard guard(means.device());
            const hipStream_t stream = current_stream(means);
            const PlanPtrs pp = plan_ptrs(plan.get(), stream);
            check(pigs_residual_backward(dtype_code(means), (int)d, (int)c, N, M, ptr(means), ptr(conics), ptr(values), ptr(samples),
                                         coeffs, ptr(gout), ptr(g_means), ptr(g_conics), ptr(g_values), pp.pw, pp.pb, pp.sw, pp.sb,
                                         stream),
                  "pigs_residual_backward");
        } else {
            g_means.zero_(); g_values.zero_(); g_conics.zero_();
        }
        if (debug) device_sync(means);
        res[0] = g_means; res[1] = g_values; res[2] = g_conics;
        if (has_target && task_should_compute_output(3)) res[3] = gout.neg().to(target_dtype);
        return res;
    }
};

at::Tensor residual_apply(const at::Tensor& means, const at::Tensor& values, const at::Tensor& conics, const at::Tensor& samples,
                          const std::array<double, 4>& coeffs, const c10::optional<at::Tensor>& target, bool debug,
                          const std::shared_ptr<Plan>& plan) {
    const int64_t N = means.size(0), d = means.size(1), c = values.size(1), M = samples.size(0);
    at::Tensor tgt;
    if (target.has_value()) {
        at::AutoGradMode no_grad(false);
        tgt = target->detach().to(means.scalar_type()).contiguous();
    }
    at::Tensor out = at::empty({M, c}, means.options());
    if (M > 0) {
        c10::DeviceGuard guard(means.device());
        const hipStream_t stream = current_stream(means);
        const PlanPtrs pp = plan_ptrs(plan.get(), stream);
        check(pigs_residual_forward(dtype_code(means), (int)d, (int)c, N, M, ptr(means), ptr(conics), ptr(values), ptr(samples),
                                    coeffs.data(), ptr(tgt), ptr(out), pp.pw, pp.pb, pp.sw, pp.sb, stream),
              "pigs_residual_forward");
    }
    if (debug) device_sync(means);
    const bool tgrad = target.has_value() && target->requires_grad();
    if (at::GradMode::is_enabled() && (means.requires_grad() || values.requires_grad() || conics.requires_grad() || tgrad)) {
        std::shared_ptr<ResidualBackward> node(new ResidualBackward(), torch::autograd::deleteNode);
        if (target.has_value()) node->set_next_edges(torch::autograd::collect_next_edges(means, values, conics, *target));
        else node->set_next_edges(torch::autograd::collect_next_edges(means, values, conics));
        node->means = means; node->values = values; node->conics = conics; node->samples = samples;
        node->versions[0] = means._version(); node->versions[1] = values._version();
        node->versions[2] = conics._version(); node->versions[3] = samples._version();
        for (int k = 0; k < 4; ++k) node->coeffs[k] = coeffs[k];
        node->debug = debug; node->plan = plan;
        node->has_target = target.has_value();
        if (target.has_value()) node->target_dtype = target->scalar_type();
        torch::autograd::create_gradient_edge(out, node);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// preprocess_aggregate / aggregate_neighbors (model_pn.py:257-264; parity unpinned: this repository's own
// definition, pigs_amd/csrc/aggregate.hip).  Same structure as pigs_amd/aggregate.py.
// ---------------------------------------------------------------------------------------------
constexpr int64_t AGG_BRUTE_MAX = 2048;      // aggregate.hip: up to here every pair is tested and cap = N

struct NeighborLists {
    at::Tensor means, conics, workspace, row_counts, col_counts, row_lists, col_lists, overflow;
    int64_t N = 0, cap = 1;

    NeighborLists(const at::Tensor& means_in, const at::Tensor& conics_in, double q_max, int64_t cap_in) {
        if (means_in.dim() != 2 || means_in.size(1) != 2) raise_py(PyExc_NotImplementedError, "aggregate_neighbors is implemented for d = 2");
        at::AutoGradMode no_grad(false);
        means = means_in.detach().contiguous();
        N = means.size(0);
        conics = conics_in.detach().reshape({N, 3}).contiguous();
        const int dt = dtype_code(means);
        const size_t nbytes = pigs_aggregate_workspace_bytes(dt, N);
        if (nbytes == 0) throw PigsFailure("aggregate_neighbors does not support N=" + std::to_string(N));
        const auto iopt = means.options().dtype(at::kInt);
        workspace = at::empty({(int64_t)nbytes}, means.options().dtype(at::kByte));
        row_counts = at::empty({N}, iopt);
        col_counts = at::empty({N}, iopt);
        overflow = at::zeros({1}, iopt);
        c10::DeviceGuard guard(means.device());
        const hipStream_t stream = current_stream(means);
        auto run = [&](int flags, int64_t cap_, bool with_lists) {
            check(pigs_aggregate_lists(dt, N, cap_, ptr(means), ptr(conics), q_max, workspace.data_ptr(), nbytes, flags,
                                       (int32_t*)ptr(row_counts), with_lists ? (int32_t*)ptr(row_lists) : nullptr,
                                       (int32_t*)ptr(col_counts), with_lists ? (int32_t*)ptr(col_lists) : nullptr,
                                       (int32_t*)ptr(overflow), stream),
                  "pigs_aggregate_lists");
        };
        int flags = PIGS_AGGREGATE_BUILD_GRID;
        int64_t c = cap_in;
        if (c <= 0 && N <= AGG_BRUTE_MAX) c = N > 0 ? N : 1;
        if (c <= 0 && capturing(stream)) c = std::min<int64_t>(N, means.scalar_type() == at::kDouble ? 4096 : 8192);
        if (c <= 0) {        // counting pass; the longest list is read back once (the one synchronisation)
            run(flags, 1, false);
            flags = 0;
            const int64_t longest = at::maximum(row_counts.max(), col_counts.max()).item<int64_t>();
            c = std::max<int64_t>(64, (longest + 63) / 64 * 64);
        }
        cap = std::max<int64_t>(1, c);
        row_lists = at::empty({N, cap}, iopt);
        col_lists = at::empty({N, cap}, iopt);
        if (N > 0) run(flags, cap, true);
    }
    void check_overflow() const {
        if (overflow.item<int32_t>() != 0)
            throw PigsFailure("aggregate: a Gaussian has more than " + std::to_string(cap) + " neighbours (list truncated)");
    }
};

struct AggregateBackward : public torch::autograd::Node {
    std::shared_ptr<NeighborLists> nb;
    at::Tensor f, tr, q, k, fr, dist, lse, acc;     // converted, contiguous, detached
    uint32_t versions[6] = {0, 0, 0, 0, 0, 0};
    at::ScalarType in_dtypes[6];
    int64_t N = 0;
    int L = 0, K = 0, F = 0;

    std::string name() const override { return "PigsAggregateBackward"; }

    torch::autograd::variable_list apply(torch::autograd::variable_list&& grads) override {
        if (!f.defined()) throw std::runtime_error("aggregate_neighbors: backward through the graph a second time (saved tensors were freed)");
        const at::Tensor* saved[6] = {&f, &tr, &q, &k, &fr, &dist};
        for (int x = 0; x < 6; ++x)
            if (saved[x]->_version() != versions[x])
                throw std::runtime_error("one of the tensors handed to aggregate_neighbors() has been modified in place before its backward");
        if (grads.empty() || !grads[0].defined()) return torch::autograd::variable_list(6);
        at::AutoGradMode no_grad(false);
        if (grads[0].requires_grad()) throw std::runtime_error("aggregate_neighbors is differentiable once");
        const at::Tensor gout = grads[0].to(f.scalar_type()).contiguous();
        at::Tensor g_f = at::empty_like(f), g_tr = at::empty_like(tr), g_q = at::empty_like(q), g_k = at::empty_like(k),
                   g_fr = at::empty_like(fr), g_dist = at::empty_like(dist);
        const int dt = dtype_code(f);
        if (N > 0) {
            const size_t nbytes = pigs_aggregate_backward_scratch_bytes(dt, N, L, F);
            at::Tensor scratch = at::empty({(int64_t)nbytes}, f.options().dtype(at::kByte));
            c10::DeviceGuard guard(f.device());
            check(pigs_aggregate_backward(dt, N, nb->cap, L, K, F, ptr(nb->means), ptr(nb->conics), (const int32_t*)ptr(nb->row_counts),
                                          (const int32_t*)ptr(nb->row_lists), (const int32_t*)ptr(nb->col_counts),
                                          (const int32_t*)ptr(nb->col_lists), ptr(f), ptr(tr), ptr(q), ptr(k), ptr(fr), ptr(dist),
                                          ptr(lse), ptr(acc), ptr(gout), scratch.data_ptr(), nbytes, ptr(g_f), ptr(g_tr), ptr(g_q),
                                          ptr(g_k), ptr(g_fr), ptr(g_dist), current_stream(f)),
                  "pigs_aggregate_backward");
        } else {
            g_tr.zero_(); g_fr.zero_(); g_dist.zero_();
        }
        return {g_f.to(in_dtypes[0]), g_tr.to(in_dtypes[1]), g_q.to(in_dtypes[2]), g_k.to(in_dtypes[3]), g_fr.to(in_dtypes[4]),
                g_dist.to(in_dtypes[5])};
    }
    void release_variables() override {
        f = tr = q = k = fr = dist = lse = acc = at::Tensor();
    }
};

at::Tensor aggregate_apply(const std::shared_ptr<NeighborLists>& nb, const at::Tensor& features, const at::Tensor& transform,
                           const at::Tensor& queries, const at::Tensor& keys, const at::Tensor& frequencies,
                           const at::Tensor& distance_transform) {
    const int64_t N = nb->N;
    if (features.dim() != 2 || features.size(0) != N)
        raise_py(PyExc_ValueError, "features must be [N=" + std::to_string(N) + ", L], got " + shape_str(features));
    if (queries.dim() != 2 || keys.dim() != 2 || frequencies.dim() != 1 || transform.dim() != 2 || distance_transform.dim() != 2)
        raise_py(PyExc_ValueError, "aggregate_neighbors: transform, queries, keys, distance_transform must be 2-d, frequencies 1-d");
    const int64_t L = features.size(1), K = queries.size(1), F = frequencies.size(0), E = 4 * F + 1;
    if (transform.size(0) != L || transform.size(1) != L || queries.size(0) != N || keys.size(0) != N || keys.size(1) != K ||
        distance_transform.size(0) != L || distance_transform.size(1) != 2 * E)
        raise_py(PyExc_ValueError, "aggregate_neighbors: expected transform [" + std::to_string(L) + "," + std::to_string(L) +
                                       "], queries/keys [" + std::to_string(N) + "," + std::to_string(K) + "], distance_transform [" +
                                       std::to_string(L) + "," + std::to_string(2 * E) + "] (E = 2*d*F + 1 = " + std::to_string(E) + ")");
    const at::Tensor* ins[6] = {&features, &transform, &queries, &keys, &frequencies, &distance_transform};
    const char* names[6] = {"features", "transform", "queries", "keys", "frequencies", "distance_transform"};
    for (int x = 0; x < 6; ++x)
        if (!ins[x]->is_cuda())
            raise_py(PyExc_RuntimeError, std::string(names[x]) + " is on " + ins[x]->device().str() +
                                             ": aggregate_neighbors runs on the GPU only (no CPU fallback)");
    if (L + 2 * E > 128) raise_py(PyExc_NotImplementedError, "L + 2E = " + std::to_string(L + 2 * E) + " > 128 is not supported");
    const auto dt = nb->means.scalar_type();
    at::Tensor c[6];
    {
        at::AutoGradMode no_grad(false);
        for (int x = 0; x < 6; ++x) c[x] = ins[x]->detach().to(dt).contiguous();
    }
    const auto opt = c[0].options();
    at::Tensor out = at::empty({N, L}, opt), lse = at::empty({N}, opt), acc = at::empty({N, L + 2 * E}, opt);
    if (N > 0) {
        c10::DeviceGuard guard(c[0].device());
        check(pigs_aggregate_forward(dtype_code(c[0]), N, nb->cap, (int)L, (int)K, (int)F, ptr(nb->means), ptr(nb->conics),
                                     (const int32_t*)ptr(nb->row_counts), (const int32_t*)ptr(nb->row_lists), ptr(c[0]), ptr(c[1]),
                                     ptr(c[2]), ptr(c[3]), ptr(c[4]), ptr(c[5]), ptr(out), ptr(lse), ptr(acc), current_stream(c[0])),
              "pigs_aggregate_forward");
    }
    if (out.scalar_type() != features.scalar_type()) out = out.to(features.scalar_type());
    bool need = false;
    for (int x = 0; x < 6; ++x) need = need || ins[x]->requires_grad();
    if (at::GradMode::is_enabled() && need) {
        std::shared_ptr<AggregateBackward> node(new AggregateBackward(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(features, transform, queries, keys, frequencies, distance_transform));
        node->nb = nb;
        node->f = c[0]; node->tr = c[1]; node->q = c[2]; node->k = c[3]; node->fr = c[4]; node->dist = c[5];
        node->lse = lse; node->acc = acc;
        for (int x = 0; x < 6; ++x) { node->versions[x] = c[x]._version(); node->in_dtypes[x] = ins[x]->scalar_type(); }
        node->N = N; node->L = (int)L; node->K = (int)K; node->F = (int)F;
        torch::autograd::create_gradient_edge(out, node);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// the state behind GaussianSampler
// ---------------------------------------------------------------------------------------------
enum { FUSE_AUTO = 0, FUSE_ALL = 1, FUSE_NONE = 2 };
enum { BACKEND_AUTO = 0, BACKEND_DENSE = 1, BACKEND_BINNED = 2 };

struct Core {
    static constexpr int64_t FUSE_AUTO_MAX_POINTS = 1 << 16;
    static constexpr int64_t BINNED_AUTO_MIN_PAIRS = 1 << 26;

    bool debug;
    int fuse, backend;
    float q_max, q_max3, q_max_b;
    int reuse;
    bool defer_lists = false;         // PIGS_BUILD_DEFER_LISTS (measured: the one-launch lists + forward is not faster; an option)
    bool static_samples = false;      // a capture may reuse a remembered (eagerly built) SamplePlan: the caller promises
                                      // not to modify the samples tensor between replays (GraphedStep(static_samples=True))
    bool bound = false;
    at::Tensor means, values, conics, samples, samples_source;
    std::shared_ptr<Plan> plan, plan3;
    std::vector<std::shared_ptr<SamplePlan>> sample_plans;      // most recently used first
    std::shared_ptr<PlanPool> pool = std::make_shared<PlanPool>();
    Outs cache;
    std::shared_ptr<NeighborLists> neighbors;
    static bool warned_samples_grad;

    Core(bool debug_, int fuse_, int backend_, double q_max_, double q_max3_, double q_max_b_, int reuse_)
        : debug(debug_), fuse(fuse_), backend(backend_), q_max((float)q_max_), q_max3((float)q_max3_),
          q_max_b((float)q_max_b_), reuse(reuse_) {
        if (pigs_abi_version() != PIGS_ABI_VERSION)
            throw PigsFailure("libpigs_amd.so: ABI version " + std::to_string(pigs_abi_version()) + " != " +
                              std::to_string(PIGS_ABI_VERSION) + "; rebuild");
    }

    void preprocess(const at::Tensor& means_in, at::Tensor values_in, const py::object& covariances, at::Tensor conics_in,
                    at::Tensor samples_in) {
        if (means_in.dim() != 2) raise_py(PyExc_ValueError, "means must be [N, d], got " + shape_str(means_in));
        const int64_t N = means_in.size(0), d = means_in.size(1);
        if (d != 1 && d != 2) raise_py(PyExc_NotImplementedError, "d = " + std::to_string(d) + " is not supported (d in {1, 2})");
        const int64_t nf = d * (d + 1) / 2;
        const std::pair<const char*, const at::Tensor*> named[4] = {
            {"means", &means_in}, {"values", &values_in}, {"conics", &conics_in}, {"samples", &samples_in}};
        for (auto& nt : named) {
            const at::Tensor& t = *nt.second;
            if (!t.is_cuda())
                raise_py(PyExc_RuntimeError, std::string(nt.first) + " is on " + t.device().str() +
                                                 ": GaussianSampler runs on the GPU only (no CPU fallback)");
            if (t.device() != means_in.device())
                raise_py(PyExc_RuntimeError, std::string(nt.first) + " is on " + t.device().str() + ", means on " + means_in.device().str());
            if (t.scalar_type() != means_in.scalar_type())
                raise_py(PyExc_TypeError, std::string(nt.first) + " has dtype " + c10::toString(t.scalar_type()) +
                                              ", means " + c10::toString(means_in.scalar_type()));
        }
        if (means_in.scalar_type() != at::kFloat && means_in.scalar_type() != at::kDouble)
            raise_py(PyExc_TypeError, std::string("dtype ") + c10::toString(means_in.scalar_type()) + " is not supported (float32 / float64)");
        if (values_in.dim() == 1) values_in = values_in.reshape({N, 1});
        if (values_in.dim() != 2 || values_in.size(0) != N) raise_py(PyExc_ValueError, "values must be [N, c], got " + shape_str(values_in));
        const int64_t c = values_in.size(1);
        if (c < 1 || c > 4) raise_py(PyExc_NotImplementedError, "c = " + std::to_string(c) + " channels is not supported (1..4)");
        if (conics_in.numel() != N * nf)
            raise_py(PyExc_ValueError, "conics must hold N*" + std::to_string(nf) + " elements (flat upper triangle), got " + shape_str(conics_in));
        conics_in = conics_in.reshape({N, nf});
        if (!covariances.is_none() && THPVariable_Check(covariances.ptr())) {
            const at::Tensor& cov = THPVariable_Unpack(covariances.ptr());
            if (cov.numel() != N * nf)
                raise_py(PyExc_ValueError, "covariances must hold N*" + std::to_string(nf) + " elements, got " + shape_str(cov));
        }
        if (samples_in.dim() == 1 && d == 1) samples_in = samples_in.reshape({-1, 1});
        if (samples_in.dim() != 2 || samples_in.size(1) != d)
            raise_py(PyExc_ValueError, "samples must be [M, " + std::to_string(d) + "], got " + shape_str(samples_in));
        // no gradient flows to the sample points (the reference requests none from the sampler:
        // test_derivatives.py:123 asks for (means, values, conics) only)
        if (samples_in.requires_grad() && at::GradMode::is_enabled() && !warned_samples_grad) {
            warned_samples_grad = true;
            if (PyErr_WarnEx(PyExc_UserWarning,
                             "GaussianSampler: samples.requires_grad is set, but the sampler returns no gradient with respect to "
                             "the sample points (as the reference, whose tests ask for the gradients of means, values and "
                             "conics only); use the derivative outputs instead", 2) < 0)
                throw py::error_already_set();
        }
        means = means_in.contiguous();
        values = values_in.contiguous();
        conics = conics_in.contiguous();
        samples = samples_in.detach().contiguous();
        samples_source = samples_in;
        bound = true;
        for (auto& t : cache) t = at::Tensor();
        plan.reset();
        plan3.reset();
        neighbors.reset();
        const bool use_plan = backend == BACKEND_BINNED || (backend == BACKEND_AUTO && N * samples.size(0) >= BINNED_AUTO_MIN_PAIRS);
        if (use_plan && plan_supported(means, values, samples))
            plan = make_plan(q_max, nullptr);
        else if (backend == BACKEND_BINNED && N > 0 && samples.size(0) > 0)
            raise_py(PyExc_NotImplementedError, "backend='binned' needs float32, d = 2, c <= 2");
    }

    // A plan for the bound inputs; the samples half is reused when preprocess was handed an unmodified
    // samples tensor it remembers.  While a hipGraph is being captured nothing is looked up and nothing
    // is remembered: the capture must record the samples build itself (a replay after an in-place
    // update of the static samples input has to re-sort them), its workspaces belong to the graph (they
    // neither come from the pool nor go back to it), and a SamplePlan that was only RECORDED has not
    // been built as far as later eager calls are concerned.
    std::shared_ptr<Plan> make_plan(float q, std::shared_ptr<SamplePlan> sp) {
        at::AutoGradMode no_grad(false);
        const bool cap = capturing(current_stream(means));
        if (!sp && (!cap || static_samples) && reuse > 0)
            for (auto& p : sample_plans)
                if (p->built && p->matches(samples_source)) { sp = p; break; }
        auto pl = build_plan(means.detach(), values.detach(), conics.detach(), samples, q, q_max_b > q ? q_max_b : q, sp,
                             samples_source, cap ? nullptr : pool, defer_lists);
        if (reuse > 0 && !cap) {
            std::vector<std::shared_ptr<SamplePlan>> next{pl->samples};
            for (auto& p : sample_plans)
                if (p != pl->samples && (int)next.size() < reuse) next.push_back(p);
            sample_plans.swap(next);
        }
        if (debug && !cap) device_sync(means);
        return pl;
    }

    void require_inputs() const {
        if (!bound) raise_py(PyExc_RuntimeError, "preprocess() must be called before sampling");
    }

    std::shared_ptr<Plan> plan_for(int mask) {
        if (!plan) return plan;
        const bool cap = capturing(current_stream(means));
        // An eager call behind a capture (no preprocess in between) must not sample what the capture only RECORDED:
        // the plan is rebuilt eagerly, on a samples half of its own (the recorded one belongs to the graph, whose
        // replays re-sort it).
        if (plan->recorded_only && !cap) {
            plan = make_plan(q_max, nullptr);
            plan3.reset();
        }
        if (!(mask & 8) || q_max3 == q_max) return plan;
        if (plan3 && plan3->recorded_only && !cap) plan3.reset();
        if (!plan3) plan3 = make_plan(q_max3, plan->samples);      // same points: the sorted samples are shared
        return plan3;
    }

    void compute(int mask) {
        require_inputs();
        Outs outs = sample_apply(means, values, conics, samples, mask, debug, plan_for(mask));
        for (int k = 0; k < 5; ++k)
            if (mask >> k & 1) cache[k] = outs[k];
    }

    at::Tensor get(int order) {
        require_inputs();
        if (!cache[order].defined()) {
            int mask = 1 << order;
            if (order <= 2 && (fuse == FUSE_ALL || (fuse == FUSE_AUTO && samples.size(0) <= FUSE_AUTO_MAX_POINTS))) {
                int have = 0;
                for (int k = 0; k < 5; ++k)
                    if (cache[k].defined()) have |= 1 << k;
                mask = (7 & ~have) | (1 << order);
            }
            compute(mask);
        }
        return cache[order];
    }

    // orders: 0..3 and TRACE (the Python wrapper maps "lap")
    py::tuple sample(const std::vector<int>& orders) {
        require_inputs();
        int want = 0;
        for (int o : orders) {
            if (o < 0 || o > TRACE) raise_py(PyExc_ValueError, "orders must be in 0..3 or \"lap\"");
            if (!cache[o].defined()) want |= 1 << o;
        }
        if ((want & 16) && ((want & 4) || cache[2].defined())) want &= ~16;      // the Hessian is (being) computed: take its diagonal
        if ((want & 16) && (want & 8)) {                                         // no fused kernel for trace + order 3
            compute(8);
            want &= ~8;
        }
        if (want) compute(want);
        py::tuple res(orders.size());
        for (size_t i = 0; i < orders.size(); ++i) {
            if (orders[i] == TRACE && !cache[TRACE].defined()) cache[TRACE] = cache[2].diagonal(0, 1, 2).sum(-1);
            res[i] = cache[orders[i]];
        }
        return res;
    }

    at::Tensor residual(const std::array<double, 4>& coeffs, const c10::optional<at::Tensor>& target) {
        require_inputs();
        return residual_apply(means, values, conics, samples, coeffs, target, debug, plan_for(0));
    }

    void preprocess_aggregate(int64_t cap) {
        require_inputs();
        if (means.size(1) != 2) raise_py(PyExc_NotImplementedError, "aggregate_neighbors is implemented for d = 2");
        neighbors = std::make_shared<NeighborLists>(means, conics, (double)q_max, cap);
        if (debug) neighbors->check_overflow();
    }

    at::Tensor aggregate_neighbors(const at::Tensor& features, const at::Tensor& transform, const at::Tensor& queries,
                                   const at::Tensor& keys, const at::Tensor& frequencies, const at::Tensor& distance_transform) {
        if (!neighbors) raise_py(PyExc_RuntimeError, "preprocess_aggregate() must be called before aggregate_neighbors()");
        return aggregate_apply(neighbors, features, transform, queries, keys, frequencies, distance_transform);
    }

    py::object inputs() const {
        if (!bound) return py::none();
        return py::make_tuple(means, values, conics, samples);
    }
};
bool Core::warned_samples_grad = false;

// raw launches for tools and the bench (kernel-only timing on a built plan)
py::list py_forward_raw(const at::Tensor& means, const at::Tensor& values, const at::Tensor& conics, const at::Tensor& samples,
                        int mask, std::shared_ptr<Plan> plan) {
    Outs o = forward_raw(means, values, conics, samples, mask, plan.get());
    py::list l;
    for (auto& t : o) l.append(t.defined() ? py::cast(t) : py::none());
    return l;
}

py::tuple py_backward_raw(const at::Tensor& means, const at::Tensor& values, const at::Tensor& conics, const at::Tensor& samples,
                          const std::vector<c10::optional<at::Tensor>>& gouts, int mask, std::shared_ptr<Plan> plan) {
    Outs g;
    for (size_t k = 0; k < 5 && k < gouts.size(); ++k)
        if (gouts[k].has_value()) g[k] = *gouts[k];
    auto r = backward_raw(means, values, conics, samples, g, mask, plan.get());
    return py::make_tuple(r[0], r[1], r[2]);
}

uint32_t error_flag(const at::Tensor& workspace, size_t offset) {
    return (uint32_t)workspace.slice(0, (int64_t)offset, (int64_t)offset + 4).view(at::kInt).item<int32_t>();
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "native host side of pigs_amd.GaussianSampler (C++ over the C ABI of include/pigs_amd.h)";
    // failures of the C ABI surface as pigs_amd._lib.PigsError (a RuntimeError), like on the ctypes host
    static py::object pigs_error = py::module_::import("pigs_amd._lib").attr("PigsError");
    py::register_exception_translator([](std::exception_ptr p) {
        try {
            if (p) std::rethrow_exception(p);
        } catch (const PigsFailure& e) {
            PyErr_SetString(pigs_error.ptr(), e.what());
        }
    });
    m.attr("ABI_VERSION") = PIGS_ABI_VERSION;
    m.attr("TRACE") = TRACE;

    py::class_<SamplePlan, std::shared_ptr<SamplePlan>>(m, "SamplePlan")
        .def_readonly("workspace", &SamplePlan::workspace)
        .def_readonly("M", &SamplePlan::M)
        .def_readonly("source", &SamplePlan::source)
        .def_readonly("version", &SamplePlan::version)
        .def_readonly("built", &SamplePlan::built)
        .def("matches", &SamplePlan::matches)
        .def("scan_took_slow_path", [](const SamplePlan& s) {      // diagnostic (synchronises): include/pigs_amd.h
            return error_flag(s.workspace, pigs_samples_error_offset()) != 0;
        });
    py::class_<Plan, std::shared_ptr<Plan>>(m, "Plan")
        .def_readonly("workspace", &Plan::workspace)
        .def_readonly("samples", &Plan::samples)
        .def_readonly("N", &Plan::N)
        .def_readonly("M", &Plan::M)
        .def_readonly("c", &Plan::c)
        .def_readonly("q_max", &Plan::q_max)
        .def_readonly("q_max_backward", &Plan::q_max_backward)
        .def_readonly("other_stream_used", &Plan::other_stream_used)
        .def_readonly("recorded_only", &Plan::recorded_only)
        .def("scan_took_slow_path", [](const Plan& p) {
            return error_flag(p.samples->workspace, pigs_samples_error_offset()) != 0 ||
                   error_flag(p.workspace, pigs_plan_error_offset()) != 0;
        });
    py::class_<NeighborLists, std::shared_ptr<NeighborLists>>(m, "NeighborLists")
        .def_readonly("N", &NeighborLists::N)
        .def_readonly("cap", &NeighborLists::cap)
        .def_readonly("row_counts", &NeighborLists::row_counts)
        .def_readonly("col_counts", &NeighborLists::col_counts)
        .def_readonly("row_lists", &NeighborLists::row_lists)
        .def_readonly("col_lists", &NeighborLists::col_lists)
        .def_readonly("overflow", &NeighborLists::overflow)
        .def("check", &NeighborLists::check_overflow);
    py::class_<Core>(m, "SamplerCore")
        .def(py::init<bool, int, int, double, double, double, int>(), py::arg("debug"), py::arg("fuse"), py::arg("backend"),
             py::arg("q_max"), py::arg("q_max_order3"), py::arg("q_max_backward"), py::arg("reuse_samples"))
        .def("preprocess", &Core::preprocess)
        .def("get", &Core::get)
        .def("sample", &Core::sample)
        .def("inputs", &Core::inputs)
        .def("residual", &Core::residual, py::arg("coeffs"), py::arg("target") = c10::optional<at::Tensor>())
        .def("preprocess_aggregate", &Core::preprocess_aggregate, py::arg("cap") = -1)
        .def("aggregate_neighbors", &Core::aggregate_neighbors)
        .def_readonly("neighbors", &Core::neighbors)
        .def_readonly("plan", &Core::plan)
        .def_readonly("plan3", &Core::plan3)
        .def_readwrite("defer_lists", &Core::defer_lists)
        .def_readwrite("static_samples", &Core::static_samples)
        .def_property_readonly("sample_plans", [](const Core& c) { return c.sample_plans; })
        .def_property_readonly("pool_size", [](const Core& c) { return c.pool->size(); })
        .def("cached_orders", [](const Core& c) {
            std::vector<int> v;
            for (int k = 0; k < 5; ++k)
                if (c.cache[k].defined()) v.push_back(k);
            return v;
        });
    m.def("forward_raw", &py_forward_raw, py::arg("means"), py::arg("values"), py::arg("conics"), py::arg("samples"),
          py::arg("mask"), py::arg("plan") = std::shared_ptr<Plan>());
    m.def("backward_raw", &py_backward_raw, py::arg("means"), py::arg("values"), py::arg("conics"), py::arg("samples"),
          py::arg("gouts"), py::arg("mask"), py::arg("plan") = std::shared_ptr<Plan>());
}

// extern "C" entry points declared in include/pigs_amd.h: argument checks + dispatch.
#include "launch.h"

using namespace pigs;

namespace pigs {
thread_local hipError_t g_last_hip_error = hipSuccess;
}

static int check_common(int dtype, int d, int c, int mask, int64_t N, int64_t M, const void* means,
                        const void* conics, const void* values, const void* samples) {
    if (dtype != PIGS_F32 && dtype != PIGS_F64) return PIGS_ERR_UNSUPPORTED;
    if (d < 1 || c < 1 || N < 0 || M < 0 || !mask_valid(mask)) return PIGS_ERR_INVALID;
    if (d > 2 || c > 4) return PIGS_ERR_UNSUPPORTED;
    if (N > 0 && (!means || !conics || !values)) return PIGS_ERR_INVALID;
    if (M > 0 && !samples) return PIGS_ERR_INVALID;
    return PIGS_OK;
}

extern "C" {

int pigs_abi_version(void) { return PIGS_ABI_VERSION; }

const char* pigs_last_hip_error(void) { return hipGetErrorString(g_last_hip_error); }

const char* pigs_status_string(int status) {
    switch (status) {
        case PIGS_OK: return "ok";
        case PIGS_ERR_INVALID: return "invalid argument";
        case PIGS_ERR_UNSUPPORTED: return "unsupported d/c/dtype/orders combination";
        case PIGS_ERR_LAUNCH: return "HIP launch error";
        case PIGS_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

int pigs_sample_forward(int dtype, int d, int c, int orders_mask, int64_t N, int64_t M, const void* means,
                        const void* conics, const void* values, const void* samples, void* out0, void* out1,
                        void* out2, void* out3, void* stream) {
    int rc = check_common(dtype, d, c, orders_mask, N, M, means, conics, values, samples);
    if (rc != PIGS_OK) return rc;
    void* outs[4] = {out0, out1, out2, out3};
    for (int k = 0; k < 4; ++k)
        if (mask_uses_slot(orders_mask, k) && M > 0 && !outs[k]) return PIGS_ERR_INVALID;
    SampleArgs a{};
    a.dtype = dtype; a.d = d; a.c = c; a.orders_mask = orders_mask; a.N = N; a.M = M;
    a.means = means; a.conics = conics; a.values = values; a.samples = samples;
    for (int k = 0; k < 4; ++k) a.out[k] = mask_uses_slot(orders_mask, k) ? outs[k] : nullptr;
    return dense_dispatch(false, a, (hipStream_t)stream);
}

int pigs_sample_backward(int dtype, int d, int c, int orders_mask, int64_t N, int64_t M, const void* means,
                         const void* conics, const void* values, const void* samples, const void* gout0,
                         const void* gout1, const void* gout2, const void* gout3, void* g_means, void* g_conics,
                         void* g_values, void* stream) {
    int rc = check_common(dtype, d, c, orders_mask, N, M, means, conics, values, samples);
    if (rc != PIGS_OK) return rc;
    const void* gs[4] = {gout0, gout1, gout2, gout3};
    for (int k = 0; k < 4; ++k)
        if (mask_uses_slot(orders_mask, k) && M > 0 && !gs[k]) return PIGS_ERR_INVALID;
    if (N > 0 && (!g_means || !g_conics || !g_values)) return PIGS_ERR_INVALID;
    SampleArgs a{};
    a.dtype = dtype; a.d = d; a.c = c; a.orders_mask = orders_mask; a.N = N; a.M = M;
    a.means = means; a.conics = conics; a.values = values; a.samples = samples;
    for (int k = 0; k < 4; ++k) a.gout[k] = mask_uses_slot(orders_mask, k) ? gs[k] : nullptr;
    a.g_means = g_means; a.g_conics = g_conics; a.g_values = g_values;
    return dense_dispatch(true, a, (hipStream_t)stream);
}

int pigs_build_covariances(int dtype, int64_t N, const void* scaling, const void* transform, void* covariances,
                           void* conics, void* stream) {
    if (N < 0) return PIGS_ERR_INVALID;
    if (N > 0 && (!scaling || !transform || (!covariances && !conics))) return PIGS_ERR_INVALID;
    return covariances_dispatch(false, dtype, N, scaling, transform, nullptr, nullptr, covariances, conics,
                                (hipStream_t)stream);
}

int pigs_build_covariances_backward(int dtype, int64_t N, const void* scaling, const void* transform,
                                    const void* g_covariances, const void* g_conics, void* g_scaling,
                                    void* g_transform, void* stream) {
    if (N < 0) return PIGS_ERR_INVALID;
    if (N > 0 && (!scaling || !transform || !g_scaling || !g_transform)) return PIGS_ERR_INVALID;
    return covariances_dispatch(true, dtype, N, scaling, transform, g_covariances, g_conics, g_scaling, g_transform,
                                (hipStream_t)stream);
}

size_t pigs_samples_workspace_bytes(int64_t M) { return samples_workspace_bytes(M); }
size_t pigs_plan_workspace_bytes(int64_t N, int64_t M, int c) { return plan_workspace_bytes(N, M, c); }
size_t pigs_samples_error_offset(void) { return samples_error_offset(); }
size_t pigs_plan_error_offset(void) { return plan_error_offset(); }
size_t pigs_samples_lattice_offset(void) { return samples_lattice_offset(); }
size_t pigs_plan_strips_offset(void) { return plan_strips_offset(); }

int pigs_plan_layout_info(int64_t N, int64_t M, int c, int64_t info[6]) {
    if (!info) return PIGS_ERR_INVALID;
    return plan_layout_info(N, M, c, info);
}

int pigs_samples_build(void* samples_ws, size_t samples_ws_bytes, int64_t M, const void* samples, void* stream) {
    if (M < 0 || !samples) return PIGS_ERR_INVALID;
    return samples_build(samples_ws, samples_ws_bytes, M, samples, (hipStream_t)stream);
}

int pigs_samples_order_hint(int64_t M) { return samples_order_hint(M); }

int pigs_plan_build(void* workspace, size_t workspace_bytes, void* samples_ws, size_t samples_ws_bytes,
                    int flags, int64_t N, int64_t M, int c, float q_max, float q_max_backward, const void* means,
                    const void* conics, const void* values, const void* samples, void* stream) {
    if (N < 0 || M < 0 || c < 1) return PIGS_ERR_INVALID;
    if (!means || !conics || !values || ((flags & PIGS_BUILD_SAMPLES) && !samples)) return PIGS_ERR_INVALID;
    return plan_build(workspace, workspace_bytes, samples_ws, samples_ws_bytes, flags, N, M, c, q_max, q_max_backward, means,
                      conics, values, samples, (hipStream_t)stream);
}

int pigs_plan_forward(void* workspace, size_t workspace_bytes, const void* samples_ws, size_t samples_ws_bytes,
                      int64_t N, int64_t M, int c, float q_max, int orders_mask, void* out0, void* out1, void* out2,
                      void* out3, void* stream) {
    if (!mask_valid(orders_mask)) return PIGS_ERR_INVALID;
    void* outs[4] = {out0, out1, out2, out3};
    for (int k = 0; k < 4; ++k)
        if (mask_uses_slot(orders_mask, k) && !outs[k]) return PIGS_ERR_INVALID;
    return plan_forward(workspace, workspace_bytes, samples_ws, samples_ws_bytes, N, M, c, q_max, orders_mask, outs,
                        (hipStream_t)stream);
}

int pigs_plan_backward(void* workspace, size_t workspace_bytes, const void* samples_ws, size_t samples_ws_bytes,
                       int64_t N, int64_t M, int c, float q_max, int orders_mask, const void* gout0,
                       const void* gout1, const void* gout2, const void* gout3, void* g_means, void* g_conics,
                       void* g_values, void* stream) {
    if (!mask_valid(orders_mask)) return PIGS_ERR_INVALID;
    const void* gs[4] = {gout0, gout1, gout2, gout3};
    for (int k = 0; k < 4; ++k)
        if (mask_uses_slot(orders_mask, k) && !gs[k]) return PIGS_ERR_INVALID;
    if (!g_means || !g_conics || !g_values) return PIGS_ERR_INVALID;
    return plan_backward(workspace, workspace_bytes, samples_ws, samples_ws_bytes, N, M, c, q_max, orders_mask, gs,
                         g_means, g_conics, g_values, (hipStream_t)stream);
}

// ---- linear residual (pair_math.h ORDR): dense when plan_ws is null, else through the plan
int pigs_residual_forward(int dtype, int d, int c, int64_t N, int64_t M, const void* means, const void* conics,
                          const void* values, const void* samples, const double* coeffs, const void* target, void* out,
                          void* plan_ws, size_t plan_ws_bytes, const void* samples_ws, size_t samples_ws_bytes, void* stream) {
    if (!coeffs || (M > 0 && !out)) return PIGS_ERR_INVALID;
    if (plan_ws) {
        if (dtype != PIGS_F32 || d != 2) return PIGS_ERR_UNSUPPORTED;
        void* outs[4] = {out, nullptr, nullptr, nullptr};
        return plan_forward(plan_ws, plan_ws_bytes, samples_ws, samples_ws_bytes, N, M, c, 0.f, 32, outs, (hipStream_t)stream,
                            coeffs, target);
    }
    int rc = check_common(dtype, d, c, 1, N, M, means, conics, values, samples);
    if (rc != PIGS_OK) return rc;
    SampleArgs a{};
    a.dtype = dtype; a.d = d; a.c = c; a.orders_mask = 32; a.N = N; a.M = M;
    a.means = means; a.conics = conics; a.values = values; a.samples = samples;
    a.out[0] = out;
    for (int k = 0; k < 4; ++k) a.resid[k] = coeffs[k];
    a.target = target;
    return dense_dispatch(false, a, (hipStream_t)stream);
}

int pigs_residual_backward(int dtype, int d, int c, int64_t N, int64_t M, const void* means, const void* conics,
                           const void* values, const void* samples, const double* coeffs, const void* gout, void* g_means,
                           void* g_conics, void* g_values, void* plan_ws, size_t plan_ws_bytes, const void* samples_ws,
                           size_t samples_ws_bytes, void* stream) {
    if (!coeffs || (M > 0 && !gout) || (N > 0 && (!g_means || !g_conics || !g_values))) return PIGS_ERR_INVALID;
    if (plan_ws) {
        if (dtype != PIGS_F32 || d != 2) return PIGS_ERR_UNSUPPORTED;
        const void* gs[4] = {gout, nullptr, nullptr, nullptr};
        return plan_backward(plan_ws, plan_ws_bytes, samples_ws, samples_ws_bytes, N, M, c, 0.f, 32, gs, g_means, g_conics,
                             g_values, (hipStream_t)stream, coeffs);
    }
    int rc = check_common(dtype, d, c, 1, N, M, means, conics, values, samples);
    if (rc != PIGS_OK) return rc;
    SampleArgs a{};
    a.dtype = dtype; a.d = d; a.c = c; a.orders_mask = 32; a.N = N; a.M = M;
    a.means = means; a.conics = conics; a.values = values; a.samples = samples;
    a.gout[0] = gout;
    a.g_means = g_means; a.g_conics = g_conics; a.g_values = g_values;
    for (int k = 0; k < 4; ++k) a.resid[k] = coeffs[k];
    return dense_dispatch(true, a, (hipStream_t)stream);
}

static int aggregate_sizes_ok(int dtype, int64_t N, int64_t cap, int L, int K, int F) {
    if (dtype != PIGS_F32 && dtype != PIGS_F64) return PIGS_ERR_UNSUPPORTED;
    if (N < 0 || cap < 1 || L < 1 || K < 1 || F < 0) return PIGS_ERR_INVALID;
    if (L + 2 * (4 * F + 1) > 128 || L + K > 128 || K + F > 128 || N > 0x7fffffffLL) return PIGS_ERR_UNSUPPORTED;   // two components per lane
    return PIGS_OK;
}

size_t pigs_aggregate_workspace_bytes(int dtype, int64_t N) {
    if ((dtype != PIGS_F32 && dtype != PIGS_F64) || N < 0) return 0;
    return aggregate_workspace_bytes(dtype, N);
}

int pigs_aggregate_lists(int dtype, int64_t N, int64_t cap, const void* means, const void* conics, double q_max,
                         void* workspace, size_t workspace_bytes, int flags, int32_t* row_counts, int32_t* row_lists,
                         int32_t* col_counts, int32_t* col_lists, int32_t* overflow, void* stream) {
    if (dtype != PIGS_F32 && dtype != PIGS_F64) return PIGS_ERR_UNSUPPORTED;
    if (N < 0 || cap < 1 || !(q_max > 0)) return PIGS_ERR_INVALID;
    if ((row_lists == nullptr) != (col_lists == nullptr)) return PIGS_ERR_INVALID;
    if (N > 0 && (!means || !conics || !row_counts || !col_counts || (row_lists && !overflow))) return PIGS_ERR_INVALID;
    return aggregate_lists(dtype, N, cap, means, conics, q_max, workspace, workspace_bytes, flags, row_counts, row_lists,
                           col_counts, col_lists, overflow, (hipStream_t)stream);
}

int pigs_aggregate_forward(int dtype, int64_t N, int64_t cap, int L, int K, int F, const void* means, const void* conics,
                           const int32_t* row_counts, const int32_t* row_lists, const void* features,
                           const void* transform, const void* queries, const void* keys, const void* frequencies,
                           const void* distance_transform, void* out, void* lse, void* acc, void* stream) {
    const int rc = aggregate_sizes_ok(dtype, N, cap, L, K, F);
    if (rc != PIGS_OK) return rc;
    if (N > 0 && (!means || !conics || !row_counts || !row_lists || !features || !transform || !queries || !keys ||
                  (F > 0 && !frequencies) || !distance_transform || !out || !lse || !acc))
        return PIGS_ERR_INVALID;
    AggregateArgs a{};
    a.dtype = dtype; a.N = N; a.cap = cap; a.L = L; a.K = K; a.F = F;
    a.means = means; a.conics = conics; a.row_counts = row_counts; a.row_lists = row_lists;
    a.features = features; a.transform = transform; a.queries = queries; a.keys = keys; a.frequencies = frequencies;
    a.distance_transform = distance_transform; a.out = out; a.lse = lse; a.acc = acc;
    return aggregate_forward(a, (hipStream_t)stream);
}

size_t pigs_aggregate_backward_scratch_bytes(int dtype, int64_t N, int L, int F) {
    if ((dtype != PIGS_F32 && dtype != PIGS_F64) || N < 0 || L < 1 || F < 0) return 0;
    return aggregate_backward_scratch_bytes(dtype, N, L, F);
}

int pigs_aggregate_backward(int dtype, int64_t N, int64_t cap, int L, int K, int F, const void* means,
                            const void* conics, const int32_t* row_counts, const int32_t* row_lists,
                            const int32_t* col_counts, const int32_t* col_lists, const void* features,
                            const void* transform, const void* queries, const void* keys, const void* frequencies,
                            const void* distance_transform, const void* lse, const void* acc, const void* gout,
                            void* scratch, size_t scratch_bytes, void* g_features, void* g_transform, void* g_queries,
                            void* g_keys, void* g_frequencies, void* g_distance_transform, void* stream) {
    const int rc = aggregate_sizes_ok(dtype, N, cap, L, K, F);
    if (rc != PIGS_OK) return rc;
    if (N > 0 && (!means || !conics || !row_counts || !row_lists || !col_counts || !col_lists || !features || !transform ||
                  !queries || !keys || (F > 0 && !frequencies) || !distance_transform || !lse || !acc || !gout || !scratch ||
                  !g_features || !g_transform || !g_queries || !g_keys || (F > 0 && !g_frequencies) || !g_distance_transform))
        return PIGS_ERR_INVALID;
    if (N > 0 && scratch_bytes < aggregate_backward_scratch_bytes(dtype, N, L, F)) return PIGS_ERR_WORKSPACE;
    AggregateArgs a{};
    a.dtype = dtype; a.N = N; a.cap = cap; a.L = L; a.K = K; a.F = F;
    a.means = means; a.conics = conics; a.row_counts = row_counts; a.row_lists = row_lists;
    a.col_counts = col_counts; a.col_lists = col_lists;
    a.features = features; a.transform = transform; a.queries = queries; a.keys = keys; a.frequencies = frequencies;
    a.distance_transform = distance_transform;
    a.lse = const_cast<void*>(lse); a.acc = const_cast<void*>(acc); a.gout = gout; a.scratch = scratch;
    a.g_features = g_features; a.g_transform = g_transform; a.g_queries = g_queries; a.g_keys = g_keys;
    a.g_frequencies = g_frequencies; a.g_distance_transform = g_distance_transform;
    return aggregate_backward(a, (hipStream_t)stream);
}

}  // extern "C"

// Host-side argument block shared by the launchers behind the C ABI (include/pigs_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pigs_amd.h"

namespace pigs {

struct SampleArgs {
    int dtype, d, c, orders_mask;
    int64_t N, M;
    const void *means, *conics, *values, *samples;
    void* out[4];          // forward outputs (orders 0..3)
    const void* gout[4];   // backward: incoming gradients
    void *g_means, *g_conics, *g_values;
    double resid[4];       // orders_mask == 32 (linear residual): a0, a1x, a1y, aL
    const void* target;    //   and its target [M][c] (or null)
};

int dense_dispatch(bool backward, const SampleArgs& a, hipStream_t stream);

// covariances.hip
int covariances_dispatch(bool backward, int dtype, int64_t N, const void* scaling, const void* transform,
                         const void* a, const void* b, void* o0, void* o1, hipStream_t stream);

// aggregate.hip
struct AggregateArgs {
    int dtype;
    int64_t N, cap;
    int L, K, F;
    const void *means, *conics;
    const int32_t *row_counts, *row_lists, *col_counts, *col_lists;
    const void *features, *transform, *queries, *keys, *frequencies, *distance_transform;
    void *out, *lse, *acc;                         // forward outputs (the backward reads lse and acc)
    const void* gout;                              // backward: incoming gradient [N][L]
    void* scratch;                                 // backward: dacc [N][W], D [N], per-row d frequencies [N][F]
    void *g_features, *g_transform, *g_queries, *g_keys, *g_frequencies, *g_distance_transform;
};
size_t aggregate_backward_scratch_bytes(int dtype, int64_t N, int L, int F);
size_t aggregate_workspace_bytes(int dtype, int64_t N);
int aggregate_lists(int dtype, int64_t N, int64_t cap, const void* means, const void* conics, double q_max, void* workspace,
                    size_t workspace_bytes, int flags, int32_t* row_counts, int32_t* row_lists, int32_t* col_counts,
                    int32_t* col_lists, int32_t* overflow, hipStream_t stream);
int aggregate_forward(const AggregateArgs& a, hipStream_t stream);
int aggregate_backward(const AggregateArgs& a, hipStream_t stream);

// plan.hip
size_t samples_workspace_bytes(int64_t M);
size_t plan_workspace_bytes(int64_t N, int64_t M, int c);
int samples_build(void* sws, size_t sws_bytes, int64_t M, const void* samples, hipStream_t stream);
int samples_order_hint(int64_t M);
int plan_build(void* ws, size_t ws_bytes, void* sws, size_t sws_bytes, int flags, int64_t N, int64_t M, int c,
               float q_max, float q_max_backward, const void* means, const void* conics, const void* values, const void* samples,
               hipStream_t stream);
int plan_forward(void* ws, size_t ws_bytes, const void* sws, size_t sws_bytes, int64_t N, int64_t M, int c,
                 float q_max, int mask, void* const* out, hipStream_t stream, const double* resid = nullptr,
                 const void* target = nullptr);
int plan_backward(void* ws, size_t ws_bytes, const void* sws, size_t sws_bytes, int64_t N, int64_t M, int c,
                  float q_max, int mask, const void* const* gout, void* g_means, void* g_conics, void* g_values,
                  hipStream_t stream, const double* resid = nullptr);
int plan_layout_info(int64_t N, int64_t M, int c, int64_t* info);
// the Gaussian grid alone (aggregate.hip): plan.hip
size_t aggregate_grid_bytes(int64_t N);
int aggregate_grid_build(void* ws, size_t ws_bytes, int64_t N, float q_grid, const float* means, const float* conics,
                         hipStream_t stream);
size_t samples_error_offset();
size_t samples_lattice_offset();
size_t plan_strips_offset();
size_t plan_error_offset();

// Order masks: bit k < 4 = derivative order k (pointer slot k); bit 4 (16) = the TRACE of the order-2
// output (the Laplacian), which takes pointer slot 2 in place of the full Hessian, [M][c].
// 32 = the linear residual (pair_math.h ORDR), alone, in slot 0 -- reachable through pigs_residual_* only.
inline bool mask_valid(int m) { return m > 0 && m < 32 && !((m & 4) && (m & 16)); }
inline bool mask_uses_slot(int m, int k) { return m == 32 ? k == 0 : (m >> k & 1) || (k == 2 && (m & 16)); }
// Smallest compiled mask covering the request (compiled: single orders, 0..2, 0..3, the trace alone
// and orders 0, 1 + trace); 0 = no compiled kernel (trace together with order 3).
inline int covering_mask_of(int mask) {
    if (mask == 32) return 32;
    if (mask & 16) return mask == 16 ? 16 : (mask & ~19) == 0 ? 19 : 0;
    if (mask == 1 || mask == 2 || mask == 4 || mask == 8) return mask;
    if ((mask & ~7) == 0) return 7;
    return 15;
}

// The HIP "last error" is sticky per thread and the host process (PyTorch) makes its own HIP
// calls: clear it before a launch, read it after.
extern thread_local hipError_t g_last_hip_error;      // capi.hip; reported by pigs_last_hip_error()
inline void clear_hip_error() { (void)hipGetLastError(); }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return PIGS_OK;
    g_last_hip_error = e;
    return PIGS_ERR_LAUNCH;
}

}  // namespace pigs

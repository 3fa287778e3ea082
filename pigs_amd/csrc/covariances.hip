// Fused covariance builder: the caller-side step right before preprocess().
//
// Replaces gaussians.build_covariances / build_full_covariances + flatten_covariances
// (/root/reference/gaussians.py:163-193; called once per step at model_pn.py:499-502, 538-541,
// 607-610, 695-698 and by every test driver) -- there a chain of ~10 small torch launches (tanh,
// prod, sqrt, diag_embed, two index writes, a batched 2x2 inverse, two gathers) plus as many again
// in its backward.  Here: one launch forward, one launch backward, one thread per Gaussian.
//
//   h = tanh(t), r = sqrt(s0 s1), k = 1 / (1 - h^2)
//   covariance = (s0, h r, s1)           conic = covariance^-1 = (k / s0, -h k / r, k / s1)
// 1 / (1 - h^2) = cosh(t)^2 is evaluated as (1 + e)^2 / (4 e) with e = exp(-2 |t|): no cancellation for large |t|.
#include <hip/hip_runtime.h>

#include "launch.h"

namespace pigs {

template <typename T>
struct CovTerms {
    T s0, s1, h, r, k;     // k = 1 / (1 - h^2)
    __device__ __forceinline__ CovTerms(const T* __restrict__ scaling, const T* __restrict__ transform, int64_t i) {
        s0 = scaling[2 * i];
        s1 = scaling[2 * i + 1];
        const T t = transform[i];
        const T e = exp(T(-2) * fabs(t));
        h = tanh(t);                                       // accurate near 0, where (1 - e) / (1 + e) cancels
        k = (T(1) + e) * (T(1) + e) / (T(4) * e);          // cosh(t)^2: accurate where 1 - h^2 cancels
        r = sqrt(s0 * s1);
    }
};

template <typename T>
__global__ __launch_bounds__(256) void build_covariances_kernel(int64_t N, const T* __restrict__ scaling,
                                                                const T* __restrict__ transform, T* __restrict__ cov,
                                                                T* __restrict__ conic) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const CovTerms<T> c(scaling, transform, i);
    if (cov) {
        cov[3 * i] = c.s0;
        cov[3 * i + 1] = c.h * c.r;
        cov[3 * i + 2] = c.s1;
    }
    if (conic) {
        conic[3 * i] = c.k / c.s0;
        conic[3 * i + 1] = -c.h * c.k / c.r;
        conic[3 * i + 2] = c.k / c.s1;
    }
}

// gradients of <g_cov, cov> + <g_conic, conic> (null = zero) wrt scaling [N][2] and transform [N]
template <typename T>
__global__ __launch_bounds__(256) void build_covariances_backward_kernel(
    int64_t N, const T* __restrict__ scaling, const T* __restrict__ transform, const T* __restrict__ g_cov,
    const T* __restrict__ g_conic, T* __restrict__ g_scaling, T* __restrict__ g_transform) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const CovTerms<T> c(scaling, transform, i);
    T gxx = 0, gxy = 0, gyy = 0, qxx = 0, qxy = 0, qyy = 0;
    if (g_cov) { gxx = g_cov[3 * i]; gxy = g_cov[3 * i + 1]; gyy = g_cov[3 * i + 2]; }
    if (g_conic) { qxx = g_conic[3 * i]; qxy = g_conic[3 * i + 1]; qyy = g_conic[3 * i + 2]; }
    const T tau = c.h * c.r, hk = c.h * c.k;
    const T half = T(0.5);
    g_scaling[2 * i] = gxx + half * gxy * tau / c.s0 - qxx * c.k / (c.s0 * c.s0) + half * qxy * hk / (c.r * c.s0);
    g_scaling[2 * i + 1] = gyy + half * gxy * tau / c.s1 - qyy * c.k / (c.s1 * c.s1) + half * qxy * hk / (c.r * c.s1);
    const T g_h = gxy * c.r + T(2) * hk * c.k * (qxx / c.s0 + qyy / c.s1) - qxy * c.k * c.k * (T(1) + c.h * c.h) / c.r;
    g_transform[i] = g_h / c.k;     // dh/dt = 1 - h^2
}

template <typename T>
static int launch_cov(bool backward, int64_t N, const void* scaling, const void* transform, const void* a,
                      const void* b, void* o0, void* o1, hipStream_t stream) {
    if (N == 0) return PIGS_OK;
    const int64_t blocks = (N + 255) / 256;
    if (blocks > 0x7fffffffLL) return PIGS_ERR_INVALID;
    clear_hip_error();
    if (!backward)
        hipLaunchKernelGGL(build_covariances_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, stream, N,
                           (const T*)scaling, (const T*)transform, (T*)o0, (T*)o1);
    else
        hipLaunchKernelGGL(build_covariances_backward_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, stream, N,
                           (const T*)scaling, (const T*)transform, (const T*)a, (const T*)b, (T*)o0, (T*)o1);
    return launch_status();
}

int covariances_dispatch(bool backward, int dtype, int64_t N, const void* scaling, const void* transform,
                         const void* a, const void* b, void* o0, void* o1, hipStream_t stream) {
    if (dtype == PIGS_F32) return launch_cov<float>(backward, N, scaling, transform, a, b, o0, o1, stream);
    if (dtype == PIGS_F64) return launch_cov<double>(backward, N, scaling, transform, a, b, o0, o1, stream);
    return PIGS_ERR_UNSUPPORTED;
}

}  // namespace pigs

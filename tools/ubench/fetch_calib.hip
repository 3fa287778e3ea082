// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths of the forward kernel:
// coalesced loads of W bytes per lane over a buffer far larger than L2 (W = 4, 8, 12, 16).
// Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- tools/ubench/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct P12 { float x, y; uint32_t m; };
template <typename T> __global__ __launch_bounds__(256) void rd(const T* __restrict__ in, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const T v = in[i];
    float s;
    if constexpr (sizeof(T) == 4) s = v;
    else if constexpr (sizeof(T) == 8) s = v.x + v.y;
    else if constexpr (sizeof(T) == 12) s = v.x + v.y + (float)v.m;
    else s = v.x + v.y + v.z + v.w;
    if (s == 12345.678f) out[0] = s;        // never true: keeps the load
}
int main() {
    const uint32_t n = 1u << 24;            // 16 M elements: 64 .. 256 MB per pass
    void* buf; float* out;
    (void)hipMalloc(&buf, (size_t)n * 16); (void)hipMalloc(&out, 64);
    (void)hipMemset(buf, 0, (size_t)n * 16);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rd<float>, dim3(n / 256), dim3(256), 0, 0, (const float*)buf, out, n);
        hipLaunchKernelGGL(rd<float2>, dim3(n / 256), dim3(256), 0, 0, (const float2*)buf, out, n);
        hipLaunchKernelGGL(rd<P12>, dim3(n / 256), dim3(256), 0, 0, (const P12*)buf, out, n);
        hipLaunchKernelGGL(rd<float4>, dim3(n / 256), dim3(256), 0, 0, (const float4*)buf, out, n);
    }
    (void)hipDeviceSynchronize();
    printf("bytes per launch: 4 B/lane %zu, 8 B/lane %zu, 12 B/lane %zu, 16 B/lane %zu\n", (size_t)n * 4, (size_t)n * 8,
           (size_t)n * 12, (size_t)n * 16);
    return 0;
}

// Micro-benchmark: streaming read/write with and without scattered returning global atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int ATOMICS_PER_BLOCK, bool SPREAD>
__global__ __launch_bounds__(1024) void k(const float2* in, uint2* out, uint32_t* counters, uint32_t ncounters) {
    const uint32_t i0 = blockIdx.x * 4096 + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63);
    float2 p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = in[i0 + 64 * q];
    uint32_t base = 0;
    if (threadIdx.x < ATOMICS_PER_BLOCK) {
        const uint32_t c = SPREAD ? (blockIdx.x * 977u + threadIdx.x * 16u) % ncounters : (blockIdx.x / 4 * 128 + threadIdx.x) % ncounters;
        base = atomicAdd(&counters[c], 1u);
    }
    __shared__ uint32_t sh[1024];
    sh[threadIdx.x] = base;
    __syncthreads();
    base = sh[threadIdx.x & 127];
#pragma unroll
    for (int q = 0; q < 4; ++q) out[i0 + 64 * q] = make_uint2(__float_as_uint(p[q].x) + base, __float_as_uint(p[q].y));
}
template <int A, bool S>
void run(const char* name, float2* in, uint2* out, uint32_t* c) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<A, S>), dim3(256), dim3(1024), 0, 0, in, out, c, 16384u);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k<A, S>), dim3(256), dim3(1024), 0, 0, in, out, c, 16384u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.2f us per launch\n", name, ms / 20 * 1e3);
}
int main() {
    float2* in; uint2* out; uint32_t* c;
    (void)hipMalloc(&in, 1 << 23); (void)hipMalloc(&out, 1 << 23); (void)hipMalloc(&c, 16384 * 4);
    (void)hipMemset(in, 0, 1 << 23); (void)hipMemset(c, 0, 16384 * 4);
    run<0, false>("8MB read + 8MB write, no atomics", in, out, c);
    run<128, false>("+128 atomics/block, dense counters", in, out, c);
    run<128, true>("+128 atomics/block, spread counters", in, out, c);
    run<1024, false>("+1024 atomics/block, dense", in, out, c);
    run<1024, true>("+1024 atomics/block, spread", in, out, c);
    return 0;
}

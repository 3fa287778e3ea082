// Micro-benchmark: throughput of scattered device-scope atomics (the counting sort of the plan build):
// T threads, one atomicAdd each on a pseudo-random counter out of NC, counters STRIDE words apart,
// returning (value used) or not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <bool RET>
__global__ __launch_bounds__(256) void k(uint32_t* counters, uint32_t nc, uint32_t stride, uint32_t* out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = (hash(i) % nc) * stride;
    if (RET) out[i] = atomicAdd(&counters[c], 1u);
    else { atomicAdd(&counters[c], 1u); }
}
template <bool RET>
void run(const char* name, uint32_t T, uint32_t nc, uint32_t stride, uint32_t* c, uint32_t* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<RET>, dim3(T / 256), dim3(256), 0, 0, c, nc, stride, out);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k<RET>, dim3(T / 256), dim3(256), 0, 0, c, nc, stride, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s T=%8u NC=%6u stride=%3u : %8.2f us per launch  (%.2f atomics/ns)\n", name, T, nc, stride, ms / 20 * 1e3,
           T / (ms / 20 * 1e6));
}
int main() {
    uint32_t *c, *out;
    (void)hipMalloc(&c, (size_t)1 << 26); (void)hipMalloc(&out, (size_t)1 << 24);
    (void)hipMemset(c, 0, (size_t)1 << 26);
    const uint32_t M = 1u << 20;
    for (uint32_t stride : {1u, 4u, 16u, 32u, 64u}) {
        run<true>("returning", M, 16384, stride, c, out);
        run<false>("fire-and-forget", M, 16384, stride, c, out);
    }
    for (uint32_t nc : {262144u, 65536u, 4096u, 1280u, 256u})
        for (uint32_t stride : {1u, 32u}) run<true>("returning", M, nc, stride, c, out);
    for (uint32_t stride : {1u, 32u}) run<true>("returning", 65536, 1280, stride, c, out);
    for (uint32_t stride : {1u, 32u}) run<true>("returning", 65536, 16384, stride, c, out);
    return 0;
}

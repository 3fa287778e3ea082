// Micro-benchmark: the backward's flush pattern.  W waves; in each, ACTIVE lanes add NV floats into
// acc[k][base + lane] (SoA, base pseudo-random per wave: consecutive lanes hit consecutive words).
//   device : atomicAdd at agent scope (what tile_backward_kernel does)
//   xcd    : 8 copies of acc, one per XCD (HW_REG_XCC_ID), atomics at workgroup scope (stay in the XCD's L2)
// The second form is checked: the copies must add up to the same totals.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
constexpr int NV = 6;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* acc, uint32_t N, uint32_t active, uint32_t* xcc_seen) {
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t base = hash(wave) % (N - 64);
    uint32_t xcc = 0;
    if (MODE == 1) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        if (lane == 0) xcc_seen[blockIdx.x] = xcc | ((blockIdx.x & 7u) << 8);
    }
    if (lane >= active) return;
    float* a = acc + (size_t)xcc * NV * N + base + lane;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        if (MODE == 0) atomicAdd(a + (size_t)q * N, 1.0f);
        else __hip_atomic_fetch_add(a + (size_t)q * N, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__global__ void fold(float* acc, uint32_t N) {      // copy 0 += copies 1..7
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NV * N) return;
    float s = 0.f;
    for (int x = 0; x < 8; ++x) s += acc[(size_t)x * NV * N + i];
    acc[i] = s;
}
int main() {
    const uint32_t N = 65536, W = 16384;
    float *acc, *ref; uint32_t* seen;
    (void)hipMalloc(&acc, (size_t)8 * NV * N * 4); (void)hipMalloc(&ref, (size_t)NV * N * 4); (void)hipMalloc(&seen, W);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t active : {45u, 64u, 16u}) {
        for (int mode = 0; mode < 2; ++mode) {
            (void)hipMemset(acc, 0, (size_t)8 * NV * N * 4);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(W / 4), dim3(256), 0, 0, acc, N, active, seen);
                else hipLaunchKernelGGL(k<1>, dim3(W / 4), dim3(256), 0, 0, acc, N, active, seen);
            };
            launch();
            if (mode == 0) (void)hipMemcpy(ref, acc, (size_t)NV * N * 4, hipMemcpyDeviceToDevice);
            else {
                hipLaunchKernelGGL(fold, dim3(NV * N / 256), dim3(256), 0, 0, acc, N);
                std::vector<float> a(NV * N), b(NV * N);
                (void)hipMemcpy(a.data(), acc, a.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(b.data(), ref, b.size() * 4, hipMemcpyDeviceToHost);
                size_t bad = 0; double tot = 0;
                for (size_t i = 0; i < a.size(); ++i) { bad += a[i] != b[i]; tot += a[i]; }
                std::vector<uint32_t> s(W / 4);
                (void)hipMemcpy(s.data(), seen, s.size() * 4, hipMemcpyDeviceToHost);
                size_t off = 0; for (auto v : s) off += (v & 0xff) != (v >> 8);
                printf("  check: %zu of %zu words differ from the device-scope result (total %.0f); %zu of %zu workgroups ran on an XCD other than blockIdx %% 8\n",
                       bad, a.size(), tot, off, s.size());
            }
            for (int w = 0; w < 3; ++w) launch();
            hipEventRecord(e0);
            for (int w = 0; w < 20; ++w) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-8s active=%2u : %8.2f us per launch (%.1f lane-atomics/ns)\n", mode ? "xcd" : "device", active, ms / 20 * 1e3,
                   (double)W * active * NV / (ms / 20 * 1e6));
        }
    }
    return 0;
}

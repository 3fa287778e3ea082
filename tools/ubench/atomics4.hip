// Are float atomics cheaper when they stay in the issuing XCD's L2?  Same traffic shape as the backward
// (4.4 M lane-atomics in runs of 16 consecutive indices, 6 values per index, SoA over N = 65 536; every
// workgroup works on a contiguous strip of indices so that an index is met by few XCDs):
//   (a) agent-scope atomics into one array (what tile_backward_kernel does),
//   (b) workgroup-scope atomics into the copy of the XCD the wave runs on (HW_REG_XCC_ID), 8 copies;
//       the L2 is shared by all CUs of an XCD, so within a copy the L2 is the point of coherence.
// Also prints whether workgroup i really ran on XCD i % 8.
// hipcc -O3 --offload-arch=gfx950 atomics4.hip -o atomics4 && ./atomics4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* acc, uint32_t N, uint32_t per_wave, uint32_t* xcd_mismatch) {
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t x = xcc_id();
    if (lane == 0 && (threadIdx.x >> 6) == 0 && x != (blockIdx.x & 7u)) atomicAdd(xcd_mismatch, 1u);
    // strip of the domain for this wave: base index follows the wave id (neighbouring waves overlap heavily)
    const uint32_t nwaves = gridDim.x * 4;
    const uint32_t base = (uint32_t)((uint64_t)wave * (N - 256) / nwaves);
    uint32_t s = wave * 2654435761u + lane / 16 * 97u;
    float* dst = MODE == 0 ? acc : acc + (size_t)x * 6 * N;
    for (uint32_t it = 0; it < per_wave; ++it) {
        s = s * 1664525u + 1013904223u;
        const uint32_t idx = base + ((s >> 8) % 240u) + (lane & 15u);     // runs of 16 consecutive indices
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (MODE == 0) atomicAdd(&dst[(size_t)q * N + idx], 1.0f);
            else __hip_atomic_fetch_add(&dst[(size_t)q * N + idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

__global__ void reduce8(const float* acc, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int x = 0; x < 8; ++x) s += acc[(size_t)x * n + i];
    out[i] = s;
}

int main() {
    const uint32_t N = 65536, waves = 16384, per_wave = 12;      // 16384 * 64 * 12 * 6 = 75 M lane-atomics? no: see below
    // backward at C3: 16384 tiles x ~45 entries x 6 values = 4.4 M lane-atomic INSTRUCTION lanes; here every
    // wave issues per_wave x 6 instructions of 64 lanes: 16384 x 12 x 6 x 64 = 75 M lanes -- scale down per_wave
    float *a0, *a1, *out;
    uint32_t* mm;
    CHECK(hipMalloc(&a0, sizeof(float) * 6 * N));
    CHECK(hipMalloc(&a1, sizeof(float) * 6 * N * 8));
    CHECK(hipMalloc(&out, sizeof(float) * 6 * N));
    CHECK(hipMalloc(&mm, 4));
    CHECK(hipMemset(mm, 0, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (uint32_t pw : {1u, 2u}) {
        for (int mode = 0; mode < 2; ++mode) {
            CHECK(hipMemset(a0, 0, sizeof(float) * 6 * N));
            CHECK(hipMemset(a1, 0, sizeof(float) * 6 * N * 8));
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves / 4), dim3(256), 0, 0, a0, N, pw, mm);
                else hipLaunchKernelGGL(k<1>, dim3(waves / 4), dim3(256), 0, 0, a1, N, pw, mm);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double lanes = (double)waves * pw * 6 * 64;
            printf("per_wave %u mode %s: %.1f us  (%.2f M lane-atomics, %.1f per ns)\n", pw, mode ? "workgroup-scope, per-XCD copy" : "agent-scope, one array",
                   best * 1e3, lanes / 1e6, lanes / (best * 1e6));
        }
    }
    // correctness of mode 1: the 8 copies summed must equal the expected total count
    CHECK(hipMemset(a1, 0, sizeof(float) * 6 * N * 8));
    hipLaunchKernelGGL(k<1>, dim3(waves / 4), dim3(256), 0, 0, a1, N, 2u, mm);
    hipLaunchKernelGGL(reduce8, dim3(6 * N / 256), dim3(256), 0, 0, a1, out, 6 * N);
    std::vector<float> h(6 * N);
    CHECK(hipMemcpy(h.data(), out, sizeof(float) * 6 * N, hipMemcpyDeviceToHost));
    double tot = 0;
    for (float v : h) tot += v;
    uint32_t hm;
    CHECK(hipMemcpy(&hm, mm, 4, hipMemcpyDeviceToHost));
    printf("sum over copies %.0f, expected %.0f; workgroups not on XCD blockIdx %% 8: %u\n", tot, (double)waves * 2 * 6 * 64, hm);
    return 0;
}

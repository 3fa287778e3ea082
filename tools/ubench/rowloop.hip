// Micro-benchmark: the forward kernel's row loop (evaluate_rows<1,7>) on a resident LDS queue, no global
// traffic: what one (row of 16 points x 1 Gaussian) x 4 costs per SIMD at a given occupancy.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -o rowloop rowloop.hip && ./rowloop
#include "../../pigs_amd/csrc/plan.hip"
#include <cstdio>
namespace pigs { thread_local hipError_t g_last_hip_error = hipSuccess; }
using namespace pigs;

template <int C, int MASK>
__global__ __launch_bounds__(256, 8) void k(float* out, int rows, int iters) {
    using L = FwdLayout<2, C, MASK>;
    __shared__ FwdLds lds_all[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    FwdLds& lds = lds_all[wave];
    float* f = (float*)lds.rec;
    for (int i = lane; i < (int)(sizeof(FwdLds) / 4); i += 64) f[i] = 0.01f * (float)((i * 37 + wave) % 101) + 0.5f;
    __syncthreads();
    float acc[L::N];
    for (int q = 0; q < L::N; ++q) acc[q] = 0.f;
    const float s[2] = {0.3f + lane * 1e-3f, 0.7f - lane * 1e-3f};
    for (int it = 0; it < iters; ++it) evaluate_rows<C, MASK>(acc, s, lds, rows, lane);
    float t = 0;
    for (int q = 0; q < L::N; ++q) t += acc[q];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 16 * 256 * 4);
    const int rows = 32, iters = 400;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<1, 7>), dim3(blocks), dim3(256), 0, 0, out, rows, 10);
        hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<1, 7>), dim3(blocks), dim3(256), 0, 0, out, rows, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double wave_rows_per_simd = (double)rows * iters * wps;
        printf("waves/SIMD=%d  %.3f ms  ns per wave-row per SIMD = %.2f\n", wps, best, best * 1e6 / wave_rows_per_simd);
    }
    return 0;
}

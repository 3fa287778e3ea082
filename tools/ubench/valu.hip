// Micro-benchmark: VALU issue rates on gfx950 (v_fma_f32, v_pk_fma_f32, v_exp_f32) by occupancy.
// hipcc --offload-arch=gfx950 -O3 -o valu valu.hip && ./valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[8];
    float2v y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = float2v{x[i], x[i] + 1.f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[i]) : "v"(float2v{a, a}), "v"(float2v{b, b}));
                if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "s"(a), "v"(b));
                if (MODE == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
                if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(y[i]) : "v"(float2v{a, a}));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 2000;
    for (int wpsimd : {1, 2, 4, 8}) {
        // 256 CUs x 4 SIMDs; blocks of 256 threads = 4 waves = 1 wave per SIMD
        const int blocks = 256 * wpsimd;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double)iters * 32;
        const double cyc = ms * 1e-3 * 2.4e9;   // at nominal 2.4 GHz
        printf("%-14s waves/SIMD=%d  %.3f ms  cycles/instr/SIMD (at 2.4GHz) = %.2f\n", name, wpsimd, ms,
               cyc / (instr_per_wave * wpsimd));
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", out);
    run<3>("v_fma_f32 sgpr", out);
    run<4>("v_mul_f32", out);
    run<1>("v_pk_fma_f32", out);
    run<5>("v_pk_mul_f32", out);
    run<2>("v_exp_f32", out);
    return 0;
}

// Probe: lane semantics of v_permlane32_swap / v_permlane16_swap on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
    for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[t * 64 + i]); printf("\n"); }
    return 0;
}

// Probe: the 8-record transposing butterfly of binned_backward (swap32, swap16, row DPP).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
// hipcc (ROCm 7.2) mis-lowers __builtin_amdgcn_permlane{32,16}_swap when both results feed one
// add (it emits v_add v, v, v with the FIRST result twice), so the swaps are inline asm.  The
// s_nop covers the VALU-write -> permlane-swap-read wait states the compiler would insert.
__device__ __forceinline__ float swap32_add(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;     // lanes 0-31: a[l] + a[l+32]; lanes 32-63: b[l-32] + b[l]
}
__device__ __forceinline__ float swap16_add(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;     // rows 0,2: a[row] + a[row+1]; rows 1,3: b[row-1] + b[row]
}
__device__ float row_sum_builtin(float v) {
    v += dpp_f32<0xB1>(v); v += dpp_f32<0x4E>(v); v += dpp_f32<0x141>(v); v += dpp_f32<0x140>(v);
    return v;
}
__device__ float row_sum(float v) {      // the asm form used by binned.hip
    asm volatile(
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
__global__ void k(float* o) {
    const int lane = threadIdx.x;
    float part[8];
    for (int u = 0; u < 8; ++u) part[u] = (float)((u + 1) * 1000 + lane);   // sum over lanes = 64000(u+1) + 2016
    const float x0 = swap32_add(part[0], part[4]), x1 = swap32_add(part[1], part[5]);
    const float x2 = swap32_add(part[2], part[6]), x3 = swap32_add(part[3], part[7]);
    const float y0 = row_sum(swap16_add(x0, x2)), y1 = row_sum(swap16_add(x1, x3));
    o[lane] = y0; o[64 + lane] = y1;
}
int main() {
    float* d; float h[128];
    (void)hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r) printf("row %d: y0 = %.0f (expect record %d: %d)  y1 = %.0f (expect record %d: %d)\n", r, h[16 * r], 2 * r, 64000 * (2 * r + 1) + 2016, h[64 + 16 * r], 2 * r + 1, 64000 * (2 * r + 2) + 2016);
    return 0;
}

// A plain C++ host of the C ABI (include/pigs_amd.h): hipMalloc'ed buffers, a hipStream_t, no torch.
// What a native caller of the reference's sampler would write against this library.  Used by
// tests/test_abi_host_gpu.py: reads one case from a binary file, runs the dense forward, the binned
// preprocess (cold, then a second build into the same plan workspace on the existing samples
// workspace with PIGS_BUILD_PLAN_WS_CLEAN), the binned forward and backward, and writes the results.
//
//   in :  int64 N, M, c; float means[N][2], conics[N][3], values[N][c], samples[M][2], gout0[M][c], gout1[M][2][c], gout2[M][2][2][c]
//   out:  dense out0, out1, out2; binned out0, out1, out2; binned g_means, g_conics, g_values
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pigs_amd.h"

#define CHECK_HIP(x)                                                                   \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define CHECK_PIGS(x)                                                                  \
    do {                                                                               \
        int rc_ = (x);                                                                 \
        if (rc_ != 0) { std::fprintf(stderr, "%s: %s (%s)\n", #x, pigs_status_string(rc_), pigs_last_hip_error()); return 3; } \
    } while (0)

static float* upload(const std::vector<float>& h) {
    float* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(float) + 16) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s <case.bin> <result.bin>\n", argv[0]); return 1; }
    if (pigs_abi_version() != PIGS_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    int64_t hdr[3];
    if (std::fread(hdr, sizeof(int64_t), 3, f) != 3) return 1;
    const int64_t N = hdr[0], M = hdr[1];
    const int c = (int)hdr[2];
    auto rd = [&](size_t n) { std::vector<float> v(n); if (std::fread(v.data(), sizeof(float), n, f) != n) std::exit(1); return v; };
    const auto means = rd(N * 2), conics = rd(N * 3), values = rd(N * c), samples = rd(M * 2);
    const auto gout0 = rd(M * c), gout1 = rd(M * 2 * c), gout2 = rd(M * 4 * c);
    std::fclose(f);

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    float *d_means = upload(means), *d_conics = upload(conics), *d_values = upload(values), *d_samples = upload(samples);
    float *d_g0 = upload(gout0), *d_g1 = upload(gout1), *d_g2 = upload(gout2);
    if (!d_means || !d_conics || !d_values || !d_samples || !d_g0 || !d_g1 || !d_g2) return 2;
    float *o0, *o1, *o2, *b0, *b1, *b2, *gm, *gc, *gv;
    CHECK_HIP(hipMalloc(&o0, M * c * 4)); CHECK_HIP(hipMalloc(&o1, M * 2 * c * 4)); CHECK_HIP(hipMalloc(&o2, M * 4 * c * 4));
    CHECK_HIP(hipMalloc(&b0, M * c * 4)); CHECK_HIP(hipMalloc(&b1, M * 2 * c * 4)); CHECK_HIP(hipMalloc(&b2, M * 4 * c * 4));
    CHECK_HIP(hipMalloc(&gm, N * 2 * 4)); CHECK_HIP(hipMalloc(&gc, N * 3 * 4)); CHECK_HIP(hipMalloc(&gv, N * c * 4));

    // dense: exact sums, orders 0..2 in one launch
    CHECK_PIGS(pigs_sample_forward(PIGS_F32, 2, c, 0x7, N, M, d_means, d_conics, d_values, d_samples, o0, o1, o2, nullptr, stream));

    // binned: caller-owned workspaces
    const size_t sb = pigs_samples_workspace_bytes(M), pb = pigs_plan_workspace_bytes(N, M, c);
    if (!sb || !pb) { std::fprintf(stderr, "unsupported sizes\n"); return 1; }
    void *sws, *ws;
    CHECK_HIP(hipMalloc(&sws, sb)); CHECK_HIP(hipMalloc(&ws, pb));
    const float q_max = 36.f;
    CHECK_PIGS(pigs_plan_build(ws, pb, sws, sb, PIGS_BUILD_SAMPLES, N, M, c, q_max, q_max + 4.f, d_means, d_conics, d_values, d_samples, stream));
    // the same Gaussians again on the existing samples workspace, into the same (now clean) plan workspace
    CHECK_PIGS(pigs_plan_build(ws, pb, sws, sb, PIGS_BUILD_PLAN_WS_CLEAN, N, M, c, q_max, q_max + 4.f, d_means, d_conics, d_values, d_samples, stream));
    CHECK_PIGS(pigs_plan_forward(ws, pb, sws, sb, N, M, c, q_max, 0x7, b0, b1, b2, nullptr, stream));
    CHECK_PIGS(pigs_plan_backward(ws, pb, sws, sb, N, M, c, q_max, 0x7, d_g0, d_g1, d_g2, nullptr, gm, gc, gv, stream));
    // the samples half on its own (pigs_samples_build), then a plan on top of it: the same forward results
    void *sws2, *ws2;
    float* b0_2;
    CHECK_HIP(hipMalloc(&sws2, sb)); CHECK_HIP(hipMalloc(&ws2, pb)); CHECK_HIP(hipMalloc(&b0_2, M * c * 4));
    CHECK_PIGS(pigs_samples_build(sws2, sb, M, d_samples, stream));
    CHECK_PIGS(pigs_plan_build(ws2, pb, sws2, sb, 0, N, M, c, q_max, q_max + 4.f, d_means, d_conics, d_values, nullptr, stream));
    CHECK_PIGS(pigs_plan_forward(ws2, pb, sws2, sb, N, M, c, q_max, 0x1, b0_2, nullptr, nullptr, nullptr, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    {
        std::vector<float> x(M * c), y(M * c);
        CHECK_HIP(hipMemcpy(x.data(), b0, M * c * 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(y.data(), b0_2, M * c * 4, hipMemcpyDeviceToHost));
        double worst = 0, top = 0;
        for (size_t k = 0; k < x.size(); ++k) { worst = std::fmax(worst, std::fabs((double)x[k] - y[k])); top = std::fmax(top, std::fabs((double)x[k])); }
        if (worst > 1e-6 * top) { std::fprintf(stderr, "samples_build + plan_build differ from the fused build: %g of %g\n", worst, top); return 5; }
    }
    uint32_t err = 0;
    CHECK_HIP(hipMemcpy(&err, (char*)ws + pigs_plan_error_offset(), 4, hipMemcpyDeviceToHost));
    // a diagnostic since ABI 6 (the scan recomputes what it does not receive in time; the result is valid)
    if (err) std::fprintf(stderr, "note: the plan build's scan took its recompute path\n");

    std::FILE* g = std::fopen(argv[2], "wb");
    if (!g) return 1;
    auto wr = [&](const float* d, size_t n) {
        std::vector<float> h(n);
        if (hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) std::exit(2);
        std::fwrite(h.data(), 4, n, g);
    };
    wr(o0, M * c); wr(o1, M * 2 * c); wr(o2, M * 4 * c);
    wr(b0, M * c); wr(b1, M * 2 * c); wr(b2, M * 4 * c);
    wr(gm, N * 2); wr(gc, N * 3); wr(gv, N * c);
    std::fclose(g);
    std::printf("ok N=%lld M=%lld c=%d\n", (long long)N, (long long)M, c);
    return 0;
}

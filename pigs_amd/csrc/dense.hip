// Dense (no culling) sampler kernels: every sample point against every Gaussian.
//
// This is the exact restatement of the reference semantics (gaussians.py:48-58, 89-116) on the
// GPU and the path taken for small problems (the PINN training loops call the sampler with
// N ~ 1e3 Gaussians x M ~ 1e3 points: model_pn.py:768-772, test_no_mlp.py:104-125) and for
// float64 (gradcheck, test_derivatives.py:96-106).  Large N x M goes through the binned path
// (plan.hip).
//
// Forward:  lane = sample point, Gaussian parameters are wave-uniform and arrive through the
//           scalar data path (s_load) -- no LDS or VGPR traffic per Gaussian.  The NW waves of
//           a workgroup split the Gaussian range and are summed through LDS in a fixed order.
// Backward: lane = Gaussian (its parameters and 5+c gradient accumulators live in VGPRs),
//           sample points and incoming gradients are wave-uniform scalar loads, so no
//           cross-lane reduction is needed.  Waves split the point range; workgroups that
//           share a Gaussian block (gridDim.y > 1) combine with float atomics.
#include "pair_math.h"
#include "launch.h"

namespace pigs {

template <typename T, int D, int C, int MASK, int NW>
__global__ __launch_bounds__(NW * 64) void dense_forward_kernel(
    int64_t N, int64_t M, const T* __restrict__ means, const T* __restrict__ conics,
    const T* __restrict__ values, const T* __restrict__ samples, T* __restrict__ o0, T* __restrict__ o1,
    T* __restrict__ o2, T* __restrict__ o3, Resid<T> rz) {
    using L = FwdLayout<D, C, MASK>;
    constexpr int NF = Sym<D>::NF;
    __shared__ T red[(NW > 1 ? NW - 1 : 1) * L::N * 64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t m = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = m < M;

    T s[D];
#pragma unroll
    for (int i = 0; i < D; ++i) s[i] = valid ? samples[m * D + i] : T(0);

    T acc[L::N];
#pragma unroll
    for (int k = 0; k < L::N; ++k) acc[k] = T(0);

    // four Gaussians per iteration: their scalar loads are issued together, so one memory round
    // trip covers four evaluations (a single-record loop waits on every record)
    constexpr int U = 4;
    int64_t n = (int64_t)wave * U;
    for (; n + U <= N; n += (int64_t)NW * U) {
        T mu[U][D], con[U][NF], v[U][C];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < D; ++i) mu[u][i] = means[(n + u) * D + i];
#pragma unroll
            for (int i = 0; i < NF; ++i) con[u][i] = conics[(n + u) * NF + i];
#pragma unroll
            for (int i = 0; i < C; ++i) v[u][i] = values[(n + u) * C + i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) fwd_accumulate<T, D, C, MASK>(acc, s, mu[u], con[u], v[u], &rz);
    }
    for (int64_t m2 = n; m2 < N && m2 < n + U; ++m2) {          // ragged tail of this wave's last chunk
        T mu[D], con[NF], v[C];
#pragma unroll
        for (int i = 0; i < D; ++i) mu[i] = means[m2 * D + i];
#pragma unroll
        for (int i = 0; i < NF; ++i) con[i] = conics[m2 * NF + i];
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = values[m2 * C + i];
        fwd_accumulate<T, D, C, MASK>(acc, s, mu, con, v, &rz);
    }

    if constexpr (NW > 1) {
        if (wave > 0) {
#pragma unroll
            for (int k = 0; k < L::N; ++k) red[((wave - 1) * L::N + k) * 64 + lane] = acc[k];
        }
        __syncthreads();
        if (wave == 0) {
            for (int w = 0; w < NW - 1; ++w) {
#pragma unroll
                for (int k = 0; k < L::N; ++k) acc[k] += red[(w * L::N + k) * 64 + lane];
            }
        }
    }
    if (wave == 0 && valid) fwd_store<T, D, C, MASK>(acc, m, o0, o1, o2, o3, &rz);
}

template <typename T, int D, int C, int MASK, int NW>
__global__ __launch_bounds__(NW * 64) void dense_backward_kernel(
    int64_t N, int64_t M, const T* __restrict__ means, const T* __restrict__ conics,
    const T* __restrict__ values, const T* __restrict__ samples, const T* __restrict__ G0,
    const T* __restrict__ G1, const T* __restrict__ G2, const T* __restrict__ G3, T* __restrict__ g_means,
    T* __restrict__ g_conics, T* __restrict__ g_values, Resid<T> rz) {
    using L = BwdLayout<D, C>;
    constexpr int EM = MASK == ORDR ? ORDR_AS : MASK;      // a residual's backward = orders 0, 1, trace
    auto load = [&](Gsym<T, D, C, EM>& G, int64_t m) {
        if constexpr (MASK == ORDR) G.load_residual(m, G0, rz);
        else G.load(m, G0, G1, G2, G3);
    };
    constexpr int NF = Sym<D>::NF;
    __shared__ T red[(NW > 1 ? NW - 1 : 1) * L::N * 64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = n < N;
    const int64_t nn = valid ? n : 0;

    T mu[D], con[NF], v[C];
#pragma unroll
    for (int i = 0; i < D; ++i) mu[i] = means[nn * D + i];
#pragma unroll
    for (int i = 0; i < NF; ++i) con[i] = conics[nn * NF + i];
#pragma unroll
    for (int i = 0; i < C; ++i) v[i] = values[nn * C + i];

    T acc[L::N];
#pragma unroll
    for (int k = 0; k < L::N; ++k) acc[k] = T(0);

    // this workgroup's slice of the points, split again over its waves
    const int64_t per = (M + gridDim.y - 1) / gridDim.y;
    const int64_t m_begin = (int64_t)blockIdx.y * per;
    const int64_t m_end = m_begin + per < M ? m_begin + per : M;
    // two points per iteration: their scalar loads (coordinates + incoming gradients) are issued
    // together, halving the exposed memory round trips
    int64_t m = m_begin + (int64_t)wave * 2;
    for (; m + 2 <= m_end; m += (int64_t)NW * 2) {
        T s0[D], s1[D];
#pragma unroll
        for (int i = 0; i < D; ++i) { s0[i] = samples[m * D + i]; s1[i] = samples[(m + 1) * D + i]; }
        Gsym<T, D, C, EM> Ga, Gb;
        load(Ga, m);
        load(Gb, m + 1);
        bwd_accumulate<T, D, C, EM, true>(acc, s0, mu, con, v, Ga);
        bwd_accumulate<T, D, C, EM, true>(acc, s1, mu, con, v, Gb);
    }
    if (m < m_end) {                       // odd point of this wave's last pair
        T s0[D];
#pragma unroll
        for (int i = 0; i < D; ++i) s0[i] = samples[m * D + i];
        Gsym<T, D, C, EM> Ga;
        load(Ga, m);
        bwd_accumulate<T, D, C, EM, true>(acc, s0, mu, con, v, Ga);
    }

    if constexpr (NW > 1) {
        if (wave > 0) {
#pragma unroll
            for (int k = 0; k < L::N; ++k) red[((wave - 1) * L::N + k) * 64 + lane] = acc[k];
        }
        __syncthreads();
        if (wave == 0) {
            for (int w = 0; w < NW - 1; ++w) {
#pragma unroll
                for (int k = 0; k < L::N; ++k) acc[k] += red[(w * L::N + k) * 64 + lane];
            }
        }
    }
    if (wave == 0 && valid) {
        if (gridDim.y == 1) {
#pragma unroll
            for (int i = 0; i < D; ++i) g_means[n * D + i] = acc[L::MU + i];
#pragma unroll
            for (int i = 0; i < NF; ++i) g_conics[n * NF + i] = acc[L::CON + i];
#pragma unroll
            for (int i = 0; i < C; ++i) g_values[n * C + i] = acc[L::VAL + i];
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) atomicAdd(&g_means[n * D + i], acc[L::MU + i]);
#pragma unroll
            for (int i = 0; i < NF; ++i) atomicAdd(&g_conics[n * NF + i], acc[L::CON + i]);
#pragma unroll
            for (int i = 0; i < C; ++i) atomicAdd(&g_values[n * C + i], acc[L::VAL + i]);
        }
    }
}


// ------------------------------------------------------------------------------------------
// FEW POINTS (the reference's own training sizes: N ~ 1e3 Gaussians, 1 024 collocation points, main_pn.py:57,103;
// model_pn.py:766-788): the kernels above give such a launch 16 workgroups on a 256-CU chip, each wave walking
// its share of the Gaussians through a chain of scalar loads -- 16 us forward, 13-22 us backward where the
// arithmetic is worth one.  These two spread the same sums over the whole chip.
// ------------------------------------------------------------------------------------------

// Forward: a workgroup of ROWS_WAVES waves takes 16 points; lane = 16 * row + i: point i of the workgroup, and
// every (wave, row) pair one of 4 * ROWS_WAVES interleaved slices of the Gaussians (row-uniform loads: the 16
// lanes of a row read the same record).  The four rows of a wave meet by two shuffles, the waves in LDS, in a
// fixed order.  (8 waves: 256 VGPRs a wave -- the widest accumulator sets fit without spilling.)
constexpr int ROWS_WAVES = 8;
constexpr int ROWS_CHUNK_BYTES = 36 * 1024;      // Gaussian parameters staged per chunk (1 536 Gaussians of d = 2, c = 1, float32)
template <typename T, int D, int C, int MASK>
__global__ __launch_bounds__(64 * ROWS_WAVES) void dense_forward_rows_kernel(
    int64_t N, int64_t M, const T* __restrict__ means, const T* __restrict__ conics, const T* __restrict__ values,
    const T* __restrict__ samples, T* __restrict__ o0, T* __restrict__ o1, T* __restrict__ o2, T* __restrict__ o3,
    Resid<T> rz) {
    using L = FwdLayout<D, C, MASK>;
    constexpr int NF = Sym<D>::NF;
    constexpr int SLICES = 4 * ROWS_WAVES;
    __shared__ T red[ROWS_WAVES][L::N][16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = lane >> 4, i = lane & 15;
    const int64_t m = (int64_t)blockIdx.x * 16 + i;
    const bool valid = m < M;
    T s[D];
#pragma unroll
    for (int k = 0; k < D; ++k) s[k] = samples[(valid ? m : M - 1) * D + k];
    T acc[L::N];
#pragma unroll
    for (int k = 0; k < L::N; ++k) acc[k] = T(0);
    // The Gaussians come through LDS, a chunk at a time: the workgroup copies the chunk's three parameter arrays in
    // one coalesced round trip, then every (wave, row) slice walks ITS Gaussians of the chunk with row-uniform LDS
    // reads (fetched straight from global memory, a slice's 50 Gaussians at N = 1 600 were a chain of 25
    // dependent round trips: 12 us for 1.6 M pairs).
    constexpr int CHUNK = ROWS_CHUNK_BYTES / (int)((D + NF + C) * sizeof(T));
    __shared__ T smu[CHUNK * D], scon[CHUNK * NF], sval[CHUNK * C];
    for (int64_t n0 = 0; n0 < N; n0 += CHUNK) {
        const int cnt = (int)(N - n0 < CHUNK ? N - n0 : CHUNK);
        if (n0 > 0) __syncthreads();                      // the previous chunk has been read by everyone
        for (int k = threadIdx.x; k < cnt * D; k += 64 * ROWS_WAVES) smu[k] = means[n0 * D + k];
        for (int k = threadIdx.x; k < cnt * NF; k += 64 * ROWS_WAVES) scon[k] = conics[n0 * NF + k];
        for (int k = threadIdx.x; k < cnt * C; k += 64 * ROWS_WAVES) sval[k] = values[n0 * C + k];
        __syncthreads();
#pragma unroll 2
        for (int n = wave * 4 + row; n < cnt; n += SLICES) {
            T mu[D], con[NF], v[C];
#pragma unroll
            for (int k = 0; k < D; ++k) mu[k] = smu[n * D + k];
#pragma unroll
            for (int k = 0; k < NF; ++k) con[k] = scon[n * NF + k];
#pragma unroll
            for (int k = 0; k < C; ++k) v[k] = sval[n * C + k];
            fwd_accumulate<T, D, C, MASK>(acc, s, mu, con, v, &rz);
        }
    }
#pragma unroll
    for (int k = 0; k < L::N; ++k) {
        acc[k] += __shfl_xor(acc[k], 16);
        acc[k] += __shfl_xor(acc[k], 32);
    }
    if (row == 0) {
#pragma unroll
        for (int k = 0; k < L::N; ++k) red[wave][k][i] = acc[k];
    }
    __syncthreads();
    if (wave == 0 && row == 0) {
#pragma unroll
        for (int k = 0; k < L::N; ++k) {
            T t = red[0][k][i];
            for (int w = 1; w < ROWS_WAVES; ++w) t += red[w][k][i];
            acc[k] = t;
        }
        if (valid) fwd_store<T, D, C, MASK>(acc, m, o0, o1, o2, o3, &rz);
    }
}

// Backward: lane = Gaussian (its parameters and 5 + c accumulators in VGPRs, as above); a workgroup = 4 waves =
// 256 Gaussians against ONE slice of 64 points, which its threads fetch together into LDS (coordinates + the
// symmetrised incoming gradients: what the scalar-load chain delivered two points at a time); every wave then
// runs down the 64 points (wave-uniform LDS reads).  gridDim.y = slices of the points; partial sums meet by
// atomics in buffers the caller zeroed.
template <typename T, int D, int C, int MASK>
__global__ __launch_bounds__(256) void dense_backward_staged_kernel(
    int64_t N, int64_t M, const T* __restrict__ means, const T* __restrict__ conics, const T* __restrict__ values,
    const T* __restrict__ samples, const T* __restrict__ G0, const T* __restrict__ G1, const T* __restrict__ G2,
    const T* __restrict__ G3, T* __restrict__ g_means, T* __restrict__ g_conics, T* __restrict__ g_values, Resid<T> rz,
    int slice) {      // points per slice: 64, or 32 where 64 would leave most of the chip without a workgroup
    using L = BwdLayout<D, C>;
    constexpr int NF = Sym<D>::NF;
    constexpr int EM = MASK == ORDR ? ORDR_AS : MASK;
    struct Pt {
        T s[D];
        Gsym<T, D, C, EM> G;
    };
    __shared__ Pt pts[64];
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = n < N;
    const int64_t nn = valid ? n : 0;
    T mu[D], con[NF], v[C];
#pragma unroll
    for (int k = 0; k < D; ++k) mu[k] = means[nn * D + k];
#pragma unroll
    for (int k = 0; k < NF; ++k) con[k] = conics[nn * NF + k];
#pragma unroll
    for (int k = 0; k < C; ++k) v[k] = values[nn * C + k];
    const int64_t m0 = (int64_t)blockIdx.y * slice;
    const int cnt = (int)(M - m0 < slice ? M - m0 : slice);
    if ((int)threadIdx.x < cnt) {
        const int64_t m = m0 + threadIdx.x;
        Pt p;
#pragma unroll
        for (int k = 0; k < D; ++k) p.s[k] = samples[m * D + k];
        if constexpr (MASK == ORDR) p.G.load_residual(m, G0, rz);
        else p.G.load(m, G0, G1, G2, G3);
        pts[threadIdx.x] = p;
    }
    __syncthreads();
    T acc[L::N];
#pragma unroll
    for (int k = 0; k < L::N; ++k) acc[k] = T(0);
    for (int t = 0; t < cnt; ++t) bwd_accumulate<T, D, C, EM, true>(acc, pts[t].s, mu, con, v, pts[t].G);
    if (valid) {
#pragma unroll
        for (int k = 0; k < D; ++k) atomicAdd(&g_means[n * D + k], acc[L::MU + k]);
#pragma unroll
        for (int k = 0; k < NF; ++k) atomicAdd(&g_conics[n * NF + k], acc[L::CON + k]);
#pragma unroll
        for (int k = 0; k < C; ++k) atomicAdd(&g_values[n * C + k], acc[L::VAL + k]);
    }
}

// zero three word arrays (the atomically accumulated gradient buffers) in one launch
__global__ __launch_bounds__(256) void zero_grads_kernel(uint32_t* __restrict__ p0, uint64_t n0, uint32_t* __restrict__ p1,
                                                         uint64_t n1, uint32_t* __restrict__ p2, uint64_t n2) {
    const uint64_t total = n0 + n1 + n2;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        if (i < n0) p0[i] = 0u;
        else if (i < n0 + n1) p1[i - n0] = 0u;
        else p2[i - n0 - n1] = 0u;
    }
}

// ------------------------------------------------------------------------------------------
// Host-side launchers
// ------------------------------------------------------------------------------------------

template <typename T, int D, int C, int MASK>
static int launch_dense_forward(const SampleArgs& a, hipStream_t stream) {
    const int64_t blocks = (a.M + 63) / 64;
    if (blocks == 0 ) return PIGS_OK;
    if (blocks > 0x7fffffffLL) return PIGS_ERR_INVALID;
    clear_hip_error();
    const T* means = (const T*)a.means; const T* conics = (const T*)a.conics;
    const T* values = (const T*)a.values; const T* samples = (const T*)a.samples;
    T* o0 = (T*)a.out[0]; T* o1 = (T*)a.out[1]; T* o2 = (T*)a.out[2]; T* o3 = (T*)a.out[3];
    const Resid<T> rz{(T)a.resid[0], {(T)a.resid[1], (T)a.resid[2]}, (T)a.resid[3], (const T*)a.target};
    constexpr int NACC = FwdLayout<D, C, MASK>::N;
    // few point blocks: spread the Gaussian loop over 16 waves per workgroup
    // (a 1 024-thread workgroup leaves 128 VGPRs per wave: the wide accumulator sets -- several channels, third
    // derivatives, float64 -- spilled there, so they keep the four-wave variant)
    constexpr bool can16 = (15 * NACC * 64 * sizeof(T) <= 60 * 1024) && NACC * (sizeof(T) / 4) <= 7;
    // few points (<= 256 of the 64-point blocks above): 16 points per workgroup, the Gaussians in 32 slices
    constexpr bool can_rows = NACC * (sizeof(T) / 4) <= 28;      // beyond, the two records in flight spill even at 256 VGPRs
    if constexpr (can_rows) {
        if (blocks <= 256 && a.N >= 128) {
            hipLaunchKernelGGL((dense_forward_rows_kernel<T, D, C, MASK>), dim3((unsigned)((a.M + 15) / 16)), dim3(64 * ROWS_WAVES), 0, stream,
                               a.N, a.M, means, conics, values, samples, o0, o1, o2, o3, rz);
            return launch_status();
        }
    }
    if constexpr (can16) {
      if (blocks < 1024 && a.N >= 64) {
        hipLaunchKernelGGL((dense_forward_kernel<T, D, C, MASK, 16>), dim3((unsigned)blocks), dim3(1024), 0, stream,
                           a.N, a.M, means, conics, values, samples, o0, o1, o2, o3, rz);
        return launch_status();
      }
    }
    {
        hipLaunchKernelGGL((dense_forward_kernel<T, D, C, MASK, 4>), dim3((unsigned)blocks), dim3(256), 0, stream,
                           a.N, a.M, means, conics, values, samples, o0, o1, o2, o3, rz);
    }
    return launch_status();
}

template <typename T, int D, int C, int MASK>
static int launch_dense_backward(const SampleArgs& a, hipStream_t stream) {
    constexpr int NF = Sym<D>::NF;
    const int64_t gblocks = (a.N + 63) / 64;
    if (gblocks == 0) return PIGS_OK;
    if (gblocks > 0x7fffffffLL) return PIGS_ERR_INVALID;
    clear_hip_error();
    T* gm = (T*)a.g_means; T* gc = (T*)a.g_conics; T* gv = (T*)a.g_values;
    // The gradient buffers are zeroed by a kernel of our own, one launch for all three:
    // hipMemsetAsync nodes replayed inside a hipGraph were seen (ROCm 7.2) to fill with a stale
    // 16-byte pattern now and then (tests/test_graph_gpu.py), and this is one launch, not three.
    auto zero_grads = [&]() {
        const uint64_t wm = sizeof(T) / 4 * (uint64_t)a.N * D, wc = sizeof(T) / 4 * (uint64_t)a.N * NF,
                       wv = sizeof(T) / 4 * (uint64_t)a.N * C;
        const uint64_t blocks = (wm + wc + wv + 1023) / 1024;
        hipLaunchKernelGGL(zero_grads_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream,
                           (uint32_t*)gm, wm, (uint32_t*)gc, wc, (uint32_t*)gv, wv);
    };
    if (a.M == 0) {
        zero_grads();
        return launch_status();
    }
    // few points: 64-point slices staged in LDS, 256 Gaussians per workgroup (dense_backward_staged_kernel)
    if (a.M <= 16384 && (a.M + 63) / 64 <= 65535) {
        zero_grads();
        const int64_t gx = (a.N + 255) / 256;
        const int slice = gx * ((a.M + 63) / 64) < 512 ? 32 : 64;      // N = 1 600, 1 024 points: 112 workgroups of 64-point slices
        hipLaunchKernelGGL((dense_backward_staged_kernel<T, D, C, MASK>), dim3((unsigned)gx, (unsigned)((a.M + slice - 1) / slice)),
                           dim3(256), 0, stream, a.N, a.M, (const T*)a.means, (const T*)a.conics, (const T*)a.values,
                           (const T*)a.samples, (const T*)a.gout[0], (const T*)a.gout[1], (const T*)a.gout[2], (const T*)a.gout[3],
                           gm, gc, gv, Resid<T>{(T)a.resid[0], {(T)a.resid[1], (T)a.resid[2]}, (T)a.resid[3], nullptr}, slice);
        return launch_status();
    }
    // split the point range over gridDim.y so that ~2048 workgroups exist; each wave should
    // still see >= 64 points
    int64_t ysplit = 2048 / gblocks;
    const int64_t max_split = (a.M + 4 * 64 - 1) / (4 * 64);
    if (ysplit > max_split) ysplit = max_split;
    if (ysplit < 1) ysplit = 1;
    if (ysplit > 65535) ysplit = 65535;
    if (ysplit > 1) zero_grads();
    hipLaunchKernelGGL((dense_backward_kernel<T, D, C, MASK, 4>), dim3((unsigned)gblocks, (unsigned)ysplit), dim3(256),
                       0, stream, a.N, a.M, (const T*)a.means, (const T*)a.conics, (const T*)a.values,
                       (const T*)a.samples, (const T*)a.gout[0], (const T*)a.gout[1], (const T*)a.gout[2],
                       (const T*)a.gout[3], gm, gc, gv,
                       Resid<T>{(T)a.resid[0], {(T)a.resid[1], (T)a.resid[2]}, (T)a.resid[3], nullptr});
    return launch_status();
}

template <typename T, int D, int C>
static int dispatch_mask(bool backward, const SampleArgs& a, hipStream_t stream) {
    // Orders inside the covering mask that were not requested have null output / gradient
    // pointers: the forward skips their stores, the backward reads their gradients as zero.
    const int mask = covering_mask_of(a.orders_mask);
#define PIGS_CASE(MK)                                                                        \
    case MK:                                                                                 \
        return backward ? launch_dense_backward<T, D, C, MK>(a, stream)                      \
                        : launch_dense_forward<T, D, C, MK>(a, stream);
    switch (mask) {
        PIGS_CASE(1) PIGS_CASE(2) PIGS_CASE(4) PIGS_CASE(8) PIGS_CASE(7) PIGS_CASE(15) PIGS_CASE(16) PIGS_CASE(19)
        PIGS_CASE(32)
        default: break;
    }
#undef PIGS_CASE
    return PIGS_ERR_UNSUPPORTED;
}

template <typename T, int D>
static int dispatch_c(bool backward, const SampleArgs& a, hipStream_t stream) {
    switch (a.c) {
        case 1: return dispatch_mask<T, D, 1>(backward, a, stream);
        case 2: return dispatch_mask<T, D, 2>(backward, a, stream);
        case 3: return dispatch_mask<T, D, 3>(backward, a, stream);
        case 4: return dispatch_mask<T, D, 4>(backward, a, stream);
        default: return PIGS_ERR_UNSUPPORTED;
    }
}

template <typename T>
static int dispatch_d(bool backward, const SampleArgs& a, hipStream_t stream) {
    switch (a.d) {
        case 1: return dispatch_c<T, 1>(backward, a, stream);
        case 2: return dispatch_c<T, 2>(backward, a, stream);
        default: return PIGS_ERR_UNSUPPORTED;
    }
}

int dense_dispatch(bool backward, const SampleArgs& a, hipStream_t stream) {
    if (a.dtype == PIGS_F32) return dispatch_d<float>(backward, a, stream);
    if (a.dtype == PIGS_F64) return dispatch_d<double>(backward, a, stream);
    return PIGS_ERR_UNSUPPORTED;
}

}  // namespace pigs

// preprocess_aggregate / aggregate_neighbors of the sampler surface (call sites
// /root/reference/model_pn.py:257-264, test_neighbor_aggregation.py:75-98), float32 and float64, d = 2.
//
// PARITY UNPINNED: the arithmetic of these two methods exists only in the reference's absent CUDA
// source.  The definition implemented here is this repository's own (DESIGN.md "aggregate_neighbors",
// checker: oracle/aggregate_torch.py):
//   neighbours of Gaussian i   N(i) = { j : (mu_i - mu_j)^T C_j (mu_i - mu_j) <= q_max }  (j's ellipse reaches i's centre)
//   attention                  a_ij = softmax_{j in N(i)} <queries_i, keys_j> / sqrt(K)
//   embedding of d = mu_j - mu_i   e_ij = (sin f_k dx, cos f_k dx, sin f_k dy, cos f_k dy)_{k < F}, 1     [E = 4F + 1]
//   message                    m_ij = transform features_j + distance_transform [e_ij ; g_ij e_ij],  g_ij = exp(-q_ij / 2)
//   out_i = sum_j a_ij m_ij = transform fbar_i + distance_transform ebar_i,
//           fbar_i = sum_j a_ij features_j  [L],   ebar_i = sum_j a_ij [e_ij ; g_ij e_ij]  [2E]
//
// Sparse by construction: the neighbour relation is stored as index lists (one row per Gaussian, by
// rows i and -- for the backward's gather -- by columns j); nothing of size [N, N, ...] is ever
// materialised.  One wave per Gaussian, 64 neighbours per round:
//   forward   scores with lane = neighbour (online softmax across rounds), then the weighted sums
//             with lane = component of [fbar ; ebar] and the neighbours broadcast by __shfl
//   backward  per (i, j) pair: da_ij = <dfbar_i, features_j> + <debar_i, emb_ij>, ds_ij = a_ij (da_ij - D_i)
//             with D_i = <dacc_i, acc_i> (the softmax's sum, known without a pass over the pairs);
//             by rows: d queries_i, the per-row partial of d frequencies; by columns (gather over the
//             Gaussians i that have j as a neighbour -- no atomics, deterministic): d features_j, d keys_j.
// d transform, d distance_transform and dacc = gout [transform | distance_transform] are plain
// GEMMs and are left to the caller (pigs_amd/aggregate.py uses torch.matmul).
#include "launch.h"

namespace pigs {

template <typename T> __device__ __forceinline__ T exp_(T x);
template <> __device__ __forceinline__ float exp_<float>(float x) { return __expf(x); }
template <> __device__ __forceinline__ double exp_<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ void sincos_(T x, T* s, T* c);
template <> __device__ __forceinline__ void sincos_<float>(float x, float* s, float* c) { sincosf(x, s, c); }
template <> __device__ __forceinline__ void sincos_<double>(double x, double* s, double* c) { sincos(x, s, c); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}

// q of centre `at` under Gaussian `of`: (mu_at - mu_of)^T C_of (mu_at - mu_of)
template <typename T>
__device__ __forceinline__ T q_of(const T* means, const T* conics, int64_t at, int64_t of, T* dx, T* dy) {
    *dx = means[2 * of] - means[2 * at];          // delta = mu_j - mu_i with i = at, j = of
    *dy = means[2 * of + 1] - means[2 * at + 1];
    const T a = conics[3 * of], b = conics[3 * of + 1], c = conics[3 * of + 2];
    return a * *dx * *dx + T(2) * b * *dx * *dy + c * *dy * *dy;
}

// ---- neighbour lists: row i = { j : q_j(mu_i) <= q_max } and column j = { i : q_j(mu_i) <= q_max },
// each a slab of `cap` indices; one wave per row, candidates tested 64 at a time, order ascending.
template <typename T, bool BY_COLUMN>
__global__ __launch_bounds__(256) void aggregate_lists_kernel(int64_t N, int64_t cap, const T* __restrict__ means,
                                                              const T* __restrict__ conics, T q_max,
                                                              int32_t* __restrict__ counts, int32_t* __restrict__ lists,
                                                              int32_t* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    int64_t n = 0;
    for (int64_t o0 = 0; o0 < N; o0 += 64) {
        const int64_t o = o0 + lane;
        bool in = false;
        if (o < N) {
            T dx, dy;
            const T q = BY_COLUMN ? q_of(means, conics, o, r, &dx, &dy) : q_of(means, conics, r, o, &dx, &dy);
            in = q <= q_max;
        }
        const uint64_t m = __ballot(in);
        const int64_t pos = n + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (in && pos < cap) lists[r * cap + pos] = (int32_t)o;
        n += __builtin_popcountll(m);
    }
    if (lane == 0) {
        counts[r] = (int32_t)(n < cap ? n : cap);
        if (n > cap) atomicOr(overflow, 1);
    }
}

// component c of [features_j ; e_ij ; g e_ij] (c < L: feature; then the 2E embedding entries)
template <typename T>
__device__ __forceinline__ T component(int c, int L, int E, const T* __restrict__ feat_j, const T* __restrict__ freq,
                                       T dx, T dy, T g) {
    if (c < L) return feat_j[c];
    int e = c - L;
    T scale = T(1);
    if (e >= E) { e -= E; scale = g; }
    if (e == E - 1) return scale;                  // the constant 1 of the embedding
    const int k = e >> 2, axis = (e >> 1) & 1, cs = e & 1;
    T s, co;
    sincos_<T>(freq[k] * (axis ? dy : dx), &s, &co);
    return scale * (cs ? co : s);
}

// forward: out [N, L], lse [N] (log-sum-exp of the scaled scores), acc [N, L + 2E] = [fbar ; ebar]
template <typename T>
__global__ __launch_bounds__(64) void aggregate_forward_kernel(
    int64_t N, int L, int K, int F, int64_t cap, const T* __restrict__ means, const T* __restrict__ conics,
    const int32_t* __restrict__ counts, const int32_t* __restrict__ lists, const T* __restrict__ features,
    const T* __restrict__ transform, const T* __restrict__ queries, const T* __restrict__ keys,
    const T* __restrict__ freq, const T* __restrict__ dist, T* __restrict__ out, T* __restrict__ lse,
    T* __restrict__ acc_out) {
    extern __shared__ unsigned char smem_raw[];
    T* sh = (T*)smem_raw;                       // [L + 2E] the row's sums, for the final mat-vec
    const int lane = threadIdx.x;
    const int64_t i = blockIdx.x;
    const int E = 4 * F + 1, W = L + 2 * E;
    const T inv_sqrt_k = T(1) / sqrt((T)K);
    const int n = counts[i];
    const int32_t* row = lists + i * cap;
    T m = -INFINITY, l = 0;
    T acc0 = 0, acc1 = 0;                       // components lane and lane + 64 (W <= 128)
    for (int j0 = 0; j0 < n; j0 += 64) {
        // lane = neighbour: its score, offset and density
        const bool have = j0 + lane < n;
        const int64_t j = have ? row[j0 + lane] : i;
        T dx, dy;
        const T q = q_of(means, conics, i, j, &dx, &dy);
        const T g = exp_<T>(T(-0.5) * q);
        T s = 0;
        for (int k = 0; k < K; ++k) s += queries[i * K + k] * keys[j * K + k];
        s = have ? s * inv_sqrt_k : -INFINITY;
        // online softmax: new maximum, old sums rescaled
        const T mnew = fmax(m, wave_max(s));
        const T resc = exp_<T>(m - mnew);       // m = -inf on the first round: exp(-inf) = 0, the sums are 0 anyway
        const T w = have ? exp_<T>(s - mnew) : T(0);
        l = l * resc + wave_sum(w);
        acc0 *= resc; acc1 *= resc;
        m = mnew;
        // lane = component: the round's neighbours one after the other
        const int cnt = n - j0 < 64 ? n - j0 : 64;
        for (int t = 0; t < cnt; ++t) {
            const T wt = __shfl(w, t), dxt = __shfl(dx, t), dyt = __shfl(dy, t), gt = __shfl(g, t);
            const int64_t jt = __shfl((int)j, t);
            const T* fj = features + jt * L;
            if (lane < W) acc0 += wt * component<T>(lane, L, E, fj, freq, dxt, dyt, gt);
            if (lane + 64 < W) acc1 += wt * component<T>(lane + 64, L, E, fj, freq, dxt, dyt, gt);
        }
    }
    const T inv_l = T(1) / l;                   // n >= 1: every Gaussian is its own neighbour (q_ii = 0)
    acc0 *= inv_l; acc1 *= inv_l;
    if (lane < W) { sh[lane] = acc0; acc_out[i * W + lane] = acc0; }
    if (lane + 64 < W) { sh[lane + 64] = acc1; acc_out[i * W + lane + 64] = acc1; }
    if (lane == 0) lse[i] = m + log(l);
    __syncthreads();
    for (int r = lane; r < L; r += 64) {        // out_i = transform fbar + distance_transform ebar
        T o = 0;
        for (int c = 0; c < L; ++c) o += transform[r * L + c] * sh[c];
        for (int c = 0; c < 2 * E; ++c) o += dist[r * 2 * E + c] * sh[L + c];
        out[i * L + r] = o;
    }
}

// One (i, j) pair of the backward, lane = the other index: a_ij, ds_ij and what the frequency gradient needs.
template <typename T>
struct PairTerms {
    T a, ds, dx, dy, g;
};
template <typename T>
__device__ __forceinline__ PairTerms<T> pair_terms(int64_t i, int64_t j, int L, int K, int F, const T* __restrict__ means,
                                                   const T* __restrict__ conics, const T* __restrict__ features,
                                                   const T* __restrict__ queries, const T* __restrict__ keys,
                                                   const T* __restrict__ freq, const T* __restrict__ lse,
                                                   const T* __restrict__ dacc, const T* __restrict__ D) {
    PairTerms<T> p;
    const int E = 4 * F + 1, W = L + 2 * E;
    const T q = q_of(means, conics, i, j, &p.dx, &p.dy);
    p.g = exp_<T>(T(-0.5) * q);
    T s = 0;
    for (int k = 0; k < K; ++k) s += queries[i * K + k] * keys[j * K + k];
    p.a = exp_<T>(s / sqrt((T)K) - lse[i]);
    const T* di = dacc + i * W;
    T da = 0;
    for (int c = 0; c < L; ++c) da += di[c] * features[j * L + c];
    for (int k = 0; k < F; ++k) {
        T sx, cx, sy, cy;
        sincos_<T>(freq[k] * p.dx, &sx, &cx);
        sincos_<T>(freq[k] * p.dy, &sy, &cy);
        const T* d0 = di + L + 4 * k;           // plain half
        const T* d1 = d0 + E;                   // density-weighted half
        da += (d0[0] + p.g * d1[0]) * sx + (d0[1] + p.g * d1[1]) * cx + (d0[2] + p.g * d1[2]) * sy + (d0[3] + p.g * d1[3]) * cy;
    }
    da += di[L + E - 1] + p.g * di[L + 2 * E - 1];
    p.ds = p.a * (da - D[i]);
    return p;
}

// backward by rows: d queries_i [K] and the row's share of d frequencies [F].  The pair terms are
// computed once (lane = neighbour) and parked in LDS; the sums then run value by value.
template <typename T>
__global__ __launch_bounds__(64) void aggregate_backward_rows_kernel(
    int64_t N, int L, int K, int F, int64_t cap, const T* __restrict__ means, const T* __restrict__ conics,
    const int32_t* __restrict__ counts, const int32_t* __restrict__ lists, const T* __restrict__ features,
    const T* __restrict__ queries, const T* __restrict__ keys, const T* __restrict__ freq, const T* __restrict__ lse,
    const T* __restrict__ dacc, const T* __restrict__ D, T* __restrict__ g_queries, T* __restrict__ g_freq_rows) {
    extern __shared__ unsigned char smem_raw[];
    T* sh_ds = (T*)smem_raw;                    // [cap]
    T* sh_a = sh_ds + cap;                      // [cap]
    const int lane = threadIdx.x;
    const int64_t i = blockIdx.x;
    const int E = 4 * F + 1, W = L + 2 * E;
    const T inv_sqrt_k = T(1) / sqrt((T)K);
    const int n = counts[i];
    const int32_t* row = lists + i * cap;
    const T* di = dacc + i * W;
    for (int t = lane; t < n; t += 64) {
        const PairTerms<T> p = pair_terms<T>(i, row[t], L, K, F, means, conics, features, queries, keys, freq, lse, dacc, D);
        sh_ds[t] = p.ds;
        sh_a[t] = p.a;
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        T sum = 0;
        for (int t = lane; t < n; t += 64) sum += sh_ds[t] * keys[(int64_t)row[t] * K + k];
        sum = wave_sum(sum);
        if (lane == 0) g_queries[i * K + k] = sum * inv_sqrt_k;
    }
    for (int k = 0; k < F; ++k) {
        T sum = 0;
        for (int t = lane; t < n; t += 64) {
            T dx, dy;
            const T q = q_of(means, conics, i, (int64_t)row[t], &dx, &dy);
            const T g = exp_<T>(T(-0.5) * q);
            T sx, cx, sy, cy;
            sincos_<T>(freq[k] * dx, &sx, &cx);
            sincos_<T>(freq[k] * dy, &sy, &cy);
            const T* d0 = di + L + 4 * k;
            const T* d1 = d0 + E;
            // d/df of a (d_sin sin(f x) + d_cos cos(f x)) = a x (d_sin cos(f x) - d_cos sin(f x)), per axis
            sum += sh_a[t] * (dx * ((d0[0] + g * d1[0]) * cx - (d0[1] + g * d1[1]) * sx) +
                              dy * ((d0[2] + g * d1[2]) * cy - (d0[3] + g * d1[3]) * sy));
        }
        sum = wave_sum(sum);
        if (lane == 0) g_freq_rows[i * F + k] = sum;
    }
}

// backward by columns: d features_j [L] and d keys_j [K], gathered over the rows i that hold j
template <typename T>
__global__ __launch_bounds__(64) void aggregate_backward_cols_kernel(
    int64_t N, int L, int K, int F, int64_t cap, const T* __restrict__ means, const T* __restrict__ conics,
    const int32_t* __restrict__ counts, const int32_t* __restrict__ lists, const T* __restrict__ features,
    const T* __restrict__ queries, const T* __restrict__ keys, const T* __restrict__ freq, const T* __restrict__ lse,
    const T* __restrict__ dacc, const T* __restrict__ D, T* __restrict__ g_features, T* __restrict__ g_keys) {
    extern __shared__ unsigned char smem_raw[];
    T* sh_ds = (T*)smem_raw;                    // [cap]
    T* sh_a = sh_ds + cap;                      // [cap]
    const int lane = threadIdx.x;
    const int64_t j = blockIdx.x;
    const int E = 4 * F + 1, W = L + 2 * E;
    const T inv_sqrt_k = T(1) / sqrt((T)K);
    const int n = counts[j];
    const int32_t* col = lists + j * cap;
    for (int t = lane; t < n; t += 64) {
        const PairTerms<T> p = pair_terms<T>(col[t], j, L, K, F, means, conics, features, queries, keys, freq, lse, dacc, D);
        sh_ds[t] = p.ds;
        sh_a[t] = p.a;
    }
    __syncthreads();
    for (int c = 0; c < L; ++c) {
        T sum = 0;
        for (int t = lane; t < n; t += 64) sum += sh_a[t] * dacc[(int64_t)col[t] * W + c];
        sum = wave_sum(sum);
        if (lane == 0) g_features[j * L + c] = sum;
    }
    for (int k = 0; k < K; ++k) {
        T sum = 0;
        for (int t = lane; t < n; t += 64) sum += sh_ds[t] * queries[(int64_t)col[t] * K + k];
        sum = wave_sum(sum);
        if (lane == 0) g_keys[j * K + k] = sum * inv_sqrt_k;
    }
}

template <typename T>
static int aggregate_lists_t(int64_t N, int64_t cap, const void* means, const void* conics, double q_max, int32_t* row_counts,
                             int32_t* row_lists, int32_t* col_counts, int32_t* col_lists, int32_t* overflow,
                             hipStream_t stream) {
    const dim3 grid((unsigned)((N + 3) / 4)), block(256);
    clear_hip_error();
    hipLaunchKernelGGL((aggregate_lists_kernel<T, false>), grid, block, 0, stream, N, cap, (const T*)means, (const T*)conics,
                       (T)q_max, row_counts, row_lists, overflow);
    hipLaunchKernelGGL((aggregate_lists_kernel<T, true>), grid, block, 0, stream, N, cap, (const T*)means, (const T*)conics,
                       (T)q_max, col_counts, col_lists, overflow);
    return launch_status();
}

int aggregate_lists(int dtype, int64_t N, int64_t cap, const void* means, const void* conics, double q_max,
                    int32_t* row_counts, int32_t* row_lists, int32_t* col_counts, int32_t* col_lists, int32_t* overflow,
                    hipStream_t stream) {
    if (N == 0) return PIGS_OK;
    return dtype == PIGS_F32 ? aggregate_lists_t<float>(N, cap, means, conics, q_max, row_counts, row_lists, col_counts,
                                                        col_lists, overflow, stream)
                             : aggregate_lists_t<double>(N, cap, means, conics, q_max, row_counts, row_lists, col_counts,
                                                         col_lists, overflow, stream);
}

template <typename T>
static int aggregate_forward_t(const AggregateArgs& a, hipStream_t stream) {
    const int W = a.L + 2 * (4 * a.F + 1);
    clear_hip_error();
    hipLaunchKernelGGL((aggregate_forward_kernel<T>), dim3((unsigned)a.N), dim3(64), sizeof(T) * W, stream, a.N, a.L, a.K, a.F,
                       a.cap, (const T*)a.means, (const T*)a.conics, a.row_counts, a.row_lists, (const T*)a.features,
                       (const T*)a.transform, (const T*)a.queries, (const T*)a.keys, (const T*)a.frequencies,
                       (const T*)a.distance_transform, (T*)a.out, (T*)a.lse, (T*)a.acc);
    return launch_status();
}

template <typename T>
static int aggregate_backward_t(const AggregateArgs& a, hipStream_t stream) {
    clear_hip_error();
    const size_t lds = 2 * sizeof(T) * (size_t)a.cap;
    hipLaunchKernelGGL((aggregate_backward_rows_kernel<T>), dim3((unsigned)a.N), dim3(64), lds, stream, a.N, a.L, a.K, a.F, a.cap,
                       (const T*)a.means, (const T*)a.conics, a.row_counts, a.row_lists, (const T*)a.features,
                       (const T*)a.queries, (const T*)a.keys, (const T*)a.frequencies, (const T*)a.lse, (const T*)a.dacc,
                       (const T*)a.D, (T*)a.g_queries, (T*)a.g_freq_rows);
    hipLaunchKernelGGL((aggregate_backward_cols_kernel<T>), dim3((unsigned)a.N), dim3(64), lds, stream, a.N, a.L, a.K, a.F, a.cap,
                       (const T*)a.means, (const T*)a.conics, a.col_counts, a.col_lists, (const T*)a.features,
                       (const T*)a.queries, (const T*)a.keys, (const T*)a.frequencies, (const T*)a.lse, (const T*)a.dacc,
                       (const T*)a.D, (T*)a.g_features, (T*)a.g_keys);
    return launch_status();
}

int aggregate_forward(const AggregateArgs& a, hipStream_t stream) {
    if (a.N == 0) return PIGS_OK;
    return a.dtype == PIGS_F32 ? aggregate_forward_t<float>(a, stream) : aggregate_forward_t<double>(a, stream);
}

int aggregate_backward(const AggregateArgs& a, hipStream_t stream) {
    if (a.N == 0) return PIGS_OK;
    return a.dtype == PIGS_F32 ? aggregate_backward_t<float>(a, stream) : aggregate_backward_t<double>(a, stream);
}

}  // namespace pigs

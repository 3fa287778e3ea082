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
// rows i and -- for the backward's gather -- by columns j), `cap` slots per Gaussian with `cap` sized
// by a counting pass (the host reads the longest list once per preprocess_aggregate).
//
// List build: the Gaussians are binned into the multi-level grid of the sampler's plan (plan.h; built
// here on float32 copies of the centres and conics with an inflated cut-off -- it only nominates
// candidates), one wave per Gaussian walks it (grid_walk.h) around its centre (row: whose ellipses reach
// my centre?) or around its ellipse's box (column: whose centres lie inside my ellipse?) and applies the
// exact test in the caller's dtype.  List order = grid order (level, cell, rank inside the cell): the
// ranks come from atomics, so the order -- and with it the last bit of the sums -- may differ between
// two builds.
//
// All three sampling kernels share one shape, one wave per Gaussian, 64 pairs per round:
//   phase 1, lane = pair:       offsets, density, the 4F sin / cos values (ONCE per pair), score, softmax
//                               weight; what phase 2 sums is parked in the wave's LDS, one row per pair
//   phase 2, lane = component:  runs down the parked rows (one LDS read + one FMA per pair; the pair's
//                               weight arrives by v_readlane)
//   forward   [fbar ; ebar] from {features_j, sin/cos}; the two constant embedding entries are sum a = 1
//             and sum a g, reduced in phase 1
//   backward  per pair: da_ij = <dfbar_i, features_j> + <debar_i, emb_ij>, ds_ij = a_ij (da_ij - D_i) with
//             D_i = <dacc_i, acc_i> (the softmax's inner sum, known without a pass over the pairs);
//             by rows: d queries_i and the row's share of d frequencies (from the same sin / cos values);
//             by columns, as a gather (no atomics): d features_j, d keys_j.
// d transform, d distance_transform and dacc = gout [transform | distance_transform] are plain
// GEMMs and are left to the caller (pigs_amd/aggregate.py uses torch.matmul).
#include "launch.h"
#include "plan.h"
#include "grid_walk.h"

namespace pigs {

PlanView aggregate_grid_view(void* ws, int64_t N, float q_grid);      // plan.hip

template <typename T> __device__ __forceinline__ T exp_(T x);
template <> __device__ __forceinline__ float exp_<float>(float x) { return __expf(x); }
template <> __device__ __forceinline__ double exp_<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ void sincos_(T x, T* s, T* c);
// float32: the hardware's sin / cos of an argument in revolutions, reduced to [0, 1) first (v_fract_f32; the
// instructions take |r| <= 256), ~4e-7 absolute on values of order one -- the embedding's consumers are
// float32 sums of them.  sincosf's exact range reduction costs ~150 instructions a call, 12 calls a pair:
// it was most of the forward.  float64 (the reference's gradcheck dtype) keeps the exact sincos.
template <> __device__ __forceinline__ void sincos_<float>(float x, float* s, float* c) {
    const float r = __builtin_amdgcn_fractf(x * 0.15915494309189535f);
    *s = __builtin_amdgcn_sinf(r);
    *c = __builtin_amdgcn_cosf(r);
}
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
// value of lane t (wave-uniform t) in every lane, through an SGPR
__device__ __forceinline__ float lane_value(float v, int t) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
}
__device__ __forceinline__ double lane_value(double v, int t) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, t);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), t);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// n consecutive values of a row in global memory -> dst (LDS or registers' array): 16-byte loads where the
// row allows them (n a multiple of the vector width and the row start aligned: rows of L or K = 16 values)
template <typename T, typename Put>
__device__ __forceinline__ void for_row(const T* __restrict__ src, int n, Put&& put) {
    constexpr int V = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    if (n % V == 0 && ((uintptr_t)src & 15u) == 0) {
        for (int k = 0; k < n; k += V) {
            const vec_t v = *(const vec_t*)(src + k);
#pragma unroll
            for (int u = 0; u < V; ++u) put(k + u, v[u]);
        }
    } else {
        for (int k = 0; k < n; ++k) put(k, src[k]);
    }
}

// q of centre `at` under Gaussian `of`: (mu_at - mu_of)^T C_of (mu_at - mu_of)
template <typename T>
__device__ __forceinline__ T q_of(const T* means, const T* conics, int64_t at, int64_t of, T* dx, T* dy) {
    *dx = means[2 * of] - means[2 * at];          // delta = mu_j - mu_i with i = at, j = of
    *dy = means[2 * of + 1] - means[2 * at + 1];
    const T a = conics[3 * of], b = conics[3 * of + 1], c = conics[3 * of + 2];
    return a * *dx * *dx + T(2) * b * *dx * *dy + c * *dy * *dy;
}

// ------------------------------------------------------------------------------------------
// neighbour lists
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void aggregate_cast_kernel(int64_t N, const T* __restrict__ means, const T* __restrict__ conics,
                                                             float* __restrict__ means32, float* __restrict__ conics32) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    means32[2 * i] = (float)means[2 * i];
    means32[2 * i + 1] = (float)means[2 * i + 1];
#pragma unroll
    for (int k = 0; k < 3; ++k) conics32[3 * i + k] = (float)conics[3 * i + k];
}

// Few Gaussians (N <= AGG_BRUTE_MAX): every pair is tested, one wave per row, candidates 64 at a time, list
// order ascending (deterministic); no grid, no workspace traffic.  At the model's N = 1 600 this is two
// launches of ~13 us where the grid costs six.
constexpr int64_t AGG_BRUTE_MAX = 2048;
template <typename T, bool BY_COLUMN>
__global__ __launch_bounds__(256) void aggregate_lists_brute_kernel(int64_t N, int64_t cap, const T* __restrict__ means,
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
        if (in && lists && pos < cap) lists[r * cap + pos] = (int32_t)o;
        n += __builtin_popcountll(m);
    }
    if (lane == 0) {
        counts[r] = (int32_t)(lists && n > cap ? cap : n);
        if (lists && n > cap) atomicOr(overflow, 1);
    }
}

// row r = { j : q_j(mu_r) <= q_max } (BY_COLUMN: column r = { i : q_r(mu_i) <= q_max }), a slab of `cap`
// caller indices in grid order; `lists` == nullptr: count only (counts[] then holds the full lengths).
template <typename T, bool BY_COLUMN>
__global__ __launch_bounds__(256) void aggregate_lists_kernel(PlanView pv, int64_t N, int64_t cap, const T* __restrict__ means,
                                                              const T* __restrict__ conics, T q_max, float q_grid,
                                                              int32_t* __restrict__ counts, int32_t* __restrict__ lists,
                                                              int32_t* __restrict__ overflow) {
    __shared__ TravLds lds_all[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= N) return;
    const GaussGrid gg = pv.params->gg;
    const uint32_t level_mask = pv.params->level_mask;
    const uint32_t loff = pv.params->level_off[lane < PLAN_MAX_LEVELS ? lane : 0];
    const float mx = (float)means[2 * r], my = (float)means[2 * r + 1];
    float bx0, by0, bx1, by1;
    if (BY_COLUMN) {       // the box of my q <= q_grid ellipse: the centres inside my ellipse lie in it
        const T a = conics[3 * r], b = conics[3 * r + 1], c = conics[3 * r + 2];
        const T k = (T)q_grid / (a * c - b * b);
        float hx = (float)sqrt(k * c) * 1.001f, hy = (float)sqrt(k * a) * 1.001f;
        if (!(hx < 3.0e38f)) hx = 3.0e38f;      // NaN / inf (degenerate conic): everybody is a candidate
        if (!(hy < 3.0e38f)) hy = 3.0e38f;
        bx0 = mx - hx; bx1 = mx + hx; by0 = my - hy; by1 = my + hy;
    } else {               // my centre (a float32 neighbourhood of it: the grid holds float32 roundings)
        const float e = 4.0e-7f * fmaxf(fmaxf(fabsf(mx), fabsf(my)), 1.0e-30f);
        bx0 = mx - e; bx1 = mx + e; by0 = my - e; by1 = my + e;
    }
    int64_t n = 0;
    traverse(pv, gg, level_mask, loff, bx0, by0, bx1, by1, lane, lds_all[wave], true,
             [](int, uint32_t, uint32_t) {},
             [&](const float4, const float4, uint64_t mask, uint32_t j) __attribute__((always_inline)) {
        bool in = false;
        int64_t o = 0;
        if (mask >> lane & 1ull) {
            o = (int64_t)pv.g2o[j];
            T dx, dy;
            const T q = BY_COLUMN ? q_of(means, conics, o, r, &dx, &dy) : q_of(means, conics, r, o, &dx, &dy);
            in = q <= q_max;
        }
        const uint64_t m = __ballot(in);
        const int64_t pos = n + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (in && lists && pos < cap) lists[r * cap + pos] = (int32_t)o;
        n += __builtin_popcountll(m);
    });
    if (lane == 0) {
        counts[r] = (int32_t)(lists && n > cap ? cap : n);
        if (lists && n > cap) atomicOr(overflow, 1);
    }
}

// ------------------------------------------------------------------------------------------
// Sampling kernels.  A workgroup = four waves; `wpg` (1, 2 or 4, wave-uniform) of them share one Gaussian:
// wave g of a Gaussian takes the rounds g, g + wpg, ... of 64 pairs, the partial results meet in LDS and
// the Gaussian's first wave finishes.  The host picks wpg = 4 for few Gaussians (at the model's N = 1 600 a
// launch is a single generation of waves and the three rounds of ~140 neighbours run side by side instead
// of one after the other) and 1 for many (throughput: no idle waves).
// ------------------------------------------------------------------------------------------
constexpr int PART = 136;       // values of a wave's partial result: 128 components + scalars

// phase 2 of every kernel: acc += scale(t) * X[t][idx] over the cnt parked rows; `scale_of(t)` returns the
// two candidates (wave-uniform, read from lane t), `second` chooses per lane.  Four rows per iteration:
// their LDS reads are in flight together.
template <typename T, typename Scale>
__device__ __forceinline__ T run_rows(const T* X, int xs, int idx, int cnt, bool second, Scale&& scale_of) {
    T acc = 0;
    int t = 0;
    for (; t + 4 <= cnt; t += 4) {
        T x[4], sA[4], sB[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x[u] = X[(size_t)(t + u) * xs + idx];
            scale_of(t + u, &sA[u], &sB[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (second ? sB[u] : sA[u]) * x[u];
    }
    for (; t < cnt; ++t) {
        T sA, sB;
        scale_of(t, &sA, &sB);
        acc += (second ? sB : sA) * X[(size_t)t * xs + idx];
    }
    return acc;
}

struct WaveSlot {
    int lane, wave, g;          // lane; wave of the workgroup; wave's index inside its Gaussian (0 .. wpg - 1)
    int64_t i;                  // the Gaussian
    bool valid;
};
__device__ __forceinline__ WaveSlot wave_slot(int64_t N, int wpg) {
    WaveSlot w;
    w.lane = threadIdx.x & 63;
    w.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.g = w.wave % wpg;
    w.i = (int64_t)blockIdx.x * (4 / wpg) + w.wave / wpg;
    w.valid = w.i < N;
    return w;
}

// forward: out [N, L], lse [N] (log-sum-exp of the scaled scores), acc [N, L + 2E] = [fbar ; ebar]
// LDS row of a pair: [features_j (L) ; sin f_k dx, cos f_k dx, sin f_k dy, cos f_k dy (4F)], stride odd
template <typename T>
__global__ __launch_bounds__(256) void aggregate_forward_kernel(
    int64_t N, int L, int K, int F, int64_t cap, int wpg, const T* __restrict__ means, const T* __restrict__ conics,
    const int32_t* __restrict__ counts, const int32_t* __restrict__ lists, const T* __restrict__ features,
    const T* __restrict__ transform, const T* __restrict__ queries, const T* __restrict__ keys,
    const T* __restrict__ freq, const T* __restrict__ dist, T* __restrict__ out, T* __restrict__ lse,
    T* __restrict__ acc_out) {
    extern __shared__ unsigned char smem_raw[];
    const WaveSlot ws = wave_slot(N, wpg);
    const int lane = ws.lane;
    const int64_t i = ws.valid ? ws.i : 0;
    const int C4 = 4 * F, E = C4 + 1, W = L + 2 * E;
    const int WC = L + 2 * C4;                   // components that phase 2 sums (the two constants come from phase 1)
    const int xs = (L + C4) | 1;                 // row stride (odd: lanes writing their own row hit different banks)
    const int region = 64 * xs > PART ? 64 * xs : PART;      // per wave: the parked rows, later its partial result
    T* X = (T*)smem_raw + (size_t)ws.wave * region;
    const T inv_sqrt_k = T(1) / sqrt((T)K);
    const int n = ws.valid ? counts[i] : 0;
    const int32_t* row = lists + i * cap;
    const T* qi = queries + i * K;               // wave-uniform: scalar loads
    T m = -INFINITY, l = 0, lg = 0;              // running maximum, sum of weights, sum of weights x density (per lane partial)
    T acc0 = 0, acc1 = 0;                        // components lane and lane + 64
    const int c0 = lane, c1 = lane + 64;
    const int idx0 = c0 < L + C4 ? c0 : c0 - C4, idx1 = c1 < L + C4 ? c1 : c1 - C4;
    const bool dens0 = c0 >= L + C4, dens1 = c1 >= L + C4;
    for (int j0 = ws.g * 64; j0 < n; j0 += 64 * wpg) {
        const bool have = j0 + lane < n;
        const int64_t j = have ? row[j0 + lane] : i;
        T dx, dy;
        const T q = q_of(means, conics, i, j, &dx, &dy);
        const T g = exp_<T>(T(-0.5) * q);
        T s = 0;
        for_row<T>(keys + j * K, K, [&](int k, T v) { s += qi[k] * v; });
        s = have ? s * inv_sqrt_k : -INFINITY;
        const T mnew = fmax(m, wave_max(s));
        const T resc = exp_<T>(m - mnew);        // m = -inf on the first round: exp(-inf) = 0, the sums are 0 anyway
        const T w = have ? exp_<T>(s - mnew) : T(0);
        const T wg = w * g;
        l = l * resc + w;                        // per-lane partials, reduced after the last round
        lg = lg * resc + wg;
        acc0 *= resc; acc1 *= resc;
        m = mnew;
        wave_lds_fence();
        T* xr = X + (size_t)lane * xs;
        for_row<T>(features + j * L, L, [&](int c, T v) { xr[c] = v; });
        for (int k = 0; k < F; ++k) {
            T sx, cx, sy, cy;
            sincos_<T>(freq[k] * dx, &sx, &cx);
            sincos_<T>(freq[k] * dy, &sy, &cy);
            xr[L + 4 * k] = sx; xr[L + 4 * k + 1] = cx; xr[L + 4 * k + 2] = sy; xr[L + 4 * k + 3] = cy;
        }
        wave_lds_fence();
        const int cnt = n - j0 < 64 ? n - j0 : 64;
        auto scale_of = [&](int t, T* a, T* b) { *a = lane_value(w, t); *b = lane_value(wg, t); };
        if (c0 < WC) acc0 += run_rows<T>(X, xs, idx0, cnt, dens0, scale_of);
        if (WC > 64 && c1 < WC) acc1 += run_rows<T>(X, xs, idx1, cnt, dens1, scale_of);
    }
    // the wave's partial result: [acc (128) ; m ; sum w ; sum w g]
    l = wave_sum(l);
    lg = wave_sum(lg);
    wave_lds_fence();
    X[c0] = acc0; X[c1] = acc1;
    if (lane == 0) { X[128] = m; X[129] = l; X[130] = lg; }
    __syncthreads();
    if (ws.g != 0 || !ws.valid) return;
    T mstar = -INFINITY;
    for (int w = 0; w < wpg; ++w) mstar = fmax(mstar, X[(size_t)w * region + 128]);
    T lt = 0, lgt = 0;
    acc0 = 0; acc1 = 0;
    for (int w = 0; w < wpg; ++w) {
        const T* Pw = X + (size_t)w * region;
        const T f = exp_<T>(Pw[128] - mstar);    // a wave without a round: m = -inf, factor 0
        lt += f * Pw[129]; lgt += f * Pw[130];
        acc0 += f * Pw[c0]; acc1 += f * Pw[c1];
    }
    const T inv_l = T(1) / lt;                   // n >= 1: every Gaussian is its own neighbour (q_ii = 0)
    acc0 *= inv_l; acc1 *= inv_l;
    // layout of acc: [fbar (L) ; e (4F), 1 ; g e (4F), sum a g]
    wave_lds_fence();
    T* sh = X;                                   // this wave's region (its partial has been read by every lane above)
    auto put = [&](int c, T v) {
        const int o = c < L + C4 ? c : c + 1;    // the constant 1 sits between the two halves
        sh[o] = v;
        acc_out[i * W + o] = v;
    };
    if (c0 < WC) put(c0, acc0);
    if (c1 < WC) put(c1, acc1);
    if (lane == 0) {
        sh[L + E - 1] = T(1); acc_out[i * W + L + E - 1] = T(1);
        sh[L + 2 * E - 1] = lgt * inv_l; acc_out[i * W + L + 2 * E - 1] = lgt * inv_l;
        lse[i] = mstar + log(lt);
    }
    wave_lds_fence();
    for (int r = lane; r < L; r += 64) {         // out_i = transform fbar + distance_transform ebar
        T o = 0;
        for (int c = 0; c < L; ++c) o += transform[r * L + c] * sh[c];
        for (int c = 0; c < 2 * E; ++c) o += dist[r * 2 * E + c] * sh[L + c];
        out[i * L + r] = o;
    }
}

// One (i, j) pair of the backward: a_ij, ds_ij; `fterm(k, v)` receives, per frequency, the pair's share of
// the frequency gradient.  `s` = <queries_i, keys_j>, `dfeat` = <dfbar_i, features_j> (the callers have the
// rows in hand); de = the embedding part of dacc_i: [plain (4F), const ; density-weighted (4F), const].
template <typename T, typename De, typename Fterm>
__device__ __forceinline__ void pair_terms(int F, T dx, T dy, T g, T s, T dfeat, De&& de, const T* __restrict__ freq, T lse_i,
                                           T D_i, T inv_sqrt_k, T* a_out, T* ds_out, Fterm&& fterm) {
    const int E = 4 * F + 1;
    const T a = exp_<T>(s * inv_sqrt_k - lse_i);
    T da = dfeat;
    for (int k = 0; k < F; ++k) {
        T sx, cx, sy, cy;
        sincos_<T>(freq[k] * dx, &sx, &cx);
        sincos_<T>(freq[k] * dy, &sy, &cy);
        const T e0 = de(4 * k) + g * de(E + 4 * k), e1 = de(4 * k + 1) + g * de(E + 4 * k + 1);
        const T e2 = de(4 * k + 2) + g * de(E + 4 * k + 2), e3 = de(4 * k + 3) + g * de(E + 4 * k + 3);
        da += e0 * sx + e1 * cx + e2 * sy + e3 * cy;
        // d/df of (e0 sin(f x) + e1 cos(f x)) = x (e0 cos(f x) - e1 sin(f x)), per axis
        fterm(k, a * (dx * (e0 * cx - e1 * sx) + dy * (e2 * cy - e3 * sy)));
    }
    da += de(E - 1) + g * de(2 * E - 1);
    *a_out = a;
    *ds_out = a * (da - D_i);
}

// backward by rows.  Prologue: dacc_i = gout_i [transform | distance_transform]  [W] and D_i = <dacc_i, acc_i>
// (kept in the wave's LDS for the pairs; written out for the column pass by the Gaussian's first wave).
// Then d queries_i [K] and the row's share of d frequencies [F].
// LDS row of a pair: [keys_j (K) ; frequency terms (F)].
template <typename T>
__global__ __launch_bounds__(256) void aggregate_backward_rows_kernel(
    int64_t N, int L, int K, int F, int64_t cap, int wpg, const T* __restrict__ means, const T* __restrict__ conics,
    const int32_t* __restrict__ counts, const int32_t* __restrict__ lists, const T* __restrict__ features,
    const T* __restrict__ transform, const T* __restrict__ queries, const T* __restrict__ keys, const T* __restrict__ freq,
    const T* __restrict__ dist, const T* __restrict__ lse, const T* __restrict__ acc_in, const T* __restrict__ gout,
    T* __restrict__ dacc_out, T* __restrict__ D_out, T* __restrict__ g_queries, T* __restrict__ g_freq_rows) {
    extern __shared__ unsigned char smem_raw[];
    const WaveSlot ws = wave_slot(N, wpg);
    const int lane = ws.lane;
    const int64_t i = ws.valid ? ws.i : 0;
    const int E = 4 * F + 1, W = L + 2 * E;
    const int xs = (K + F) | 1;
    const int region = PART + (64 * xs > PART ? 64 * xs : PART);     // per wave: dacc_i, then the parked rows / the partial
    T* DA = (T*)smem_raw + (size_t)ws.wave * region;
    T* X = DA + PART;
    // ---- prologue, per wave (2 components per lane, L MACs each)
    T Dp = 0;
    for (int c = lane; c < W; c += 64) {
        T v = 0;
        for (int r = 0; r < L; ++r) v += gout[i * L + r] * (c < L ? transform[r * L + c] : dist[r * 2 * E + (c - L)]);
        DA[c] = v;
        if (ws.g == 0 && ws.valid) dacc_out[i * W + c] = v;
        Dp += v * acc_in[i * W + c];
    }
    const T D_i = wave_sum(Dp);
    if (lane == 0 && ws.g == 0 && ws.valid) D_out[i] = D_i;
    wave_lds_fence();
    const T inv_sqrt_k = T(1) / sqrt((T)K);
    const int n = ws.valid ? counts[i] : 0;
    const int32_t* row = lists + i * cap;
    const T* qi = queries + i * K;
    const T lse_i = lse[i];
    T acc0 = 0, acc1 = 0;                        // components lane, lane + 64 of [d queries (K) ; d frequencies (F)]
    const int WC = K + F;
    for (int j0 = ws.g * 64; j0 < n; j0 += 64 * wpg) {
        const bool have = j0 + lane < n;
        const int64_t j = have ? row[j0 + lane] : i;
        T dx, dy;
        const T q = q_of(means, conics, i, j, &dx, &dy);
        const T g = exp_<T>(T(-0.5) * q);
        wave_lds_fence();
        T* xr = X + (size_t)lane * xs;
        T s = 0, dfeat = 0;
        for_row<T>(keys + j * K, K, [&](int k, T v) { xr[k] = v; s += qi[k] * v; });
        for_row<T>(features + j * L, L, [&](int c, T v) { dfeat += DA[c] * v; });
        T a, ds;
        pair_terms<T>(F, dx, dy, g, s, dfeat, [&](int e) { return DA[L + e]; }, freq, lse_i, D_i, inv_sqrt_k, &a, &ds,
                      [&](int k, T v) { xr[K + k] = v; });
        if (!have) ds = 0;
        wave_lds_fence();
        const int cnt = n - j0 < 64 ? n - j0 : 64;
        auto scale_of = [&](int t, T* sa, T* sb) { *sa = lane_value(ds, t); *sb = T(1); };
        if (lane < WC) acc0 += run_rows<T>(X, xs, lane, cnt, lane >= K, scale_of);
        if (WC > 64 && lane + 64 < WC) acc1 += run_rows<T>(X, xs, lane + 64, cnt, lane + 64 >= K, scale_of);
    }
    wave_lds_fence();
    X[lane] = acc0; X[lane + 64] = acc1;
    __syncthreads();
    if (ws.g != 0 || !ws.valid) return;
    acc0 = 0; acc1 = 0;
    for (int w = 0; w < wpg; ++w) { acc0 += X[(size_t)w * region + lane]; acc1 += X[(size_t)w * region + lane + 64]; }
    auto put = [&](int c, T v) {
        if (c < K) g_queries[i * K + c] = v * inv_sqrt_k;
        else g_freq_rows[i * F + (c - K)] = v;
    };
    if (lane < WC) put(lane, acc0);
    if (lane + 64 < WC) put(lane + 64, acc1);
}

// backward by columns: d features_j [L] and d keys_j [K], gathered over the rows i that hold j.
// LDS row of a pair: [dfbar_i (L) ; queries_i (K)]
template <typename T>
__global__ __launch_bounds__(256) void aggregate_backward_cols_kernel(
    int64_t N, int L, int K, int F, int64_t cap, int wpg, const T* __restrict__ means, const T* __restrict__ conics,
    const int32_t* __restrict__ counts, const int32_t* __restrict__ lists, const T* __restrict__ features,
    const T* __restrict__ queries, const T* __restrict__ keys, const T* __restrict__ freq, const T* __restrict__ lse,
    const T* __restrict__ dacc, const T* __restrict__ D, T* __restrict__ g_features, T* __restrict__ g_keys) {
    extern __shared__ unsigned char smem_raw[];
    const WaveSlot ws = wave_slot(N, wpg);
    const int lane = ws.lane;
    const int64_t j = ws.valid ? ws.i : 0;
    const int E = 4 * F + 1, W = L + 2 * E;
    const int xs = (L + K) | 1;
    const int region = 64 * xs > PART ? 64 * xs : PART;
    T* X = (T*)smem_raw + (size_t)ws.wave * region;
    const T inv_sqrt_k = T(1) / sqrt((T)K);
    const int n = ws.valid ? counts[j] : 0;
    const int32_t* col = lists + j * cap;
    const T* kj = keys + j * K;                  // wave-uniform
    const T* fj = features + j * L;
    T acc0 = 0, acc1 = 0;                        // components lane, lane + 64 of [d features (L) ; d keys (K)]
    const int WC = L + K;
    for (int s0 = ws.g * 64; s0 < n; s0 += 64 * wpg) {
        const bool have = s0 + lane < n;
        const int64_t i = have ? col[s0 + lane] : j;
        T dx, dy;
        const T q = q_of(means, conics, i, j, &dx, &dy);
        const T g = exp_<T>(T(-0.5) * q);
        const T* di = dacc + i * W;
        wave_lds_fence();
        T* xr = X + (size_t)lane * xs;
        T s = 0, dfeat = 0;
        for_row<T>(di, L, [&](int c, T v) { xr[c] = v; dfeat += v * fj[c]; });
        for_row<T>(queries + i * K, K, [&](int k, T v) { xr[L + k] = v; s += v * kj[k]; });
        T a, ds;
        pair_terms<T>(F, dx, dy, g, s, dfeat, [&](int e) { return di[L + e]; }, freq, lse[i], D[i], inv_sqrt_k, &a, &ds,
                      [](int, T) {});
        if (!have) { a = 0; ds = 0; }
        wave_lds_fence();
        const int cnt = n - s0 < 64 ? n - s0 : 64;
        auto scale_of = [&](int t, T* sa, T* sb) { *sa = lane_value(a, t); *sb = lane_value(ds, t); };
        if (lane < WC) acc0 += run_rows<T>(X, xs, lane, cnt, lane >= L, scale_of);
        if (WC > 64 && lane + 64 < WC) acc1 += run_rows<T>(X, xs, lane + 64, cnt, lane + 64 >= L, scale_of);
    }
    wave_lds_fence();
    X[lane] = acc0; X[lane + 64] = acc1;
    __syncthreads();
    if (ws.g != 0 || !ws.valid) return;
    acc0 = 0; acc1 = 0;
    for (int w = 0; w < wpg; ++w) { acc0 += X[(size_t)w * region + lane]; acc1 += X[(size_t)w * region + lane + 64]; }
    auto put = [&](int c, T v) {
        if (c < L) g_features[j * L + c] = v;
        else g_keys[j * K + (c - L)] = v * inv_sqrt_k;
    };
    if (lane < WC) put(lane, acc0);
    if (lane + 64 < WC) put(lane + 64, acc1);
}

// The sums over the Gaussians: d [transform | distance_transform] = gout^T acc  [L][L + 2E] and
// d frequencies = column sums of the per-row shares.  One workgroup per (output column, split of the
// Gaussians); thread = (row r of the output, slice of the split's Gaussians), the slices meet in LDS.
// splits > 1 add their partial sums atomically into outputs zeroed by aggregate_zero_kernel.
template <typename T>
__global__ __launch_bounds__(256) void aggregate_outer_kernel(int64_t N, int L, int F, int splits, const T* __restrict__ gout,
                                                              const T* __restrict__ acc, const T* __restrict__ g_freq_rows,
                                                              T* __restrict__ g_transform, T* __restrict__ g_dist,
                                                              T* __restrict__ g_freq) {
    __shared__ T sh[256];
    const int E = 4 * F + 1, W = L + 2 * E;
    const int c = blockIdx.x / splits, sp = blockIdx.x % splits;
    const int64_t per = (N + splits - 1) / splits;
    const int64_t i0 = sp * per, i1 = i0 + per < N ? i0 + per : N;
    const int R = c < W ? L : 1;                 // a column of gout^T acc has L entries, a frequency one
    int Rp = 1;
    while (Rp < R) Rp <<= 1;                     // L <= 126: Rp <= 128
    const int S = 256 / Rp;                      // slices of the Gaussians
    const int r = threadIdx.x % Rp, sl = threadIdx.x / Rp;
    T v = 0;
    if (r < R) {
        // a latency-bound loop (few workgroups, strided loads): eight iterations' loads in flight together
        if (c < W) {
#pragma unroll 8
            for (int64_t i = i0 + sl; i < i1; i += S) v += gout[i * L + r] * acc[i * W + c];
        } else {
#pragma unroll 8
            for (int64_t i = i0 + sl; i < i1; i += S) v += g_freq_rows[i * F + (c - W)];
        }
    }
    sh[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < R) {
        T tot = 0;
        for (int k = 0; k < S; ++k) tot += sh[k * Rp + threadIdx.x];
        T* dst = c >= W ? g_freq + (c - W) : c < L ? g_transform + threadIdx.x * L + c : g_dist + threadIdx.x * 2 * E + (c - L);
        if (splits == 1) *dst = tot;
        else atomicAdd(dst, tot);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void aggregate_zero_kernel(T* a, int64_t na, T* b, int64_t nb, T* c, int64_t nc) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < na) a[t] = 0;
    if (t < nb) b[t] = 0;
    if (t < nc) c[t] = 0;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static size_t cast_bytes(int dtype, int64_t N) {       // float32 copies of centres and conics (float64 callers)
    return dtype == PIGS_F64 ? align_up(sizeof(float) * 5 * (size_t)N, 256) : 0;
}

size_t aggregate_workspace_bytes(int dtype, int64_t N) {
    if (N <= AGG_BRUTE_MAX) return 256;          // every pair is tested: no grid
    const size_t g = aggregate_grid_bytes(N);
    return g == 0 ? 0 : cast_bytes(dtype, N) + g;
}

// The grid's cut-off only nominates candidates; it must cover q <= q_max however the float32 grid
// arithmetic rounds: a generous margin costs a few candidates, nothing else.
static float grid_cutoff(int dtype, double q_max) { return (float)(q_max * (dtype == PIGS_F64 ? 1.05 : 1.02) + 1e-3); }

template <typename T>
static int aggregate_lists_t(int dtype, int64_t N, int64_t cap, const void* means, const void* conics, double q_max,
                             void* workspace, size_t workspace_bytes, int flags, int32_t* row_counts, int32_t* row_lists,
                             int32_t* col_counts, int32_t* col_lists, int32_t* overflow, hipStream_t stream) {
    const dim3 grid((unsigned)((N + 3) / 4)), block(256);
    if (N <= AGG_BRUTE_MAX) {
        clear_hip_error();
        hipLaunchKernelGGL((aggregate_lists_brute_kernel<T, false>), grid, block, 0, stream, N, cap, (const T*)means,
                           (const T*)conics, (T)q_max, row_counts, row_lists, overflow);
        hipLaunchKernelGGL((aggregate_lists_brute_kernel<T, true>), grid, block, 0, stream, N, cap, (const T*)means,
                           (const T*)conics, (T)q_max, col_counts, col_lists, overflow);
        return launch_status();
    }
    const size_t need = aggregate_workspace_bytes(dtype, N);
    if (need == 0) return PIGS_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < need) return PIGS_ERR_WORKSPACE;
    const float q_grid = grid_cutoff(dtype, q_max);
    char* grid_ws = (char*)workspace + cast_bytes(dtype, N);
    const size_t grid_bytes = workspace_bytes - cast_bytes(dtype, N);
    if (flags & PIGS_AGGREGATE_BUILD_GRID) {
        const float *m32 = (const float*)means, *c32 = (const float*)conics;
        if (dtype == PIGS_F64) {
            float* mm = (float*)workspace;
            float* cc = mm + 2 * N;
            clear_hip_error();
            hipLaunchKernelGGL((aggregate_cast_kernel<T>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, N,
                               (const T*)means, (const T*)conics, mm, cc);
            const int rc = launch_status();
            if (rc != PIGS_OK) return rc;
            m32 = mm; c32 = cc;
        }
        const int rc = aggregate_grid_build(grid_ws, grid_bytes, N, q_grid, m32, c32, stream);
        if (rc != PIGS_OK) return rc;
    }
    const PlanView pv = aggregate_grid_view(grid_ws, N, q_grid);
    clear_hip_error();
    hipLaunchKernelGGL((aggregate_lists_kernel<T, false>), grid, block, 0, stream, pv, N, cap, (const T*)means, (const T*)conics,
                       (T)q_max, q_grid, row_counts, row_lists, overflow);
    hipLaunchKernelGGL((aggregate_lists_kernel<T, true>), grid, block, 0, stream, pv, N, cap, (const T*)means, (const T*)conics,
                       (T)q_max, q_grid, col_counts, col_lists, overflow);
    return launch_status();
}

int aggregate_lists(int dtype, int64_t N, int64_t cap, const void* means, const void* conics, double q_max, void* workspace,
                    size_t workspace_bytes, int flags, int32_t* row_counts, int32_t* row_lists, int32_t* col_counts,
                    int32_t* col_lists, int32_t* overflow, hipStream_t stream) {
    if (N == 0) return PIGS_OK;
    return dtype == PIGS_F32 ? aggregate_lists_t<float>(dtype, N, cap, means, conics, q_max, workspace, workspace_bytes, flags,
                                                        row_counts, row_lists, col_counts, col_lists, overflow, stream)
                             : aggregate_lists_t<double>(dtype, N, cap, means, conics, q_max, workspace, workspace_bytes, flags,
                                                         row_counts, row_lists, col_counts, col_lists, overflow, stream);
}

// LDS of a sampling kernel: four wave regions of `region` values; beyond 64 KB the launch has to ask for it.
template <typename Kernel>
static size_t sampling_lds(Kernel kernel, size_t elem, size_t region) {
    const size_t bytes = elem * 4 * region;
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return bytes;
}
static size_t rows_region(int stride) { return (size_t)(64 * stride > PART ? 64 * stride : PART); }
// waves per Gaussian: a launch of few Gaussians is one generation of waves bound by a wave's serial life
// (split every Gaussian's rounds over four waves); many Gaussians are a throughput problem (no idle waves)
static int waves_per_gaussian(int64_t N) { return N <= 4096 ? 4 : N <= 8192 ? 2 : 1; }

template <typename T>
static int aggregate_forward_t(const AggregateArgs& a, hipStream_t stream) {
    const size_t lds = sampling_lds(aggregate_forward_kernel<T>, sizeof(T), rows_region((a.L + 4 * a.F) | 1));
    const int wpg = waves_per_gaussian(a.N), gpw = 4 / wpg;
    clear_hip_error();
    hipLaunchKernelGGL((aggregate_forward_kernel<T>), dim3((unsigned)((a.N + gpw - 1) / gpw)), dim3(256), lds, stream, a.N, a.L, a.K,
                       a.F, a.cap, wpg, (const T*)a.means, (const T*)a.conics, a.row_counts, a.row_lists, (const T*)a.features,
                       (const T*)a.transform, (const T*)a.queries, (const T*)a.keys, (const T*)a.frequencies,
                       (const T*)a.distance_transform, (T*)a.out, (T*)a.lse, (T*)a.acc);
    return launch_status();
}

size_t aggregate_backward_scratch_bytes(int dtype, int64_t N, int L, int F) {
    const size_t e = dtype == PIGS_F64 ? 8 : 4;
    const int W = L + 2 * (4 * F + 1);
    return align_up(e * (size_t)N * (size_t)(W + 1 + F), 256);      // dacc [N][W], D [N], per-row d frequencies [N][F]
}

template <typename T>
static int aggregate_backward_t(const AggregateArgs& a, hipStream_t stream) {
    const int E = 4 * a.F + 1, W = a.L + 2 * E;
    T* dacc = (T*)a.scratch;
    T* D = dacc + (size_t)a.N * W;
    T* gfr = D + a.N;
    const size_t lds_r = sampling_lds(aggregate_backward_rows_kernel<T>, sizeof(T), PART + rows_region((a.K + a.F) | 1));
    const size_t lds_c = sampling_lds(aggregate_backward_cols_kernel<T>, sizeof(T), rows_region((a.L + a.K) | 1));
    const int wpg = waves_per_gaussian(a.N), gpw = 4 / wpg;
    const dim3 grid((unsigned)((a.N + gpw - 1) / gpw)), block(256);
    // the sums over N: one split per ~2048 Gaussians (a single split is a plain, deterministic sum)
    int splits = (int)((a.N + 2047) / 2048);
    splits = splits < 1 ? 1 : splits > 64 ? 64 : splits;
    clear_hip_error();
    if (splits > 1) {
        const int64_t mx = (int64_t)a.L * (2 * E > a.L ? 2 * E : a.L);
        hipLaunchKernelGGL((aggregate_zero_kernel<T>), dim3((unsigned)((mx + 255) / 256)), dim3(256), 0, stream, (T*)a.g_transform,
                           (int64_t)a.L * a.L, (T*)a.g_distance_transform, (int64_t)a.L * 2 * E, (T*)a.g_frequencies, (int64_t)a.F);
    }
    hipLaunchKernelGGL((aggregate_backward_rows_kernel<T>), grid, block, lds_r, stream, a.N, a.L, a.K, a.F,
                       a.cap, wpg, (const T*)a.means, (const T*)a.conics, a.row_counts, a.row_lists, (const T*)a.features,
                       (const T*)a.transform, (const T*)a.queries, (const T*)a.keys, (const T*)a.frequencies,
                       (const T*)a.distance_transform, (const T*)a.lse, (const T*)a.acc, (const T*)a.gout, dacc, D,
                       (T*)a.g_queries, gfr);
    hipLaunchKernelGGL((aggregate_backward_cols_kernel<T>), grid, block, lds_c, stream, a.N, a.L, a.K, a.F,
                       a.cap, wpg, (const T*)a.means, (const T*)a.conics, a.col_counts, a.col_lists, (const T*)a.features,
                       (const T*)a.queries, (const T*)a.keys, (const T*)a.frequencies, (const T*)a.lse, (const T*)dacc,
                       (const T*)D, (T*)a.g_features, (T*)a.g_keys);
    hipLaunchKernelGGL((aggregate_outer_kernel<T>), dim3((unsigned)((W + a.F) * splits)), dim3(256), 0, stream, a.N, a.L, a.F, splits,
                       (const T*)a.gout, (const T*)a.acc, (const T*)gfr, (T*)a.g_transform, (T*)a.g_distance_transform,
                       (T*)a.g_frequencies);
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

// Binned (culled) sampler: preprocess (plan build) + forward + backward, float32, d = 2.
// Data structures and the cut-off rule: plan.h.  Per-pair arithmetic: pair_math.h.
//
// Sampling kernels: one wave = one sample cell (<= 64 points per pass, lane = point).  The wave
// (1) reduces the bounding box of its points, (2) walks, per occupied Gaussian level, the few
// contiguous record ranges whose cells lie within one cell of that box -- 64 candidates per
// step, one per lane, coalesced 2 x 16 B loads -- and tests each candidate's q <= q_max ellipse
// against the box exactly, (3) for every accepted candidate (a bit of the 64-bit ballot) pulls
// the 32-byte record through the SCALAR data path (wave-uniform address) and evaluates it on
// all 64 points from SGPR operands.  No LDS, no barriers; HBM traffic is the point stream
// (perm + coordinates in, outputs out) plus record re-reads that hit L2.
#include "pair_math.h"
#include "plan.h"
#include "launch.h"

namespace pigs {

// ------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes, all active)
// ------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
// DPP controls: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140, row_bcast15 = 0x142, row_bcast31 = 0x143.
__device__ __forceinline__ float wave_sum(float v) {   // result valid in lane 63
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    v += dpp_f32<0x142, 0xA>(v);
    v += dpp_f32<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ float wave_sum_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum(v)), 63));
}
__device__ __forceinline__ float wave_min_bcast(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float wave_max_bcast(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float uniform_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// ------------------------------------------------------------------------------------------
// preprocess kernels
// ------------------------------------------------------------------------------------------
struct BuildArgs {
    PlanHeader* header;
    uint32_t* starts;     // counts, then (in place) their exclusive scan
    uint32_t* cursor;
    uint32_t* blocksum;
    uint32_t* gkey;
    uint32_t* skey;
    float4* rec;
    uint32_t* g2o;
    uint32_t* perm;
    const float* means;
    const float* conics;
    const float* values;
    const float* samples;
    uint32_t N, M;
    int c, G0, L;
    uint32_t gcells, scells_cap, ncounts;
    uint32_t level_off[PLAN_MAX_LEVELS + 1];
    float q_max;
};

__global__ __launch_bounds__(256) void plan_init_kernel(BuildArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i <= a.ncounts) a.starts[i] = 0;
    if (i == 0) {
        a.header->gbox[0] = a.header->gbox[1] = a.header->sbox[0] = a.header->sbox[1] = 0x7fffffff;
        a.header->gbox[2] = a.header->gbox[3] = a.header->sbox[2] = a.header->sbox[3] = (int32_t)0x80000000;
        a.header->level_mask = 0;
    }
}

__device__ __forceinline__ void bbox_accumulate(const float2* __restrict__ pts, uint32_t n, int32_t* box) {
    const float INF = __builtin_huge_valf();
    float x0 = INF, y0 = INF, x1 = -INF, y1 = -INF;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float2 p = pts[i];
        if (fabsf(p.x) < INF) { x0 = fminf(x0, p.x); x1 = fmaxf(x1, p.x); }
        if (fabsf(p.y) < INF) { y0 = fminf(y0, p.y); y1 = fmaxf(y1, p.y); }
    }
    x0 = wave_min_bcast(x0); y0 = wave_min_bcast(y0);
    x1 = wave_max_bcast(x1); y1 = wave_max_bcast(y1);
    if ((threadIdx.x & 63) == 0) {
        if (x0 <= x1) { atomicMin(&box[0], float_to_ordered(x0)); atomicMax(&box[2], float_to_ordered(x1)); }
        if (y0 <= y1) { atomicMin(&box[1], float_to_ordered(y0)); atomicMax(&box[3], float_to_ordered(y1)); }
    }
}

__global__ __launch_bounds__(256) void plan_bbox_kernel(BuildArgs a) {
    bbox_accumulate((const float2*)a.means, a.N, a.header->gbox);
    bbox_accumulate((const float2*)a.samples, a.M, a.header->sbox);
}

__global__ __launch_bounds__(256) void plan_count_kernel(BuildArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.N) {
        const GaussGrid g = gauss_grid(a.header, a.G0);
        const float mx = a.means[2 * i], my = a.means[2 * i + 1];
        const float ca = a.conics[3 * i], cb = a.conics[3 * i + 1], cc = a.conics[3 * i + 2];
        // half extents of the q <= q_max ellipse: sqrt(q_max * Sigma_xx), Sigma = C^-1
        const float det = ca * cc - cb * cb;
        const float R = sqrtf(a.q_max * fmaxf(ca, cc) / det);   // NaN / inf (degenerate conic) -> top level
        int l = 0;
        float s = g.s0;
        while (l < a.L - 1 && !(R <= s)) { ++l; s *= 2.f; }
        const int G = a.G0 >> l;
        const float inv_s = 1.f / s;
        const int cx = (int)clampf((mx - g.ox) * inv_s, 0.f, (float)(G - 1));   // NaN -> 0
        const int cy = (int)clampf((my - g.oy) * inv_s, 0.f, (float)(G - 1));
        const uint32_t key = a.level_off[l] + (uint32_t)(cy * G + cx);
        a.gkey[i] = key;
        atomicAdd(&a.starts[key], 1u);
        if (!(*(volatile uint32_t*)&a.header->level_mask >> l & 1u)) atomicOr(&a.header->level_mask, 1u << l);
    }
    if (i < a.M) {
        const SampleGrid sg = sample_grid(a.header, a.M, a.scells_cap);
        const float x = a.samples[2 * i], y = a.samples[2 * i + 1];
        const int cx = (int)clampf((x - sg.ox) * sg.inv_w, 0.f, (float)(sg.nx - 1));
        const int cy = (int)clampf((y - sg.oy) * sg.inv_w, 0.f, (float)(sg.ny - 1));
        const uint32_t id = sample_cell_id(cx, cy, sg.nx);
        a.skey[i] = id;
        atomicAdd(&a.starts[a.gcells + id], 1u);
    }
}

// exclusive scan of starts[0 .. ncounts) in place, 4096 elements per block, two launches
constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_PER_BLOCK = 256 * SCAN_PER_THREAD;

__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* sh) {
    // returns the block total in every thread
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void plan_scan_blocksum_kernel(BuildArgs a) {
    __shared__ uint32_t sh[4];
    const uint32_t base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) s += (base + k < a.ncounts) ? a.starts[base + k] : 0u;
    const uint32_t t = block_sum_256(s, sh);
    if (threadIdx.x == 0) a.blocksum[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void plan_scan_apply_kernel(BuildArgs a) {
    __shared__ uint32_t sh[4];
    __shared__ uint32_t wave_tot[4];
    // offset of this block = sum of the preceding block sums
    uint32_t pre = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256) pre += a.blocksum[b];
    const uint32_t offset = block_sum_256(pre, sh);
    const uint32_t base = blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    uint32_t v[SCAN_PER_THREAD];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        v[k] = (base + k < a.ncounts) ? a.starts[base + k] : 0u;
        s += v[k];
    }
    // exclusive scan of the per-thread sums across the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t run = offset + inc - s;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; ++k) {
        if (base + k < a.ncounts) {
            a.starts[base + k] = run;
            a.cursor[base + k] = run;
        }
        run += v[k];
    }
    // total = N + M: every Gaussian and every point was counted exactly once
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.starts[a.ncounts] = a.N + a.M;
}

__global__ __launch_bounds__(256) void plan_scatter_kernel(BuildArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.N) {
        const uint32_t pos = atomicAdd(&a.cursor[a.gkey[i]], 1u);
        float v[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < a.c; ++k) v[k] = a.values[(size_t)i * a.c + k];
        a.rec[2 * pos] = make_float4(a.means[2 * i], a.means[2 * i + 1], a.conics[3 * i], a.conics[3 * i + 1]);
        a.rec[2 * pos + 1] = make_float4(a.conics[3 * i + 2], v[0], v[1], v[2]);
        a.g2o[pos] = i;
    }
    if (i < a.M) {
        const uint32_t pos = atomicAdd(&a.cursor[a.gcells + a.skey[i]], 1u) - a.N;
        a.perm[pos] = i;
    }
}

// ------------------------------------------------------------------------------------------
// candidate test: does the ellipse q <= q_max of a Gaussian reach the rectangle [x0,x1]x[y0,y1]?
// q is convex with its minimum at the centre, so the minimum over the rectangle lies on the
// edge(s) facing the centre; each facing edge is minimised in closed form.  Comparisons are
// written so that NaN (degenerate conic) accepts.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ellipse_reaches_rect(float4 A, float cc, float x0, float y0, float x1, float y1,
                                                     float q_max) {
    const float l = x0 - A.x, r = x1 - A.x, bt = y0 - A.y, tp = y1 - A.y;
    const float a = A.z, b = A.w;
    const float xe = clampf(0.f, l, r), ye = clampf(0.f, bt, tp);
    const float ys = clampf(-b * xe * __builtin_amdgcn_rcpf(cc), bt, tp);
    const float xs = clampf(-b * ye * __builtin_amdgcn_rcpf(a), l, r);
    const float q1 = a * xe * xe + (2.f * b * xe + cc * ys) * ys;
    const float q2 = cc * ye * ye + (2.f * b * ye + a * xs) * xs;
    return !(fminf(q1, q2) > q_max);
}

// ------------------------------------------------------------------------------------------
// the traversal shared by forward and backward: calls `visit(j)` (j = sorted Gaussian index,
// wave-uniform) for every Gaussian whose ellipse reaches the box, and `batch_begin(j0)` /
// `batch_end(j0, mask)` around each step of 64 candidates.
// ------------------------------------------------------------------------------------------
template <typename BatchBegin, typename Visit, typename BatchEnd>
__device__ __forceinline__ void for_each_reaching_gaussian(const PlanView& pv, const GaussGrid& gg, uint32_t level_mask,
                                                           float bx0, float by0, float bx1, float by1, int lane,
                                                           BatchBegin&& batch_begin, Visit&& visit,
                                                           BatchEnd&& batch_end) {
    for (int l = 0; l < pv.L; ++l) {
        if (!(level_mask >> l & 1u)) continue;
        const int G = pv.G0 >> l;
        const float inv_s = gg.inv_s0 * (1.f / (float)(1 << l));
        const float gmax = (float)(G - 1);
        // cells within one cell of the box (R <= s_l on this level); the top level has one cell
        const int cx0 = (int)clampf(floorf((bx0 - gg.ox) * inv_s) - 1.f, 0.f, gmax);
        const int cx1 = (int)clampf(floorf((bx1 - gg.ox) * inv_s) + 1.f, 0.f, gmax);
        const int cy0 = (int)clampf(floorf((by0 - gg.oy) * inv_s) - 1.f, 0.f, gmax);
        const int cy1 = (int)clampf(floorf((by1 - gg.oy) * inv_s) + 1.f, 0.f, gmax);
        for (int cy = cy0; cy <= cy1; ++cy) {
            const uint32_t row = pv.level_off[l] + (uint32_t)(cy * G);
            const uint32_t jb = pv.starts[row + cx0];
            const uint32_t je = pv.starts[row + cx1 + 1];
            for (uint32_t j0 = jb; j0 < je; j0 += 64) {
                const uint32_t j = j0 + lane;
                bool ok = j < je;
                const uint32_t jj = ok ? j : jb;
                const float4 A = pv.rec[2 * jj];
                const float4 B = pv.rec[2 * jj + 1];
                ok = ok && ellipse_reaches_rect(A, B.x, bx0, by0, bx1, by1, pv.q_max);
                uint64_t mask = __ballot(ok);
                batch_begin(j0);
                uint64_t rest = mask;
                while (rest) {
                    const int b = __builtin_ctzll(rest);
                    rest &= rest - 1;
                    visit(j0 + (uint32_t)b, b);
                }
                batch_end(j0, mask);
            }
        }
    }
}

// scalar-path record fetch (j is wave-uniform)
struct Rec {
    float mu[2], con[3], v[3];
};
__device__ __forceinline__ Rec load_rec(const float4* __restrict__ rec, uint32_t j) {
    const float4 A = rec[2 * j], B = rec[2 * j + 1];
    Rec r;
    r.mu[0] = A.x; r.mu[1] = A.y; r.con[0] = A.z; r.con[1] = A.w; r.con[2] = B.x;
    r.v[0] = B.y; r.v[1] = B.z; r.v[2] = B.w;
    return r;
}

template <int C, int MASK>
__global__ __launch_bounds__(256) void binned_forward_kernel(PlanView pv, const float* __restrict__ samples,
                                                             float* __restrict__ o0, float* __restrict__ o1,
                                                             float* __restrict__ o2, float* __restrict__ o3) {
    using L = FwdLayout<2, C, MASK>;
    const int lane = threadIdx.x & 63;
    const uint32_t cell = blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t sbeg = pv.starts[pv.gcells + cell] - pv.N;
    const uint32_t send = pv.starts[pv.gcells + cell + 1] - pv.N;
    if (sbeg >= send) return;
    const GaussGrid gg = gauss_grid(pv.header, pv.G0);
    const uint32_t level_mask = pv.header->level_mask;
    const float INF = __builtin_huge_valf();

    for (uint32_t base = sbeg; base < send; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < send;
        const uint32_t m = valid ? pv.perm[i] : 0u;
        float s[2] = {0.f, 0.f};
        if (valid) {
            const float2 p = ((const float2*)samples)[m];
            s[0] = p.x; s[1] = p.y;
        }
        const float bx0 = wave_min_bcast(valid ? s[0] : INF), bx1 = wave_max_bcast(valid ? s[0] : -INF);
        const float by0 = wave_min_bcast(valid ? s[1] : INF), by1 = wave_max_bcast(valid ? s[1] : -INF);
        if (!valid) { s[0] = bx0; s[1] = by0; }

        float acc[L::N];
#pragma unroll
        for (int k = 0; k < L::N; ++k) acc[k] = 0.f;

        for_each_reaching_gaussian(
            pv, gg, level_mask, bx0, by0, bx1, by1, lane, [](uint32_t) {},
            [&](uint32_t j, int) {
                const Rec r = load_rec(pv.rec, j);
                fwd_accumulate<float, 2, C, MASK>(acc, s, r.mu, r.con, r.v);
            },
            [](uint32_t, uint64_t) {});

        if (valid) fwd_store<float, 2, C, MASK>(acc, (int64_t)m, o0, o1, o2, o3);
    }
}

// Backward: same traversal.  For every accepted Gaussian the per-point contributions are summed
// over the 64 lanes (DPP) and parked in the lane that tested that candidate; after the step
// those lanes add their 5+c sums to the sorted-order scratch gacc[k][j] (consecutive lanes ->
// consecutive addresses).  plan_unpermute_kernel then writes the caller's gradient layout.
template <int C, int MASK>
__global__ __launch_bounds__(256) void binned_backward_kernel(PlanView pv, const float* __restrict__ samples,
                                                              const float* __restrict__ G0p,
                                                              const float* __restrict__ G1p,
                                                              const float* __restrict__ G2p,
                                                              const float* __restrict__ G3p) {
    using BL = BwdLayout<2, C>;
    const int lane = threadIdx.x & 63;
    const uint32_t cell = blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t sbeg = pv.starts[pv.gcells + cell] - pv.N;
    const uint32_t send = pv.starts[pv.gcells + cell + 1] - pv.N;
    if (sbeg >= send) return;
    const GaussGrid gg = gauss_grid(pv.header, pv.G0);
    const uint32_t level_mask = pv.header->level_mask;
    const float INF = __builtin_huge_valf();

    for (uint32_t base = sbeg; base < send; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < send;
        const uint32_t m = valid ? pv.perm[i] : 0u;
        float s[2] = {0.f, 0.f};
        if (valid) {
            const float2 p = ((const float2*)samples)[m];
            s[0] = p.x; s[1] = p.y;
        }
        const float bx0 = wave_min_bcast(valid ? s[0] : INF), bx1 = wave_max_bcast(valid ? s[0] : -INF);
        const float by0 = wave_min_bcast(valid ? s[1] : INF), by1 = wave_max_bcast(valid ? s[1] : -INF);
        if (!valid) { s[0] = bx0; s[1] = by0; }

        Gsym<float, 2, C, MASK> G;
        G.load((int64_t)m, G0p, G1p, G2p, G3p);
        if (!valid) {   // lanes without a point contribute nothing
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                G.g0[ch] = 0.f;
                G.g1[0][ch] = G.g1[1][ch] = 0.f;
                G.g2[0][ch] = G.g2[1][ch] = G.g2[2][ch] = 0.f;
                G.g3[0][ch] = G.g3[1][ch] = G.g3[2][ch] = G.g3[3][ch] = 0.f;
            }
        }

        float mine[BL::N];
        for_each_reaching_gaussian(
            pv, gg, level_mask, bx0, by0, bx1, by1, lane,
            [&](uint32_t) {
#pragma unroll
                for (int k = 0; k < BL::N; ++k) mine[k] = 0.f;
            },
            [&](uint32_t j, int b) {
                const Rec r = load_rec(pv.rec, j);
                float part[BL::N];
#pragma unroll
                for (int k = 0; k < BL::N; ++k) part[k] = 0.f;
                bwd_accumulate<float, 2, C, MASK>(part, s, r.mu, r.con, r.v, G);
#pragma unroll
                for (int k = 0; k < BL::N; ++k) {
                    const float tot = wave_sum_bcast(part[k]);
                    mine[k] = (lane == b) ? tot : mine[k];
                }
            },
            [&](uint32_t j0, uint64_t mask) {
                if (mask >> lane & 1ull) {
#pragma unroll
                    for (int k = 0; k < BL::N; ++k) atomicAdd(&pv.gacc[(size_t)k * pv.N + j0 + lane], mine[k]);
                }
            });
    }
}

template <int C>
__global__ __launch_bounds__(256) void plan_unpermute_kernel(PlanView pv, float* __restrict__ g_means,
                                                             float* __restrict__ g_conics,
                                                             float* __restrict__ g_values) {
    using BL = BwdLayout<2, C>;
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= pv.N) return;
    const uint32_t n = pv.g2o[j];
    g_means[2 * n] = pv.gacc[(size_t)(BL::MU + 0) * pv.N + j];
    g_means[2 * n + 1] = pv.gacc[(size_t)(BL::MU + 1) * pv.N + j];
#pragma unroll
    for (int k = 0; k < 3; ++k) g_conics[3 * n + k] = pv.gacc[(size_t)(BL::CON + k) * pv.N + j];
#pragma unroll
    for (int k = 0; k < C; ++k) g_values[(size_t)C * n + k] = pv.gacc[(size_t)(BL::VAL + k) * pv.N + j];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static PlanView make_view(const PlanLayout& p, void* ws, float q_max) {
    char* b = (char*)ws;
    PlanView v{};
    v.header = (const PlanHeader*)(b + p.off_header);
    v.starts = (const uint32_t*)(b + p.off_starts);
    v.rec = (const float4*)(b + p.off_rec);
    v.g2o = (const uint32_t*)(b + p.off_g2o);
    v.perm = (const uint32_t*)(b + p.off_perm);
    v.N = (uint32_t)p.N; v.M = (uint32_t)p.M;
    v.G0 = p.G0; v.L = p.L;
    v.gcells = p.gcells; v.scells_cap = p.scells_cap;
    for (int l = 0; l <= PLAN_MAX_LEVELS; ++l) v.level_off[l] = p.level_off[l];
    v.q_max = q_max;
    v.gacc = (float*)(b + p.off_gacc);
    return v;
}

static bool plan_supported(int64_t N, int64_t M, int c) {
    return N >= 1 && M >= 1 && c >= 1 && c <= 3 && N < (1LL << 30) && M < (1LL << 31) - 64 &&
           N + M < (1LL << 32) - 1;
}

size_t plan_workspace_bytes(int64_t N, int64_t M, int c) {
    if (!plan_supported(N, M, c)) return 0;
    return make_plan_layout(N, M, c).total_bytes;
}

int plan_build(void* ws, size_t ws_bytes, int64_t N, int64_t M, int c, float q_max, const void* means,
               const void* conics, const void* values, const void* samples, hipStream_t stream) {
    if (!plan_supported(N, M, c)) return PIGS_ERR_UNSUPPORTED;
    if (!(q_max > 0.f)) return PIGS_ERR_INVALID;
    const PlanLayout p = make_plan_layout(N, M, c);
    if (!ws || ws_bytes < p.total_bytes) return PIGS_ERR_WORKSPACE;
    char* b = (char*)ws;
    BuildArgs a{};
    a.header = (PlanHeader*)(b + p.off_header);
    a.starts = (uint32_t*)(b + p.off_starts);
    a.cursor = (uint32_t*)(b + p.off_cursor);
    a.blocksum = (uint32_t*)(b + p.off_blocksum);
    a.gkey = (uint32_t*)(b + p.off_gkey);
    a.skey = (uint32_t*)(b + p.off_skey);
    a.rec = (float4*)(b + p.off_rec);
    a.g2o = (uint32_t*)(b + p.off_g2o);
    a.perm = (uint32_t*)(b + p.off_perm);
    a.means = (const float*)means; a.conics = (const float*)conics;
    a.values = (const float*)values; a.samples = (const float*)samples;
    a.N = (uint32_t)N; a.M = (uint32_t)M; a.c = c; a.G0 = p.G0; a.L = p.L;
    a.gcells = p.gcells; a.scells_cap = p.scells_cap; a.ncounts = p.ncounts;
    for (int l = 0; l <= PLAN_MAX_LEVELS; ++l) a.level_off[l] = p.level_off[l];
    a.q_max = q_max;

    clear_hip_error();
    const uint32_t nmax = (uint32_t)(N > M ? N : M);
    const uint32_t scan_blocks = (p.ncounts + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    hipLaunchKernelGGL(plan_init_kernel, dim3(p.ncounts / 256 + 1), dim3(256), 0, stream, a);
    uint32_t bbox_blocks = (nmax + 255) / 256;
    if (bbox_blocks > 1024) bbox_blocks = 1024;
    hipLaunchKernelGGL(plan_bbox_kernel, dim3(bbox_blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(plan_count_kernel, dim3((nmax + 255) / 256), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(plan_scan_blocksum_kernel, dim3(scan_blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(plan_scan_apply_kernel, dim3(scan_blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(plan_scatter_kernel, dim3((nmax + 255) / 256), dim3(256), 0, stream, a);
    return launch_status();
}

template <int C>
static int plan_forward_c(const PlanView& pv, int mask, const float* samples, float* const* out, hipStream_t stream) {
    const dim3 grid(pv.scells_cap / 4), block(256);
    clear_hip_error();
#define PIGS_CASE(MK)                                                                                        \
    case MK:                                                                                                 \
        hipLaunchKernelGGL((binned_forward_kernel<C, MK>), grid, block, 0, stream, pv, samples, out[0], out[1], \
                           out[2], out[3]);                                                                  \
        break;
    switch (mask) {
        PIGS_CASE(1) PIGS_CASE(2) PIGS_CASE(4) PIGS_CASE(8) PIGS_CASE(7) PIGS_CASE(15)
        default: return PIGS_ERR_UNSUPPORTED;
    }
#undef PIGS_CASE
    return launch_status();
}

template <int C>
static int plan_backward_c(const PlanView& pv, int mask, const float* samples, const float* const* g, float* gm,
                           float* gc, float* gv, hipStream_t stream) {
    const dim3 grid(pv.scells_cap / 4), block(256);
    clear_hip_error();
    if (hipMemsetAsync(pv.gacc, 0, sizeof(float) * 8 * (size_t)pv.N, stream) != hipSuccess) return PIGS_ERR_LAUNCH;
#define PIGS_CASE(MK)                                                                                         \
    case MK:                                                                                                  \
        hipLaunchKernelGGL((binned_backward_kernel<C, MK>), grid, block, 0, stream, pv, samples, g[0], g[1], g[2], \
                           g[3]);                                                                             \
        break;
    switch (mask) {
        PIGS_CASE(1) PIGS_CASE(2) PIGS_CASE(4) PIGS_CASE(8) PIGS_CASE(7) PIGS_CASE(15)
        default: return PIGS_ERR_UNSUPPORTED;
    }
#undef PIGS_CASE
    hipLaunchKernelGGL((plan_unpermute_kernel<C>), dim3((pv.N + 255) / 256), dim3(256), 0, stream, pv, gm, gc, gv);
    return launch_status();
}

static int covering_mask_b(int mask) {
    if (mask == 1 || mask == 2 || mask == 4 || mask == 8) return mask;
    if ((mask & ~7) == 0) return 7;
    return 15;
}

int plan_forward(void* ws, size_t ws_bytes, int64_t N, int64_t M, int c, float q_max, int mask,
                 const void* samples, void* const* out, hipStream_t stream) {
    if (!plan_supported(N, M, c)) return PIGS_ERR_UNSUPPORTED;
    const PlanLayout p = make_plan_layout(N, M, c);
    if (!ws || ws_bytes < p.total_bytes) return PIGS_ERR_WORKSPACE;
    const PlanView pv = make_view(p, ws, q_max);
    float* o[4];
    for (int k = 0; k < 4; ++k) o[k] = (mask >> k & 1) ? (float*)out[k] : nullptr;
    const int cm = covering_mask_b(mask);
    switch (c) {
        case 1: return plan_forward_c<1>(pv, cm, (const float*)samples, o, stream);
        case 2: return plan_forward_c<2>(pv, cm, (const float*)samples, o, stream);
        case 3: return plan_forward_c<3>(pv, cm, (const float*)samples, o, stream);
    }
    return PIGS_ERR_UNSUPPORTED;
}

int plan_backward(void* ws, size_t ws_bytes, int64_t N, int64_t M, int c, float q_max, int mask,
                  const void* samples, const void* const* gout, void* g_means, void* g_conics, void* g_values,
                  hipStream_t stream) {
    if (!plan_supported(N, M, c)) return PIGS_ERR_UNSUPPORTED;
    const PlanLayout p = make_plan_layout(N, M, c);
    if (!ws || ws_bytes < p.total_bytes) return PIGS_ERR_WORKSPACE;
    const PlanView pv = make_view(p, ws, q_max);
    const float* g[4];
    for (int k = 0; k < 4; ++k) g[k] = (mask >> k & 1) ? (const float*)gout[k] : nullptr;
    const int cm = covering_mask_b(mask);
    switch (c) {
        case 1: return plan_backward_c<1>(pv, cm, (const float*)samples, g, (float*)g_means, (float*)g_conics, (float*)g_values, stream);
        case 2: return plan_backward_c<2>(pv, cm, (const float*)samples, g, (float*)g_means, (float*)g_conics, (float*)g_values, stream);
        case 3: return plan_backward_c<3>(pv, cm, (const float*)samples, g, (float*)g_means, (float*)g_conics, (float*)g_values, stream);
    }
    return PIGS_ERR_UNSUPPORTED;
}

}  // namespace pigs

// Binned (culled) sampler: preprocess (plan build) + forward + backward, float32, d = 2.
// Data structures and the cut-off rule: plan.h.  Per-pair arithmetic: pair_math.h.
//
// Preprocess = 4 launches (bbox partials -> cell key + rank -> scan -> scatter), no memset, no
// host synchronisation, static memory.
// Sampling kernels: one wave = one sample cell (<= 64 points per pass, lane = point).  The wave
// reduces the bounding box of its points; forms, lane-parallel, the cell rectangle to visit on
// every occupied Gaussian level and fetches all the row ranges with one gather; culls the
// candidates in two stages (16-byte box records, then the exact ellipse-vs-box test on the
// survivors' full records); compacts the accepted records into a wave-private LDS queue and
// evaluates them from there with wave-uniform (broadcast) LDS reads.  No workgroup barriers in
// the sampling kernels; HBM traffic is the point stream (sorted points in, outputs out through
// the points' original indices) plus record reads that mostly hit L2.
//
// Build-time knobs (defaults measured on MI355X): PIGS_FWD_WAVES, PIGS_FWD_UNROLL,
// PIGS_FWD_BLOCK_WAVES, PIGS_FWD_QCAP, PIGS_TRAV_STEPS, PIGS_BWD_WAVES, PIGS_XCD_CHUNK; PIGS_STAMPS=1 builds the
// diagnostic variant read by tools/stamps.py.
#include "pair_math.h"
#include "plan.h"
#include "launch.h"

#ifndef PIGS_FWD_WAVES
#define PIGS_FWD_WAVES 6   // waves per SIMD the forward kernel's register budget is held to
#endif
#ifndef PIGS_STAMPS
#define PIGS_STAMPS 0   // diagnostic build: per-wave s_memtime stamps of the forward kernel phases
#endif
#ifndef PIGS_FWD_BLOCK_WAVES
#define PIGS_FWD_BLOCK_WAVES 4   // waves (= sample cells) per forward workgroup; divides 4
#endif
#ifndef PIGS_FWD_UNROLL
#define PIGS_FWD_UNROLL 2     // accepted records evaluated per loop iteration
#endif
#ifndef PIGS_BWD_WAVES
#define PIGS_BWD_WAVES 4      // waves per SIMD the backward kernel's register budget is held to
#endif
#ifndef PIGS_TRAV_STEPS
#define PIGS_TRAV_STEPS 2     // candidate steps whose box records are in flight together (2, 4, 8 measured equal)
#endif


namespace pigs {

#if PIGS_STAMPS
__device__ unsigned long long g_stamps[32768][6];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif

// ------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes, all active)
// ------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
// DPP controls: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140, row_bcast15 = 0x142, row_bcast31 = 0x143.
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
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// ------------------------------------------------------------------------------------------
// preprocess kernels
// ------------------------------------------------------------------------------------------
struct BuildArgs {
    PlanParams* params;
    BoxPartial* boxes;    // [PLAN_BBOX_BLOCKS]
    uint32_t* counts;     // Gaussian cell counters at [0, gcells), sample cell counters at
                          // [sbase, sbase + scells_cap); followed by the scan aggregates
    unsigned long long* agg;   // [scan_blocks] {1 << 32 | workgroup total}, zero before the scan
    uint32_t* starts;     // [ncounts + 1] exclusive scan of counts
    uint2* gkey;          // per Gaussian {cell key, rank inside the cell}
    uint2* skey;          // per point    {cell id,  rank inside the cell}
    float4* rec;
    float4* gbox;
    float* gacc;
    uint32_t* g2o;
    SPoint* spts;
    const float* means;
    const float* conics;
    const float* values;
    const float* samples;
    uint32_t N, M;
    int c, G0, L;
    uint32_t sbase, scells_cap, ncounts, zero_words;
    uint32_t level_off[PLAN_MAX_LEVELS + 1];
    float q_max;
};

// Launch 1 (PLAN_BBOX_BLOCKS workgroups): zero the cell counters; per-workgroup bounding boxes
// of the Gaussian centres and of the sample points, 8 float4 loads (16 points) in flight per
// thread, written as plain partials.
__device__ __forceinline__ void bbox_partial(const float4* __restrict__ pts2, const float2* __restrict__ pts,
                                             uint32_t n, float* out, float (*sh)[4]) {
    const float INF = __builtin_huge_valf();
    float x0 = INF, y0 = INF, x1 = -INF, y1 = -INF;
    auto take = [&](float x, float y) {
        if (fabsf(x) < INF) { x0 = fminf(x0, x); x1 = fmaxf(x1, x); }
        if (fabsf(y) < INF) { y0 = fminf(y0, y); y1 = fmaxf(y1, y); }
    };
    const uint32_t npair = n / 2;                 // float4 = two points
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < npair; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t j = i + k * stride;
            v[k] = pts2[j < npair ? j : i];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { take(v[k].x, v[k].y); take(v[k].z, v[k].w); }
    }
    if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) take(pts[n - 1].x, pts[n - 1].y);
    x0 = wave_min_bcast(x0); y0 = wave_min_bcast(y0);
    x1 = wave_max_bcast(x1); y1 = wave_max_bcast(y1);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[wave][0] = x0; sh[wave][1] = y0; sh[wave][2] = x1; sh[wave][3] = y1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            x0 = fminf(x0, sh[w][0]); y0 = fminf(y0, sh[w][1]);
            x1 = fmaxf(x1, sh[w][2]); y1 = fmaxf(y1, sh[w][3]);
        }
        out[0] = x0; out[1] = y0; out[2] = x1; out[3] = y1;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void plan_bbox_kernel(BuildArgs a) {
    __shared__ float sh[4][4];
    uint4* c4 = (uint4*)a.counts;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.zero_words / 4; i += gridDim.x * 256)
        c4[i] = make_uint4(0, 0, 0, 0);
    bbox_partial((const float4*)a.means, (const float2*)a.means, a.N, a.boxes[blockIdx.x].g, sh);
    bbox_partial((const float4*)a.samples, (const float2*)a.samples, a.M, a.boxes[blockIdx.x].s, sh);
}

// every workgroup of the count kernel reduces the PLAN_BBOX_BLOCKS partials (4 KB, L2 resident)
__device__ __forceinline__ void reduce_boxes(const BoxPartial* boxes, float* gbox, float* sbox, float (*sh)[8]) {
    const float INF = __builtin_huge_valf();
    float v[8] = {INF, INF, -INF, -INF, INF, INF, -INF, -INF};
    if (threadIdx.x < PLAN_BBOX_BLOCKS) {
        const float4* p = (const float4*)&boxes[threadIdx.x];
        const float4 g = p[0], s = p[1];
        v[0] = g.x; v[1] = g.y; v[2] = g.z; v[3] = g.w; v[4] = s.x; v[5] = s.y; v[6] = s.z; v[7] = s.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (k & 2) ? wave_max_bcast(v[k]) : wave_min_bcast(v[k]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) sh[wave][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float r = sh[0][k];
        for (int w = 1; w < (PLAN_BBOX_BLOCKS + 63) / 64; ++w) r = (k & 2) ? fmaxf(r, sh[w][k]) : fminf(r, sh[w][k]);
        (k < 4 ? gbox : sbox)[k & 3] = r;
    }
}

// Launch 2: cell key of every Gaussian / point and its rank inside the cell, with ONE returning
// atomic per run of equal keys in a wave (points of a regular grid arrive in runs that share a
// cell): the run leader adds the run length to the cell counter, the others take consecutive
// ranks behind it.  run_* split the step so that several independent atomics are in flight.
struct Run { int start; uint32_t len; bool leader; };
__device__ __forceinline__ Run run_of(uint32_t k, int lane) {
    const uint32_t prev = __shfl_up(k, 1);
    Run r;
    r.leader = lane == 0 || k != prev;
    const uint64_t lm = __ballot(r.leader);
    const uint64_t upto = (2ull << lane) - 1ull;          // bits 0..lane (lane 63: all ones)
    r.start = 63 - __builtin_clzll(lm & upto);
    const uint64_t above = lm & ~upto;
    r.len = (uint32_t)((above ? __builtin_ctzll(above) : 64) - lane);   // meaningful for leaders
    return r;
}

__global__ __launch_bounds__(256) void plan_count_kernel(BuildArgs a) {
    __shared__ float shb[4][8];
    const int lane = threadIdx.x & 63;
    // Every dependent memory round trip costs 2-4 us in this kernel (in-kernel stamps): issue the
    // workgroup's own loads first, so they fly while the bounding-box partials are reduced.
    const uint32_t gblocks = (a.N + 255) / 256;
    const bool gpart = blockIdx.x < gblocks;
    float gm[2] = {0.f, 0.f}, gc[3] = {0.f, 0.f, 0.f};
    float2 pt[4];
    const uint32_t gi = blockIdx.x * 256 + threadIdx.x;
    const uint32_t i0 = ((blockIdx.x - gblocks) * 4 + (threadIdx.x >> 6)) * 256 + lane;
    if (gpart) {
        if (gi < a.N) {
            gm[0] = a.means[2 * gi]; gm[1] = a.means[2 * gi + 1];
            gc[0] = a.conics[3 * gi]; gc[1] = a.conics[3 * gi + 1]; gc[2] = a.conics[3 * gi + 2];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + 64 * k;
            pt[k] = i < a.M ? ((const float2*)a.samples)[i] : make_float2(0.f, 0.f);
        }
    }
    float gbox[4], sbox[4];
    reduce_boxes(a.boxes, gbox, sbox, shb);
    const GaussGrid g = gauss_grid(gbox, a.G0);
    const SampleGrid sg = sample_grid(sbox, a.M, a.scells_cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.params->gg = g;
        a.params->sg = sg;
#pragma unroll
        for (int l = 0; l <= PLAN_MAX_LEVELS; ++l) a.params->level_off[l] = a.level_off[l];
    }
    // Gaussian workgroups first, sample workgroups after them: the two halves are independent
    // latency chains (load -> returning atomic -> store) and run concurrently on different CUs
    if (gpart) {                        // block-uniform: whole waves enter
        const uint32_t i = gi;
        const bool valid = i < a.N;
        uint32_t key = 0xffffffffu;
        int l = 0;
        if (valid) {
            const float mx = gm[0], my = gm[1];
            const float ca = gc[0], cb = gc[1], cc = gc[2];
            // half extents of the q <= q_max ellipse: sqrt(q_max * Sigma_xx), Sigma = C^-1
            const float det = ca * cc - cb * cb;
            const float R = sqrtf(a.q_max * fmaxf(ca, cc) / det);   // NaN / inf (degenerate conic) -> top level
            float s = g.s0;
            while (l < a.L - 1 && !(R <= s)) { ++l; s *= 2.f; }
            const int G = a.G0 >> l;
            const float inv_s = 1.f / s;
            const int cx = (int)clampf((mx - g.ox) * inv_s, 0.f, (float)(G - 1));   // NaN -> 0
            const int cy = (int)clampf((my - g.oy) * inv_s, 0.f, (float)(G - 1));
            key = a.level_off[l] + ((uint32_t)(cy * G + cx) << level_shift((uint32_t)(G * G)));
        }
        const Run r = run_of(key, lane);
        uint32_t base = 0;
        if (r.leader && valid) base = atomicAdd(&a.counts[key], r.len);
        base = __shfl(base, r.start);
        if (valid) a.gkey[i] = make_uint2(key, base + (uint32_t)(lane - r.start));
    } else {
        // each wave: 4 steps of 64 consecutive points, their atomics issued back to back
        uint32_t id[4], base[4];
        Run r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + 64 * k;
            id[k] = 0xffffffffu;
            if (i < a.M) {
                const float2 p = pt[k];
                const int cx = (int)clampf((p.x - sg.ox) * sg.inv_w, 0.f, (float)(sg.nx - 1));
                const int cy = (int)clampf((p.y - sg.oy) * sg.inv_w, 0.f, (float)(sg.ny - 1));
                id[k] = sample_cell_id(cx, cy, sg.nx);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r[k] = run_of(id[k], lane);
            base[k] = 0;
            if (r[k].leader && id[k] != 0xffffffffu)
                base[k] = atomicAdd(&a.counts[a.sbase + id[k]], r[k].len);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + 64 * k;
            const uint32_t b = __shfl(base[k], r[k].start);
            if (i < a.M) a.skey[i] = make_uint2(id[k], b + (uint32_t)(lane - r[k].start));
        }
    }
}

// Launch 3: exclusive scan counts -> starts in ONE launch.  A workgroup scans PLAN_SCAN_BLOCK
// counters (one coalesced uint4 per thread), publishes its total as one 8-byte {flag, total}
// granule (single agent-scope store: data and flag travel together, no fence needed) and sums
// the granules of the workgroups before it; nobody waits on a later workgroup, so dispatch
// order cannot deadlock it.
__global__ __launch_bounds__(256) void plan_scan_kernel(BuildArgs a) {
    __shared__ uint32_t sh[4];
    __shared__ uint32_t sh2[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;          // uint4 index
    const uint4 v = ((const uint4*)a.counts)[q];
    const uint32_t s = v.x + v.y + v.z + v.w;
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(&a.agg[blockIdx.x], (1ull << 32) | (sh[0] + sh[1] + sh[2] + sh[3]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    uint32_t pre = 0;
    for (uint32_t t = threadIdx.x; t < blockIdx.x; t += 256) {
        unsigned long long x;
        do {
            x = __hip_atomic_load(&a.agg[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(x >> 32)) __builtin_amdgcn_s_sleep(1);
        } while (!(x >> 32));
        pre += (uint32_t)x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o);
    if (lane == 0) sh2[wave] = pre;
    __syncthreads();
    uint32_t run = inc - s + sh2[0] + sh2[1] + sh2[2] + sh2[3];
    for (int w = 0; w < wave; ++w) run += sh[w];
    uint4 o4;
    o4.x = run; o4.y = run + v.x; o4.z = o4.y + v.y; o4.w = o4.z + v.z;
    ((uint4*)a.starts)[q] = o4;      // counters beyond ncounts are zero: starts[ncounts] = total
}

// Launch 4: scatter into sorted order (no atomics: position = cell start + rank) and publish the
// level mask.
__global__ __launch_bounds__(256) void plan_scatter_kernel(BuildArgs a) {
    const uint32_t gblocks = (a.N + 255) / 256;
    const bool gpart = blockIdx.x < gblocks;
    const uint32_t i = (gpart ? blockIdx.x : blockIdx.x - gblocks) * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        // level l holds a Gaussian iff its cells' scanned range is not empty (no atomics, no scratch)
        const int l = (int)threadIdx.x;
        const bool occ = l < a.L && a.starts[a.level_off[l + 1 <= a.L ? l + 1 : a.L]] != a.starts[a.level_off[l < a.L ? l : 0]];
        const uint64_t m = __ballot(occ);
        if (threadIdx.x == 0) a.params->level_mask = (uint32_t)m;
    }
    if (gpart && i < a.N) {
        const uint2 kr = a.gkey[i];
        const uint32_t pos = a.starts[kr.x] + kr.y;
        float v[2] = {0.f, 0.f};
        for (int k = 0; k < a.c; ++k) v[k] = a.values[(size_t)i * a.c + k];
        // {mux, muy, a, b}, {b, c, v0, v1}: (a, b) and (b, c) are register pairs for packed math
        a.rec[2 * pos] = make_float4(a.means[2 * i], a.means[2 * i + 1], a.conics[3 * i], a.conics[3 * i + 1]);
        a.rec[2 * pos + 1] = make_float4(a.conics[3 * i + 1], a.conics[3 * i + 2], v[0], v[1]);
        {   // bounding box of the q <= q_max ellipse: half extents sqrt(q_max Sigma_xx), sqrt(q_max Sigma_yy)
            const float ca = a.conics[3 * i], cb = a.conics[3 * i + 1], cc = a.conics[3 * i + 2];
            const float k = a.q_max / (ca * cc - cb * cb);
            float hx = sqrtf(k * cc), hy = sqrtf(k * ca);
            if (!(hx < 3.0e38f)) hx = 3.0e38f;      // NaN / inf (degenerate conic): always a candidate
            if (!(hy < 3.0e38f)) hy = 3.0e38f;
            a.gbox[pos] = make_float4(a.means[2 * i], a.means[2 * i + 1], hx * 1.0001f, hy * 1.0001f);
        }
        a.g2o[pos] = i;
        // the backward's scratch starts zeroed (and plan_unpermute_kernel re-zeroes what it
        // reads), so the backward needs no memset launch
#pragma unroll
        for (int k = 0; k < 8; ++k) a.gacc[(size_t)k * a.N + pos] = 0.f;
    }
    if (!gpart && i < a.M) {
        const uint2 kr = a.skey[i];
        const float2 p = ((const float2*)a.samples)[i];
        SPoint sp;
        sp.x = p.x; sp.y = p.y; sp.m = i;
        a.spts[a.starts[a.sbase + kr.x] - a.N + kr.y] = sp;
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
struct Rec {
    float mu[2], con[3], v[2];
};
__device__ __forceinline__ Rec make_rec(const float4 A, const float4 B) {
    Rec r;
    r.mu[0] = A.x; r.mu[1] = A.y; r.con[0] = A.z; r.con[1] = A.w; r.con[2] = B.y;
    r.v[0] = B.z; r.v[1] = B.w;
    return r;
}

// Workgroups are dispatched round-robin over the 8 XCDs (workgroup i runs on XCD i % 8) and every
// XCD has its own L2.  Cell ids follow the domain row by row, so inside every group of
// 8 * PIGS_XCD_CHUNK consecutive cell blocks XCD x takes the x-th contiguous run of PIGS_XCD_CHUNK
// blocks: each L2 then holds the Gaussian records of a strip of the domain instead of all of them,
// while the launch still sweeps the domain once from top to bottom.  Bijective for any grid size
// (blocks behind the last whole group keep their index).  0 = no remapping.
#ifndef PIGS_XCD_CHUNK
#define PIGS_XCD_CHUNK 256
#endif
__device__ __forceinline__ uint32_t xcd_block() {
#if PIGS_XCD_CHUNK > 0
    constexpr uint32_t GROUP = 8u * PIGS_XCD_CHUNK;
    const uint32_t b = blockIdx.x, g = b / GROUP, r = b % GROUP;
    if ((g + 1) * GROUP > gridDim.x) return b;
    return g * GROUP + (r & 7u) * PIGS_XCD_CHUNK + (r >> 3);
#else
    return blockIdx.x;
#endif
}

__device__ __forceinline__ int lanes_below(uint64_t mask) {   // set bits of mask below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ------------------------------------------------------------------------------------------
// Forward: one wave per sample cell (lane = point), lean on VALU issue slots (the kernel is
// bound by them, ~1.2 ns per wave-instruction per SIMD) and on dependent memory round trips:
//   1. cell bounds (scalar loads) -> the cell's sorted points (one 12-byte load per lane)
//   2. bounding box of the points (DPP min/max)
//   3. per-level cell rectangles computed lane-parallel (lane = level), their rows scattered
//      into an LDS table, then ONE gather fetches every row's record range
//   4. rows walked in wave-uniform order, 64 candidates per step; the next step's records are
//      in flight while the current step is tested exactly against the box
//   5. accepted records (already in registers) are compacted into a wave-private LDS queue and
//      evaluated from there with wave-uniform addresses (LDS broadcast -> VGPR operands; an
//      SGPR operand makes a VALU op ~1.6x slower on gfx950), two per iteration, the next two
//      prefetched
//   6. outputs stored through the point's original index.
// ------------------------------------------------------------------------------------------
#ifndef PIGS_FWD_QCAP
#define PIGS_FWD_QCAP 128
#endif
constexpr int QCAP = PIGS_FWD_QCAP;   // accepted records queued per wave before an evaluation run

constexpr int CCAP = 256;   // bbox-accepted candidate indices buffered per wave before the exact test

template <int Q>
struct WaveLdsT {
    static constexpr int CAP = Q;
    float4 queue[Q + 8][2];
    uint32_t row_a0[64];
    uint32_t row_a1[64];
    uint32_t cand[CCAP + 64];
};
using WaveLds = WaveLdsT<QCAP>;

// Wave-wide bounding box {min x, max x, min y, max y} with the DPP modifier fused into the min / max
// (hipcc emits v_mov_dpp + a canonicalising v_max + v_min per step from the builtin form: 4x the
// instructions).  The four reductions are independent chains and are interleaved step by step, so
// the two wait states a DPP read needs after the VALU write of its source are filled by the other
// three chains: one s_nop at the head instead of one per step (an s_nop costs an issue slot like a
// VALU instruction).  Results broadcast from lane 63.
#define PIGS_BOX_STEP(MOD)                                \
    "v_min_f32_dpp %0, %0, %0 " MOD " bank_mask:0xf\n\t" \
    "v_max_f32_dpp %1, %1, %1 " MOD " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %2, %2, %2 " MOD " bank_mask:0xf\n\t" \
    "v_max_f32_dpp %3, %3, %3 " MOD " bank_mask:0xf\n\t"
__device__ __forceinline__ void wave_box_dpp(float& x0, float& x1, float& y0, float& y1) {
    asm volatile("s_nop 1\n\t"
                 PIGS_BOX_STEP("quad_perm:[1,0,3,2] row_mask:0xf")
                 PIGS_BOX_STEP("quad_perm:[2,3,0,1] row_mask:0xf")
                 PIGS_BOX_STEP("row_half_mirror row_mask:0xf")
                 PIGS_BOX_STEP("row_mirror row_mask:0xf")
                 PIGS_BOX_STEP("row_bcast:15 row_mask:0xa")
                 PIGS_BOX_STEP("row_bcast:31 row_mask:0xc")
                 "s_nop 1"
                 : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));
    x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x0), 63));
    x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1), 63));
    y0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y0), 63));
    y1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y1), 63));
}

typedef float f2 __attribute__((ext_vector_type(2)));

// One record on one point for c = 1, orders 0..2, written on register PAIRS so that hipcc emits
// v_pk_* without the v_mov shuffles its SLP packing of the scalar form needs: the record keeps
// (a, b) and (b, c) adjacent, the accumulators are (ux, uy), (hxx, hxy) and scalars u, hyy.
struct AccPk {
    float u, hyy;
    f2 g1, h2;      // (ux, uy), (hxx, hxy); signs applied on store
};
__device__ __forceinline__ void accumulate_pk(AccPk& a, f2 s, const float4 A, const float4 B) {
    const f2 mu = {A.x, A.y}, ab = {A.z, A.w}, bc = {B.x, B.y};
    const f2 d = s - mu;
    f2 p = ab * d.x;
    p = __builtin_elementwise_fma(bc, (f2){d.y, d.y}, p);            // p = C d
    const float q = fma_<float>(d.y, p.y, d.x * p.x);
    const float w = B.z * exp_neg_half<float>(q);
    const f2 t = __builtin_elementwise_fma((f2){p.x, p.x}, p, -ab);   // (px px - a, px py - b)
    const float tyy = fma_<float>(p.y, p.y, -B.y);
    a.u += w;
    a.g1 = __builtin_elementwise_fma(p, (f2){w, w}, a.g1);
    a.h2 = __builtin_elementwise_fma(t, (f2){w, w}, a.h2);
    a.hyy = fma_<float>(tyy, w, a.hyy);
}

template <int C, int MASK>
__device__ __forceinline__ void evaluate_queue(float* acc, const float* s, float4 (*q)[2], int n, int lane) {
    constexpr int U = PIGS_FWD_UNROLL;       // records per iteration (independent chains for ILP)
    if (n == 0) return;
    if (lane < 2 * U - 1) {   // neutral records (v = 0) behind the last one: ragged n and the prefetch
        q[n + lane][0] = make_float4(0.f, 0.f, 0.f, 0.f);
        q[n + lane][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { a[u] = q[u][0]; b[u] = q[u][1]; }
    if constexpr (C == 1 && MASK == 7) {
        using L = FwdLayout<2, 1, 7>;
        AccPk A = {acc[L::O0], acc[L::O2 + 2], {acc[L::O1], acc[L::O1 + 1]}, {acc[L::O2], acc[L::O2 + 1]}};
        const f2 sp = {s[0], s[1]};
        for (int k = 0; k < n; k += U) {
            float4 na[U], nb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { na[u] = q[k + U + u][0]; nb[u] = q[k + U + u][1]; }
#pragma unroll
            for (int u = 0; u < U; ++u) accumulate_pk(A, sp, a[u], b[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = na[u]; b[u] = nb[u]; }
        }
        acc[L::O0] = A.u; acc[L::O1] = A.g1.x; acc[L::O1 + 1] = A.g1.y;
        acc[L::O2] = A.h2.x; acc[L::O2 + 1] = A.h2.y; acc[L::O2 + 2] = A.hyy;
    } else {
        for (int k = 0; k < n; k += U) {
            float4 na[U], nb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { na[u] = q[k + U + u][0]; nb[u] = q[k + U + u][1]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const Rec r = make_rec(a[u], b[u]);
                fwd_accumulate<float, 2, C, MASK>(acc, s, r.mu, r.con, r.v);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = na[u]; b[u] = nb[u]; }
        }
    }
}

// Steps 3-4 for one pass.  `batch(A, B, mask, j)` is called for every step with accepted
// candidates (A, B: this lane's record; j: its sorted Gaussian index).
template <typename Lds, typename Batch>
__device__ __forceinline__ void traverse(const PlanView& pv, const GaussGrid& gg, uint32_t level_mask, uint32_t loff,
                                         float bx0, float by0, float bx1, float by1, int lane, Lds& lds,
                                         Batch&& batch) {
    // 3. lane = level: rectangle of cells within one cell of the box; rows scanned over lanes
    const bool occ = lane < pv.L && (level_mask >> lane & 1u);
    const int sh = lane < pv.L ? lane : 0;
    const int G = pv.G0 >> sh;
    const float inv_s = gg.inv_s0 * __builtin_amdgcn_ldexpf(1.f, -sh);
    const float gmax = (float)(G - 1);
    const int cx0 = (int)clampf(floorf((bx0 - gg.ox) * inv_s) - 1.f, 0.f, gmax);
    const int cx1 = (int)clampf(floorf((bx1 - gg.ox) * inv_s) + 1.f, 0.f, gmax);
    const int cy0 = (int)clampf(floorf((by0 - gg.oy) * inv_s) - 1.f, 0.f, gmax);
    const int cy1 = (int)clampf(floorf((by1 - gg.oy) * inv_s) + 1.f, 0.f, gmax);
    const int nr = occ ? cy1 - cy0 + 1 : 0;
    const int csh = level_shift((uint32_t)(G * G));      // the level's counter spacing
    int inc = nr;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {           // levels live in lanes 0..11
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    const int cum = inc - nr;
    const int R = __builtin_amdgcn_readlane(inc, 15);

    for (int r0 = 0; r0 < R; r0 += 64) {
        for (int k = 0; k < nr; ++k) {
            const int r = cum + k - r0;
            if (r >= 0 && r < 64) {
                const uint32_t row = (uint32_t)((cy0 + k) * G);
                lds.row_a0[r] = loff + ((row + (uint32_t)cx0) << csh);
                lds.row_a1[r] = loff + ((row + (uint32_t)cx1 + 1u) << csh);
            }
        }
        const int nrow = R - r0 < 64 ? R - r0 : 64;
        uint32_t jbv = 0, lenv = 0;
        if (lane < nrow) {
            jbv = pv.starts[lds.row_a0[lane]];
            lenv = pv.starts[lds.row_a1[lane]] - jbv;
        }
        // 4. two-stage culling.  Stage 1: the rows' ranges are walked in wave-uniform order, PIGS_TRAV_STEPS
        //    steps (64 candidates each) at a time -- their 16-byte {centre, half extents} records
        //    are all in flight together (every dependent round trip costs microseconds here) --
        //    and tested box against box (8 instructions); survivors' indices go to an LDS list.
        //    Stage 2: survivors' full records are gathered 64 at a time, tested exactly (ellipse
        //    against box) and handed to `batch`.
        int r = -1;
        uint32_t j0 = 0, je = 0;
        auto advance = [&]() -> bool {
            j0 += 64;
            while (j0 >= je) {
                if (++r >= nrow) return false;
                j0 = (uint32_t)__builtin_amdgcn_readlane((int)jbv, r);
                je = j0 + (uint32_t)__builtin_amdgcn_readlane((int)lenv, r);
            }
            return true;
        };
        int cn = 0;
        auto exact_stage = [&]() {
            for (int b0 = 0; b0 < cn; b0 += 128) {
                // two gathers in flight
                const bool in0 = b0 + lane < cn, in1 = b0 + 64 + lane < cn;
                const uint32_t i0 = in0 ? lds.cand[b0 + lane] : lds.cand[0];
                const uint32_t i1 = in1 ? lds.cand[b0 + 64 + lane] : lds.cand[0];
                const float4 A0 = pv.rec[2 * i0], B0 = pv.rec[2 * i0 + 1];
                float4 A1 = A0, B1 = B0;
                if (b0 + 64 < cn) { A1 = pv.rec[2 * i1]; B1 = pv.rec[2 * i1 + 1]; }
                const uint64_t m0 = __ballot(in0 && ellipse_reaches_rect(A0, B0.y, bx0, by0, bx1, by1, pv.q_max));
                if (m0) batch(A0, B0, m0, i0);
                if (b0 + 64 < cn) {
                    const uint64_t m1 = __ballot(in1 && ellipse_reaches_rect(A1, B1.y, bx0, by0, bx1, by1, pv.q_max));
                    if (m1) batch(A1, B1, m1, i1);
                }
            }
            cn = 0;
        };
        bool have = advance();
        while (have) {
            uint32_t sj[PIGS_TRAV_STEPS], se[PIGS_TRAV_STEPS];
            float4 T[PIGS_TRAV_STEPS];
            int ns = 0;
#pragma unroll
            for (int u = 0; u < PIGS_TRAV_STEPS; ++u) {
                sj[u] = j0; se[u] = je;
                if (have) {
                    ns = u + 1;
                    const uint32_t j = j0 + lane < je ? j0 + lane : j0;
                    T[u] = pv.gbox[j];
                    have = advance();
                } else {
                    se[u] = sj[u];          // empty step
                    T[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < PIGS_TRAV_STEPS; ++u) {
                if (u < ns) {
                    const float ex = fmaxf(fmaxf(bx0 - T[u].x, T[u].x - bx1), 0.f);
                    const float ey = fmaxf(fmaxf(by0 - T[u].y, T[u].y - by1), 0.f);
                    const bool ok = (sj[u] + lane < se[u]) && ex <= T[u].z && ey <= T[u].w;
                    const uint64_t mask = __ballot(ok);
                    if (mask) {
                        if (ok) lds.cand[cn + lanes_below(mask)] = sj[u] + lane;
                        cn += __builtin_popcountll(mask);
                        if (cn > CCAP - 64) exact_stage();
                    }
                }
            }
        }
        exact_stage();
    }
}

// register budget: 6 waves/SIMD (80 VGPRs) for up to 10 accumulators per point, fewer waves for the
// wide variants (c = 2 with orders up to 3: 12-20 accumulators) so that they do not spill
template <int C, int MASK>
constexpr int fwd_waves() {
    constexpr int n = FwdLayout<2, C, MASK>::N;
    return n > 12 ? 4 : (n > 10 || (n >= 8 && (MASK & ORD3)) || (C == 1 && MASK == 2)) ? 5 : PIGS_FWD_WAVES;
}
template <int C, int MASK>
__global__ __launch_bounds__(64 * PIGS_FWD_BLOCK_WAVES, (fwd_waves<C, MASK>())) void binned_forward_kernel(
    PlanView pv, float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ o3) {
    using L = FwdLayout<2, C, MASK>;
    __shared__ WaveLds lds_all[PIGS_FWD_BLOCK_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds& lds = lds_all[wave];
    const GaussGrid gg = pv.params->gg;
    const uint32_t level_mask = pv.params->level_mask;
    const uint32_t loff = pv.params->level_off[lane < PLAN_MAX_LEVELS ? lane : 0];   // lane = level: its first counter
    const float INF = __builtin_huge_valf();
    const uint32_t cell = xcd_block() * PIGS_FWD_BLOCK_WAVES + (uint32_t)wave;
    const uint32_t sbeg = pv.starts[pv.sbase + cell] - pv.N;
    const uint32_t send = pv.starts[pv.sbase + cell + 1] - pv.N;
#if PIGS_STAMPS
    const unsigned long long T0 = stamp();
#endif
    if (sbeg >= send) return;
    if (send - sbeg <= 64) {
        // ---- the usual cell: one pass, accumulators live in registers across queue flushes ----
        const bool valid = sbeg + lane < send;
        SPoint sp = {0.f, 0.f, 0u};
        if (valid) sp = pv.spts[sbeg + lane];
        float s[2] = {sp.x, sp.y};
        float bx0 = valid ? s[0] : INF, bx1 = valid ? s[0] : -INF, by0 = valid ? s[1] : INF, by1 = valid ? s[1] : -INF;
        wave_box_dpp(bx0, bx1, by0, by1);
        if (!valid) { s[0] = bx0; s[1] = by0; }

        float acc[L::N];
#pragma unroll
        for (int k = 0; k < L::N; ++k) acc[k] = 0.f;
        int qn = 0;
#if PIGS_STAMPS
        const unsigned long long T1 = stamp();
#endif
        traverse(pv, gg, level_mask, loff, bx0, by0, bx1, by1, lane, lds,
                 [&](const float4 A, const float4 B, uint64_t mask, uint32_t) {
            const int cnt = __builtin_popcountll(mask);
            if (qn + cnt > QCAP) {
                evaluate_queue<C, MASK>(acc, s, lds.queue, qn, lane);
                qn = 0;
            }
            const int slot = qn + lanes_below(mask);
            if (mask >> lane & 1ull) { lds.queue[slot][0] = A; lds.queue[slot][1] = B; }
            qn += cnt;
        });
#if PIGS_STAMPS
        const unsigned long long T2 = stamp();
#endif
        evaluate_queue<C, MASK>(acc, s, lds.queue, qn, lane);
#if PIGS_STAMPS
        const unsigned long long T3 = stamp();
#endif
        if (valid) fwd_store<float, 2, C, MASK>(acc, (int64_t)sp.m, o0, o1, o2, o3);
#if PIGS_STAMPS
        const unsigned long long T4 = stamp();
        if (lane == 0 && cell < 32768) {
            g_stamps[cell][0] = T0; g_stamps[cell][1] = T1; g_stamps[cell][2] = T2; g_stamps[cell][3] = T3;
            g_stamps[cell][4] = T4; g_stamps[cell][5] = (unsigned long long)qn;
        }
#endif
        return;
    }
    // ---- a cell holding more than 64 points (Poisson occupancy, anisotropic grids, clusters): ONE
    // traversal against the box of all its points; every queue flush is evaluated for each 64-point
    // chunk in turn, the chunk's partial sums parked in its output rows between flushes ----
    float bx0 = INF, bx1 = -INF, by0 = INF, by1 = -INF;
    for (uint32_t base = sbeg; base < send; base += 64) {
        if (base + lane < send) {
            const SPoint q = pv.spts[base + lane];
            bx0 = fminf(bx0, q.x); bx1 = fmaxf(bx1, q.x);
            by0 = fminf(by0, q.y); by1 = fmaxf(by1, q.y);
        }
    }
    wave_box_dpp(bx0, bx1, by0, by1);
    int qn = 0;
    bool first = true;
    auto flush = [&]() {
        for (uint32_t base = sbeg; base < send; base += 64) {
            const bool valid = base + lane < send;
            SPoint sp = {bx0, by0, 0u};
            if (valid) sp = pv.spts[base + lane];
            const float s[2] = {sp.x, sp.y};
            float acc[L::N];
#pragma unroll
            for (int k = 0; k < L::N; ++k) acc[k] = 0.f;
            if (!first && valid) fwd_load<float, 2, C, MASK>(acc, (int64_t)sp.m, o0, o1, o2, o3);
            evaluate_queue<C, MASK>(acc, s, lds.queue, qn, lane);
            if (valid) fwd_store<float, 2, C, MASK>(acc, (int64_t)sp.m, o0, o1, o2, o3);
        }
        first = false;
        qn = 0;
    };
    traverse(pv, gg, level_mask, loff, bx0, by0, bx1, by1, lane, lds,
             [&](const float4 A, const float4 B, uint64_t mask, uint32_t) {
        const int cnt = __builtin_popcountll(mask);
        if (qn + cnt > QCAP) flush();
        const int slot = qn + lanes_below(mask);
        if (mask >> lane & 1ull) { lds.queue[slot][0] = A; lds.queue[slot][1] = B; }
        qn += cnt;
    });
    if (qn > 0 || first) flush();
}

// ------------------------------------------------------------------------------------------
// Backward: same cell / traversal / LDS queue as the forward.  Every (wave, Gaussian) row yields
// NV = 5 + c per-lane contributions that must be summed over the 64 points.  Rows are processed
// four at a time and reduced by a transposing butterfly (v_permlane32_swap, v_permlane16_swap,
// then 4 DPP steps inside a row): 2.5 NV instructions per Gaussian instead of 6 NV for four
// separate wave reductions.  The sums are parked in LDS by queue slot and flushed with one atomic
// per lane and value into gacc[k][j] (queue order follows the sorted order, so consecutive lanes
// hit near-consecutive addresses); plan_unpermute_kernel writes the caller's layout.
// ------------------------------------------------------------------------------------------
constexpr int QCAP_BWD = 64;      // smaller queue than the forward's: LDS, not registers, caps the occupancy here
struct WaveLdsBwd {
    WaveLdsT<QCAP_BWD> t;         // queue + row tables (shared code with the forward)
    uint32_t qj[QCAP_BWD + 8];    // sorted Gaussian index of every queue slot
    float sums[QCAP_BWD + 8][8];  // per-slot reduced contributions
};

// hipcc (ROCm 7.2) mis-lowers __builtin_amdgcn_permlane{32,16}_swap when both results feed one
// add (it emits v_add v, v, v with the FIRST result twice), so the swaps are inline asm.  The
// butterfly works on GROUPS of three or four values at a time: their swaps and DPP steps are
// independent chains, interleaved so that one s_nop at the head of a block covers the VALU-write
// -> swap / DPP-read wait states that a nop per instruction covered before (58 -> 9 nops per four
// records; each costs an issue slot).
//   swap32: lanes 0-31 <- a[l] , a[l+32] ; lanes 32-63 <- b[l-32] , b[l]   (the pair is then added)
//   swap16: rows 0,2 <- a[row] , a[row+1] ; rows 1,3 <- b[row-1] , b[row]
template <int G>
__device__ __forceinline__ void swap32_add_group(float* x, float* a, float* b) {     // x[k] = swap32_add(a[k], b[k])
    static_assert(G == 3 || G == 4, "groups of three or four");
    if constexpr (G == 3)
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                     "v_permlane32_swap_b32 %4, %5"
                     : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]));
    else
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                     "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7"
                     : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]));
#pragma unroll
    for (int k = 0; k < G; ++k) x[k] = a[k] + b[k];
}
template <int G>
__device__ __forceinline__ void swap16_add_group(float* y, float* a, float* b) {
    if constexpr (G == 3)
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                     "v_permlane16_swap_b32 %4, %5"
                     : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]));
    else
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                     "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
                     : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]));
#pragma unroll
    for (int k = 0; k < G; ++k) y[k] = a[k] + b[k];
}
// sum over the 16 lanes of a row, in every lane, for G values at once
#define PIGS_ROW3(MOD)                                                   \
    "v_add_f32_dpp %0, %0, %0 " MOD " row_mask:0xf bank_mask:0xf\n\t"   \
    "v_add_f32_dpp %1, %1, %1 " MOD " row_mask:0xf bank_mask:0xf\n\t"   \
    "v_add_f32_dpp %2, %2, %2 " MOD " row_mask:0xf bank_mask:0xf\n\t"
#define PIGS_ROW4(MOD) PIGS_ROW3(MOD) "v_add_f32_dpp %3, %3, %3 " MOD " row_mask:0xf bank_mask:0xf\n\t"
template <int G>
__device__ __forceinline__ void row_sum_group(float* y) {
    if constexpr (G == 3)
        asm volatile("s_nop 1\n\t" PIGS_ROW3("quad_perm:[1,0,3,2]") PIGS_ROW3("quad_perm:[2,3,0,1]")
                     PIGS_ROW3("row_half_mirror") PIGS_ROW3("row_mirror") "s_nop 1"
                     : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]));
    else
        asm volatile("s_nop 1\n\t" PIGS_ROW4("quad_perm:[1,0,3,2]") PIGS_ROW4("quad_perm:[2,3,0,1]")
                     PIGS_ROW4("row_half_mirror") PIGS_ROW4("row_mirror") "s_nop 1"
                     : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
}
// y[k] = sum over the 64 lanes of part[r][k], delivered in row r (r = 0..3), for values k0 .. k0+G-1
template <int G, int NV>
__device__ __forceinline__ void butterfly_group(float (*part)[NV], int k0, float* y) {
    float x0[G], x1[G];
    swap32_add_group<G>(x0, &part[0][k0], &part[2][k0]);     // lanes 0-31 <- records 0 / 1, lanes 32-63 <- 2 / 3
    swap32_add_group<G>(x1, &part[1][k0], &part[3][k0]);
    swap16_add_group<G>(y + k0, x0, x1);                     // rows {0,2} <- even, rows {1,3} <- odd records
    row_sum_group<G>(y + k0);
}

template <int C, int MASK>
__device__ __forceinline__ void backward_queue(const float* s, const Gsym<float, 2, C, MASK>& G, WaveLdsBwd& lds,
                                               int n, int lane, float* __restrict__ gacc, uint32_t N) {
    using BL = BwdLayout<2, C>;
    constexpr int NV = BL::N;
    if (n == 0) return;
    if (lane < 8) {           // neutral records behind the last one (their sums are never flushed)
        lds.t.queue[n + lane][0] = make_float4(0.f, 0.f, 0.f, 0.f);
        lds.t.queue[n + lane][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int row = lane >> 4, col = lane & 15;
    // four records per round: row 0 = record 0, row 1 = record 1, row 2 = record 2, row 3 = record 3
    for (int k0 = 0; k0 < n; k0 += 4) {
        float part[4][NV];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const Rec r = make_rec(lds.t.queue[k0 + u][0], lds.t.queue[k0 + u][1]);
#pragma unroll
            for (int k = 0; k < NV; ++k) part[u][k] = 0.f;
            bwd_accumulate<float, 2, C, MASK, (MASK & ORD3) != 0>(part[u], s, r.mu, r.con, r.v, G);
        }
        float y[NV];
        butterfly_group<3, NV>(part, 0, y);
        if constexpr (NV == 6) butterfly_group<3, NV>(part, 3, y);
        else butterfly_group<4, NV>(part, 3, y);
        if (col == 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) lds.sums[k0 + row][k] = y[k];
        }
    }
    for (int q0 = 0; q0 < n; q0 += 64) {
        const int slot = q0 + lane;
        if (slot < n) {
            const uint32_t j = lds.qj[slot];
#pragma unroll
            for (int k = 0; k < NV; ++k) atomicAdd(&gacc[(size_t)k * N + j], lds.sums[slot][k]);
        }
    }
}

template <int C, int MASK>
constexpr int bwd_waves() {      // the widest gradient sets get 3 waves (168 VGPRs): no spills
    return ((C == 2 && (MASK == 7 || MASK == 8 || MASK == 15)) || (C == 1 && MASK == 15)) ? 3 : PIGS_BWD_WAVES;
}
template <int C, int MASK>
__global__ __launch_bounds__(256, (bwd_waves<C, MASK>())) void binned_backward_kernel(PlanView pv, const float* __restrict__ G0p,
                                                              const float* __restrict__ G1p,
                                                              const float* __restrict__ G2p,
                                                              const float* __restrict__ G3p) {
    __shared__ WaveLdsBwd lds_all[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t cell = xcd_block() * 4 + (uint32_t)wave;
    const uint32_t sbeg = pv.starts[pv.sbase + cell] - pv.N;
    const uint32_t send = pv.starts[pv.sbase + cell + 1] - pv.N;
    if (sbeg >= send) return;
    WaveLdsBwd& lds = lds_all[wave];
    const GaussGrid gg = pv.params->gg;
    const uint32_t level_mask = pv.params->level_mask;
    const uint32_t loff = pv.params->level_off[lane < PLAN_MAX_LEVELS ? lane : 0];   // lane = level: its first counter
    const float INF = __builtin_huge_valf();

    // the points and incoming gradients of one 64-point chunk of the cell; lanes without a point
    // sit on a corner of the box and contribute nothing
    auto load_chunk = [&](uint32_t base, float bx0, float by0, float* s, Gsym<float, 2, C, MASK>& G) {
        const bool valid = base + lane < send;
        SPoint sp = {bx0, by0, 0u};
        if (valid) sp = pv.spts[base + lane];
        s[0] = sp.x; s[1] = sp.y;
        G.load((int64_t)sp.m, G0p, G1p, G2p, G3p);
        if (!valid) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                G.g0[ch] = 0.f;
                G.g1[0][ch] = G.g1[1][ch] = 0.f;
                G.g2[0][ch] = G.g2[1][ch] = G.g2[2][ch] = 0.f;
                G.g3[0][ch] = G.g3[1][ch] = G.g3[2][ch] = G.g3[3][ch] = 0.f;
            }
        }
    };
    // box of ALL the cell's points: a cell holding more than 64 is still traversed once, and every
    // queue flush is reduced for each of its 64-point chunks in turn
    float bx0 = INF, bx1 = -INF, by0 = INF, by1 = -INF;
    for (uint32_t base = sbeg; base < send; base += 64) {
        if (base + lane < send) {
            const SPoint q = pv.spts[base + lane];
            bx0 = fminf(bx0, q.x); bx1 = fmaxf(bx1, q.x);
            by0 = fminf(by0, q.y); by1 = fmaxf(by1, q.y);
        }
    }
    wave_box_dpp(bx0, bx1, by0, by1);
    const bool single = send - sbeg <= 64;          // the usual cell: its chunk stays in registers
    float s[2];
    Gsym<float, 2, C, MASK> G;
    if (single) load_chunk(sbeg, bx0, by0, s, G);
    int qn = 0;
    auto flush = [&]() {
        if (single) {
            backward_queue<C, MASK>(s, G, lds, qn, lane, pv.gacc, pv.N);
        } else {
            for (uint32_t base = sbeg; base < send; base += 64) {
                load_chunk(base, bx0, by0, s, G);
                backward_queue<C, MASK>(s, G, lds, qn, lane, pv.gacc, pv.N);
            }
        }
        qn = 0;
    };
    traverse(pv, gg, level_mask, loff, bx0, by0, bx1, by1, lane, lds.t,
             [&](const float4 A, const float4 B, uint64_t mask, uint32_t j) {
        const int cnt = __builtin_popcountll(mask);
        if (qn + cnt > QCAP_BWD) flush();
        const int slot = qn + lanes_below(mask);
        if (mask >> lane & 1ull) {
            lds.t.queue[slot][0] = A;
            lds.t.queue[slot][1] = B;
            lds.qj[slot] = j;
        }
        qn += cnt;
    });
    flush();
}

template <int C>
__global__ __launch_bounds__(256) void plan_unpermute_kernel(PlanView pv, float* __restrict__ g_means,
                                                             float* __restrict__ g_conics,
                                                             float* __restrict__ g_values) {
    using BL = BwdLayout<2, C>;
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= pv.N) return;
    const uint32_t n = pv.g2o[j];
    float v[BL::N];
#pragma unroll
    for (int k = 0; k < BL::N; ++k) {
        v[k] = pv.gacc[(size_t)k * pv.N + j];
        pv.gacc[(size_t)k * pv.N + j] = 0.f;       // leave the scratch zeroed for the next backward
    }
    g_means[2 * n] = v[BL::MU + 0];
    g_means[2 * n + 1] = v[BL::MU + 1];
#pragma unroll
    for (int k = 0; k < 3; ++k) g_conics[3 * n + k] = v[BL::CON + k];
#pragma unroll
    for (int k = 0; k < C; ++k) g_values[(size_t)C * n + k] = v[BL::VAL + k];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static PlanView make_view(const PlanLayout& p, void* ws, float q_max) {
    char* b = (char*)ws;
    PlanView v{};
    v.params = (const PlanParams*)(b + p.off_params);
    v.starts = (const uint32_t*)(b + p.off_starts);
    v.rec = (const float4*)(b + p.off_rec);
    v.gbox = (const float4*)(b + p.off_box);
    v.g2o = (const uint32_t*)(b + p.off_g2o);
    v.spts = (const SPoint*)(b + p.off_spts);
    v.N = (uint32_t)p.N; v.M = (uint32_t)p.M;
    v.G0 = p.G0; v.L = p.L;
    v.sbase = p.sbase; v.scells_cap = p.scells_cap;
    for (int l = 0; l <= PLAN_MAX_LEVELS; ++l) v.level_off[l] = p.level_off[l];
    v.q_max = q_max;
    v.gacc = (float*)(b + p.off_gacc);
    return v;
}

static bool plan_supported(int64_t N, int64_t M, int c) {
    return N >= 1 && M >= 1 && c >= 1 && c <= 2 && N < (1LL << 30) && M < (1LL << 31) - 64 &&
           N + M < (1LL << 32) - 1;
}


#if PIGS_STAMPS
extern "C" int pigs_debug_stamps(void* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32768 * 6);
}
#endif

size_t plan_workspace_bytes(int64_t N, int64_t M, int c) {
    if (!plan_supported(N, M, c)) return 0;
    return make_plan_layout(N, M, c).total_bytes;
}

int plan_build(void* ws, size_t ws_bytes, int64_t N, int64_t M, int c, float q_max,
               const void* means, const void* conics, const void* values, const void* samples,
               hipStream_t stream) {
    if (!plan_supported(N, M, c)) return PIGS_ERR_UNSUPPORTED;
    if (!(q_max > 0.f)) return PIGS_ERR_INVALID;
    const PlanLayout p = make_plan_layout(N, M, c);
    if (!ws || ws_bytes < p.total_bytes) return PIGS_ERR_WORKSPACE;
    char* b = (char*)ws;
    BuildArgs a{};
    a.params = (PlanParams*)(b + p.off_params);
    a.boxes = (BoxPartial*)(b + p.off_boxes);
    a.counts = (uint32_t*)(b + p.off_counts);
    a.agg = (unsigned long long*)(b + p.off_agg);
    a.starts = (uint32_t*)(b + p.off_starts);
    a.gkey = (uint2*)(b + p.off_gkey);
    a.skey = (uint2*)(b + p.off_skey);
    a.rec = (float4*)(b + p.off_rec);
    a.gbox = (float4*)(b + p.off_box);
    a.gacc = (float*)(b + p.off_gacc);
    a.g2o = (uint32_t*)(b + p.off_g2o);
    a.spts = (SPoint*)(b + p.off_spts);
    a.means = (const float*)means; a.conics = (const float*)conics;
    a.values = (const float*)values; a.samples = (const float*)samples;
    a.N = (uint32_t)N; a.M = (uint32_t)M; a.c = c; a.G0 = p.G0; a.L = p.L;
    a.sbase = p.sbase; a.scells_cap = p.scells_cap; a.ncounts = p.ncounts;
    a.zero_words = (uint32_t)((p.off_starts - p.off_counts) / 4);     // counters + aggregates
    for (int l = 0; l <= PLAN_MAX_LEVELS; ++l) a.level_off[l] = p.level_off[l];
    a.q_max = q_max;

    clear_hip_error();
    hipLaunchKernelGGL(plan_bbox_kernel, dim3(PLAN_BBOX_BLOCKS), dim3(256), 0, stream, a);
    const uint32_t gb = (uint32_t)((N + 255) / 256), sb = (uint32_t)((M + 1023) / 1024);
    hipLaunchKernelGGL(plan_count_kernel, dim3(gb + sb), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(p.scan_blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(plan_scatter_kernel, dim3(gb + (uint32_t)((M + 255) / 256)), dim3(256), 0, stream, a);
    return launch_status();
}

template <int C>
static int plan_forward_c(const PlanView& pv, int mask, float* const* out, hipStream_t stream) {
    const dim3 grid(pv.scells_cap / PIGS_FWD_BLOCK_WAVES), block(64 * PIGS_FWD_BLOCK_WAVES);
    clear_hip_error();
#define PIGS_CASE(MK)                                                                                        \
    case MK:                                                                                                 \
        hipLaunchKernelGGL((binned_forward_kernel<C, MK>), grid, block, 0, stream, pv, out[0], out[1], out[2], \
                           out[3]);                                                                          \
        break;
    switch (mask) {
        PIGS_CASE(1) PIGS_CASE(2) PIGS_CASE(4) PIGS_CASE(8) PIGS_CASE(7) PIGS_CASE(15) PIGS_CASE(16) PIGS_CASE(19)
        default: return PIGS_ERR_UNSUPPORTED;
    }
#undef PIGS_CASE
    return launch_status();
}

template <int C>
static int plan_backward_c(const PlanView& pv, int mask, const float* const* g, float* gm, float* gc, float* gv,
                           hipStream_t stream) {
    const dim3 grid(pv.scells_cap / 4), block(256);
    clear_hip_error();
#define PIGS_CASE(MK)                                                                                         \
    case MK:                                                                                                  \
        hipLaunchKernelGGL((binned_backward_kernel<C, MK>), grid, block, 0, stream, pv, g[0], g[1], g[2], g[3]); \
        break;
    switch (mask) {
        PIGS_CASE(1) PIGS_CASE(2) PIGS_CASE(4) PIGS_CASE(8) PIGS_CASE(7) PIGS_CASE(15) PIGS_CASE(16) PIGS_CASE(19)
        default: return PIGS_ERR_UNSUPPORTED;
    }
#undef PIGS_CASE
    hipLaunchKernelGGL((plan_unpermute_kernel<C>), dim3((pv.N + 255) / 256), dim3(256), 0, stream, pv, gm, gc, gv);
    return launch_status();
}

int plan_forward(void* ws, size_t ws_bytes, int64_t N, int64_t M, int c, float q_max, int mask, void* const* out,
                 hipStream_t stream) {
    if (!plan_supported(N, M, c)) return PIGS_ERR_UNSUPPORTED;
    const PlanLayout p = make_plan_layout(N, M, c);
    if (!ws || ws_bytes < p.total_bytes) return PIGS_ERR_WORKSPACE;
    const PlanView pv = make_view(p, ws, q_max);
    float* o[4];
    for (int k = 0; k < 4; ++k) o[k] = mask_uses_slot(mask, k) ? (float*)out[k] : nullptr;
    const int cm = covering_mask_of(mask);
    switch (c) {
        case 1: return plan_forward_c<1>(pv, cm, o, stream);
        case 2: return plan_forward_c<2>(pv, cm, o, stream);
    }
    return PIGS_ERR_UNSUPPORTED;
}

int plan_backward(void* ws, size_t ws_bytes, int64_t N, int64_t M, int c, float q_max, int mask,
                  const void* const* gout, void* g_means, void* g_conics, void* g_values, hipStream_t stream) {
    if (!plan_supported(N, M, c)) return PIGS_ERR_UNSUPPORTED;
    const PlanLayout p = make_plan_layout(N, M, c);
    if (!ws || ws_bytes < p.total_bytes) return PIGS_ERR_WORKSPACE;
    const PlanView pv = make_view(p, ws, q_max);
    const float* g[4];
    for (int k = 0; k < 4; ++k) g[k] = mask_uses_slot(mask, k) ? (const float*)gout[k] : nullptr;
    const int cm = covering_mask_of(mask);
    switch (c) {
        case 1: return plan_backward_c<1>(pv, cm, g, (float*)g_means, (float*)g_conics, (float*)g_values, stream);
        case 2: return plan_backward_c<2>(pv, cm, g, (float*)g_means, (float*)g_conics, (float*)g_values, stream);
    }
    return PIGS_ERR_UNSUPPORTED;
}

}  // namespace pigs

// Walking the multi-level Gaussian grid of a plan workspace (plan.h): wave-level helpers, the exact
// ellipse / rectangle test and `traverse()`.  Shared by the tile-list build (plan.hip) and the neighbour-
// list build of aggregate_neighbors (aggregate.hip).
#pragma once
#include "plan.h"

#ifndef PIGS_TRAV_STEPS
#define PIGS_TRAV_STEPS 2     // candidate steps whose box records are in flight together (2, 4, 8 measured equal)
#endif
// (Also measured and dropped: the candidates of ALL row ranges packed into chunks of 256 -- a range holds 20-40
// Gaussians at C3, so a step of 64 lanes is 40 % full and the ~12 ranges around a block take 6 dependent round
// trips where a packed walk takes 1-2: the lane -> (range, offset) mapping cost more than the round trips it
// saved, list build 21.2 vs 20.4 us; with the survivors dropped -- the walk alone -- 12.0 vs 10.8 us.)

namespace pigs {

// ------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes, all active)
// ------------------------------------------------------------------------------------------
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
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int lanes_below(uint64_t mask) {   // set bits of mask below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// orders this wave's LDS accesses for the compiler (lanes exchange data through LDS without a
// workgroup barrier: the LDS itself serves one wave's instructions in order)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}


// ------------------------------------------------------------------------------------------
// candidate test: does the ellipse q <= q_max of a Gaussian reach the rectangle [x0,x1]x[y0,y1]?
// q is convex with its minimum at the centre, so the minimum over the rectangle lies on the
// edge(s) facing the centre; each facing edge is minimised in closed form.  Comparisons are
// written so that NaN (degenerate conic) accepts.
// ------------------------------------------------------------------------------------------
struct Ellipse {
    float x, y, a, b, c, nb_c, nb_a;      // centre, conic, -b/c, -b/a
};
// what depends on the Gaussian alone (two reciprocals), prepared once and reused for the tile's box
// and its four group boxes
__device__ __forceinline__ Ellipse ellipse_of(float4 A, float cc) {
    Ellipse e;
    e.x = A.x; e.y = A.y; e.a = A.z; e.b = A.w; e.c = cc;
    e.nb_c = -A.w * __builtin_amdgcn_rcpf(cc);
    e.nb_a = -A.w * __builtin_amdgcn_rcpf(A.z);
    return e;
}
// the minimum of q over the rectangle (NaN for a degenerate conic: every comparison `!(qmin > q_max)` accepts)
__device__ __forceinline__ float ellipse_min_q_rect(const Ellipse& e, float x0, float y0, float x1, float y1) {
    // clamp(v, lo, hi) with lo <= hi is the median of the three (one v_med3_f32; like the min / max
    // pair it returns a bound when v is NaN)
    const float l = x0 - e.x, r = x1 - e.x, bt = y0 - e.y, tp = y1 - e.y;
    const float xe = __builtin_amdgcn_fmed3f(0.f, l, r), ye = __builtin_amdgcn_fmed3f(0.f, bt, tp);
    const float ys = __builtin_amdgcn_fmed3f(e.nb_c * xe, bt, tp);
    const float xs = __builtin_amdgcn_fmed3f(e.nb_a * ye, l, r);
    const float q1 = e.a * xe * xe + (2.f * e.b * xe + e.c * ys) * ys;
    const float q2 = e.c * ye * ye + (2.f * e.b * ye + e.a * xs) * xs;
    return fminf(q1, q2);
}
__device__ __forceinline__ bool ellipse_reaches_rect(const Ellipse& e, float x0, float y0, float x1, float y1, float q_max) {
    return !(ellipse_min_q_rect(e, x0, y0, x1, y1) > q_max);
}
__device__ __forceinline__ bool ellipse_reaches_rect(float4 A, float cc, float x0, float y0, float x1, float y1,
                                                     float q_max) {
    return ellipse_reaches_rect(ellipse_of(A, cc), x0, y0, x1, y1, q_max);
}


// ------------------------------------------------------------------------------------------
// Traversal of the Gaussian grid for one rectangle (used by the list build only):
//   1. lane = level: rectangle of cells within one cell of the box; its rows scattered into an
//      LDS table, then ONE gather fetches every row's record range  -> `rows(nrow, jb, len)`
//   2. two-stage culling.  Stage 1: the rows' ranges are walked in wave-uniform order,
//      PIGS_TRAV_STEPS steps (64 candidates each) at a time -- their 16-byte {centre, half
//      extents} records are all in flight together (every dependent round trip costs
//      microseconds here) -- and tested box against box (8 instructions); survivors' indices go
//      to an LDS list.  Stage 2: survivors' full records are gathered 64 at a time, tested exactly
//      (ellipse against box) and handed to `batch(A, B, mask, j)`: this lane's record and its
//      sorted Gaussian index, `mask` = the lanes that hold an accepted one.
// ------------------------------------------------------------------------------------------
constexpr int CCAP = 128 * PIGS_TRAV_STEPS;   // bbox-accepted candidate indices buffered per wave before the exact test
static_assert(CCAP >= 128 * PIGS_TRAV_STEPS, "room for one more batch of steps below the flush threshold");
struct TravLds {
    uint32_t row_a0[64];
    uint32_t row_a1[64];
    uint32_t cand[CCAP];
    uint32_t strip[256];          // traverse_strips: the hit strips of 16 super-strips
};

template <typename Rows, typename Batch>
__device__ __forceinline__ void traverse(const PlanView& pv, const GaussGrid& gg, uint32_t level_mask, uint32_t loff,
                                         float bx0, float by0, float bx1, float by1, int lane, TravLds& lds,
                                         bool walk, Rows&& rows, Batch&& batch) {
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
        wave_lds_fence();
        const int nrow = R - r0 < 64 ? R - r0 : 64;
        uint32_t jbv = 0, lenv = 0;
        if (lane < nrow) {
            jbv = pv.starts[lds.row_a0[lane]];
            lenv = pv.starts[lds.row_a1[lane]] - jbv;
        }
        wave_lds_fence();
        rows(nrow, jbv, lenv);
        if (!walk) continue;
        int r = -1;
        uint32_t j0 = 0, je = 0;
        auto advance = [&]() __attribute__((always_inline)) -> bool {
            j0 += 64;
            while (j0 >= je) {
                if (++r >= nrow) return false;
                j0 = (uint32_t)__builtin_amdgcn_readlane((int)jbv, r);
                je = j0 + (uint32_t)__builtin_amdgcn_readlane((int)lenv, r);
            }
            return true;
        };
        int cn = 0;
        auto exact_stage = [&]() __attribute__((always_inline)) {
            wave_lds_fence();
            // the next step's records are requested before the current step is tested and handed on
            uint32_t i = lane < cn ? lds.cand[lane] : lds.cand[0];
            float4 A = pv.rec[2 * i], B = pv.rec[2 * i + 1];
            for (int b0 = 0; b0 < cn; b0 += 64) {
                const bool in = b0 + lane < cn;
                const uint32_t ic = i;
                const float4 Ac = A, Bc = B;
                if (b0 + 64 < cn) {
                    i = b0 + 64 + lane < cn ? lds.cand[b0 + 64 + lane] : lds.cand[0];
                    A = pv.rec[2 * i]; B = pv.rec[2 * i + 1];
                }
                const uint64_t m = __ballot(in && ellipse_reaches_rect(Ac, Bc.x, bx0, by0, bx1, by1, pv.q_max));
                if (m) batch(Ac, Bc, m, ic);
            }
            wave_lds_fence();
            cn = 0;
        };
        bool have = advance();
        while (have || cn > 0) {
            if (have) {
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
                        }
                    }
                }
            }
            // one call site (the exact stage and everything the caller does per batch is inlined here once)
            if (cn > CCAP - 64 * PIGS_TRAV_STEPS || (!have && cn > 0)) exact_stage();
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same for a plan that kept the Gaussians in the caller's order (PlanParams::strips): candidates come from boxes,
// coarse to fine.  Stage 1: 256 super-strip boxes at a time (four loads per lane in flight), the hits to an LDS list.
// Stage 2: the 16 strip boxes of 16 hit super-strips at a time (four loads in flight), the hit strips to a second
// list.  Stage 3: the records of four hit strips per step, the next two steps' in flight under the current one's exact
// test, handed to `batch` as above (j = the Gaussian's own index).  About as many dependent round trips as the grid's
// starts -> boxes -> records.  walk == false: stages 1 and 2, then `rows` with the hit strips as record ranges, runs of
// consecutive strips merged (the RANGES fall-back of a tile whose lists do not fit).
// ------------------------------------------------------------------------------------------
template <typename Rows, typename Batch>
__device__ __forceinline__ void traverse_strips(const PlanView& pv, float bx0, float by0, float bx1, float by1, int lane,
                                                TravLds& lds, bool walk, Rows&& rows, Batch&& batch) {
    const uint32_t N = pv.N;
    const uint32_t nsuper = (N + SUPER - 1u) / SUPER, nstrip = (N + STRIP - 1u) / STRIP;
    // (a NaN in a box -- a NaN centre -- keeps it: every comparison is written to accept)
    auto reaches = [&](const float4 b) { return !(b.x > bx1) && !(b.z < bx0) && !(b.y > by1) && !(b.w < by0); };
    const uint32_t slot = (uint32_t)lane >> 4, sub = (uint32_t)lane & 15u;
    static_assert(CCAP >= 256, "the hit super-strips of a chunk of 256");
    for (uint32_t c0 = 0; c0 < nsuper; c0 += 256u) {
        // ---- stage 1
        float4 sb[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t sup = c0 + 64u * k + (uint32_t)lane;
            sb[k] = pv.sbox[sup < nsuper ? sup : 0u];
        }
        uint32_t nhs = 0;
        wave_lds_fence();
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t sup = c0 + 64u * k + (uint32_t)lane;
            const bool hit = sup < nsuper && reaches(sb[k]);
            const uint64_t hm = __ballot(hit);
            if (hit) lds.cand[nhs + (uint32_t)lanes_below(hm)] = sup;
            nhs += (uint32_t)__builtin_popcountll(hm);
        }
        wave_lds_fence();
        for (uint32_t h0 = 0; h0 < nhs; h0 += 16u) {
            // ---- stage 2: 16 hit super-strips = 256 strip boxes
            float4 pb[4];
            uint32_t sidx[4];
            bool in[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t h = h0 + 4u * k + slot;
                in[k] = h < nhs;
                sidx[k] = lds.cand[in[k] ? h : h0] * SUPER_STRIPS + sub;
                in[k] = in[k] && sidx[k] < nstrip;
                pb[k] = pv.pbox[in[k] ? sidx[k] : 0u];
            }
            uint32_t np = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const bool hit = in[k] && reaches(pb[k]);
                const uint64_t hm = __ballot(hit);
                if (hit) lds.strip[np + (uint32_t)lanes_below(hm)] = sidx[k];
                np += (uint32_t)__builtin_popcountll(hm);
            }
            if (np == 0u) continue;
            wave_lds_fence();
            if (!walk) {
                // the hit strips (ascending) as record ranges, runs of consecutive strips merged: a lattice in row order
                // gives one range per row of the lattice that the rectangle's reach crosses
                for (uint32_t k0 = 0; k0 < np; k0 += 64u) {
                    const uint32_t k = k0 + (uint32_t)lane;
                    const bool have = k < np;
                    const uint32_t st = lds.strip[have ? k : k0];
                    const uint32_t prev = lds.strip[have && lane > 0 ? k - 1u : k0];
                    const bool start = have && (lane == 0 || st != prev + 1u);
                    const uint64_t sm = __ballot(start);
                    const uint64_t hv = __ballot(have);
                    // this start's run ends in front of the next start (or of the chunk's last strip in hand)
                    const uint64_t above = lane < 63 ? sm >> (lane + 1) : 0ull;
                    const uint32_t nh = (uint32_t)__builtin_popcountll(hv);
                    const uint32_t run = above ? (uint32_t)__builtin_ctzll(above) + 1u : nh - (uint32_t)lane;
                    const uint32_t nrow = (uint32_t)__builtin_popcountll(sm);
                    wave_lds_fence();
                    if (start) {
                        const uint32_t r = (uint32_t)lanes_below(sm);
                        const uint32_t jb = st * STRIP;
                        const uint32_t len = run * STRIP;
                        lds.row_a0[r] = jb;
                        lds.row_a1[r] = N - jb < len ? N - jb : len;
                    }
                    wave_lds_fence();
                    uint32_t jb = 0, len = 0;
                    if ((uint32_t)lane < nrow) { jb = lds.row_a0[lane]; len = lds.row_a1[lane]; }
                    wave_lds_fence();
                    rows((int)nrow, jb, len);
                }
                continue;
            }
            // ---- stage 3: four strips per step, the next step's records requested before the current step is tested
            auto fetch = [&](uint32_t k0, float4& A, float4& B, uint32_t& j, bool& ok) __attribute__((always_inline)) {
                const bool hv = k0 + slot < np;
                const uint32_t st = lds.strip[hv ? k0 + slot : k0];
                j = st * STRIP + sub;
                ok = hv && j < N;
                if (!ok) j = 0u;
                A = pv.rec[2 * j]; B = pv.rec[2 * j + 1];
            };
            float4 A0, B0, A1, B1;
            uint32_t j0, j1 = 0u;
            bool ok0, ok1 = false;
            fetch(0u, A0, B0, j0, ok0);
            A1 = A0; B1 = B0;
            if (4u < np) fetch(4u, A1, B1, j1, ok1);
            for (uint32_t k0 = 0; k0 < np; k0 += 4u) {       // (two steps' records in flight under the test of a third)
                const float4 Ac = A0, Bc = B0;
                const uint32_t jc = j0;
                const bool okc = ok0;
                A0 = A1; B0 = B1; j0 = j1; ok0 = ok1;
                if (k0 + 8u < np) fetch(k0 + 8u, A1, B1, j1, ok1);
                const uint64_t m = __ballot(okc && ellipse_reaches_rect(Ac, Bc.x, bx0, by0, bx1, by1, pv.q_max));
                if (m) batch(Ac, Bc, m, jc);
            }
            wave_lds_fence();
        }
    }
}

}  // namespace pigs

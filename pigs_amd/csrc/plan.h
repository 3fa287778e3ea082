// Binned ("plan") path: data structures shared by the preprocess and sampling kernels.
//
// What `GaussianSampler.preprocess(means, values, covariances, conics, samples)` builds
// (call sites model_pn.py:648,768,784; the reference's own native preprocess is not visible --
// this design is new):
//
//  * Gaussians are binned by their CENTRE into a multi-level uniform grid.  Level l has
//    (G0 >> l)^2 square cells of side s0 * 2^l; a Gaussian whose q <= q_max ellipse has
//    half-extent R = max(hx, hy) goes to the lowest level with R <= s_l (top level: one cell,
//    anything larger).  Within a level the Gaussians are counting-sorted by cell (row-major),
//    so "all Gaussians whose ellipse can reach a rectangle" is a handful of CONTIGUOUS ranges
//    of packed 32-byte records: the cells within one cell of the rectangle, per level.
//    Memory is static (N records + the cell table): no per-call allocation, no host sync.
//  * Sample points are counting-sorted into square cells holding ~63 points (2x2-blocked cell
//    order, so the four waves of a workgroup own a 2x2 block of cells); the sorted copy keeps
//    each point's original index.  One wave = one cell = 64 lanes = 64 points.
//
// Cut-off: a (point, Gaussian) pair is evaluated iff the Gaussian's ellipse q <= q_max reaches
// the bounding box of the wave's points (exact ellipse/rectangle test).  Dropped terms are
// < exp(-q_max/2) of the term's scale (q_max = 36: 1.5e-8; see DESIGN.md "Cut-off").
//
// Correctness never depends on the grid domains: out-of-domain coordinates clamp to border
// cells and queries clamp the same (monotone) way; only speed depends on them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pigs {

constexpr int PLAN_MAX_LEVELS = 12;
constexpr uint32_t PLAN_SCAN_BLOCK = 256 * 4;     // counters scanned per workgroup (one uint4 per thread)

constexpr int PLAN_POINTS_PER_CELL = 63;   // target occupancy of a 64-lane sample cell: a regular grid then
                                            // yields cells of 49..64 points (never a second pass); for Poisson
                                            // counts the short second passes cost about what emptier cells would

// Per-workgroup partial bounding boxes written by the first build kernel (plain stores; same-
// address atomics serialise at ~10 ns each, so no atomics here) and reduced again by every
// workgroup of the second.
constexpr int PLAN_BBOX_BLOCKS = 128;
struct BoxPartial {
    float g[4];   // Gaussian centres: min x, min y, max x, max y (+-inf when empty)
    float s[4];   // sample points
};

struct GaussGrid {
    float ox, oy, inv_s0, s0;
};
struct SampleGrid {
    float ox, oy, inv_w;
    int nx, ny;   // even
};
// A sample point in sorted (cell) order: its coordinates and its index in the caller's array.
struct SPoint {
    float x, y;
    uint32_t m;
};

// Written once by the build (first bytes of the workspace), read by the sampling kernels.
struct PlanParams {
    GaussGrid gg;
    SampleGrid sg;
    uint32_t level_mask;
    uint32_t level_off[PLAN_MAX_LEVELS + 1];   // first counter of every level (copy of PlanLayout::level_off:
                                               // the sampling kernels index it by lane)
};

// Host+device view of the workspace (plain offsets; computed identically by every entry point
// from (N, M, c) alone).
struct PlanLayout {
    int64_t N, M;
    int c;
    int G0, L;                 // finest Gaussian grid is G0 x G0; L levels
    uint32_t gcells;           // total Gaussian cell counters over all levels (padded: level_shift)
    uint32_t scells_cap;       // capacity (upper bound) of sample cells, multiple of 4
    uint32_t sbase;            // index of the first sample-cell counter (gcells rounded up to a 128-B line)
    uint32_t ncounts;          // sbase + scells_cap
    uint32_t level_off[PLAN_MAX_LEVELS + 1];
    uint32_t scan_blocks;      // workgroups of the scan = ceil((ncounts + 1) / PLAN_SCAN_BLOCK)
    size_t off_params, off_boxes, off_counts, off_agg, off_starts, off_gkey, off_skey, off_rec, off_box, off_g2o, off_spts, off_gacc,
        total_bytes;
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Counter spacing of a level with `cells` cells: device-scope atomics on one 128-byte line
// serialise (~10 ns each; tools/ubench/atomics2.hip: 1 280 packed counters take 5 atomics/ns,
// 128 bytes apart 25/ns, the chip's ceiling), and the coarse levels have few cells but can
// hold most Gaussians -- so a level's counters are spread over at least 256 lines (up to one
// line per counter).  The padding counters stay zero; the scan sums over them unchanged.
__host__ __device__ inline int level_shift(uint32_t cells) {     // cells: a power of two
    const int sh = 13 - (31 - __builtin_clz(cells | 1u));
    return sh < 0 ? 0 : sh > 5 ? 5 : sh;
}

inline PlanLayout make_plan_layout(int64_t N, int64_t M, int c) {
    PlanLayout p{};
    p.N = N; p.M = M; p.c = c;
    // finest level: about 4 Gaussians per cell for a uniform cloud
    int g = 1;
    while ((int64_t)g * g * 4 < N && g < 1024) g <<= 1;
    p.G0 = g;
    p.L = 1;
    while ((g >> (p.L - 1)) > 1) ++p.L;
    uint32_t off = 0;
    for (int l = 0; l < p.L; ++l) {
        p.level_off[l] = off;
        const uint32_t gl = (uint32_t)(g >> l);
        off += (gl * gl) << level_shift(gl * gl);
    }
    p.level_off[p.L] = off;
    p.gcells = off;
    // sample cells: <= M/target + perimeter slack, rounded to whole 2x2 blocks
    int64_t cap = M / PLAN_POINTS_PER_CELL + 8 * (int64_t)(__builtin_sqrt((double)(M / PLAN_POINTS_PER_CELL + 1)) + 2) + 64;
    cap = (cap + 3) / 4 * 4;         // whole 2x2 blocks
    p.scells_cap = (uint32_t)cap;
    p.sbase = (p.gcells + 31) / 32 * 32;
    p.ncounts = p.sbase + p.scells_cap;
    size_t o = 0;
    p.scan_blocks = (p.ncounts + 1 + PLAN_SCAN_BLOCK - 1) / PLAN_SCAN_BLOCK;
    p.off_params = o;   o = align_up(o + sizeof(PlanParams), 256);
    p.off_boxes = o;    o = align_up(o + sizeof(BoxPartial) * PLAN_BBOX_BLOCKS, 256);
    // counters and the scan's per-workgroup aggregates are adjacent: zeroed together
    // (both arrays padded to whole scan blocks: the scan moves uint4s)
    p.off_counts = o;   o = align_up(o + sizeof(uint32_t) * (size_t)p.scan_blocks * PLAN_SCAN_BLOCK, 256);
    p.off_agg = o;      o = align_up(o + sizeof(uint64_t) * (size_t)p.scan_blocks, 256);
    p.off_starts = o;   o = align_up(o + sizeof(uint32_t) * (size_t)p.scan_blocks * PLAN_SCAN_BLOCK, 256);
    p.off_gkey = o;     o = align_up(o + sizeof(uint2) * (size_t)N, 256);      // {cell key, rank in cell}
    p.off_skey = o;     o = align_up(o + sizeof(uint2) * (size_t)M, 256);
    p.off_rec = o;      o = align_up(o + 32 * (size_t)N, 256);
    p.off_box = o;      o = align_up(o + 16 * (size_t)N, 256);
    p.off_g2o = o;      o = align_up(o + sizeof(uint32_t) * (size_t)N, 256);
    p.off_spts = o;     o = align_up(o + sizeof(SPoint) * (size_t)M, 256);
    p.off_gacc = o;     o = align_up(o + sizeof(float) * 8 * (size_t)N, 256);   // backward scratch [8][N]
    p.total_bytes = o;
    return p;
}

// Device view: raw pointers + the scalars kernels need.
struct PlanView {
    const PlanParams* params;
    const uint32_t* starts;       // [ncounts + 1] exclusive scan of the cell counters: Gaussian cells at
                                  // [0, gcells), sample cells at [sbase, sbase + scells_cap)
    const float4* rec;            // [2N] sorted records: {mux, muy, a, b}, {b, c, v0, v1}  (c <= 2)
    const float4* gbox;           // [N] sorted: {mux, muy, hx, hy} = centre and half extents of the q <= q_max ellipse
    const uint32_t* g2o;          // sorted Gaussian -> original index
    const SPoint* spts;           // sorted points: coordinates + original index
    uint32_t N, M;
    int G0, L;
    uint32_t sbase, scells_cap;
    uint32_t level_off[PLAN_MAX_LEVELS + 1];
    float q_max;
    float* gacc;                  // backward scratch: [8][N] sorted-order gradient sums
};

// ---- grid geometry derived (identically by every thread) from the header's bounding boxes ----

// box = {min x, min y, max x, max y}; min > max (+-inf) when there were no finite points
__device__ inline GaussGrid gauss_grid(const float* box, int G0) {
    GaussGrid g;
    const float x0 = box[0], y0 = box[1], x1 = box[2], y1 = box[3];
    float ext = fmaxf(x1 - x0, y1 - y0);
    if (!(ext > 0.f) || !(ext < 3.0e38f)) ext = 1.f;      // empty / single point / non-finite
    g.ox = (x1 >= x0) ? x0 : 0.f;
    g.oy = (y1 >= y0) ? y0 : 0.f;
    g.s0 = ext * 1.0001f / (float)G0;
    g.inv_s0 = 1.f / g.s0;
    return g;
}

__device__ inline SampleGrid sample_grid(const float* box, uint32_t M, uint32_t scells_cap) {
    SampleGrid s;
    const float x0 = box[0], y0 = box[1], x1 = box[2], y1 = box[3];
    float ex = x1 - x0, ey = y1 - y0;
    if (!(ex >= 0.f) || !(ex < 3.0e38f)) ex = 0.f;
    if (!(ey >= 0.f) || !(ey < 3.0e38f)) ey = 0.f;
    float emax = fmaxf(ex, ey);
    if (!(emax > 0.f)) emax = 1.f;
    ex = fmaxf(ex, emax * (1.f / 1024.f)) * 1.0001f;
    ey = fmaxf(ey, emax * (1.f / 1024.f)) * 1.0001f;
    s.ox = (x1 >= x0) ? x0 : 0.f;
    s.oy = (y1 >= y0) ? y0 : 0.f;
    float w = sqrtf(ex * ey * (float)PLAN_POINTS_PER_CELL / (float)(M > 0 ? M : 1));
    for (int it = 0; it < 16; ++it) {
        s.nx = ((int)ceilf(ex / w) + 1) & ~1;
        s.ny = ((int)ceilf(ey / w) + 1) & ~1;
        if (s.nx < 2) s.nx = 2;
        if (s.ny < 2) s.ny = 2;
        if ((uint64_t)s.nx * (uint64_t)s.ny <= scells_cap) break;
        w *= 1.25f;
    }
    if ((uint64_t)s.nx * (uint64_t)s.ny > scells_cap) { s.nx = 2; s.ny = 2; w = fmaxf(ex, ey); }  // never for sane input
    s.inv_w = 1.f / w;
    return s;
}

// 2x2-blocked cell id: the four cells of a block are consecutive (one workgroup = one block)
__device__ inline uint32_t sample_cell_id(int cx, int cy, int nx) {
    return (uint32_t)((((cy >> 1) * (nx >> 1) + (cx >> 1)) << 2) | ((cy & 1) << 1) | (cx & 1));
}

}  // namespace pigs

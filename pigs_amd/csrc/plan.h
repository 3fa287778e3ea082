// Binned ("plan") path: data structures shared by the preprocess and sampling kernels.
//
// What `GaussianSampler.preprocess(means, values, covariances, conics, samples)` builds
// (call sites model_pn.py:648,768,784; the reference's own native preprocess is not visible --
// this design is new).  Two workspaces, because the two halves change at different rates (the
// reference re-binds new Gaussians to an unchanged sample set every step of a roll-out,
// main_pn.py:317-324; a fixed collocation grid does the same in training):
//
//  SAMPLES workspace (built from `samples` alone; immutable afterwards, shared by any number of
//  plans):
//    * the points counting-sorted into square FINE CELLS of ~16 points (4 x 4 points of a regular
//      grid), cell ids ordered block (4x4 cells) > 2x2 > cell, each point keeping its index in the
//      caller's array.  Position in the sorted array is all the sampling kernels use:
//        TILE  = 64 consecutive sorted points = one wave   (a 2x2 block of cells on a grid)
//        GROUP = 16 consecutive sorted points = one 16-lane DPP row of that wave (one cell)
//      so every wave is full whatever the point distribution; only the tightness of the boxes
//      depends on it.
//
//  PLAN workspace (built from the Gaussians + a SAMPLES workspace, once per preprocess):
//    * Gaussians binned by their CENTRE into a multi-level uniform grid over the sample domain.
//      Level l has (G0 >> l)^2 square cells of side s0 * 2^l; a Gaussian whose q <= q_max ellipse
//      has half-extent R = max(hx, hy) goes to the lowest level with R <= s_l (top level: one
//      cell).  Within a level the Gaussians are counting-sorted by cell (row-major), so "all
//      Gaussians whose ellipse can reach a rectangle" is a handful of CONTIGUOUS ranges of packed
//      32-byte records: the cells within one cell of the rectangle, per level.
//    * per TILE the list of Gaussians whose ellipse reaches the bounding box of the tile's points
//      (exact ellipse / rectangle test), each entry carrying two 4-bit masks of the tile's GROUPS
//      whose own box it reaches (wide / narrow cut-off, see LIST_IDX_BITS; read by the backward); and per
//      GROUP the same Gaussians split by that mask into four packed index lists (read by the
//      forward: every DPP row streams its own list, no compaction at sampling time).  Forward,
//      backward and every further sample_* call of the same preprocess read these lists; none of
//      them traverses the grid again.  A list that would not fit its slab (very wide Gaussians) is
//      replaced by the grid's record ranges around the tile (every Gaussian in them is evaluated
//      for the whole tile), and by the single range [0, N) when even those do not fit.
//
// Cut-off: a (point, Gaussian) pair is evaluated iff the Gaussian's ellipse q <= q_max reaches
// the bounding box of the point's 16-point group.  Dropped terms are < exp(-q_max/2) of the
// term's scale (q_max = 36: 1.5e-8; see DESIGN.md "Cut-off").
//
// Correctness never depends on the grid domains: out-of-domain coordinates clamp to border
// cells and queries clamp the same (monotone) way; only speed depends on them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pigs {

constexpr int PLAN_MAX_LEVELS = 12;
constexpr int PLAN_BAR_WORDS = 2 * 17 + 1;   // two barriers of 17 words + the exit counter (plan.hip, grid_barrier)
constexpr uint32_t PLAN_SCAN_BLOCK = 256 * 4;     // counters scanned per workgroup (one uint4 per thread)

constexpr int TILE_POINTS = 64;            // one wave
constexpr int GROUP_POINTS = 16;           // one DPP row
constexpr int PLAN_POINTS_PER_CELL = 16;   // target occupancy of a fine sample cell: a regular res x res grid
                                            // (res a multiple of 4) then yields exactly 4 x 4 points per cell

// Per-workgroup partial bounding boxes of the sample points written by the first build kernel (plain
// stores; same-address atomics serialise at ~10 ns each, so no atomics here) and reduced again by
// every workgroup of the second.
constexpr int PLAN_BBOX_BLOCKS = 512;      // partials the samples workspace holds; the first launch has 256 or 512 workgroups

// tile list entries: sorted Gaussian index | WIDE group mask << 24 | NARROW group mask << 28.  Two cut-offs
// live in one plan: the forward evaluates a pair iff the q <= q_f ellipse reaches the point's group box
// (narrow mask, also what the group lists hold); the backward of gradients that arrive at second (or
// third) derivatives uses the q <= q_b ellipse (wide mask; q_b >= q_f): the conic gradient of such a
// term carries a q^2 prefactor AND its per-Gaussian sum nearly cancels (the plane integral of a
// derivative of a Gaussian vanishes), so the tail dropped at q = 36 was up to 2.5e-5 of the largest entry
// with thousands of points per Gaussian and one-signed incoming gradients (tests/test_fuzz_gpu.py,
// tools/fuzz_diag.py); at q_b = 40 (the host's default) it is 3.5e-6, the level of the dense kernel's own
// float32 accumulation error on the same cases (3.3e-6); 44 buys nothing more (3.4e-6).
constexpr int LIST_IDX_BITS = 24;
constexpr uint32_t LIST_IDX_MASK = (1u << LIST_IDX_BITS) - 1u;
constexpr int LIST_WIDE_SHIFT = 24, LIST_NARROW_SHIFT = 28;
// tile header, 8 words: [0] = count | mode << 30 (entries of the tile list, or record ranges), [1..4] =
// lengths of the four group lists (list mode)
constexpr int TILE_HDR_WORDS = 8;
constexpr uint32_t TILE_MODE_LIST = 0u, TILE_MODE_RANGES = 1u, TILE_MODE_GROUPS = 2u, TILE_MODE_POINTS = 3u;
// TILE_MODE_POINTS: a tile of points so far apart (the thin outskirts of a clustered cloud) that even its
// 16-point groups meet more Gaussians than a list holds although every single point meets few: no lists
// at all -- at sampling time the Gaussian grid is walked around every single point (the 3 x 3 cells of every
// occupied level) by HELPER workgroups behind the main ones of the same launch: the list build queues such
// tiles (PlanParams::n_points, `ptiles`), a helper wave takes four points at a time, 16 lanes per point
// (lane = candidate), so the few tiles of this kind spread over POINT_HELPER_BLOCKS x 4 waves instead of
// sitting in one wave each for the length of ~150 dependent loads per lane.  Chosen when the tile's box spans more than
// POINTS_MODE_MIN_CELLS finest Gaussian cells (the points are spread out; close points under wide Gaussians
// keep their lists / record ranges: the walk would meet more candidates than the lists hold) and its longest
// group list exceeds POINTS_MODE_MIN_LIST entries (a walk costs about that much: ~9 cells x a few Gaussians
// x the occupied levels of candidates per point) or does not fit at all -- provided the walk itself stays
// small (9 cells of every level at the level's mean occupancy <= 4 x the list length: very wide Gaussians sit
// in coarse levels whose 3 x 3 cells are most of the plan).
constexpr float POINTS_MODE_MIN_CELLS = 16.f;
constexpr uint32_t POINTS_MODE_MIN_LIST = 96u;
constexpr uint32_t POINT_HELPER_BLOCKS = 256u;
constexpr float POINTS_MODE_BLOCK_CELLS = 64.f;       // a 256-point block spread over more cells than this is not listed at all
constexpr int TILE_MODE_SHIFT = 30;
constexpr uint32_t TILE_COUNT_MASK = (1u << TILE_MODE_SHIFT) - 1u;

struct GaussGrid {
    float ox, oy, inv_s0, s0;
};
struct SampleGrid {
    float ox, oy, inv_w;
    int nx, ny;   // multiples of 4
};
// A sample point in sorted (cell) order: its coordinates and its index in the caller's array.
struct SPoint {
    float x, y;
    uint32_t m;
};

// Written once by the samples build (first bytes of the samples workspace).
struct SampleParams {
    float box[4];          // min x, min y, max x, max y of the finite sample coordinates (+-inf when none)
    SampleGrid sg;
    uint32_t scan_error;   // set when the scan's bounded spin ran out (never in a healthy run)
    // how ordered the caller's points are: over a sample of the build's waves, {runs of consecutive points
    // that share a fine cell, points}.  A lattice in row order has runs of ~4 (0.25 runs per point); shuffled
    // or random points have one run per point -- every point then pays its own returning atomic in the
    // one-pass build, and the next build of a point set of this size takes the coarse-bin path
    // (SAMPLES_COARSE_BINS below; the library reads this pair back without synchronising, plan.hip).
    uint32_t order_stat[2];
    // INDEX-TILED ("lattice") order (round 4).  A point set that arrives as an rf x rs lattice in row order (rf
    // points along the fast axis, both multiples of 8: meshgrid(indexing="xy").reshape(-1, 2),
    // test_gaussian_sampling.py:43-46, main_pn.py:317-324) needs no sort and no copy: its tile of a point is index
    // arithmetic (lattice_tile_xy / lattice_index below: tile = an 8 x 8 index patch, group = a 4 x 4 patch,
    // tiles in serpentine pair-rows so that 4 consecutive tiles from a multiple of 4 are a compact block) and the
    // sampling kernels read the caller's array through it (tile_point) -- no cell keys, no counters, no scan, no
    // scatter, no `spts`.  Correctness never depends on the points BEING a lattice (the lists are built from the
    // groups' real bounding boxes): the detection (the first backward step of the fast coordinate = the row length;
    // the largest steps between index neighbours along and across rows bound every index tile's extent, which must
    // stay within twice its share of the bounding box) only decides which order is compact.
    // lat_cand: {candidate rf, fast axis (0: x, 1: y)} from the first launch; lat: {rf, rs} when the points are
    // index-tiled, {0, 0} when they were sorted into cells.
    uint32_t lat_cand[2];
    uint32_t lat[2];
    // index-tiled order: the CALLER's sample array (device address).  The samples workspace then holds no copy of the
    // points -- the sampling kernels read them where the caller keeps them, through the index arithmetic -- so the
    // array must stay valid and unmodified for as long as the workspace is used (the hosts keep the tensor).
    uint64_t src;
    // two device-wide barriers among the samples' workgroups of the count launch (plan.hip, samples_sort_in_count:
    // points that were expected to be a lattice and are none are counted, scanned and scattered in that one launch);
    // zeroed by the first launch of every samples build
    uint32_t bar[2 * 17];
};

// ---- index-tiled order: position in `spts` <-> index in the caller's array -------------------
// tile -> (tx, ty) in a grid of ntx x nty tiles of 8 x 8 points: pair-rows of tiles taken alternately left to right
// and right to left, inside a pair-row column by column (so tiles 4j .. 4j+3 are the 2 x 2 block of tile columns
// 2j, 2j+1 -- or, where an odd ntx makes a block straddle two pair-rows, a 1 x 4 column at the turning edge); an
// odd nty leaves a single last row that continues the serpentine.
__host__ __device__ inline void lattice_tile_xy(uint32_t tile, uint32_t ntx, uint32_t nty, uint32_t& tx, uint32_t& ty) {
    const uint32_t pair = 2u * ntx, full = (nty >> 1) * pair;
    if (tile < full) {
        const uint32_t pr = tile / pair, k = tile - pr * pair, txs = k >> 1;
        ty = 2u * pr + (k & 1u);
        tx = (pr & 1u) ? ntx - 1u - txs : txs;
    } else {
        const uint32_t k = tile - full;
        ty = nty - 1u;
        tx = ((nty >> 1) & 1u) ? ntx - 1u - k : k;
    }
}
// lane -> the point's index in the caller's array: group g = lane / 16 is the 4 x 4 patch (g & 1, g >> 1) of the tile
__host__ __device__ inline uint32_t lattice_index(uint32_t tx, uint32_t ty, uint32_t lane, uint32_t rf) {
    const uint32_t g = lane >> 4, i = lane & 15u;
    const uint32_t col = tx * 8u + (g & 1u) * 4u + (i & 3u), row = ty * 8u + (g >> 1) * 4u + (i >> 2);
    return row * rf + col;
}

// Written once per plan build (first bytes of the plan workspace), read by the sampling kernels.
struct PlanParams {
    GaussGrid gg;
    uint32_t level_mask;
    uint32_t level_off[PLAN_MAX_LEVELS + 1];   // first counter of every level (copy of PlanLayout::level_off:
                                               // the traversal indexes it by lane)
    uint32_t scan_error;
    float q_f, q_b;        // the plan's two cut-offs (forward / backward of order >= 2 gradients), q_b >= q_f
    uint32_t bar[PLAN_BAR_WORDS];   // device-wide barriers of the one-launch Gaussian chain (zero between builds)
    uint32_t n_points;              // tiles in TILE_MODE_POINTS (queued in `ptiles` by the list build)
    // STRIPS (round 4; plan.hip, gauss_pack_part).  Gaussians that arrive in an order in which neighbours in the array are
    // neighbours in space (the reference lays them out on a meshgrid, model_pn.py:338-342, and training moves them by
    // fractions of a spacing) are not binned at all: records stay in the CALLER's order, every STRIP = 16 consecutive
    // ones are a strip with a bounding box (`pbox`), every 16 strips a super-strip (`sbox`), and the list build tests
    // a block of tiles against super-strip boxes, then strip boxes, then the strips' records (grid_walk.h,
    // traverse_strips) instead of walking grid cells -- no count, no scan, no scatter.  Always correct (a strip box bounds its Gaussians'
    // ellipses whatever their order); fast when the strips cover the samples' domain only a few times over:
    // strip_cover = sum of strip box areas / domain area, measured by every build (either kind) and remembered by the
    // library, which decides the next build's kind from it.
    float strip_cover;              // (adjacent to n_points: one copy for the library's memory)
    uint32_t strips;                // this build kept the caller's order (lists from strip boxes)
    uint32_t points_wanted;         // ... and met this many blocks / tiles of far-apart points, which a build through the
                                    // cells would have left to the per-point walk (TILE_MODE_POINTS needs the grid): the
                                    // library then goes back to the cells
};
constexpr uint32_t STRIP = 16;                      // Gaussians per strip = a row of 16 lanes of the wave that packs it
constexpr uint32_t SUPER_STRIPS = 16;               // strips per super-strip
constexpr uint32_t SUPER = STRIP * SUPER_STRIPS;    // = the 256 Gaussians of a workgroup of the packing pass

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- samples workspace ------------------------------------------------------------------------
// Two ways to sort the points into fine-cell order.  ONE PASS (points that arrive in runs sharing a cell: any
// lattice in row order): cell key + rank by one returning atomic per run, scan of the fine-cell counters,
// scatter.  COARSE BINS (points in no order: one global returning atomic per point, ~1 M of them at 20 per ns,
// and 12-byte scattered writes over the whole array): the fine cells are grouped into SAMPLES_COARSE_BINS
// contiguous id ranges (the cell path is spatially coherent, so a bin is a compact patch); a workgroup ranks
// its chunk of points inside each bin with LDS atomics and publishes one count per (bin, workgroup); the scan
// runs over that matrix; the scatter moves every point to its bin's segment of a temporary array (16-byte
// records, runs of a workgroup's points of one bin are contiguous); one workgroup per bin then counting-sorts
// its segment by fine cell in LDS into the final array -- writes that stay inside a ~50 KB window.  One more
// launch, two more passes over the points, no global atomics.
#ifndef PIGS_HIST_WGS
#define PIGS_HIST_WGS 1024
#endif
constexpr uint32_t SAMPLES_COARSE_BINS = 256;      // = threads of a build workgroup (one bin per thread where bins are walked)
constexpr uint32_t SAMPLES_MAX_HIST_WGS = PIGS_HIST_WGS;    // workgroups (chunks of the point array) of the coarse histogram
constexpr uint32_t SAMPLES_MAX_CELLS_PER_BIN = 12288;      // LDS counters of the per-bin sort (48 KB)
struct STmp {                  // a point on its way through the coarse-bin path
    float x, y;
    uint32_t m, id;            // index in the caller's array, fine cell id
};
struct SamplesLayout {
    int64_t M;
    uint32_t scells_cap;       // capacity (upper bound) of fine sample cells, multiple of 16
    uint32_t scan_blocks;      // workgroups of the scan = ceil((scells_cap + 1) / PLAN_SCAN_BLOCK)
    uint32_t ntiles;           // ceil(M / 64)
    // coarse-bin path
    uint32_t cells_per_bin;    // fine cell ids per coarse bin: bin = id / cells_per_bin < SAMPLES_COARSE_BINS
    uint32_t h_chunk;          // points per histogram workgroup (a multiple of 2048)
    uint32_t h_wgs;            // histogram workgroups (a multiple of 4: the matrix is whole scan blocks)
    uint32_t h_scan_blocks;    // = SAMPLES_COARSE_BINS * h_wgs / PLAN_SCAN_BLOCK
    size_t off_params, off_boxes, off_lat, off_counts, off_agg, off_starts, off_skey, off_spts, off_hist, off_hagg, off_hstarts,
        off_tmp, total_bytes;
};

inline SamplesLayout make_samples_layout(int64_t M) {
    SamplesLayout p{};
    p.M = M;
    // fine cells: <= M/target + perimeter slack, rounded to whole 4x4 blocks
    const int64_t base = M / PLAN_POINTS_PER_CELL + 1;
    int64_t cap = base + base / 4 + 16 * (int64_t)(__builtin_sqrt((double)base) + 4) + 64;
    cap = (cap + 15) / 16 * 16;
    p.scells_cap = (uint32_t)cap;
    p.scan_blocks = (p.scells_cap + 1 + PLAN_SCAN_BLOCK - 1) / PLAN_SCAN_BLOCK;
    p.ntiles = (uint32_t)((M + TILE_POINTS - 1) / TILE_POINTS);
    size_t o = 0;
    p.off_params = o;   o = align_up(o + sizeof(SampleParams), 256);
    p.off_boxes = o;    o = align_up(o + sizeof(float4) * PLAN_BBOX_BLOCKS, 256);
    p.off_lat = o;      o = align_up(o + sizeof(float4) * PLAN_BBOX_BLOCKS, 256);   // per-workgroup largest neighbour steps (index-tiled order)
    // counters and the scan's per-workgroup aggregates are adjacent: zeroed together
    p.off_counts = o;   o = align_up(o + sizeof(uint32_t) * (size_t)p.scan_blocks * PLAN_SCAN_BLOCK, 256);
    p.off_agg = o;      o = align_up(o + sizeof(uint64_t) * (size_t)p.scan_blocks, 256);
    p.off_starts = o;   o = align_up(o + sizeof(uint32_t) * (size_t)p.scan_blocks * PLAN_SCAN_BLOCK, 256);
    p.off_skey = o;     o = align_up(o + sizeof(uint2) * (size_t)M, 256);      // {cell id, rank in cell}
    p.off_spts = o;     o = align_up(o + sizeof(SPoint) * (size_t)M, 256);
    // coarse-bin path: the (bin, workgroup) count matrix, its scan, the temporary array
    p.cells_per_bin = (p.scells_cap + SAMPLES_COARSE_BINS - 1) / SAMPLES_COARSE_BINS;
    // chunks of 2 048 points (what the scatter's LDS stage holds) up to 2 M points, longer ones beyond
    const int64_t chunks = (M + 2047) / 2048;
    p.h_chunk = (uint32_t)((chunks + SAMPLES_MAX_HIST_WGS - 1) / SAMPLES_MAX_HIST_WGS) * 2048u;
    p.h_wgs = (uint32_t)(((M + p.h_chunk - 1) / p.h_chunk + 3) / 4 * 4);
    p.h_scan_blocks = SAMPLES_COARSE_BINS * p.h_wgs / PLAN_SCAN_BLOCK;
    p.off_hist = o;     o = align_up(o + sizeof(uint32_t) * (size_t)p.h_scan_blocks * PLAN_SCAN_BLOCK, 256);
    p.off_hagg = o;     o = align_up(o + sizeof(uint64_t) * (size_t)p.h_scan_blocks, 256);
    p.off_hstarts = o;  o = align_up(o + sizeof(uint32_t) * (size_t)p.h_scan_blocks * PLAN_SCAN_BLOCK, 256);
    p.off_tmp = o;      o = align_up(o + sizeof(STmp) * (size_t)M, 256);
    p.total_bytes = o;
    return p;
}

// ---- plan workspace ---------------------------------------------------------------------------
struct PlanLayout {
    int64_t N, M;
    int c;
    int G0, L;                 // finest Gaussian grid is G0 x G0; L levels
    uint32_t gcells;           // total Gaussian cell counters over all levels (padded: level_shift)
    uint32_t level_off[PLAN_MAX_LEVELS + 1];
    uint32_t scan_blocks;      // workgroups of the scan = ceil((gcells + 1) / PLAN_SCAN_BLOCK)
    uint32_t ntiles;           // ceil(M / 64)
    uint32_t list_cap;         // entries per list slab (one tile list + four group lists per tile), multiple of 16
    size_t off_pbox, off_sbox, off_parea;
    size_t off_params, off_counts, off_agg, off_starts, off_gkey, off_rec, off_box, off_g2o, off_gacc, off_hdr,
        off_ptiles, off_tlist, off_glist, off_stage, total_bytes;
};

// Counter spacing of a level with `cells` cells: device-scope atomics on one 128-byte line
// serialise (~10 ns each; tools/ubench/atomics2.hip: 1 280 packed counters take 5 atomics/ns,
// 128 bytes apart 25/ns, the chip's ceiling), and the coarse levels have few cells but can
// hold most Gaussians -- so a level's counters are spread over at least 256 lines (up to one
// line per counter).  The padding counters stay zero; the scan sums over them unchanged.
__host__ __device__ inline int level_shift(uint32_t cells) {     // cells: a power of two
    const int sh = 13 - (31 - __builtin_clz(cells | 1u));
    return sh < 0 ? 0 : sh > 5 ? 5 : sh;
}

// Slab of a list: room for every Gaussian when N is small (such a list can never overflow), 512
// entries beyond (5 slabs per tile: 160 bytes per sample point); what does not fit is kept as
// record ranges (header of this file).
inline uint32_t list_cap_for(int64_t N) {
    int64_t cap = (N + 15) / 16 * 16;
    if (cap < 16) cap = 16;
    if (cap > 512) cap = 512;
    return (uint32_t)cap;
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
    for (int l = p.L; l <= PLAN_MAX_LEVELS; ++l) p.level_off[l] = off;
    p.gcells = off;
    p.scan_blocks = (p.gcells + 1 + PLAN_SCAN_BLOCK - 1) / PLAN_SCAN_BLOCK;
    p.ntiles = (uint32_t)((M + TILE_POINTS - 1) / TILE_POINTS);
    p.list_cap = list_cap_for(N);
    size_t o = 0;
    p.off_params = o;   o = align_up(o + sizeof(PlanParams), 256);
    p.off_counts = o;   o = align_up(o + sizeof(uint32_t) * (size_t)p.scan_blocks * PLAN_SCAN_BLOCK, 256);
    p.off_agg = o;      o = align_up(o + sizeof(uint64_t) * (size_t)p.scan_blocks, 256);
    p.off_starts = o;   o = align_up(o + sizeof(uint32_t) * (size_t)p.scan_blocks * PLAN_SCAN_BLOCK, 256);
    p.off_gkey = o;     o = align_up(o + sizeof(uint2) * (size_t)N, 256);      // {cell key, rank in cell}
    p.off_rec = o;      o = align_up(o + 32 * ((size_t)N + 1), 256);           // + the all-zero record N
    p.off_box = o;      o = align_up(o + 16 * (size_t)N, 256);
    p.off_g2o = o;      o = align_up(o + sizeof(uint32_t) * (size_t)N, 256);
    p.off_pbox = o;     o = align_up(o + sizeof(float4) * (((size_t)N + STRIP - 1) / STRIP), 256);     // strip boxes {min x, min y, max x, max y}
    p.off_sbox = o;     o = align_up(o + sizeof(float4) * (((size_t)N + SUPER - 1) / SUPER), 256);     // super-strip boxes
    p.off_parea = o;    o = align_up(o + sizeof(float) * (((size_t)N + STRIP - 1) / STRIP), 256);      // strip box areas / domain area
    p.off_gacc = o;     o = align_up(o + sizeof(float) * 8 * (size_t)N, 256);   // backward scratch [N][8]
    p.off_hdr = o;      o = align_up(o + sizeof(uint32_t) * TILE_HDR_WORDS * (size_t)p.ntiles, 256);
    p.off_ptiles = o;   o = align_up(o + sizeof(uint32_t) * (size_t)p.ntiles, 256);           // queue of the TILE_MODE_POINTS tiles
    p.off_tlist = o;    o = align_up(o + sizeof(uint32_t) * (size_t)p.ntiles * p.list_cap, 256);
    p.off_glist = o;    o = align_up(o + sizeof(uint32_t) * 4 * (size_t)p.ntiles * p.list_cap, 256);
    // one 32-byte record per sample point, in the CALLER's order: what points that arrive in no order send their
    // outputs / fetch their incoming gradients through (STAGE_* below); untouched for lattices
    p.off_stage = o;    o = align_up(o + 32 * (size_t)M, 256);
    p.total_bytes = o;
    return p;
}

// Device view of a samples workspace.
struct SamplesView {
    const SampleParams* params;
    const SPoint* spts;           // sorted points: coordinates + original index (not in index-tiled order)
    uint32_t M, ntiles;
};

// How the points of a samples workspace are ordered, read once per wave (wave-uniform).
struct PointOrder {
    const float2* src;            // index-tiled: the caller's array
    uint32_t rf, ntx, nty;        // rf == 0: sorted into cells (SamplesView::spts)
};
__device__ inline PointOrder point_order(const SamplesView& sv) {
    PointOrder po;
    po.rf = sv.params->lat[0];
    const uint32_t rs = sv.params->lat[1];
    po.ntx = po.rf >> 3;
    po.nty = rs >> 3;
    po.src = (const float2*)(uintptr_t)sv.params->src;
    return po;
}
// the point at position `lane` of tile `tile` (tile wave-uniform in the tile kernels: its (tx, ty) stay scalar); a
// position behind the last point repeats the last one (never stored)
__device__ inline SPoint tile_point(const SamplesView& sv, const PointOrder& po, uint32_t tile, uint32_t lane) {
    if (po.rf != 0u) {            // M is a multiple of 64 here: every position holds a point
        uint32_t tx, ty;
        lattice_tile_xy(tile, po.ntx, po.nty, tx, ty);
        SPoint sp;
        sp.m = lattice_index(tx, ty, lane, po.rf);
        const float2 p = po.src[sp.m];
        sp.x = p.x; sp.y = p.y;
        return sp;
    }
    const uint32_t m = tile * TILE_POINTS + lane;
    return sv.spts[m < sv.M ? m : sv.M - 1];
}

// Device view of a plan workspace: raw pointers + the scalars kernels need.
struct PlanView {
    const PlanParams* params;
    const uint32_t* starts;       // [gcells + 1] exclusive scan of the Gaussian cell counters
    const float4* rec;            // [2N + 2] sorted records: {mux, muy, a, b}, {c, v0, v1, 0}  (c <= 2); record N is all zero
    const float4* gbox;           // [N] sorted: {mux, muy, hx, hy} = centre and half extents of the q <= q_max ellipse
    const uint32_t* g2o;          // sorted Gaussian -> original index
    const uint32_t* hdr;          // [ntiles][TILE_HDR_WORDS]
    const uint32_t* tlist;        // [ntiles][list_cap]     tile lists (entries with group masks) / record ranges
    const uint32_t* glist;        // [ntiles][4][list_cap]  group lists (sorted Gaussian indices)
    const uint32_t* ptiles;       // [params->n_points]     the tiles in TILE_MODE_POINTS
    uint32_t N, list_cap;
    int G0, L;
    uint32_t level_off[PLAN_MAX_LEVELS + 1];
    float q_max;                  // list build only: the WIDE cut-off max(q_f, q_b) (the sampling kernels read params->q_f / q_b)
    float* gacc;                  // backward scratch: [N][8] sorted-order gradient sums, one 32-byte row per Gaussian
    const float4* pbox;           // [ceil(N / 16)] strip boxes (meaningful when params->strips)
    const float4* sbox;           // [ceil(N / 256)] super-strip boxes
    // Staging for points in no order (null: not in this launch).  A tile of such points sends its outputs to three
    // arrays through the points' original indices: three scattered 4 / 8 / 16-byte stores per point, each of which
    // costs the memory a whole 32-byte sector (96 MB written for 28 MB of outputs at 1 M points: forward 54 us where
    // a lattice takes 32), and the backward fetches the incoming gradients the same way.  Staged, a point writes ONE
    // 32-byte record at its original index -- {u, u_x, u_y, H_xx, H_xy, H_yx, H_yy, 0}, or {u, u_x, u_y, lap} --
    // and a streaming launch behind the forward deals the records out to the output arrays (stage_to_outputs_kernel);
    // the backward is preceded by the opposite launch (gradients_to_stage_kernel) and reads one record per point.
    // c = 1, orders (0, 1, 2) or (0, 1, trace); chosen by the library when it remembers the point set's size as
    // unordered (plan.hip, samples_take_coarse).
    float4* stage;
};

// ---- grid geometry derived (identically by every thread) from the sample bounding box ----

// box = {min x, min y, max x, max y}; min > max (+-inf) when there were no finite points.
// The Gaussian grid spans the sample domain plus 1/16 of its extent on every side: centres
// further out only matter through ellipses that reach back in, and clamp to the border cells.
__device__ inline GaussGrid gauss_grid(const float* box, int G0) {
    GaussGrid g;
    const float x0 = box[0], y0 = box[1], x1 = box[2], y1 = box[3];
    float ext = fmaxf(x1 - x0, y1 - y0);
    if (!(ext > 0.f) || !(ext < 3.0e38f)) ext = 1.f;      // empty / single point / non-finite
    const float pad = ext * 0.0625f;
    g.ox = ((x1 >= x0) ? x0 : 0.f) - pad;
    g.oy = ((y1 >= y0) ? y0 : 0.f) - pad;
    g.s0 = ext * 1.1251f / (float)G0;
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
    for (int it = 0; it < 24; ++it) {
        s.nx = ((int)ceilf(ex / w) + 3) & ~3;
        s.ny = ((int)ceilf(ey / w) + 3) & ~3;
        if (s.nx < 4) s.nx = 4;
        if (s.ny < 4) s.ny = 4;
        if ((uint64_t)s.nx * (uint64_t)s.ny <= scells_cap) break;
        w *= 1.25f;
    }
    if ((uint64_t)s.nx * (uint64_t)s.ny > scells_cap) { s.nx = 4; s.ny = 4; w = fmaxf(ex, ey); }  // never for sane input
    s.inv_w = 1.f / w;
    return s;
}

// Cell ids follow a path through the grid whose consecutive cells are neighbours (almost) everywhere:
// tiles and groups are runs of 64 / 16 consecutive sorted points, and where cells do not hold
// exactly 16 points (any point set but a lattice) a run straddles consecutive cells -- with a
// row-major order of blocks that meant, at the end of every block row, groups whose box spanned the
// whole domain (random points at C3: 56 tiles fell back to record ranges and one launch took
// 395 us instead of 45).  The path: blocks of 4 x 4 cells, block rows taken in turn left to right
// and right to left; inside a block the order-2 Hilbert curve from the bottom corner the path
// enters at to the other bottom corner (mirrored in the right-to-left rows).  Any 4 consecutive
// cells starting at a multiple of 4 are a 2 x 2 quad, any 16 at a multiple of 16 a block: on a
// lattice with 4 x 4 points per cell a tile is an 8 x 8 patch, four consecutive tiles 16 x 16.
__device__ inline uint32_t sample_cell_id(int cx, int cy, int nx) {
    const int nbx = nx >> 2, bx = cx >> 2, by = cy >> 2;
    const bool back = (by & 1) != 0;
    const uint32_t block = (uint32_t)(by * nbx + (back ? nbx - 1 - bx : bx));
    const int lx = back ? 3 - (cx & 3) : (cx & 3), ly = cy & 3;
    // Hilbert index of (lx, ly) in the 4 x 4 block, one nibble per cell, cell (lx, ly) at nibble ly * 4 + lx:
    // (0,0) (1,0) (1,1) (0,1) (0,2) (0,3) (1,3) (1,2) (2,2) (2,3) (3,3) (3,2) (3,1) (2,1) (2,0) (3,0)
    const uint32_t in = (uint32_t)(0xA965B874CD23FE10ull >> (4 * (ly * 4 + lx))) & 15u;
    return (block << 4) | in;
}

}  // namespace pigs

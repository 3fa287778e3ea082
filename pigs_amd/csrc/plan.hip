// Binned (culled) sampler: preprocess (samples build + plan build) + forward + backward, float32, d = 2.
// Data structures and the cut-off rule: plan.h.  Per-pair arithmetic: pair_math.h.
//
// Samples build = 4 launches (bbox partials -> cell key + rank -> scan -> scatter); plan build =
// the same chain for the Gaussians (sharing the launches of a samples build that runs with it) +
// one launch in which every wave walks the Gaussian grid once for four consecutive 64-point tiles
// and writes their lists (tile list with group masks for the backward, four group lists for the
// forward).  No memset, no host synchronisation, static memory.
// Sampling kernels: one wave = one tile of 64 consecutive sorted points (lane = point), one DPP row =
// one 16-point group.  Forward: every row streams its own group list, gathers the 32-byte records
// into its LDS queue and the rows are evaluated together -- in one instruction every row works on
// its OWN Gaussian, read from LDS with a row-uniform address.  Backward: the tile list 64 entries at
// a time, split by the group masks into four per-row lists (ballot + mbcnt), rows reduced by a
// transposing DPP fold into an LDS table, one atomic per entry and value.  No workgroup barriers;
// HBM traffic is the point stream (sorted points in, outputs out through the points' original
// indices) plus list and record reads that mostly hit L2.
//
// Build-time knobs (defaults measured on MI355X): PIGS_FWD_WAVES, PIGS_FWD_UNROLL,
// PIGS_GROUP_CAP, PIGS_BWD_WAVES, PIGS_BWD_STEP, PIGS_BWD_SPREAD, PIGS_LISTS_TPW, PIGS_TRAV_STEPS,
// PIGS_XCD_CHUNK.
#ifndef PIGS_FWD_STAGGER
#define PIGS_FWD_STAGGER 0
#endif
#ifndef PIGS_FWD_STAGGER_PHASES
#define PIGS_FWD_STAGGER_PHASES 2
#endif
#ifndef PIGS_FWD_STAGGER_FIRST
#define PIGS_FWD_STAGGER_FIRST 2048
#endif
#include "pair_math.h"
#include "plan.h"
#include "grid_walk.h"
#include "launch.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <mutex>
#include <unordered_map>

#ifndef PIGS_FWD_WAVES
#define PIGS_FWD_WAVES 8      // waves per SIMD the forward kernel's register budget is held to
#endif
#ifndef PIGS_FWD_WG_WAVES
#define PIGS_FWD_WG_WAVES 4   // waves (= tiles) per workgroup of the forward kernel
#endif
#ifndef PIGS_FWD_UNROLL
#define PIGS_FWD_UNROLL 2     // list rows evaluated per loop iteration
#endif
#ifndef PIGS_FUSED_BUILD
#define PIGS_FUSED_BUILD 0    // 1: count + scan + scatter of the Gaussians in ONE launch when the samples half is reused
                              // (plan_gauss_build_kernel).  Measured SLOWER than the three launches it replaces: warm step
                              // 59.4 us with three launches, 80.4 with release / acquire barriers, 66.3 with relaxed barriers
                              // + L2-bypassing accesses, 70.4 with per-XCD arrival counters -- a device-wide barrier on 8
                              // mutually incoherent L2s costs more than the ~3.7 us of a kernel boundary.  Kept as the record.
#endif
#ifndef PIGS_BWD_BLOCK
#define PIGS_BWD_BLOCK 0      // 1: the backward walks BLOCK lists (one wave = four tiles, one atomic per entry and value for all
                              // four: 2.2x fewer atomics); 0: tile lists (one wave = one tile).  Measured on one box, C3,
                              // kappa 0.5, kernel + unpermute, q_b = 36 / 40 / 44: block 92.2 / 97.6 / 103.7 us, tile 76.5 /
                              // 84.5 / 87.4 -- with gradients at orders 0..2 the kernel is bound by its arithmetic and the
                              // serial life of its waves, and 4 096 four-tile waves (one generation) overlap worse than
                              // 16 384 one-tile waves; only the light order-0 backward gains (66 vs 71-76 us).  Kept as the
                              // record (tests pass with either setting).
#endif
#ifndef PIGS_BWD_WAVES
#define PIGS_BWD_WAVES 6      // waves per SIMD the backward kernel's register budget is held to (its LDS allows 6 workgroups per CU)
#endif

namespace pigs {

// ------------------------------------------------------------------------------------------
// preprocess kernels
// ------------------------------------------------------------------------------------------
struct BuildArgs {
    // samples side
    SampleParams* sparams;
    float4* sboxes;       // [PLAN_BBOX_BLOCKS] per-workgroup partial boxes {min x, min y, max x, max y}
    float4* slat;         // [PLAN_BBOX_BLOCKS] per-workgroup largest neighbour steps {along x, along y, across x, across y} (index-tiled order)
    int no_lattice;       // PIGS_LATTICE=0: never index-tiled (tests, A/B)
    uint32_t rf_hint;     // the row length the last completed build of a point set of this size found (0: none): the first
                          // launch loads its tiles for that length while it is still verifying it
    uint32_t* scounts;    // [s_scan_blocks * PLAN_SCAN_BLOCK] fine-cell counters, followed by the scan aggregates
    unsigned long long* sagg;
    uint32_t* sstarts;
    uint2* skey;          // per point {cell id, rank inside the cell}
    SPoint* spts;
    const float* samples;
    uint32_t M, scells_cap, s_scan_blocks, s_zero_words;
    uint32_t s_blocks;    // blocks of 1 024 points of the one-pass count (its sample workgroups: one per block, or fewer, striding)
    uint32_t* szero;      // what the bbox launch zeroes for the samples side (s_zero_words): counters + scan aggregates
    // coarse-bin path of the samples build (plan.h): scounts / sagg / sstarts then are the (bin, workgroup) count
    // matrix, its scan aggregates and its scan
    int coarse;
    uint32_t cells_per_bin, h_chunk, h_wgs;
    STmp* tmp;
    // plan side
    PlanParams* params;
    uint32_t* counts;     // [scan_blocks * PLAN_SCAN_BLOCK] Gaussian cell counters, followed by the scan aggregates
    unsigned long long* agg;   // [scan_blocks] {1 << 32 | workgroup total}, zero before the scan
    uint32_t* starts;     // [gcells + 1] exclusive scan of counts
    uint2* gkey;          // per Gaussian {cell key, rank inside the cell}
    float4* rec;
    float4* gbox;
    float* gacc;
    uint32_t* g2o;
    const float* means;
    const float* conics;
    const float* values;
    uint32_t N;
    int c, G0, L;
    uint32_t scan_blocks, zero_words;
    uint32_t level_off[PLAN_MAX_LEVELS + 1];
    float q_max;          // the WIDE cut-off max(q_f, q_b): levels and candidate boxes are sized for it
    float q_f, q_b;
    // which halves this build covers
    int do_samples, do_plan;
    int no_lookback;      // test hook: the scan's workgroups never publish; every look-back recomputes
    int zero_gacc;        // the backward's scratch is not known to be zero (a workspace that is not PIGS_BUILD_PLAN_WS_CLEAN)
    uint32_t bbox_blocks; // workgroups of the first launch = partials in sboxes / slat: 256, or 512 from 2^19 points on
    // round 4, "the Gaussians one launch ahead" (run_build): with a lattice expected and the box of the last build of this
    // size known, the Gaussian chain does not wait for the first launch -- launch 1 = box of the samples || count of the
    // Gaussians on the REMEMBERED box (grid domains steer the quality of the binning, never the result), launch 2 = scan
    // of the Gaussian cells || the samples' lattice decision (and their count, should they be no lattice), launch 3 =
    // scatter of the Gaussians (|| scan + scatter of the samples by scan_pick, should they be no lattice): one launch
    // fewer in front of the tile lists.
    int ahead;
    float hint_box[4];
    int s_scan_in_scatter;  // launch 3 of the chain above: no scan launch ran for the samples
    int sort_in_count;      // ... or no launch 3 at all (ahead && strips: it would hold nothing but the samples' fall-back):
                            // points that are no lattice are counted, scanned AND scattered by the samples' workgroups
                            // of launch 2, behind two device-wide barriers among them (samples_sort_in_count)
    // STRIPS (plan.h, PlanParams::strips): the Gaussians keep the caller's order -- gauss_pack_part instead of count,
    // scan and scatter; `parea` is written by builds of either kind (the statistic the library decides from)
    int strips;
    float4* pbox;
    float4* sbox;
    float* parea;
};

__device__ __forceinline__ void zero_words(uint32_t* p, uint32_t words, uint32_t bid, uint32_t nb) {
    uint4* p4 = (uint4*)p;
    for (uint32_t i = bid * blockDim.x + threadIdx.x; i < words / 4; i += nb * blockDim.x) p4[i] = make_uint4(0, 0, 0, 0);
}
__device__ __forceinline__ void zero_words(uint32_t* p, uint32_t words) { zero_words(p, words, blockIdx.x, gridDim.x); }

// Bounding boxes with the DPP modifier fused into the min / max (hipcc emits v_mov_dpp + a
// canonicalising v_max + v_min per step from the builtin form: 4x the instructions).  The four
// reductions are independent chains and are interleaved step by step, so the two wait states a
// DPP read needs after the VALU write of its source are filled by the other three chains: one
// s_nop at the head instead of one per step (an s_nop costs an issue slot like a VALU
// instruction).  row_box_dpp leaves in every lane the box of the lane's own 16-lane row;
// wave_box_dpp continues from there to the box of the wave, broadcast from lane 63.
#define PIGS_BOX_STEP(MOD)                                \
    "v_min_f32_dpp %0, %0, %0 " MOD " bank_mask:0xf\n\t" \
    "v_max_f32_dpp %1, %1, %1 " MOD " bank_mask:0xf\n\t" \
    "v_min_f32_dpp %2, %2, %2 " MOD " bank_mask:0xf\n\t" \
    "v_max_f32_dpp %3, %3, %3 " MOD " bank_mask:0xf\n\t"
__device__ __forceinline__ void row_box_dpp(float& x0, float& x1, float& y0, float& y1) {
    asm volatile("s_nop 1\n\t"
                 PIGS_BOX_STEP("quad_perm:[1,0,3,2] row_mask:0xf")
                 PIGS_BOX_STEP("quad_perm:[2,3,0,1] row_mask:0xf")
                 PIGS_BOX_STEP("row_half_mirror row_mask:0xf")
                 PIGS_BOX_STEP("row_mirror row_mask:0xf")
                 "s_nop 1"
                 : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));
}
__device__ __forceinline__ void wave_box_from_rows_dpp(float& x0, float& x1, float& y0, float& y1) {
    asm volatile("s_nop 1\n\t"
                 PIGS_BOX_STEP("row_bcast:15 row_mask:0xa")
                 PIGS_BOX_STEP("row_bcast:31 row_mask:0xc")
                 "s_nop 1"
                 : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));
    x0 = readlane_f(x0, 63); x1 = readlane_f(x1, 63); y0 = readlane_f(y0, 63); y1 = readlane_f(y1, 63);
}

// Launch 1 of a samples build (PLAN_BBOX_BLOCKS workgroups): zero the cell counters (of the
// plan too, when one is built alongside); per-workgroup bounding box of the sample points, written as a plain
// partial.  And what the INDEX-TILED order (plan.h, SampleParams::lat) is decided from, in the same streaming pass:
// every workgroup finds the first index at which the fast coordinate steps backwards -- the row length rf of a
// lattice in row order (a search of the first 2 049 points, of 16 385 when those hold none; all workgroups read the
// same few KB) -- and, for a row length whose rf and M / rf are multiples of 8, the largest steps between
// neighbours: along a row (point i against i - 1, row ends left out) and across rows (point i against i - rf),
// per coordinate.  An 8 x 8 index tile is at most 7 (along + across) wide and high: the next launch holds that
// against the bounding box and decides whether index tiles are compact -- then nothing is sorted and NOTHING IS
// COPIED: the sampling kernels read the caller's array through the index arithmetic -- or the points go through
// the sort.  The pass runs on the row length of the last build of this size (the library's memory, `rf_hint`)
// while the search is still in flight, and is repeated only when the search finds another one.
constexpr uint32_t BBOX_THREADS = 256;      // (1 024-thread workgroups -- 4 096 waves to launch -- cost small point sets ~3 us)
constexpr uint32_t LAT_SEARCH0 = 2048, LAT_SEARCH1 = 16384;
constexpr uint32_t BBOX_WIDE_POINTS = 1u << 19;
__device__ __forceinline__ bool lattice_shape_ok(uint32_t rf, uint32_t n) {
    const uint32_t rs = rf ? n / rf : 0u;
    return rf >= 8u && (rf & 7u) == 0u && rs * rf == n && (rs & 7u) == 0u && n >= 64u;
}
struct GaussLoad;
__device__ __forceinline__ void gauss_count_ahead(const BuildArgs& a, uint32_t bid);
__global__ __launch_bounds__(256) void samples_bbox_kernel(BuildArgs a) {
    __shared__ float sh[4][8];
    __shared__ uint32_t shk[4];
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (a.ahead && blockIdx.x >= a.bbox_blocks) {       // (block-uniform) the Gaussians one launch ahead, on the remembered box
        gauss_count_ahead(a, blockIdx.x - a.bbox_blocks);
        return;
    }
    const uint32_t nblocks = a.bbox_blocks;             // (the launch may hold the Gaussians' workgroups behind these)
    zero_words(a.szero, a.s_zero_words, blockIdx.x, nblocks);
    if (blockIdx.x == 0 && tid < 2) a.sparams->order_stat[tid] = 0u;
    if (blockIdx.x == 0 && tid < 2 * 17) a.sparams->bar[tid] = 0u;
    if (a.do_plan && !a.ahead) {        // (ahead: a clean workspace, and its counters are being counted in this very launch)
        zero_words(a.counts, a.zero_words, blockIdx.x, nblocks);
        if (a.zero_gacc) zero_words((uint32_t*)a.gacc, 8u * a.N, blockIdx.x, nblocks);
    }
    if (a.do_plan && blockIdx.x == 0 && tid < PLAN_BAR_WORDS) a.params->bar[tid] = 0u;
    const float INF = __builtin_huge_valf();
    float x0 = INF, y0 = INF, x1 = -INF, y1 = -INF;
    auto take = [&](float x, float y) {
        if (fabsf(x) < INF) { x0 = fminf(x0, x); x1 = fmaxf(x1, x); }
        if (fabsf(y) < INF) { y0 = fminf(y0, y); y1 = fmaxf(y1, y); }
    };
    const float2* pts = (const float2*)a.samples;
    const float4* pts2 = (const float4*)a.samples;
    const uint32_t n = a.M;
    const uint32_t npair = n / 2;                 // float4 = two points
    const uint32_t stride = nblocks * BBOX_THREADS;
    // the largest neighbour steps (NaN / inf coordinates: +inf, never compact)
    float ax = 0.f, ay = 0.f, bx = 0.f, by = 0.f;
    auto step = [&](float& s, float u, float v) {
        const float d = fabsf(u - v);
        s = d == d ? fmaxf(s, d) : INF;
    };
    // one streaming pass, 8 pairs (float4 = two points) per thread in flight: the box (first time only) and, with a
    // row length rf, the neighbour steps.  With a row length a thread takes the SAME column pair of 8 consecutive rows:
    // the point above is then its own previous load (one extra load for the first of its rows), the point to the right
    // its neighbour lane's (one lane of the wave loads it) -- 9 + 1 loads where point, right and upper neighbour of
    // every pair were 24 (first launch 8.6 -> 6.x us at 1024^2).
    auto pass = [&](auto pbc, uint32_t rf, bool box) {
        constexpr int PB = decltype(pbc)::value;
        ax = ay = bx = by = 0.f;
        if (rf == 0u) {
            for (uint32_t i = blockIdx.x * BBOX_THREADS + tid; i < npair; i += PB * stride) {
                float4 v[PB];
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const uint32_t j = i + k * stride;
                    v[k] = pts2[j < npair ? j : i];
                }
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    if (i + k * stride >= npair) break;
                    if (box) { take(v[k].x, v[k].y); take(v[k].z, v[k].w); }
                }
            }
        } else {
            const uint32_t half = rf >> 1;            // pairs per row (rf is even: a float4 never straddles a row end)
            const uint32_t items = (n / rf / PB) * half;      // (column pair, block of 8 rows): rs is a multiple of 8
            for (uint32_t g = blockIdx.x * BBOX_THREADS + tid; g - (uint32_t)lane < items; g += stride) {      // whole waves stay in (the shuffles)
                const bool in = g < items;
                const uint32_t gg = in ? g : items - 1u;
                const uint32_t rb = gg / half, c = gg - rb * half;
                const uint32_t j0 = rb * PB * half + c;
                float4 v[PB];
#pragma unroll
                for (int k = 0; k < PB; ++k) v[k] = pts2[j0 + (uint32_t)k * half];
                const float4 up0 = pts2[rb ? j0 - half : j0];
                const bool last = c == half - 1u;                      // the pair at the end of a row: its right neighbour starts the next row
                const bool edge = lane == 63 && !last;                 // right neighbour in another wave: loaded
                float2 nxl[PB];
#pragma unroll
                for (int k = 0; k < PB; ++k) nxl[k] = edge ? pts[2u * (j0 + (uint32_t)k * half) + 2u] : make_float2(0.f, 0.f);
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const float rx = __shfl_down(v[k].x, 1), ry = __shfl_down(v[k].y, 1);
                    if (!in) continue;
                    if (box) { take(v[k].x, v[k].y); take(v[k].z, v[k].w); }
                    step(ax, v[k].z, v[k].x); step(ay, v[k].w, v[k].y);
                    if (!last) { step(ax, edge ? nxl[k].x : rx, v[k].z); step(ay, edge ? nxl[k].y : ry, v[k].w); }
                    const float4 u = k ? v[k > 0 ? k - 1 : 0] : up0;
                    if (k || rb) {
                        step(bx, v[k].x, u.x); step(by, v[k].y, u.y);
                        step(bx, v[k].z, u.z); step(by, v[k].w, u.w);
                    }
                }
            }
        }
        if (box && (n & 1u) && blockIdx.x == 0 && tid == 0) take(pts[n - 1].x, pts[n - 1].y);
    };
    // ---- the candidate row length.  The fast axis is the one along which the first two points differ most, its
    // direction the sign of that step; a row ends where the fast coordinate steps the other way (a jittered lattice
    // keeps its rows as long as the jitter stays below half a step).  key = that first index (NONE: none found).
    // Thread t looks at the eight steps from point 8 t on; the loads are issued HERE and looked at behind the pass:
    // one memory round trip for the launch.
    constexpr uint32_t NONE = 0xffffffffu;
    // With a row length remembered (rf_hint) there is nothing to search for: the pass below runs on it, and a point set
    // whose rows are not that long fails it (the row ends it did not expect are neighbour steps as wide as the domain) --
    // it is sorted this once, the memory forgets the row length, the next build searches again.  (The search walks up to
    // 16 385 points in every workgroup: rows of 2 880 points -- a rank's shard of bench.py's grid for 8 GPUs -- made the
    // first launch 21 us instead of 8.)
    const bool hinted = !a.no_lattice && n >= 64u && lattice_shape_ok(a.rf_hint, n);
    const bool search = !a.no_lattice && n >= 64u && !hinted;
    float2 e0 = make_float2(0.f, 0.f), e1 = e0, q[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) q[k] = e0;
    if (search) {
        e0 = pts[0]; e1 = pts[1];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if (8u * tid + (uint32_t)k < n) q[k] = pts[8u * tid + (uint32_t)k];
    }
    // the pass on the remembered row length (or, without one, for the box alone)
    const uint32_t hf = hinted ? a.rf_hint : 0u;
    if (hinted) { e0 = pts[0]; e1 = pts[1]; }        // (the fast axis: from the first two points, as the search takes it)
    // (from 2^19 points on, and with no row length expected, the launch has twice the workgroups and a thread half the
    // rows: BuildArgs::bbox_blocks)
    const bool wide = nblocks > 256u;
    if (wide) pass(std::integral_constant<int, 4>{}, hf, true); else pass(std::integral_constant<int, 8>{}, hf, true);
    const uint32_t axis = fabsf(e1.y - e0.y) > fabsf(e1.x - e0.x) ? 1u : 0u;
    const float dir = (axis ? e1.y - e0.y : e1.x - e0.x) < 0.f ? -1.f : 1.f;
    auto backward = [&](float2 p, float2 r) { return (axis ? r.y - p.y : r.x - p.x) * dir < 0.f; };
    auto block_min = [&](uint32_t k) -> uint32_t {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) k = min(k, (uint32_t)__shfl_xor((int)k, o));
        __syncthreads();
        if (lane == 0) shk[wave] = k;
        __syncthreads();
        return min(min(shk[0], shk[1]), min(shk[2], shk[3]));
    };
    uint32_t key = NONE;
    if (search) {
#pragma unroll
        for (int k = 7; k >= 0; --k)
            if (8u * tid + (uint32_t)k + 1u < n && backward(q[k], q[k + 1])) key = 8u * tid + (uint32_t)k;
        key = block_min(key);
        if (key == NONE && n > LAT_SEARCH0 + 1u) {
            uint32_t k2 = NONE;
            for (uint32_t i = LAT_SEARCH0 + tid; i < LAT_SEARCH1 && i + 1u < n; i += BBOX_THREADS)
                if (backward(pts[i], pts[i + 1u])) k2 = min(k2, i);
            key = block_min(k2);
        }
    }
    const uint32_t rf = hinted ? hf : key == NONE ? 0u : key + 1u;
    const bool cand = lattice_shape_ok(rf, n);      // block-uniform
    if (cand && rf != hf) {                         // first build of a size, or the points changed shape
        if (wide) pass(std::integral_constant<int, 4>{}, rf, false); else pass(std::integral_constant<int, 8>{}, rf, false);
    }
    if (blockIdx.x == 0 && tid == 0) {
        a.sparams->lat_cand[0] = cand ? rf : 0u;
        a.sparams->lat_cand[1] = axis;
    }
    x0 = wave_min_bcast(x0); y0 = wave_min_bcast(y0);
    x1 = wave_max_bcast(x1); y1 = wave_max_bcast(y1);
    ax = wave_max_bcast(ax); ay = wave_max_bcast(ay); bx = wave_max_bcast(bx); by = wave_max_bcast(by);
    __syncthreads();
    if (lane == 0) { sh[wave][0] = x0; sh[wave][1] = y0; sh[wave][2] = x1; sh[wave][3] = y1; sh[wave][4] = ax; sh[wave][5] = ay; sh[wave][6] = bx; sh[wave][7] = by; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            x0 = fminf(x0, sh[w][0]); y0 = fminf(y0, sh[w][1]);
            x1 = fmaxf(x1, sh[w][2]); y1 = fmaxf(y1, sh[w][3]);
            ax = fmaxf(ax, sh[w][4]); ay = fmaxf(ay, sh[w][5]); bx = fmaxf(bx, sh[w][6]); by = fmaxf(by, sh[w][7]);
        }
        a.sboxes[blockIdx.x] = make_float4(x0, y0, x1, y1);
        a.slat[blockIdx.x] = make_float4(ax, ay, bx, by);
    }
}

// a plan built on an existing samples workspace has no bbox launch in front of it: its counters
// are zeroed by this one
__global__ __launch_bounds__(256) void plan_zero_kernel(BuildArgs a) {
    zero_words(a.counts, a.zero_words);
    if (a.zero_gacc) zero_words((uint32_t*)a.gacc, 8u * a.N);
    if (blockIdx.x == 0 && threadIdx.x < PLAN_BAR_WORDS) a.params->bar[threadIdx.x] = 0u;
}

// every workgroup of the count kernel reduces the PLAN_BBOX_BLOCKS partials (4 KB, L2 resident)
// sbox[0..3] = the box; sbox[4..7] = the largest neighbour steps {along x, along y, across x, across y} (meaningful
// when the first launch had a lattice candidate; a NaN partial cannot occur: the first launch turns it into +inf)
__device__ __forceinline__ void reduce_boxes(const float4* boxes, const float4* lat, uint32_t nparts, float* sbox, float (*sh)[8]) {
    static_assert(PLAN_BBOX_BLOCKS == 512, "one or two partials per thread");
    const float4 p = boxes[threadIdx.x];
    const float4 l = lat[threadIdx.x];
    float v[8] = {p.x, p.y, p.z, p.w, l.x, l.y, l.z, l.w};
    if (nparts > 256u) {      // launch-uniform
        const float4 p2 = boxes[256 + threadIdx.x];
        const float4 l2 = lat[256 + threadIdx.x];
        v[0] = fminf(v[0], p2.x); v[1] = fminf(v[1], p2.y); v[2] = fmaxf(v[2], p2.z); v[3] = fmaxf(v[3], p2.w);
        v[4] = fmaxf(v[4], l2.x); v[5] = fmaxf(v[5], l2.y); v[6] = fmaxf(v[6], l2.z); v[7] = fmaxf(v[7], l2.w);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (k & 2) || k >= 4 ? wave_max_bcast(v[k]) : wave_min_bcast(v[k]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) sh[wave][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float r = sh[0][k];
        for (int w = 1; w < 4; ++w) r = (k & 2) || k >= 4 ? fmaxf(r, sh[w][k]) : fminf(r, sh[w][k]);
        sbox[k] = r;
    }
}
// the decision behind a lattice candidate (plan.h, SampleParams::lat): an 8 x 8 index tile is at most 7 (along +
// across) steps wide and high; it must be at most twice as wide and as high as its share of the bounding box (an
// exact lattice: 7/8 of it).  Uniform over the launch: every workgroup reduces the same partials.
__device__ __forceinline__ bool lattice_compact(const float* sbox, uint32_t rf, uint32_t rs, uint32_t axis) {
    if (rf == 0u) return false;
    const float ex = sbox[2] - sbox[0], ey = sbox[3] - sbox[1];
    const float nx = (float)(axis ? rs : rf), ny = (float)(axis ? rf : rs);      // points along x / along y
    return 7.f * (sbox[4] + sbox[6]) * nx <= 16.f * ex && 7.f * (sbox[5] + sbox[7]) * ny <= 16.f * ey;      // NaN / inf: false
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

// The coarse-bin path's first pass (plan.h): workgroup w ranks its chunk of the point array inside every coarse
// bin with LDS atomics (one per point; random points spread over the 256 counters) and publishes its 256 counts
// as column w of the (bin, workgroup) matrix; a point keeps {fine cell, rank in (bin, workgroup)}.
__device__ __forceinline__ void samples_hist_part(const BuildArgs& a, uint32_t w, const SampleGrid& sg, uint32_t* lh, int lane);

// The Gaussians' half of the count: cell key (level by size, cell by centre) and rank of every Gaussian.  `box`: the
// samples' bounding box the grid's domain is laid over -- of this build, or (BuildArgs::ahead) of the last one.
struct GaussLoad { float m[2], c[3]; };
// One Gaussian into its place (in cell order, or -- PlanParams::strips -- its own): records, the box the list build
// tests first, the way back.
__device__ __forceinline__ void gauss_scatter_one(const BuildArgs& a, uint32_t i, uint32_t pos) {
    float v[2] = {0.f, 0.f};
    for (int k = 0; k < a.c; ++k) v[k] = a.values[(size_t)i * a.c + k];
    // {mux, muy, a, b}, {c, v0, v1, 0}
    a.rec[2 * pos] = make_float4(a.means[2 * i], a.means[2 * i + 1], a.conics[3 * i], a.conics[3 * i + 1]);
    a.rec[2 * pos + 1] = make_float4(a.conics[3 * i + 2], v[0], v[1], 0.f);
    if (i == 0) {         // record N: all zero (v = 0 contributes nothing), what list positions behind a list's end read
        a.rec[2 * (size_t)a.N] = make_float4(0.f, 0.f, 0.f, 0.f);
        a.rec[2 * (size_t)a.N + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {   // bounding box of the q <= q_max ellipse: half extents sqrt(q_max Sigma_xx), sqrt(q_max Sigma_yy)
        const float ca = a.conics[3 * i], cb = a.conics[3 * i + 1], cc = a.conics[3 * i + 2];
        const float k = a.q_max / (ca * cc - cb * cb);
        float hx = sqrtf(k * cc), hy = sqrtf(k * ca);
        if (!(hx < 3.0e38f)) hx = 3.0e38f;      // NaN / inf (degenerate conic): always a candidate
        if (!(hy < 3.0e38f)) hy = 3.0e38f;
        a.gbox[pos] = make_float4(a.means[2 * i], a.means[2 * i + 1], hx * 1.0001f, hy * 1.0001f);
    }
    a.g2o[pos] = i;
    // (the backward's scratch `gacc` is zero from the workspace's first build on -- zeroed once by the first
    // launch of a build into a workspace that is not PIGS_BUILD_PLAN_WS_CLEAN, re-zeroed by plan_unpermute_kernel
    // behind every backward: no memset launch, and no 8 scattered stores per Gaussian here either)
}

// The boxes of the strips (16 consecutive Gaussians of the caller's array = a row of this wave's lanes) and of the
// super-strip (the workgroup's 256): the union of the boxes of their q <= q_max ellipses.  A strip box's area over the
// domain's goes to `parea` (summed by the list launch into PlanParams::strip_cover); the boxes themselves to `pbox` /
// `sbox` when the build keeps the caller's order (`store`; block-uniform -- a barrier inside).  Non-finite extents (a
// degenerate conic) and NaN centres make a strip reach everywhere.
__device__ __forceinline__ void strip_box(const BuildArgs& a, uint32_t i, bool valid, float mx, float my, float hx, float hy,
                                          const float* box, bool store) {
    __shared__ float4 rowbox[16];
    const float INF = __builtin_huge_valf();
    if (!(hx < 3.0e38f)) hx = INF;      // NaN too
    if (!(hy < 3.0e38f)) hy = INF;
    float x0 = valid ? mx - hx : INF, x1 = valid ? mx + hx : -INF;
    float y0 = valid ? my - hy : INF, y1 = valid ? my + hy : -INF;
    if (valid && !(mx == mx)) { x0 = -INF; x1 = INF; }      // a NaN centre: fminf / fmaxf would drop it
    if (valid && !(my == my)) { y0 = -INF; y1 = INF; }
    row_box_dpp(x0, x1, y0, y1);
    const bool first = (threadIdx.x & 15u) == 0u;
    if (first && i < a.N) {
        const uint32_t strip = i / STRIP;
        if (store) a.pbox[strip] = make_float4(x0, y0, x1, y1);
        // inside the domain only: what lies outside meets no tile
        const float dx = box[2] - box[0], dy = box[3] - box[1];
        const float w = fminf(x1, box[2]) - fmaxf(x0, box[0]), h = fminf(y1, box[3]) - fmaxf(y0, box[1]);
        float cover = (w > 0.f && h > 0.f && dx > 0.f && dy > 0.f) ? (w * h) / (dx * dy) : 0.f;
        if (!(cover == cover)) cover = 1.f;
        a.parea[strip] = cover;
    }
    if (!store) return;
    if (first) rowbox[threadIdx.x >> 4] = make_float4(x0, y0, x1, y1);
    __syncthreads();
    if (threadIdx.x == 0 && i < a.N) {
        float4 b = rowbox[0];
        for (int r = 1; r < 16; ++r) {
            const float4 q = rowbox[r];
            b.x = fminf(b.x, q.x); b.y = fminf(b.y, q.y); b.z = fmaxf(b.z, q.z); b.w = fmaxf(b.w, q.w);
        }
        a.sbox[i / SUPER] = b;
    }
}
// A build that keeps the caller's order (PlanParams::strips): records, boxes and the way back at position i itself, the
// strip's box beside them -- the whole Gaussian half of a build in one pass, no atomics.
__device__ __forceinline__ void gauss_pack_part(const BuildArgs& a, uint32_t i, const float* box, const GaussLoad& ld) {
    const bool valid = i < a.N;
    float hx = 0.f, hy = 0.f;
    if (valid) {
        gauss_scatter_one(a, i, i);
        const float k = a.q_max / (ld.c[0] * ld.c[2] - ld.c[1] * ld.c[1]);
        hx = sqrtf(k * ld.c[2]) * 1.0001f; hy = sqrtf(k * ld.c[0]) * 1.0001f;      // (as gbox holds them)
    }
    strip_box(a, i, valid, ld.m[0], ld.m[1], hx, hy, box, true);
}
__device__ __forceinline__ GaussLoad gauss_count_load(const BuildArgs& a, uint32_t bid) {      // (issued early: flies while the box is reduced)
    GaussLoad ld = {{0.f, 0.f}, {0.f, 0.f, 0.f}};
    const uint32_t i = bid * 256 + threadIdx.x;
    if (i < a.N) {
        ld.m[0] = a.means[2 * i]; ld.m[1] = a.means[2 * i + 1];
        ld.c[0] = a.conics[3 * i]; ld.c[1] = a.conics[3 * i + 1]; ld.c[2] = a.conics[3 * i + 2];
    }
    return ld;
}
__device__ __forceinline__ void gauss_count_part(const BuildArgs& a, uint32_t bid, const float* box, const GaussLoad& ld) {
    const int lane = threadIdx.x & 63;
    const uint32_t i = bid * 256 + threadIdx.x;
    const bool valid = i < a.N;
    const float* gm = ld.m;
    const float* gc = ld.c;
    const GaussGrid g = gauss_grid(box, a.G0);
    if (bid == 0 && threadIdx.x == 0) {
        a.params->gg = g;
        a.params->scan_error = 0;
        a.params->q_f = a.q_f;
        a.params->q_b = a.q_b;
        a.params->n_points = 0u;
        a.params->strips = a.strips ? 1u : 0u;
        a.params->points_wanted = 0u;
        if (a.strips) a.params->level_mask = 0u;
#pragma unroll
        for (int l = 0; l <= PLAN_MAX_LEVELS; ++l) a.params->level_off[l] = a.level_off[l];
    }
    if (a.strips) {                    // launch-uniform: the caller's order is kept (PlanParams::strips)
        gauss_pack_part(a, i, box, ld);
        return;
    }
    uint32_t key = 0xffffffffu;
    int l = 0;
    {   // the statistic the library decides the NEXT build's kind from: this wave's Gaussians as four strips
        float hx = 0.f, hy = 0.f;
        if (valid) {
            const float k = a.q_max / (gc[0] * gc[2] - gc[1] * gc[1]);
            hx = sqrtf(k * gc[2]); hy = sqrtf(k * gc[0]);
        }
        strip_box(a, i, valid, gm[0], gm[1], hx, hy, box, false);
    }
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
}

__device__ __forceinline__ void gauss_count_ahead(const BuildArgs& a, uint32_t bid) {
    gauss_count_part(a, bid, a.hint_box, gauss_count_load(a, bid));
}

template <bool COH>
__device__ __forceinline__ void scan_block(const BuildArgs& a, bool seg0, uint32_t b, uint32_t* sh, uint32_t* sh2);
__device__ __forceinline__ void grid_barrier(uint32_t* bar, uint32_t G, uint32_t id);

// BuildArgs::sort_in_count: the rest of a one-pass sort behind the count, in the count's own launch -- for points that
// were expected to be a lattice (so that the host launched nothing behind this for them) and are none.  `ns` sample
// workgroups (all resident: the host launches at most 256), this one the `sid`-th: everybody's counters are final behind
// the first barrier; the scan's blocks are dealt out in turn (every workgroup takes its blocks in rising order and a
// block looks back at lower ones only: nobody waits on somebody who waits on him), past the caches; behind the second
// barrier every thread moves the points it keyed itself.  Slow next to three launches (two device-wide barriers), and
// rare: the memory turns around.
__device__ __forceinline__ void samples_sort_in_count(const BuildArgs& a, uint32_t sid, uint32_t ns, uint32_t* sh, uint32_t* sh2, int lane) {
    grid_barrier(&a.sparams->bar[0], ns, sid);
    for (uint32_t b = sid; b < a.s_scan_blocks; b += ns) {
        scan_block<true>(a, false, b, sh, sh2);
        __syncthreads();                      // sh / sh2 are the next block's
    }
    grid_barrier(&a.sparams->bar[17], ns, sid);
    for (uint32_t sb = sid; sb < a.s_blocks; sb += ns) {
        const uint32_t i0 = (sb * 4 + (threadIdx.x >> 6)) * 256 + (uint32_t)lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + 64 * k;
            if (i < a.M) {
                const uint2 kr = a.skey[i];          // (this thread's own store)
                const float2 p = ((const float2*)a.samples)[i];
                SPoint sp;
                sp.x = p.x; sp.y = p.y; sp.m = i;
                a.spts[__hip_atomic_load(&a.sstarts[kr.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + kr.y] = sp;
            }
        }
    }
}

__global__ __launch_bounds__(256) void plan_count_kernel(BuildArgs a) {
    __shared__ float shb[4][8];
    __shared__ uint32_t lh[SAMPLES_COARSE_BINS];
    __shared__ uint32_t scan_sh[4], scan_sh2[4];
    const int lane = threadIdx.x & 63;
    // BuildArgs::ahead: the Gaussians were counted in the first launch -- the first workgroups of THIS one scan their
    // cells (look-back among the launch's first workgroups, as in plan_scan_kernel), the samples' workgroups follow
    const uint32_t shift = a.ahead && !a.strips ? a.scan_blocks : 0u;
    if (blockIdx.x < shift) {                            // block-uniform
        scan_block<false>(a, true, blockIdx.x, scan_sh, scan_sh2);
        return;
    }
    const uint32_t bid = blockIdx.x - shift, nb = gridDim.x - shift;
    // Every dependent memory round trip costs 2-4 us in this kernel (in-kernel stamps): issue the
    // workgroup's own loads first, so they fly while the bounding-box partials are reduced.
    const uint32_t gblocks = a.do_plan && !a.ahead ? (a.N + 255) / 256 : 0;
    const bool gpart = bid < gblocks;
    float2 pt[4];
    uint32_t i0 = ((bid - gblocks) * 4 + (threadIdx.x >> 6)) * 256 + lane;
    // the first launch's lattice candidate (plan.h): with one, the sample workgroups most likely have nothing to do
    const uint32_t lat_rf = a.do_samples ? a.sparams->lat_cand[0] : 0u;
    const uint32_t lat_axis = a.do_samples ? a.sparams->lat_cand[1] : 0u;
    auto load_points = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + 64 * k;
            pt[k] = i < a.M ? ((const float2*)a.samples)[i] : make_float2(0.f, 0.f);
        }
    };
    GaussLoad gld = {{0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (gpart) gld = gauss_count_load(a, bid);
    else if (!a.coarse && lat_rf == 0u) load_points();
    float sbox[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool lattice = false;
    if (a.do_samples) {
        reduce_boxes(a.sboxes, a.slat, a.bbox_blocks, sbox, shb);
        lattice = lattice_compact(sbox, lat_rf, lat_rf ? a.M / lat_rf : 0u, lat_axis);
    } else {      // the samples workspace is complete: its box is in its header
#pragma unroll
        for (int k = 0; k < 4; ++k) sbox[k] = a.sparams->box[k];
    }
    const SampleGrid sg = sample_grid(sbox, a.M, a.scells_cap);
    if (bid == gblocks && threadIdx.x == 0 && a.do_samples) {      // (the first of the samples' workgroups)
#pragma unroll
        for (int k = 0; k < 4; ++k) a.sparams->box[k] = sbox[k];
        a.sparams->sg = sg;
        a.sparams->scan_error = 0;
        a.sparams->lat[0] = lattice ? lat_rf : 0u;
        a.sparams->lat[1] = lattice ? a.M / lat_rf : 0u;
        a.sparams->src = lattice ? (uint64_t)(uintptr_t)a.samples : 0ull;
    }
    // Gaussian workgroups first, sample workgroups after them: the two halves are independent
    // latency chains (load -> returning atomic -> store) and run concurrently on different CUs
    if (gpart) {                        // block-uniform: whole waves enter
        gauss_count_part(a, bid, sbox, gld);
    } else if (lattice) {
        // index-tiled: nothing to key, count or move (block-uniform)
    } else if (a.coarse) {
        samples_hist_part(a, bid - gblocks, sg, lh, lane);
    } else {
    // the one-pass count: a sample workgroup takes 1 024 points at a time -- one block where the launch has a workgroup
    // per block; where the host expected a lattice (rf_hint) and launched an eighth of them, the workgroups stride
    // over the blocks: the points that were no lattice after all are still all counted, by fewer hands
    for (uint32_t sb = bid - gblocks; sb < a.s_blocks; sb += nb - gblocks) {
        const bool first = sb == bid - gblocks;
        i0 = (sb * 4 + (threadIdx.x >> 6)) * 256 + lane;
        if (!first || lat_rf != 0u) load_points();      // (the first block's loads were issued early unless a lattice candidate stood)
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
                base[k] = atomicAdd(&a.scounts[id[k]], r[k].len);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + 64 * k;
            const uint32_t b = __shfl(base[k], r[k].start);
            if (i < a.M) a.skey[i] = make_uint2(id[k], b + (uint32_t)(lane - r[k].start));
        }
        if ((sb & 31u) == 0u && threadIdx.x < 64u) {      // a sample of the waves (one in 128): runs per point (SampleParams::order_stat)
            uint32_t runs = 0, pts = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                runs += (uint32_t)__builtin_popcountll(__ballot(r[k].leader && id[k] != 0xffffffffu));
                pts += (uint32_t)__builtin_popcountll(__ballot(id[k] != 0xffffffffu));
            }
            if (lane == 0) { atomicAdd(&a.sparams->order_stat[0], runs); atomicAdd(&a.sparams->order_stat[1], pts); }
        }
    }
    if (a.sort_in_count) samples_sort_in_count(a, bid - gblocks, nb - gblocks, scan_sh, scan_sh2, lane);      // (launch-uniform)
    }
}

__device__ __forceinline__ void samples_hist_part(const BuildArgs& a, uint32_t w, const SampleGrid& sg, uint32_t* lh, int lane) {
    static_assert(SAMPLES_COARSE_BINS == 256, "one bin per thread of the workgroup");
    lh[threadIdx.x] = 0u;
    __syncthreads();
    const uint64_t p0 = (uint64_t)w * a.h_chunk;
    const uint64_t p1 = p0 + a.h_chunk < (uint64_t)a.M ? p0 + a.h_chunk : (uint64_t)a.M;
    // the same statistic as the one-pass build keeps, from the chunk's first 256 points (all lanes present)
    const bool sampled = (w & 31u) == 0u && p0 + 256u <= p1;
    bool first = true;
    for (uint64_t i0 = p0 + threadIdx.x; i0 < p1; i0 += 1024u) {      // four loads in flight per thread
        float2 p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t i = i0 + 256u * k;
            p[k] = i < p1 ? ((const float2*)a.samples)[i] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t i = i0 + 256u * k;
            if (i < p1) {
                const int cx = (int)clampf((p[k].x - sg.ox) * sg.inv_w, 0.f, (float)(sg.nx - 1));
                const int cy = (int)clampf((p[k].y - sg.oy) * sg.inv_w, 0.f, (float)(sg.ny - 1));
                const uint32_t id = sample_cell_id(cx, cy, sg.nx);
                const uint32_t rank = atomicAdd(&lh[id / a.cells_per_bin], 1u);
                a.skey[i] = make_uint2(id, rank);
                if (sampled && first && k == 0) {
                    const Run r = run_of(id, lane);
                    const uint32_t runs = (uint32_t)__builtin_popcountll(__ballot(r.leader));
                    if (lane == 0) { atomicAdd(&a.sparams->order_stat[0], runs); atomicAdd(&a.sparams->order_stat[1], 64u); }
                }
            }
        }
        first = false;
    }
    __syncthreads();
    a.scounts[(size_t)threadIdx.x * a.h_wgs + w] = lh[threadIdx.x];
}

// Launch 3: exclusive scan counts -> starts in ONE launch, for the Gaussian cells and (when the
// samples are built alongside) the sample cells: two independent segments.  A workgroup scans
// PLAN_SCAN_BLOCK counters (one coalesced uint4 per thread), publishes its total as one 8-byte
// {flag, total} granule (single agent-scope store: data and flag travel together, no fence
// needed) and sums the granules of the workgroups before it; nobody waits on a later workgroup.
// The wait on a predecessor is bounded, and a predecessor that has not published within the bound is
// not an error: the counters are final before this launch starts (the count kernel has completed),
// so the waiting thread sums that workgroup's PLAN_SCAN_BLOCK counters ITSELF -- slower, never
// wrong, whatever order the hardware dispatches workgroups in.  `scan_error` in the workspace header
// only records that this happened (a diagnostic; never seen with in-order dispatch).
// `no_lookback` (PIGS_BUILD_DEBUG_NO_LOOKBACK) makes every thread take that path: the test hook.
constexpr uint32_t SCAN_SPIN_LIMIT = 1u << 14;
// COH: counters read and starts written with agent-scope accesses that bypass the (per-XCD, mutually
// incoherent) L2s -- for the one-launch chain, where producer and consumer phases of one launch run on
// different XCDs with no kernel boundary in between.
template <bool COH>
__device__ __forceinline__ uint4 scan_load4(const uint32_t* p) {
    if constexpr (COH) {
        uint4 v;
        v.x = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    } else {
        return *(const uint4*)p;
    }
}
template <bool COH>
__device__ __forceinline__ void scan_block(const BuildArgs& a, bool seg0, uint32_t b, uint32_t* sh, uint32_t* sh2) {
    const uint32_t* counts = seg0 ? a.counts : a.scounts;
    unsigned long long* agg = seg0 ? a.agg : a.sagg;
    uint32_t* starts = seg0 ? a.starts : a.sstarts;
    uint32_t* err = seg0 ? &a.params->scan_error : &a.sparams->scan_error;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t q = b * 256 + threadIdx.x;          // uint4 index
    const uint4 v = scan_load4<COH>(counts + 4 * (size_t)q);
    const uint32_t s = v.x + v.y + v.z + v.w;
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0 && !a.no_lookback)
        __hip_atomic_store(&agg[b], (1ull << 32) | (sh[0] + sh[1] + sh[2] + sh[3]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    uint32_t pre = 0;
    for (uint32_t t = threadIdx.x; t < b; t += 256) {
        unsigned long long x = 0;
        if (!a.no_lookback) {
            for (uint32_t spins = 0; spins < SCAN_SPIN_LIMIT; ++spins) {
                x = __hip_atomic_load(&agg[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (x >> 32) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (!(x >> 32)) {       // not published (in time): workgroup t's total from its counters
            const uint32_t* c4 = counts + (size_t)t * 1024;
            uint32_t tot = 0;
            for (int i = 0; i < 256; ++i) {
                const uint4 w = scan_load4<COH>(c4 + 4 * i);
                tot += w.x + w.y + w.z + w.w;
            }
            x = tot;
            atomicOr(err, 1u);
        }
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
    // counters beyond the last cell are zero: starts[ncells] = total
    if constexpr (COH) {
        uint32_t* d = starts + 4 * (size_t)q;
        __hip_atomic_store(d + 0, o4.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d + 1, o4.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d + 2, o4.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d + 3, o4.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        ((uint4*)starts)[q] = o4;
    }
}
__global__ __launch_bounds__(256) void plan_scan_kernel(BuildArgs a) {
    __shared__ uint32_t sh[4];
    __shared__ uint32_t sh2[4];
    const uint32_t nb0 = a.do_plan && !a.strips ? a.scan_blocks : 0;
    const bool seg0 = blockIdx.x < nb0;
    if (!seg0 && a.sparams->lat[0] != 0u) return;      // index-tiled points: nothing was counted (block-uniform)
    scan_block<false>(a, seg0, seg0 ? blockIdx.x : blockIdx.x - nb0, sh, sh2);
}

// Launch 4: scatter into sorted order (no atomics: position = cell start + rank) and publish the
// level mask.
// The coarse-bin path's scatter (plan.h): workgroup w moves ITS chunk of the point array (the chunk it ranked in
// samples_hist_part) to the bins' segments of the temporary array.  The chunk is first laid out bin by bin in LDS
// (slot = the workgroup's own exclusive scan over its 256 bin counts + the point's rank), then written out slot
// by slot: consecutive threads write consecutive 16-byte records of a bin's run, where a direct scatter sends
// every lane of a store to another line (16.7 -> 12.2 us at 1 M random points).  count(bin, w) is the difference
// of neighbouring entries of the scanned matrix.
constexpr uint32_t SCATTER_STAGE_MAX = 2048;       // points of a chunk the LDS stage holds (32 KB)
__device__ __forceinline__ void samples_scatter_part(const BuildArgs& a, uint32_t w, uint4* stage, uint32_t* lscan, uint32_t* gbase) {
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint64_t p0 = (uint64_t)w * a.h_chunk;
    const uint64_t p1 = p0 + a.h_chunk < (uint64_t)a.M ? p0 + a.h_chunk : (uint64_t)a.M;
    if (p0 >= p1) return;                     // block-uniform: a padding workgroup of the matrix
    {   // thread = bin: this workgroup's count in the bin, scanned over the bins
        const size_t e = (size_t)tid * a.h_wgs + w;
        const uint32_t hs = a.sstarts[e];
        const uint32_t nx = e + 1 < (size_t)SAMPLES_COARSE_BINS * a.h_wgs ? a.sstarts[e + 1] : a.M;
        const uint32_t c = nx - hs;
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        __shared__ uint32_t ws[4];
        if (lane == 63) ws[wave] = inc;
        __syncthreads();
        uint32_t run = inc - c;
        for (int k = 0; k < wave; ++k) run += ws[k];
        lscan[tid] = run;
        gbase[tid] = hs;
    }
    __syncthreads();
    for (uint64_t i0 = p0 + tid; i0 < p1; i0 += 1024u) {
        uint2 kr[4];
        float2 p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t i = i0 + 256u * k;
            kr[k] = i < p1 ? a.skey[i] : make_uint2(0u, 0u);
            p[k] = i < p1 ? ((const float2*)a.samples)[i] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t i = i0 + 256u * k;
            if (i < p1)
                stage[lscan[kr[k].x / a.cells_per_bin] + kr[k].y] =
                    make_uint4(__float_as_uint(p[k].x), __float_as_uint(p[k].y), (uint32_t)i, kr[k].x);
        }
    }
    __syncthreads();
    const uint32_t n = (uint32_t)(p1 - p0);
    uint4* tmp4 = (uint4*)a.tmp;
    for (uint32_t slot = tid; slot < n; slot += 256u) {
        const uint4 r = stage[slot];
        const uint32_t bin = r.w / a.cells_per_bin;
        tmp4[gbase[bin] + (slot - lscan[bin])] = r;
    }
}

// A workgroup scans ALL `nblocks` blocks of 1 024 counters itself, a block at a time through LDS (the counters are
// final: the count launch has completed), and every thread picks the start of ITS key out of the block that holds
// it.  ~0.7 us per block and workgroup: the slow way round, taken by the samples' workgroups of launch 3 of
// BuildArgs::ahead when the points they expected to be a lattice are none (the memory then turns around).  (As a
// replacement of the Gaussians' scan launch it was measured: 43 blocks, scatter 5.4 -> 28.9 us)
__device__ __forceinline__ uint32_t scan_pick(const uint32_t* counts, uint32_t nblocks, uint32_t key, uint32_t* lds, uint32_t* ws) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0, mine = 0;
    for (uint32_t c0 = 0; c0 < nblocks; c0 += 8u) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = c0 + (uint32_t)k < nblocks ? ((const uint4*)counts)[(size_t)(c0 + (uint32_t)k) * 256 + threadIdx.x] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t c = c0 + (uint32_t)k;
            if (c >= nblocks) break;                       // block-uniform
            const uint32_t sum = v[k].x + v[k].y + v[k].z + v[k].w;
            uint32_t inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            if (lane == 63) ws[wave] = inc;
            __syncthreads();
            uint32_t base = carry + inc - sum;
            for (int w = 0; w < wave; ++w) base += ws[w];
            const uint32_t tot = ws[0] + ws[1] + ws[2] + ws[3];
            ((uint4*)lds)[threadIdx.x] = make_uint4(base, base + v[k].x, base + v[k].x + v[k].y, base + v[k].x + v[k].y + v[k].z);
            __syncthreads();
            if ((key >> 10) == c) mine = lds[key & 1023u];
            carry += tot;
            __syncthreads();                               // lds / ws are the next block's
        }
    }
    return mine;
}

__global__ __launch_bounds__(256) void plan_scatter_kernel(BuildArgs a) {
    extern __shared__ uint4 scatter_stage[];      // coarse-bin path with a chunk that fits: [h_chunk] records + 2 x 256 words
    __shared__ uint32_t scan_lds[PLAN_SCAN_BLOCK];
    __shared__ uint32_t scan_ws[4];
    const uint32_t gblocks = a.do_plan && !a.strips ? (a.N + 255) / 256 : 0;      // (strips: the Gaussians are in place already)
    const bool gpart = blockIdx.x < gblocks;
    const uint32_t i = (gpart ? blockIdx.x : blockIdx.x - gblocks) * 256 + threadIdx.x;
    if (a.do_plan && !a.strips && blockIdx.x == 0 && threadIdx.x < 64) {
        // level l holds a Gaussian iff its cells' scanned range is not empty (no atomics, no scratch)
        const int l = (int)threadIdx.x;
        const int lc = l < a.L ? l : 0;
        const bool occ = l < a.L && a.starts[a.level_off[lc + 1]] != a.starts[a.level_off[lc]];
        const uint64_t m = __ballot(occ);
        if (threadIdx.x == 0) a.params->level_mask = (uint32_t)m;
    }
    if (a.do_plan && !a.strips) {
        // the scan has consumed the counters and its own flags: leave them zeroed, so that a later
        // build into this workspace (PIGS_BUILD_PLAN_WS_CLEAN) needs no zeroing launch
        for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < a.zero_words; k += gridDim.x * 256) a.counts[k] = 0u;
    }
    if (!gpart && a.sparams->lat[0] != 0u) return;     // index-tiled points: nothing to move (block-uniform)
    if (!gpart && a.coarse && a.h_chunk <= SCATTER_STAGE_MAX) {
        // (the sample workgroups of this launch are then the chunks' workgroups: h_wgs of them)
        uint32_t* words = (uint32_t*)(scatter_stage + a.h_chunk);
        samples_scatter_part(a, blockIdx.x - gblocks, scatter_stage, words, words + SAMPLES_COARSE_BINS);
        return;
    }
    if (gpart && i < a.N) {
        const uint2 kr = a.gkey[i];
        gauss_scatter_one(a, i, a.starts[kr.x] + kr.y);
    }
    uint32_t sstart = 0;
    if (!gpart && a.s_scan_in_scatter) {      // block-uniform: an expected lattice that was none, and no scan launch ran
        const uint2 kr = i < a.M ? a.skey[i] : make_uint2(0xffffffffu, 0u);
        sstart = scan_pick(a.scounts, a.s_scan_blocks, kr.x, scan_lds, scan_ws);
    }
    if (!gpart && i < a.M) {
        const uint2 kr = a.skey[i];
        const float2 p = ((const float2*)a.samples)[i];
        if (a.s_scan_in_scatter) {
            SPoint sp;
            sp.x = p.x; sp.y = p.y; sp.m = i;
            a.spts[sstart + kr.y] = sp;
        } else if (a.coarse) {
            // coarse-bin path: to the point's bin segment of the temporary array, behind the points that earlier
            // workgroups (chunks) sent to this bin; the fine cell travels along
            const uint32_t w = i / a.h_chunk, bin = kr.x / a.cells_per_bin;
            STmp t;
            t.x = p.x; t.y = p.y; t.m = i; t.id = kr.x;
            a.tmp[a.sstarts[(size_t)bin * a.h_wgs + w] + kr.y] = t;
        } else {
            SPoint sp;
            sp.x = p.x; sp.y = p.y; sp.m = i;
            a.spts[a.sstarts[kr.x] + kr.y] = sp;
        }
    }
}

// Last launch of the coarse-bin path: one workgroup per coarse bin counting-sorts the bin's segment of the
// temporary array by fine cell into the final array.  LDS: one counter per fine cell of the bin (count, then
// -- scanned in place -- cursor).  Segments up to 8 192 points are read once (registers); the writes stay inside
// the segment.  SUB = 16 (where the LDS holds 16 counters per cell: up to ~2 M points) also orders the points of a
// cell along the cell path continued into the cell (key_of below): a group is 16 consecutive sorted points, cells
// of unordered points hold 16 +- 4, so most groups straddle two cells, and with a cell's points in no order such
// a group's box spans both cells whole.  (A 4 x 4 Z-order of sub-cells was tried first: 41.9 -> 41.0 Gaussians
// per point at 1 M random points -- the halves of a Z-order are full-width strips.)  Order inside a key: as the
// atomics fall.
template <int SUB>
__global__ __launch_bounds__(1024) void samples_binsort_kernel(BuildArgs a) {
    extern __shared__ uint32_t cnt[];       // [cells_per_bin * SUB]
    __shared__ uint32_t wsum[16];
    constexpr int B = 8;                    // points a thread keeps in registers
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (a.sparams->lat[0] != 0u) return;               // index-tiled points: nothing to sort (block-uniform)
    const uint32_t seg0 = a.sstarts[(size_t)b * a.h_wgs];
    const uint32_t seg1 = b + 1 < SAMPLES_COARSE_BINS ? a.sstarts[(size_t)(b + 1) * a.h_wgs] : a.M;
    const uint32_t id0 = b * a.cells_per_bin, nkey = a.cells_per_bin * (uint32_t)SUB;
    const bool one_batch = seg1 - seg0 <= (uint32_t)B * 1024u;      // block-uniform
    const uint4* tmp4 = (const uint4*)a.tmp;
    const SampleGrid sg = a.sparams->sg;
    auto key_of = [&](const uint4 t) -> uint32_t {
        uint32_t k = (t.w - id0) * (uint32_t)SUB;
        if constexpr (SUB == 16) {
            // The point's place on the cell path continued INTO the cell: the path through a 4 x 4 block of cells is the
            // order-2 Hilbert curve (sample_cell_id), whose order-4 refinement runs through the 4 x 4 sub-cells of every
            // cell from the side the path enters the cell to the side it leaves (its top nibble IS the cell's index
            // inside the block) -- so consecutive points stay neighbours across a cell border.  Coordinates: 4 bits per
            // axis inside the block, from the same clamped cell coordinates the cell id came from; x mirrored in the
            // right-to-left block rows, as there.
            const float u = clampf((__uint_as_float(t.x) - sg.ox) * sg.inv_w, 0.f, (float)(sg.nx - 1));
            const float v = clampf((__uint_as_float(t.y) - sg.oy) * sg.inv_w, 0.f, (float)(sg.ny - 1));
            const int cx = (int)u, cy = (int)v;
            uint32_t x = (uint32_t)(cx & 3) * 4u + min(3u, (uint32_t)((u - (float)cx) * 4.f));
            uint32_t y = (uint32_t)(cy & 3) * 4u + min(3u, (uint32_t)((v - (float)cy) * 4.f));
            if ((cy >> 2) & 1) x = 15u - x;
            uint32_t d = 0;
#pragma unroll
            for (uint32_t sbit = 8u; sbit > 0u; sbit >>= 1) {
                const uint32_t rx = (x & sbit) ? 1u : 0u, ry = (y & sbit) ? 1u : 0u;
                d += sbit * sbit * ((3u * rx) ^ ry);
                if (ry == 0u) {
                    if (rx == 1u) { x = 15u - x; y = 15u - y; }
                    const uint32_t tt = x; x = y; y = tt;
                }
            }
            k += d & 15u;      // NaN coordinates: cell (0, 0), sub-cell 0
        }
        return k;
    };
    uint4 r[B];
    uint32_t rk[B];
    if (one_batch) {                         // the loads fly while the counters are cleared
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const uint32_t p = seg0 + (uint32_t)k * 1024u + tid;
            r[k] = p < seg1 ? tmp4[p] : make_uint4(0u, 0u, 0u, 0xffffffffu);
        }
    }
    for (uint32_t t = tid; t < nkey; t += 1024u) cnt[t] = 0u;
    __syncthreads();
    if (one_batch) {
#pragma unroll
        for (int k = 0; k < B; ++k) {
            rk[k] = r[k].w != 0xffffffffu ? key_of(r[k]) : 0xffffffffu;
            if (rk[k] != 0xffffffffu) atomicAdd(&cnt[rk[k]], 1u);
        }
    } else {
        // A segment longer than one batch: a dense patch of a clustered cloud, 100 k points and more in one bin -- 73 us for
        // this launch at sigma = 0.15, a third of that cloud's cold step.  What bounds it is ONE compute unit's memory
        // stream (the segment is read twice, 16 B per point: 3.7 MB at ~55 GB/s): batches of 4 loads with the next batch
        // in flight were measured slower (118 us), rounds of 8 loads the same (70), eight workgroups per bin each taking
        // a slice of the cells but reading the whole segment the same again (74, and 2 048 workgroups to launch cost the
        // uniform case 60 us).  Sixteen loads per thread and round:
        for (uint32_t p0 = seg0; p0 < seg1; p0 += 16u * 1024u) {
            uint4 t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t p = p0 + (uint32_t)k * 1024u + tid;
                t[k] = p < seg1 ? tmp4[p] : make_uint4(0u, 0u, 0u, 0xffffffffu);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (t[k].w != 0xffffffffu) atomicAdd(&cnt[key_of(t[k])], 1u);
        }
    }
    __syncthreads();
    // exclusive scan in place: thread t owns the `per` consecutive counters from t * per
    const uint32_t per = (nkey + 1023u) / 1024u;
    const uint32_t lo = tid * per, hi = lo + per < nkey ? lo + per : nkey;
    uint32_t sum = 0;
    for (uint32_t t = lo; t < hi; ++t) sum += cnt[t];
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int w2 = 0; w2 < wave; ++w2) run += wsum[w2];
    for (uint32_t t = lo; t < hi; ++t) {
        const uint32_t c = cnt[t];
        cnt[t] = run;
        run += c;
    }
    __syncthreads();
    auto place = [&](const uint4 t, uint32_t key) {
        const uint32_t k = atomicAdd(&cnt[key], 1u);
        SPoint sp;
        sp.x = __uint_as_float(t.x); sp.y = __uint_as_float(t.y); sp.m = t.z;
        a.spts[seg0 + k] = sp;
    };
    if (one_batch) {
#pragma unroll
        for (int k = 0; k < B; ++k)
            if (rk[k] != 0xffffffffu) place(r[k], rk[k]);
    } else {
        for (uint32_t p0 = seg0; p0 < seg1; p0 += 16u * 1024u) {
            uint4 t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t p = p0 + (uint32_t)k * 1024u + tid;
                t[k] = p < seg1 ? tmp4[p] : make_uint4(0u, 0u, 0u, 0xffffffffu);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (t[k].w != 0xffffffffu) place(t[k], key_of(t[k]));
        }
    }
}

// ------------------------------------------------------------------------------------------
// The Gaussian chain count -> scan -> scatter in ONE launch (a build on an existing samples workspace: the
// warm path, main_pn.py:317-324).  Each of the three launches it replaces sits at the floor of a dependent
// launch on this chip (~3.7 us: every kernel boundary is an L2 write-back and invalidate across 8 XCDs)
// while doing ~1 us of work; here the phases are separated by two device-wide barriers instead, and a
// Gaussian's {cell, rank} stays in its thread's registers between count and scatter.
// The barriers need every workgroup resident at once: the grid is ceil(N / 256) workgroups of 256 threads
// with a few hundred bytes of LDS (2 048 fit the chip), and the host uses this kernel only up to
// FUSED_BUILD_MAX_BLOCKS workgroups -- several such builds on different streams still fit side by side;
// larger N keep the three launches.  What one phase hands to the next across XCDs (the counters, starts[])
// travels through agent-scope accesses that bypass the per-XCD L2s; the barrier itself is a relaxed counter
// behind a workgroup-scope fence (a release / acquire pair at agent scope writes back and invalidates the
// whole L2 once per workgroup and barrier: measured +21 us over the three launches it was to replace).
// ------------------------------------------------------------------------------------------
constexpr uint32_t FUSED_BUILD_MAX_BLOCKS = 256;
// One barrier = 17 words: arrivals are counted per XCD-sized group of workgroups (id & 7: 32 arrivals per
// address instead of 256 -- same-address atomics retire one every ~10 ns), the last arrival of a group
// counts the group in, the last group raises eight release flags and every workgroup polls its own group's
// (32 pollers per address).
constexpr int BAR_WORDS = 17;
__device__ __forceinline__ void grid_barrier(uint32_t* bar, uint32_t G, uint32_t id);
__device__ __forceinline__ void grid_barrier(uint32_t* bar, uint32_t G) { grid_barrier(bar, G, blockIdx.x); }
__device__ __forceinline__ void grid_barrier(uint32_t* bar, uint32_t G, uint32_t id) {      // id: this workgroup among the G that meet
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's memory operations have completed
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t grp = id & 7u;
        const uint32_t in_grp = (G - grp + 7u) >> 3, groups = G < 8u ? G : 8u;
        if (__hip_atomic_fetch_add(&bar[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_grp - 1u) {
            if (__hip_atomic_fetch_add(&bar[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1u) {
#pragma unroll
                for (int k = 0; k < 8; ++k) __hip_atomic_store(&bar[9 + k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        while (__hip_atomic_load(&bar[9 + grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void plan_gauss_build_kernel(BuildArgs a) {
    __shared__ uint32_t sh[4];
    __shared__ uint32_t sh2[4];
    const int lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < a.N;
    float mx = 0.f, my = 0.f, ca = 0.f, cb = 0.f, cc = 0.f;
    if (valid) {
        mx = a.means[2 * i]; my = a.means[2 * i + 1];
        ca = a.conics[3 * i]; cb = a.conics[3 * i + 1]; cc = a.conics[3 * i + 2];
    }
    float sbox[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sbox[k] = a.sparams->box[k];
    const GaussGrid g = gauss_grid(sbox, a.G0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.params->gg = g;
        a.params->scan_error = 0;
        a.params->q_f = a.q_f;
        a.params->q_b = a.q_b;
        a.params->n_points = 0u;
        a.params->strips = 0u;
        a.params->points_wanted = 0u;
#pragma unroll
        for (int l = 0; l <= PLAN_MAX_LEVELS; ++l) a.params->level_off[l] = a.level_off[l];
    }
    // ---- count: cell key and rank (plan_count_kernel's Gaussian half)
    uint32_t key = 0xffffffffu;
    const float det = ca * cc - cb * cb;
    if (valid) {
        const float R = sqrtf(a.q_max * fmaxf(ca, cc) / det);   // NaN / inf (degenerate conic) -> top level
        float s = g.s0;
        int l = 0;
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
    const uint32_t rank = base + (uint32_t)(lane - r.start);
    grid_barrier(&a.params->bar[0], gridDim.x);
    // ---- scan (the first scan_blocks workgroups; their look-back is among resident workgroups)
    if (blockIdx.x < a.scan_blocks) scan_block<true>(a, true, blockIdx.x, sh, sh2);
    grid_barrier(&a.params->bar[BAR_WORDS], gridDim.x);
    // ---- scatter (plan_scatter_kernel's Gaussian half)
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const int l = (int)threadIdx.x;
        const int lc = l < a.L ? l : 0;
        const bool occ = l < a.L && __hip_atomic_load(&a.starts[a.level_off[lc + 1]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
                                        __hip_atomic_load(&a.starts[a.level_off[lc]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t m = __ballot(occ);
        if (threadIdx.x == 0) a.params->level_mask = (uint32_t)m;
    }
    // the scan has consumed the counters and its own flags: leave them zeroed (PIGS_BUILD_PLAN_WS_CLEAN)
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < a.zero_words; k += gridDim.x * 256) a.counts[k] = 0u;
    if (valid) {
        const uint32_t pos = __hip_atomic_load(&a.starts[key], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + rank;
        float v[2] = {0.f, 0.f};
        for (int k = 0; k < a.c; ++k) v[k] = a.values[(size_t)i * a.c + k];
        a.rec[2 * pos] = make_float4(mx, my, ca, cb);
        a.rec[2 * pos + 1] = make_float4(cc, v[0], v[1], 0.f);
        if (i == 0) {
            a.rec[2 * (size_t)a.N] = make_float4(0.f, 0.f, 0.f, 0.f);
            a.rec[2 * (size_t)a.N + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float k = a.q_max / det;
        float hx = sqrtf(k * cc), hy = sqrtf(k * ca);
        if (!(hx < 3.0e38f)) hx = 3.0e38f;      // NaN / inf (degenerate conic): always a candidate
        if (!(hy < 3.0e38f)) hy = 3.0e38f;
        a.gbox[pos] = make_float4(mx, my, hx * 1.0001f, hy * 1.0001f);
        a.g2o[pos] = i;
    }
    // ---- the last workgroup out re-arms the barriers for the next build into this workspace
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t done = __hip_atomic_fetch_add(&a.params->bar[2 * BAR_WORDS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            for (int q = 0; q < PLAN_BAR_WORDS; ++q) __hip_atomic_store(&a.params->bar[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Workgroups are dispatched round-robin over the 8 XCDs (workgroup i runs on XCD i % 8) and every
// XCD has its own L2.  Tiles follow the domain block row by block row, so inside every group of
// 8 * PIGS_XCD_CHUNK consecutive workgroups XCD x takes the x-th contiguous run of PIGS_XCD_CHUNK:
// each L2 then holds the Gaussian records and lists of a strip of the domain instead of all of
// them, while the launch still sweeps the domain once from top to bottom.  Bijective for any grid
// size (blocks behind the last whole group keep their index).  0 = no remapping.
#ifndef PIGS_XCD_CHUNK
#define PIGS_XCD_CHUNK 256
#endif
template <uint32_t CHUNK>
__device__ __forceinline__ uint32_t xcd_block_chunk(uint32_t nblocks, uint32_t b) {      // nblocks: the blocks that take part; b: this one's index among them
    if constexpr (CHUNK > 0) {                                                            //   (helper workgroups in front of them: a multiple of 8, keep out)
        constexpr uint32_t GROUP = 8u * CHUNK;
        const uint32_t g = b / GROUP, r = b % GROUP;
        if ((g + 1) * GROUP > nblocks) return b;
        return g * GROUP + (r & 7u) * CHUNK + (r >> 3);
    } else {
        return b;
    }
}
template <uint32_t CHUNK>
__device__ __forceinline__ uint32_t xcd_block_chunk(uint32_t nblocks) { return xcd_block_chunk<CHUNK>(nblocks, blockIdx.x); }
__device__ __forceinline__ uint32_t xcd_block(uint32_t nblocks, uint32_t b) { return xcd_block_chunk<PIGS_XCD_CHUNK>(nblocks, b); }

// ------------------------------------------------------------------------------------------
// Launch 5 of a plan build: the tile lists.  One wave = LISTS_TPW consecutive tiles (4 tiles = 256
// consecutive sorted points = one 4 x 4 block of sample cells): ONE traversal of the Gaussian grid
// against the box of all of them -- the traversal is a chain of dependent loads and most of the
// kernel's instructions, so it is shared -- whose survivors (exact ellipse-vs-box test) wait in
// LDS with what the ellipse test needs of them; every 128 survivors, and at the end, each tile's
// four 16-point groups are tested against them and the accepted ones appended to the tile's list
// and group lists; entries with an empty mask are dropped.
// ------------------------------------------------------------------------------------------
#ifndef PIGS_LISTS_TPW
#define PIGS_LISTS_TPW 4
#endif
constexpr int LISTS_TPW = PIGS_LISTS_TPW;
constexpr int SURV_CAP = 128;
template <int TPW>
struct ListsLds {
    TravLds trav;
    float4 sa[SURV_CAP];              // survivor: {mux, muy, a, b}
    float4 sb[SURV_CAP];              //           {c, -b/c, -b/a, sorted index (bits)}
    float4 gbox[TPW * 4];       // boxes of the groups: {x0, y0, x1, y1}
    uint32_t sel[SURV_CAP];           // positions of the survivors that reach the tile in hand
    uint32_t bmask[SURV_CAP];         // per survivor: the four tiles' masks, byte t = wide << 4 | narrow
};
struct ListArgs {
    PlanView pv;
    SamplesView sv;
    uint32_t* hdr;
    uint32_t* tlist;
    uint32_t* glist;
    uint32_t* ptiles;     // queue of the tiles in TILE_MODE_POINTS
    uint32_t* n_points;   //   and its length (PlanParams::n_points, zeroed by the count kernel)
    uint32_t* points_wanted;   // PlanParams::points_wanted
    float q_f;            // the narrow cut-off (pv.q_max is the wide one)
    const float* parea;   // per strip: box area / domain area (written by the build's Gaussian pass)
    float* strip_cover;   //   their sum (PlanParams::strip_cover)
};

// the lists of the TPW tiles from tile0 (one wave).  TPW = 4 (a 4 x 4 block of sample cells: the traversal of the grid is
// shared by four tiles) where the launch fills the chip; TPW = 1 for small point sets (LISTS_SMALL_TILES): the launch's
// time is the serial life of ONE wave there (21 us at 65 536 points with four tiles per wave, the chip nearly idle), and
// a wave with a quarter of the work has a shorter life.
template <int TPW, bool STRIPS = false>
__device__ __forceinline__ void build_block_lists(const ListArgs& a, ListsLds<TPW>& lds, uint32_t tile0, int lane) {
    const PlanView& pv = a.pv;
    const uint32_t ntiles = a.sv.ntiles;
    const GaussGrid gg = pv.params->gg;
    const uint32_t level_mask = pv.params->level_mask;
    constexpr bool strips = STRIPS;                       // candidates from strip boxes, not from grid cells (PlanParams::strips)
    const uint32_t loff = pv.params->level_off[lane < PLAN_MAX_LEVELS ? lane : 0];   // lane = level: its first counter
    const float INF = __builtin_huge_valf();
    SPoint sp[TPW];
    bool valid[TPW];
    const PointOrder po = point_order(a.sv);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const uint32_t m = (tile0 + (uint32_t)t) * TILE_POINTS + (uint32_t)lane;
        valid[t] = m < a.sv.M;           // also false for every point of a tile behind the last one
        sp[t] = SPoint{0.f, 0.f, 0u};
        if (tile0 + (uint32_t)t < ntiles) sp[t] = tile_point(a.sv, po, tile0 + (uint32_t)t, (uint32_t)lane);     // wave-uniform
    }
    float bx0 = INF, bx1 = -INF, by0 = INF, by1 = -INF;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        float x0 = valid[t] ? sp[t].x : INF, x1 = valid[t] ? sp[t].x : -INF;
        float y0 = valid[t] ? sp[t].y : INF, y1 = valid[t] ? sp[t].y : -INF;
        row_box_dpp(x0, x1, y0, y1);
        if ((lane & 15) == 0) lds.gbox[t * 4 + (lane >> 4)] = make_float4(x0, y0, x1, y1);
        bx0 = fminf(bx0, x0); bx1 = fmaxf(bx1, x1); by0 = fminf(by0, y0); by1 = fmaxf(by1, y1);
    }
    wave_box_from_rows_dpp(bx0, bx1, by0, by1);

    const uint32_t cap = pv.list_cap;
    // what a per-point walk would meet: 9 cells of every level at the level's mean occupancy (wave-uniform;
    // all lanes call it together)
    auto walk_candidates = [&]() -> float {
        float e = 0.f;
        if (lane < pv.L) {
            const float cells = (float)(pv.G0 >> lane) * (float)(pv.G0 >> lane);
            e = 9.f * (float)(pv.starts[pv.level_off[lane + 1]] - pv.starts[pv.level_off[lane]]) / cells;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) e += __shfl_xor(e, o);      // levels live in lanes 0..11
        return __shfl(e, 0);
    };
    // A block of 256 points spread over more than POINTS_MODE_BLOCK_CELLS finest Gaussian cells (a group of 16 then
    // spans dozens of cells: its list would run to hundreds) goes to the per-point walk without being listed at
    // all: the traversal of such a box is the list build's own tail (thousands of candidates in one wave).
    if (strips && (bx1 - bx0) * gg.inv_s0 * ((by1 - by0) * gg.inv_s0) > POINTS_MODE_BLOCK_CELLS) {
        // far-apart points: the cells would have sent this block to the per-point walk -- say so (PlanParams::points_wanted)
        if (lane == 0) atomicAdd(a.points_wanted, 1u);
    }
    if (!strips && (bx1 - bx0) * gg.inv_s0 * ((by1 - by0) * gg.inv_s0) > POINTS_MODE_BLOCK_CELLS && walk_candidates() <= 4.f * (float)cap) {
        for (int t = 0; t < TPW; ++t) {
            if (tile0 + (uint32_t)t >= ntiles) break;
            if (lane < TILE_HDR_WORDS)
                a.hdr[(size_t)(tile0 + (uint32_t)t) * TILE_HDR_WORDS + lane] = lane == 0 ? (TILE_MODE_POINTS << TILE_MODE_SHIFT) : 0u;
            if (lane == 0) a.ptiles[atomicAdd(a.n_points, 1u)] = tile0 + (uint32_t)t;
        }
        return;
    }
    uint32_t n[TPW], ng[TPW][4];
    bool overflow[TPW], goverflow[TPW];       // the tile list / one of the group lists is full
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        n[t] = 0; overflow[t] = false; goverflow[t] = false;
#pragma unroll
        for (int g = 0; g < 4; ++g) ng[t][g] = 0;
    }
    int sn = 0;
#if PIGS_BWD_BLOCK
    // The block list lives in the tile-list slabs of the block's tiles (8-byte entries {sorted index, masks}):
    // capacity = tiles x cap / 2.
    uint2* const blist = (uint2*)(a.tlist + (size_t)tile0 * cap);
    const uint32_t tiles_here = ntiles - tile0 < (uint32_t)TPW ? ntiles - tile0 : (uint32_t)TPW;
    const uint32_t cap_b = tiles_here * cap / 2;
    uint32_t nb = 0;
    bool boverflow = false;
#endif
    // the tiles' tests on the survivors in LDS: per tile, (A) the survivors that reach the tile's
    // box, compacted (their positions, one byte each would do: 128 survivors), then (B) the four
    // group tests on those: usually one step of 64 instead of two
    auto flush = [&]() __attribute__((always_inline)) {
        wave_lds_fence();
#if PIGS_BWD_BLOCK
        lds.bmask[lane] = 0u;
        lds.bmask[lane + 64] = 0u;
        wave_lds_fence();
#endif
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            if (tile0 + (uint32_t)t >= ntiles) continue;
            float4 gb[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) gb[g] = lds.gbox[t * 4 + g];
            const float tx0 = fminf(fminf(gb[0].x, gb[1].x), fminf(gb[2].x, gb[3].x));
            const float ty0 = fminf(fminf(gb[0].y, gb[1].y), fminf(gb[2].y, gb[3].y));
            const float tx1 = fmaxf(fmaxf(gb[0].z, gb[1].z), fmaxf(gb[2].z, gb[3].z));
            const float ty1 = fmaxf(fmaxf(gb[0].w, gb[1].w), fmaxf(gb[2].w, gb[3].w));
            int sel = 0;
            for (int s0 = 0; s0 < sn; s0 += 64) {
                const int k = s0 + lane < sn ? s0 + lane : 0;
                const float4 A = lds.sa[k], B = lds.sb[k];
                Ellipse e;
                e.x = A.x; e.y = A.y; e.a = A.z; e.b = A.w; e.c = B.x; e.nb_c = B.y; e.nb_a = B.z;
                const bool hit = s0 + lane < sn && ellipse_reaches_rect(e, tx0, ty0, tx1, ty1, pv.q_max);
                const uint64_t hm = __ballot(hit);
                if (hit) lds.sel[sel + lanes_below(hm)] = (uint32_t)k;
                sel += __builtin_popcountll(hm);
            }
            wave_lds_fence();
            uint32_t* tl = a.tlist + (size_t)(tile0 + (uint32_t)t) * cap;
            uint32_t* gl = a.glist + (size_t)(tile0 + (uint32_t)t) * 4 * cap;
            for (int s0 = 0; s0 < sel; s0 += 64) {
                const int k = (int)lds.sel[s0 + lane < sel ? s0 + lane : 0];
                const float4 A = lds.sa[k], B = lds.sb[k];
                Ellipse e;
                e.x = A.x; e.y = A.y; e.a = A.z; e.b = A.w; e.c = B.x; e.nb_c = B.y; e.nb_a = B.z;
                const uint32_t j = __builtin_bit_cast(uint32_t, B.w);
                uint32_t gm = 0, gf = 0;       // wide (tile list, backward) and narrow (group lists, forward) masks
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // a group without a point (the ragged last tile) has an inverted box: never needed
                    const float qmin = ellipse_min_q_rect(e, gb[g].x, gb[g].y, gb[g].z, gb[g].w);
                    if (gb[g].x <= gb[g].z) {
                        if (!(qmin > pv.q_max)) gm |= 1u << g;
                        if (!(qmin > a.q_f)) gf |= 1u << g;
                    }
                }
                if (s0 + lane >= sel) gm = gf = 0u;
                const uint64_t km = __ballot(gm != 0u);
                const uint32_t cnt = (uint32_t)__builtin_popcountll(km);
#if PIGS_BWD_BLOCK
                (void)tl;
                if (gm != 0u) lds.bmask[k] |= ((gm << 4) | gf) << (8 * t);      // one lane per survivor k in this pass
#else
                if (n[t] + cnt <= cap) {
                    if (gm != 0u) tl[n[t] + (uint32_t)lanes_below(km)] = j | (gm << LIST_WIDE_SHIFT) | (gf << LIST_NARROW_SHIFT);
                } else {
                    overflow[t] = true;
                }
#endif
                n[t] += cnt;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint64_t mg = __ballot(gf >> g & 1u);
                    const uint32_t cg = (uint32_t)__builtin_popcountll(mg);
                    if (ng[t][g] + cg <= cap) {
                        if (gf >> g & 1u) gl[g * cap + ng[t][g] + (uint32_t)lanes_below(mg)] = j;
                    } else {
                        goverflow[t] = true;
                    }
                    ng[t][g] += cg;
                }
            }
            wave_lds_fence();
        }
#if PIGS_BWD_BLOCK
        // the block list: every survivor that reaches a group of any of the four tiles, once, with all its masks
        for (int s0 = 0; s0 < sn; s0 += 64) {
            const int k = s0 + lane;
            const uint32_t bm = k < sn ? lds.bmask[k] : 0u;
            const uint64_t km = __ballot(bm != 0u);
            const uint32_t cnt = (uint32_t)__builtin_popcountll(km);
            if (nb + cnt <= cap_b) {
                if (bm != 0u) blist[nb + (uint32_t)lanes_below(km)] = make_uint2(__builtin_bit_cast(uint32_t, lds.sb[k].w), bm);
            } else {
                boverflow = true;
            }
            nb += cnt;
        }
        wave_lds_fence();
#endif
        sn = 0;
    };
    auto walk_rect = [&](float x0, float y0, float x1, float y1, bool walk, auto&& rows, auto&& batch) __attribute__((always_inline)) {
        if constexpr (STRIPS) traverse_strips(pv, x0, y0, x1, y1, lane, lds.trav, walk, rows, batch);
        else traverse(pv, gg, level_mask, loff, x0, y0, x1, y1, lane, lds.trav, walk, rows, batch);
    };
    walk_rect(bx0, by0, bx1, by1, true,
             [](int, uint32_t, uint32_t) {},
             [&](const float4 A, const float4 B, uint64_t mask, uint32_t j) __attribute__((always_inline)) {
        // The survivors wait until the buffer cannot take the batch in hand (round 4: it used to be flushed as soon as
        // a FULL batch might not fit any more, i.e. from 65 on -- a block's ~100 survivors then went through the per-tile
        // filter and the group tests in two portions, twice the steps of one; a walk that finds them in batches of ~25,
        // four strips of 16 candidates, more often still).
        const int add = __builtin_popcountll(mask);
#ifdef PIGS_LISTS_PROBE_NO_FLUSH          // probe build: the traversal alone (survivors dropped)
        if (sn + add > SURV_CAP) sn = 0;
#else
        if (sn + add > SURV_CAP) flush();
#endif
        if (mask >> lane & 1ull) {
            const Ellipse e = ellipse_of(A, B.x);
            const int k = sn + lanes_below(mask);
            lds.sa[k] = A;
            lds.sb[k] = make_float4(B.x, e.nb_c, e.nb_a, __builtin_bit_cast(float, j));
        }
        sn += add;
    });
#ifndef PIGS_LISTS_PROBE_NO_FLUSH
    if (sn > 0) flush();
#endif

    // A tile list that does not fit while the four group lists do (64 scattered points of a sparse
    // region share few Gaussians: up to 4 x cap distinct ones) is no reason to give the lists up: the
    // forward reads the group lists only, and the backward walks them as four single-group lists
    // (TILE_MODE_GROUPS) -- which must then hold the WIDE set: they are rebuilt below.
    const bool two_cuts = pv.q_max > a.q_f;
    bool any_rare = false;
    bool rebuild[TPW];
#if PIGS_BWD_BLOCK
    // The block list serves the backward of all four tiles when it fits and every group list (the forward's) does;
    // otherwise (very wide Gaussians, scattered points) the tiles fall back one by one: group lists only
    // (rebuilt with the wide cut-off: the backward then walks those) or record ranges.
    bool block_ok = !boverflow;
#pragma unroll
    for (int t = 0; t < TPW; ++t) block_ok = block_ok && !(tile0 + (uint32_t)t < ntiles && goverflow[t]);
    if (lane == 0) a.hdr[(size_t)tile0 * TILE_HDR_WORDS + 5] = block_ok ? (nb | 0x80000000u) : 0u;
#pragma unroll
    for (int t = 0; t < TPW; ++t) overflow[t] = !block_ok;      // as if every tile list had overflowed
#endif
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        rebuild[t] = false;
        if (tile0 + (uint32_t)t >= ntiles) { overflow[t] = false; continue; }
        const bool tl_over = overflow[t];
        overflow[t] = tl_over && goverflow[t];            // from here on: the tile needs the ranges fallback
        rebuild[t] = tl_over && !goverflow[t] && two_cuts;
        // spread-out points with long lists: the per-point walk is cheaper than the lists (plan.h)
        uint32_t longest = ng[t][0] > ng[t][1] ? ng[t][0] : ng[t][1];
        longest = ng[t][2] > longest ? ng[t][2] : longest;
        longest = ng[t][3] > longest ? ng[t][3] : longest;
        if (longest > POINTS_MODE_MIN_LIST) {      // (strips: the per-point walk needs the grid -- only noted)
            float4 gb[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) gb[g] = lds.gbox[t * 4 + g];
            const float wx = fmaxf(fmaxf(gb[0].z, gb[1].z), fmaxf(gb[2].z, gb[3].z)) - fminf(fminf(gb[0].x, gb[1].x), fminf(gb[2].x, gb[3].x));
            const float wy = fmaxf(fmaxf(gb[0].w, gb[1].w), fmaxf(gb[2].w, gb[3].w)) - fminf(fminf(gb[0].y, gb[1].y), fminf(gb[2].y, gb[3].y));
            if (strips && wx * gg.inv_s0 * (wy * gg.inv_s0) > POINTS_MODE_MIN_CELLS) {
                if (lane == 0) atomicAdd(a.points_wanted, 1u);
            }
            if (!strips && wx * gg.inv_s0 * (wy * gg.inv_s0) > POINTS_MODE_MIN_CELLS && walk_candidates() <= 4.f * (float)(longest < cap ? longest : cap)) {
                overflow[t] = false; rebuild[t] = false;
                if (lane < 5)
                    a.hdr[(size_t)(tile0 + (uint32_t)t) * TILE_HDR_WORDS + lane] = lane == 0 ? (TILE_MODE_POINTS << TILE_MODE_SHIFT) : 0u;
                if (lane == 0) a.ptiles[atomicAdd(a.n_points, 1u)] = tile0 + (uint32_t)t;
                continue;
            }
        }
        any_rare |= overflow[t] || rebuild[t];
        if (!overflow[t] && !rebuild[t] && lane < TILE_HDR_WORDS) {
#if PIGS_BWD_BLOCK
            const bool fits = block_ok;      // LIST here = "its block list is valid" (n[t]: the tile's own entries, for statistics)
#else
            const bool fits = n[t] <= cap;
#endif
            uint32_t w = 0;
            if (lane == 0) w = fits ? ((n[t] & TILE_COUNT_MASK) | (TILE_MODE_LIST << TILE_MODE_SHIFT)) : (TILE_MODE_GROUPS << TILE_MODE_SHIFT);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (lane == 1 + g) w = ng[t][g];
            if (lane < 5) a.hdr[(size_t)(tile0 + (uint32_t)t) * TILE_HDR_WORDS + lane] = w;      // words 5..7: the block's
        }
    }
    if (!any_rare) return;
    // Rare paths, one tile at a time, not unrolled.  (1) group-lists-only tile under two cut-offs: its
    // group lists are rebuilt from a walk of the grid around ITS box with the wide cut-off (the forward
    // then evaluates a few pairs more than q_f asks for in such a tile: harmless).  (2) a group list does
    // not fit: the tile keeps the grid's record ranges around its box instead (pairs {first, length}; the
    // sampling kernels test the ranges' records against the group boxes themselves); when even those do
    // not fit, the single range of all Gaussians.
    for (int t = 0; t < TPW; ++t) {
        const bool mine_rebuild = __builtin_amdgcn_readfirstlane((int)(t == 0   ? rebuild[0]
                                                                       : t == 1 ? rebuild[TPW > 1 ? 1 : 0]
                                                                       : t == 2 ? rebuild[TPW > 2 ? 2 : 0]
                                                                                : rebuild[TPW > 3 ? 3 : 0])) != 0;
        bool mine = __builtin_amdgcn_readfirstlane((int)(t == 0   ? overflow[0]
                                                         : t == 1 ? overflow[TPW > 1 ? 1 : 0]
                                                         : t == 2 ? overflow[TPW > 2 ? 2 : 0]
                                                                  : overflow[TPW > 3 ? 3 : 0])) != 0;
        if (!mine && !mine_rebuild) continue;
        float4 gb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) gb[g] = lds.gbox[t * 4 + g];
        const float tx0 = fminf(fminf(gb[0].x, gb[1].x), fminf(gb[2].x, gb[3].x));
        const float ty0 = fminf(fminf(gb[0].y, gb[1].y), fminf(gb[2].y, gb[3].y));
        const float tx1 = fmaxf(fmaxf(gb[0].z, gb[1].z), fmaxf(gb[2].z, gb[3].z));
        const float ty1 = fmaxf(fmaxf(gb[0].w, gb[1].w), fmaxf(gb[2].w, gb[3].w));
        uint32_t* tl = a.tlist + (size_t)(tile0 + (uint32_t)t) * cap;
        if (mine_rebuild) {
            uint32_t* gl = a.glist + (size_t)(tile0 + (uint32_t)t) * 4 * cap;
            uint32_t ngw[4] = {0u, 0u, 0u, 0u};
            bool gover = false;
            walk_rect(tx0, ty0, tx1, ty1, true,
                     [](int, uint32_t, uint32_t) {},
                     [&](const float4 A, const float4 B, uint64_t mask, uint32_t j) __attribute__((always_inline)) {
                const bool have = mask >> lane & 1ull;
                const Ellipse e = ellipse_of(A, B.x);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool hit = have && gb[g].x <= gb[g].z &&
                                     !(ellipse_min_q_rect(e, gb[g].x, gb[g].y, gb[g].z, gb[g].w) > pv.q_max);
                    const uint64_t mg = __ballot(hit);
                    const uint32_t cg = (uint32_t)__builtin_popcountll(mg);
                    if (ngw[g] + cg <= cap) {
                        if (hit) gl[g * cap + ngw[g] + (uint32_t)lanes_below(mg)] = j;
                    } else {
                        gover = true;
                    }
                    ngw[g] += cg;
                }
            });
            if (!gover) {
                if (lane < 5) {
                    uint32_t w = 0;
                    if (lane == 0) w = TILE_MODE_GROUPS << TILE_MODE_SHIFT;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (lane == 1 + g) w = ngw[g];
                    a.hdr[(size_t)(tile0 + (uint32_t)t) * TILE_HDR_WORDS + lane] = w;
                }
                continue;
            }
            mine = true;          // the wide group lists do not fit either: ranges
        }
        // scattered points (a box of many Gaussian cells for 64 points): no lists, every lane walks the grid
        // around its own point at sampling time (plan.h, TILE_MODE_POINTS)
        if (!strips && (tx1 - tx0) * gg.inv_s0 * ((ty1 - ty0) * gg.inv_s0) > POINTS_MODE_MIN_CELLS && walk_candidates() <= 4.f * (float)cap) {
            if (lane < 5)
                a.hdr[(size_t)(tile0 + (uint32_t)t) * TILE_HDR_WORDS + lane] = lane == 0 ? (TILE_MODE_POINTS << TILE_MODE_SHIFT) : 0u;
            if (lane == 0) a.ptiles[atomicAdd(a.n_points, 1u)] = tile0 + (uint32_t)t;
            continue;
        }
        uint32_t nr = 0;
        bool fits = true;
        walk_rect(tx0, ty0, tx1, ty1, false,
                 [&](int nrow, uint32_t jb, uint32_t len) {
            const bool keep = lane < nrow && len > 0;
            const uint64_t km = __ballot(keep);
            const uint32_t cnt = (uint32_t)__builtin_popcountll(km);
            if (2 * (nr + cnt) <= cap) {
                if (keep) {
                    const uint32_t p = 2 * (nr + (uint32_t)lanes_below(km));
                    tl[p] = jb; tl[p + 1] = len;
                }
            } else {
                fits = false;
            }
            nr += cnt;
        },
                 [](const float4, const float4, uint64_t, uint32_t) {});
        if (!fits && lane == 0) { tl[0] = 0; tl[1] = pv.N; }
        const uint32_t cnt = fits ? nr : 1u;
        if (lane < 5)
            a.hdr[(size_t)(tile0 + (uint32_t)t) * TILE_HDR_WORDS + lane] = lane == 0 ? (cnt | (TILE_MODE_RANGES << TILE_MODE_SHIFT)) : 0u;
    }
}

// the same strips of the domain on the same XCD as in the sampling kernels, which then find a tile's
// lists in the L2 that wrote them (a workgroup here is 4 * TPW tiles; forward 27.05 -> 26.4 us)
template <int TPW>
__device__ __forceinline__ uint32_t lists_tile0(int wave) {
    return (xcd_block_chunk<PIGS_XCD_CHUNK / TPW>(gridDim.x) * 4 + (uint32_t)wave) * TPW;
}
// the first workgroup of a list launch: the strips' cover of the domain (PlanParams::strip_cover), summed from what the
// build's Gaussian pass left per strip
__device__ __forceinline__ void lists_strip_cover(const ListArgs& a) {
    __shared__ float cover_sh[4];
    if (blockIdx.x != 0) return;
    const uint32_t ns = (a.pv.N + STRIP - 1u) / STRIP;
    float sum = 0.f;
    for (uint32_t k = threadIdx.x; k < ns; k += 256u) sum += a.parea[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63u) == 0u) cover_sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) a.strip_cover[0] = cover_sh[0] + cover_sh[1] + cover_sh[2] + cover_sh[3];
}

template <int TPW, bool STRIPS>
__global__ __launch_bounds__(256) void plan_lists_kernel(ListArgs a) {
    __shared__ ListsLds<TPW> lds_all[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lists_strip_cover(a);
    const uint32_t tile0 = lists_tile0<TPW>(wave);
    if (tile0 >= a.sv.ntiles) return;
    build_block_lists<TPW, STRIPS>(a, lds_all[wave], tile0, lane);
}

// ------------------------------------------------------------------------------------------
// Sampling kernels.  One wave = one tile.
// ------------------------------------------------------------------------------------------
struct Rec {
    float mu[2], con[3], v[2];
};

// TILE_MODE_POINTS: this lane's own walk of the Gaussian grid around the point (x, y): in every occupied
// level the 3 x 3 cells around the point's cell hold every Gaussian of that level whose q <= cut ellipse can
// contain the point (a Gaussian lives in the lowest level whose cell side covers its ellipse's half extent;
// out-of-domain coordinates clamp the same monotone way the build binned them).  `body(j, A, B)` gets the
// sorted index and the record of every candidate; lanes run their own trip counts.
template <int STRIDE, typename Body>      // the lanes i = 0 .. STRIDE-1 of a point share its candidates: lane i takes j0 + i, j0 + i + STRIDE, ...
__device__ __forceinline__ void walk_point(const PlanView& pv, float x, float y, int i, Body&& body) {
    const GaussGrid gg = pv.params->gg;
    const uint32_t level_mask = pv.params->level_mask;
    for (int l = 0; l < pv.L; ++l) {
        if (!(level_mask >> l & 1u)) continue;
        const int G = pv.G0 >> l;
        const float inv_s = gg.inv_s0 * __builtin_amdgcn_ldexpf(1.f, -l);
        const float gmax = (float)(G - 1);
        const int cx = (int)clampf(floorf((x - gg.ox) * inv_s), 0.f, gmax);
        const int cy = (int)clampf(floorf((y - gg.oy) * inv_s), 0.f, gmax);
        const int cx0 = cx > 0 ? cx - 1 : 0, cx1 = cx < G - 1 ? cx + 1 : G - 1;
        const int cy0 = cy > 0 ? cy - 1 : 0, cy1 = cy < G - 1 ? cy + 1 : G - 1;
        const int csh = level_shift((uint32_t)(G * G));
        // the (up to) three rows' record ranges first -- six independent loads, one round trip -- then the records
        uint32_t j0[3], j1[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = cy0 + r;
            const uint32_t row = (uint32_t)((yy <= cy1 ? yy : cy1) * G);
            j0[r] = pv.starts[pv.level_off[l] + ((row + (uint32_t)cx0) << csh)];
            j1[r] = yy <= cy1 ? pv.starts[pv.level_off[l] + ((row + (uint32_t)cx1 + 1u) << csh)] : j0[r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
            for (uint32_t j = j0[r] + (uint32_t)i; j < j1[r]; j += STRIDE) body(j, pv.rec[2 * (size_t)j], pv.rec[2 * (size_t)j + 1]);
    }
}
// The same walk with the whole wave in step (four points per wave, 16 lanes per point, STRIDE = 16): `body(have, j, A,
// B)` is called by all 64 lanes together -- `have` says whether this lane holds a candidate (lanes without one get the
// all-zero record N) -- as many times per cell row as the longest of the four points' ranges needs, so that the body may
// exchange data between lanes.  Called with the same (wave-uniform) set of levels by every lane.
template <typename Body>
__device__ __forceinline__ void walk_point_instep(const PlanView& pv, float x, float y, int i, Body&& body) {
    const GaussGrid gg = pv.params->gg;
    const uint32_t level_mask = pv.params->level_mask;
    for (int l = 0; l < pv.L; ++l) {
        if (!(level_mask >> l & 1u)) continue;
        const int G = pv.G0 >> l;
        const float inv_s = gg.inv_s0 * __builtin_amdgcn_ldexpf(1.f, -l);
        const float gmax = (float)(G - 1);
        const int cx = (int)clampf(floorf((x - gg.ox) * inv_s), 0.f, gmax);
        const int cy = (int)clampf(floorf((y - gg.oy) * inv_s), 0.f, gmax);
        const int cx0 = cx > 0 ? cx - 1 : 0, cx1 = cx < G - 1 ? cx + 1 : G - 1;
        const int cy0 = cy > 0 ? cy - 1 : 0, cy1 = cy < G - 1 ? cy + 1 : G - 1;
        const int csh = level_shift((uint32_t)(G * G));
        uint32_t j0[3], j1[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = cy0 + r;
            const uint32_t row = (uint32_t)((yy <= cy1 ? yy : cy1) * G);
            j0[r] = pv.starts[pv.level_off[l] + ((row + (uint32_t)cx0) << csh)];
            j1[r] = yy <= cy1 ? pv.starts[pv.level_off[l] + ((row + (uint32_t)cx1 + 1u) << csh)] : j0[r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            uint32_t len = j1[r] - j0[r];                  // the same in the 16 lanes of a point
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) len = max(len, (uint32_t)__shfl_xor((int)len, o));
            len = (uint32_t)__builtin_amdgcn_readfirstlane((int)len);
            for (uint32_t o = 0; o < len; o += 16u) {
                const uint32_t j = j0[r] + o + (uint32_t)i;
                const bool have = j < j1[r];
                const size_t jj = have ? j : pv.N;
                body(have, (uint32_t)jj, pv.rec[2 * jj], pv.rec[2 * jj + 1]);
            }
        }
    }
}
__device__ __forceinline__ float pair_q(const float4 A, const float4 B, float x, float y) {
    const float dx = x - A.x, dy = y - A.y;
    return A.z * dx * dx + (2.f * A.w * dx + B.x * dy) * dy;
}
__device__ __forceinline__ Rec make_rec(const float4 A, const float4 B) {
    Rec r;
    r.mu[0] = A.x; r.mu[1] = A.y; r.con[0] = A.z; r.con[1] = A.w; r.con[2] = B.x;
    r.v[0] = B.y; r.v[1] = B.z;
    return r;
}

// Walks a tile's list (or its ranges): `step(idx, gm, have)` for every STEP entries (lane = entry; the
// lanes from STEP on hold none).  A tile in ranges mode has no masks: `ranges_mask()` is called once and
// returns the functor `mask(idx, have)` that finds an entry's (the caller's kernel keeps its `step` free
// of that rare case).
template <int STEP, bool WIDE, typename Step, typename RangesMask>
__device__ __forceinline__ void for_each_step(const PlanView& pv, uint32_t tile, int lane, Step&& step,
                                              RangesMask&& ranges_mask) {
    const uint32_t hdr = pv.hdr[(size_t)tile * TILE_HDR_WORDS];
    const uint32_t count = hdr & TILE_COUNT_MASK;
    const uint32_t* slab = pv.tlist + (size_t)tile * pv.list_cap;
    if ((hdr >> TILE_MODE_SHIFT) == TILE_MODE_LIST) {
        for (uint32_t e0 = 0; e0 < count; e0 += STEP) {
            const bool have = lane < STEP && e0 + (uint32_t)lane < count;
            const uint32_t e = have ? slab[e0 + lane] : 0u;
            step(e & LIST_IDX_MASK, WIDE ? (e >> LIST_WIDE_SHIFT) & 15u : e >> LIST_NARROW_SHIFT, have);
        }
    } else if ((hdr >> TILE_MODE_SHIFT) == TILE_MODE_GROUPS) {
        // the four group lists, one after the other, as lists of single-group entries (a Gaussian that
        // reaches two groups comes twice, each time for one of them)
        for (uint32_t g = 0; g < 4; ++g) {
            const uint32_t ng = pv.hdr[(size_t)tile * TILE_HDR_WORDS + 1 + g];
            const uint32_t* gl = pv.glist + ((size_t)tile * 4 + g) * pv.list_cap;
            for (uint32_t e0 = 0; e0 < ng; e0 += STEP) {
                const bool have = lane < STEP && e0 + (uint32_t)lane < ng;
                step(have ? gl[e0 + lane] : 0u, have ? 1u << g : 0u, have);
            }
        }
    } else {
        auto mask = ranges_mask();
        for (uint32_t r = 0; r < count; ++r) {
            const uint32_t j0 = slab[2 * r], len = slab[2 * r + 1];
            for (uint32_t o = 0; o < len; o += STEP) {
                const bool have = lane < STEP && o + (uint32_t)lane < len;
                const uint32_t idx = have ? j0 + o + (uint32_t)lane : j0;
                step(idx, mask(idx, have), have);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Forward.  Every group (DPP row) of the wave keeps its own queue of RECORDS in LDS and fills it from
// ITS OWN list (the group lists of the plan; 32 positions per chunk, two rounds of 16 gathers in
// flight), the all-zero record behind the list's end (v = 0: contributes nothing); then the queues
// are evaluated row-wise up to the longest list: in one instruction every row works on its OWN
// Gaussian, read from LDS at an address that is affine in the loop counter (no index indirection:
// the reads of the next rows are in flight while the current ones are evaluated).  A tile in
// record-range mode fills the queues by testing the ranges' records against the group boxes.
// A record in LDS is {mux, muy, a, b}, {c, v0, v1, -}: one ds_read_b128 + one ds_read_b64 (c = 1).
// The reads are inline asm: hipcc fuses 8-byte LDS reads of neighbouring rows into ds_read2_b64,
// which moves 16 bytes in 8 LDS cycles where ds_read_b128 takes 4 (MI355X_MICROARCH.md, LDS
// table), and collapses a source-level prefetch into load-then-wait.
// ------------------------------------------------------------------------------------------
#ifndef PIGS_GROUP_CAP
#define PIGS_GROUP_CAP 32        // records per group queue
#endif
constexpr int GROUP_CAP = PIGS_GROUP_CAP;
static_assert(GROUP_CAP >= 8 && GROUP_CAP % PIGS_FWD_UNROLL == 0, "queue capacity");

struct FwdLds {
    static constexpr int GSTRIDE = GROUP_CAP * 32 + 32;            // bytes; + 32: the four queues start on different banks
    float4 rec[(4 * GSTRIDE + PIGS_FWD_UNROLL * 32) / 16];        // tail: the prefetch behind the last row
};

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

// LDS byte address of a __shared__ object (for ds_* inline asm)
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
template <int C> struct LdsRec;
template <> struct LdsRec<1> { f4v a; f2v b; };
template <> struct LdsRec<2> { f4v a; f4v b; };
// issue the reads of the record at addr + OFF (no wait: lds_rec_wait before the first use)
template <int OFF>
__device__ __forceinline__ void lds_rec_issue(LdsRec<1>& r, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4"
                 : "=v"(r.a), "=v"(r.b) : "v"(addr), "i"(OFF), "i"(OFF + 16));
}
template <int OFF>
__device__ __forceinline__ void lds_rec_issue(LdsRec<2>& r, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                 : "=v"(r.a), "=v"(r.b) : "v"(addr), "i"(OFF), "i"(OFF + 16));
}
template <int C>
__device__ __forceinline__ void lds_rec_wait(LdsRec<C>* r) {      // r[0], r[1]: every pending destination
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0].a), "+v"(r[0].b), "+v"(r[1].a), "+v"(r[1].b));
}
template <int C>
__device__ __forceinline__ Rec rec_of(const LdsRec<C>& x) {
    Rec r;
    r.mu[0] = x.a.x; r.mu[1] = x.a.y; r.con[0] = x.a.z; r.con[1] = x.a.w; r.con[2] = x.b.x; r.v[0] = x.b.y;
    if constexpr (C == 2) r.v[1] = x.b.z;
    else r.v[1] = 0.f;
    return r;
}

// rows: a multiple of 2.  Two register sets take turns: while one pair of rows is evaluated the
// reads of the next pair are in flight (the last issue reads the two rows behind the queue: inside
// the LDS block, never used).
template <int C, int MASK>
__device__ __forceinline__ void evaluate_rows(float* acc, const float* s, const FwdLds& lds, int rows, int lane,
                                              const Resid<float>& rz) {
    static_assert(PIGS_FWD_UNROLL == 2, "two rows per register set");
    uint32_t q = lds_addr(lds.rec) + (uint32_t)(lane >> 4) * FwdLds::GSTRIDE;
    LdsRec<C> ra[2], rb[2];
    auto eval2 = [&](const LdsRec<C>* r) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const Rec x = rec_of<C>(r[u]);
            fwd_accumulate<float, 2, C, MASK>(acc, s, x.mu, x.con, x.v, &rz);
        }
    };
    lds_rec_issue<0>(ra[0], q);
    lds_rec_issue<32>(ra[1], q);
    int k = 0;
    for (; k + 4 <= rows; k += 4) {
        lds_rec_wait<C>(ra);
        lds_rec_issue<64>(rb[0], q);
        lds_rec_issue<96>(rb[1], q);
        eval2(ra);
        lds_rec_wait<C>(rb);
        lds_rec_issue<128>(ra[0], q);
        lds_rec_issue<160>(ra[1], q);
        q += 128;
        eval2(rb);
    }
    lds_rec_wait<C>(ra);
    if (k < rows) eval2(ra);
}

// register budget: 8 waves/SIMD (64 VGPRs) for the narrow variants, fewer waves for the wide ones
// (c = 2 with orders up to 3: 12-20 accumulators) so that they do not spill
template <int C, int MASK>
constexpr int fwd_waves() {
    constexpr int n = FwdLayout<2, C, MASK>::N;
    return n > 12 ? 4 : n > 10 ? 5 : (C == 1 && (MASK == 7 || MASK == 19 || MASK == 1 || MASK == ORDR)) ? PIGS_FWD_WAVES : 6;
}
constexpr bool fwd_can_stage(int C, int MASK) { return C == 1 && (MASK == 7 || MASK == 19); }
// staged outputs (PlanView::stage): one record per point at its original index instead of the three stores
template <int C, int MASK>
__device__ __forceinline__ void stage_store(const PlanView& pv, const float* acc, uint32_t m) {
    using L = FwdLayout<2, C, MASK>;
    if constexpr (C == 1 && MASK == 7) {
        pv.stage[2 * (size_t)m] = make_float4(acc[L::O0], -acc[L::O1], -acc[L::O1 + 1], acc[L::O2]);
        pv.stage[2 * (size_t)m + 1] = make_float4(acc[L::O2 + 1], acc[L::O2 + 1], acc[L::O2 + 2], 0.f);
    } else if constexpr (C == 1 && MASK == 19) {
        pv.stage[2 * (size_t)m] = make_float4(acc[L::O0], -acc[L::O1], -acc[L::O1 + 1], acc[L::O2]);
    }
}

// TILE_MODE_POINTS (plan.h): four points of tile `tile` at a time (quad = 0 .. 15), 16 lanes per point, lane = candidate
template <int C, int MASK>
__device__ __forceinline__ void forward_points_quad(const PlanView& pv, const SamplesView& sv, uint32_t tile, uint32_t quad, int lane,
                                                    float q_f, float* __restrict__ o0, float* __restrict__ o1,
                                                    float* __restrict__ o2, float* __restrict__ o3, const Resid<float>& rz) {
    using L = FwdLayout<2, C, MASK>;
    const int row = lane >> 4, i = lane & 15;
    const uint32_t m = tile * TILE_POINTS + quad * 4u + (uint32_t)row;
    const bool valid = m < sv.M;
    const SPoint sp = tile_point(sv, point_order(sv), tile, quad * 4u + (uint32_t)row);
    const float s[2] = {sp.x, sp.y};
    float acc[L::N];
#pragma unroll
    for (int k = 0; k < L::N; ++k) acc[k] = 0.f;
    // !(q > cut): a degenerate conic (NaN) is evaluated, as the list build's tests would have kept it
    walk_point<16>(pv, sp.x, sp.y, i, [&](uint32_t, const float4 A, const float4 B) {
        if (!(pair_q(A, B, sp.x, sp.y) > q_f)) {
            const Rec r = make_rec(A, B);
            fwd_accumulate<float, 2, C, MASK>(acc, s, r.mu, r.con, r.v, &rz);
        }
    });
#pragma unroll
    for (int k = 0; k < L::N; ++k) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) acc[k] += __shfl_xor(acc[k], o);
    }
    if (i == 0 && valid) {
        if (fwd_can_stage(C, MASK) && pv.stage) stage_store<C, MASK>(pv, acc, sp.m);
        else fwd_store<float, 2, C, MASK, false>(acc, (int64_t)sp.m, o0, o1, o2, o3, &rz);
    }
}

// one tile (one wave) through its group lists / record ranges; a tile in TILE_MODE_POINTS is left to the caller
template <int C, int MASK>
__device__ __forceinline__ void forward_tile(const PlanView& pv, const SamplesView& sv, uint32_t tile, int lane, FwdLds& lds,
                                             float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2,
                                             float* __restrict__ o3, const Resid<float>& rz) {
    using L = FwdLayout<2, C, MASK>;
    constexpr int U = PIGS_FWD_UNROLL;
    constexpr bool CAN_STAGE = fwd_can_stage(C, MASK);
    __builtin_amdgcn_s_setprio(3);
    char* const qbase = (char*)lds.rec;
    const uint32_t m = tile * TILE_POINTS + (uint32_t)lane;
    const bool valid = m < sv.M;
    const SPoint sp = tile_point(sv, point_order(sv), tile, (uint32_t)lane);      // lanes behind the last point repeat it (never stored)
    const float s[2] = {sp.x, sp.y};
    float acc[L::N];
#pragma unroll
    for (int k = 0; k < L::N; ++k) acc[k] = 0.f;
    const int g = lane >> 4, i = lane & 15;
    const uint32_t* hd = pv.hdr + (size_t)tile * TILE_HDR_WORDS;
    const uint32_t h0 = hd[0];
    // one chunk: every row fills its queue with `rows` records (its own list's, or the all-zero record
    // behind the list's end), two rounds of 16 in flight together, and the rows are evaluated
    auto chunk = [&](int rows, auto&& index_of) {
        static_assert(GROUP_CAP == 32, "two rounds of 16 records per chunk");
        rows = __builtin_amdgcn_readfirstlane((rows + U - 1) / U * U);
        const uint32_t j0 = index_of(i), j1 = index_of(16 + i);
        const float4 A0 = pv.rec[2 * (size_t)j0], B0 = pv.rec[2 * (size_t)j0 + 1];
        float4 A1 = A0, B1 = B0;
        if (rows > 16) { A1 = pv.rec[2 * (size_t)j1]; B1 = pv.rec[2 * (size_t)j1 + 1]; }
        float4* dst = (float4*)(qbase + g * FwdLds::GSTRIDE + i * 32);
        wave_lds_fence();
        dst[0] = A0;
        *(float2*)(dst + 1) = make_float2(B0.x, B0.y);
        if constexpr (C == 2) *(float2*)((char*)(dst + 1) + 8) = make_float2(B0.z, 0.f);
        if (rows > 16) {
            dst[32] = A1;
            *(float2*)(dst + 33) = make_float2(B1.x, B1.y);
            if constexpr (C == 2) *(float2*)((char*)(dst + 33) + 8) = make_float2(B1.z, 0.f);
        }
        wave_lds_fence();
#ifndef PIGS_DEBUG_SKIP_EVAL
        // waves outside the row loop (issuing loads, filling queues, storing) go first: their memory
        // requests are what the others' arithmetic hides (27.5 -> 27.3 us; the other way round 28.1)
        __builtin_amdgcn_s_setprio(0);
        evaluate_rows<C, MASK>(acc, s, lds, rows, lane, rz);
        __builtin_amdgcn_s_setprio(3);
#else
        acc[0] += (float)rows + ((const float*)lds.rec)[lane];
#endif
    };
    if ((h0 >> TILE_MODE_SHIFT) == TILE_MODE_POINTS) {
        return;                                   // scattered points: the caller's (helper workgroups / the fused launch's own walk)
    } else if ((h0 >> TILE_MODE_SHIFT) != TILE_MODE_RANGES) {            // LIST or GROUPS: the group lists are there
        const uint32_t ng = hd[1 + g];                                    // this row's list length
        const uint32_t* gl = pv.glist + ((size_t)tile * 4 + g) * pv.list_cap;
        uint32_t nmax = hd[1] > hd[2] ? hd[1] : hd[2];
        nmax = hd[3] > nmax ? hd[3] : nmax;
        nmax = hd[4] > nmax ? hd[4] : nmax;
        for (uint32_t base = 0; base < nmax; base += GROUP_CAP) {
            const int rows = (int)(nmax - base < GROUP_CAP ? nmax - base : GROUP_CAP);
            // the entry is loaded whether or not it lies inside the list (the slab has the room, and
            // the load then does not wait for the header): one dependent round trip less per tile
            chunk(rows, [&](int p) {
                const uint32_t e = gl[base + p < pv.list_cap ? base + p : 0u];
                return base + p < ng ? e : pv.N;
            });
        }
    } else {
        // Record ranges (a group list did not fit): the ranges hold every Gaussian near the tile.  Every
        // row tests them, 16 at a time, against the box of ITS group and packs the hits into its queue;
        // the queues are evaluated when one could overflow, and at the end.  With scattered points
        // (which is when lists overflow) a row keeps a small part of what the ranges hold.
        const uint32_t* slab = pv.tlist + (size_t)tile * pv.list_cap;
        const uint32_t count = h0 & TILE_COUNT_MASK;
        const float INF = __builtin_huge_valf();
        const float q_f = pv.params->q_f;
        float x0 = valid ? sp.x : INF, x1 = valid ? sp.x : -INF, y0 = valid ? sp.y : INF, y1 = valid ? sp.y : -INF;
        row_box_dpp(x0, x1, y0, y1);                  // every lane: the box of its own row's group
        const bool row_has_points = x0 <= x1;
        float4* const q = (float4*)(qbase + g * FwdLds::GSTRIDE);
        int qn = 0;                                    // records in this row's queue (the same in its 16 lanes)
        auto drain = [&]() {
            // rows = the longest queue, the others padded with all-zero records
            int rows = qn;
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) rows = max(rows, __shfl_xor(rows, o));
            rows = __builtin_amdgcn_readfirstlane((rows + U - 1) / U * U);
            wave_lds_fence();
            for (int k = qn + i; k < rows; k += 16) {
                q[2 * k] = make_float4(0.f, 0.f, 0.f, 0.f);
                q[2 * k + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            wave_lds_fence();
            if (rows > 0) {
                __builtin_amdgcn_s_setprio(0);
                evaluate_rows<C, MASK>(acc, s, lds, rows, lane, rz);
                __builtin_amdgcn_s_setprio(3);
            }
            qn = 0;
        };
        for (uint32_t r = 0; r < count; ++r) {
            const uint32_t j0 = slab[2 * r], len = slab[2 * r + 1];
            for (uint32_t base = 0; base < len; base += 16) {
                const bool in = base + (uint32_t)i < len;
                const size_t j = in ? j0 + base + (uint32_t)i : pv.N;
                const float4 A = pv.rec[2 * j], B = pv.rec[2 * j + 1];
                const bool hit = in && row_has_points && ellipse_reaches_rect(ellipse_of(A, B.x), x0, y0, x1, y1, q_f);
                const uint32_t rm = (uint32_t)(__ballot(hit) >> (16 * g)) & 0xffffu;      // this row's hits
                if (hit) {
                    const int k = qn + __builtin_popcount(rm & ((1u << i) - 1u));
                    q[2 * k] = A;
                    *(float2*)(q + 2 * k + 1) = make_float2(B.x, B.y);
                    if constexpr (C == 2) *(float2*)((char*)(q + 2 * k + 1) + 8) = make_float2(B.z, 0.f);
                }
                qn += __builtin_popcount(rm);
                if (__any(qn > GROUP_CAP - 16)) drain();
            }
        }
        drain();
    }
    // The outputs go back through the points' original indices.  Where those run in the caller's order
    // (a grid: runs of 4 or more consecutive points per cell row) a tile's stores fill whole 32..128-byte
    // segments and leave through non-temporal stores: nothing in the launch reads them again, and
    // streamed they do not wait in the L2 for the end-of-kernel write-back (28.0 -> 26.6 us).  Where
    // the points came in no order (shuffled grids, random points) every store is a lone 4..16 bytes and
    // needs the L2's write combining: streamed, the same launch takes 115 us instead of 54.
    const uint32_t m_other = (uint32_t)__shfl_xor((int)sp.m, 1);
    const uint32_t dist = sp.m > m_other ? sp.m - m_other : m_other - sp.m;
    const bool stream = __builtin_popcountll(__ballot(valid && dist == 1u)) >= 48;
    if (CAN_STAGE && pv.stage) {
        if (valid) stage_store<C, MASK>(pv, acc, sp.m);
    } else if (valid) {
        if (stream) {
            fwd_store<float, 2, C, MASK, true>(acc, (int64_t)sp.m, o0, o1, o2, o3, &rz);
            asm volatile("" ::: "memory");       // keeps the two branches' stores apart: merged into a common tail they lose the hint
        } else {
            fwd_store<float, 2, C, MASK, false>(acc, (int64_t)sp.m, o0, o1, o2, o3, &rz);
        }
    }
}

template <int C, int MASK>
__global__ __launch_bounds__(64 * PIGS_FWD_WG_WAVES, (fwd_waves<C, MASK>())) void tile_forward_kernel(
    PlanView pv, SamplesView sv, float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2,
    float* __restrict__ o3, Resid<float> rz) {
    constexpr uint32_t FW = PIGS_FWD_WG_WAVES;
    __shared__ FwdLds lds_all[FW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nmain = (sv.ntiles + FW - 1u) / FW;
    // The helper workgroups come FIRST in the launch (round 4): their walks are chains of dependent loads that take
    // many times a tile's life, and dispatched behind the main ones (round 3) they were the launch's tail -- a
    // clamped-normal cloud's forward took 56 us for 30 us of tile work.  They leave at once when the plan queued no
    // TILE_MODE_POINTS tile.  Their number is a multiple of 8: workgroup i of the main ones still runs on XCD i % 8.
    constexpr uint32_t NHELP = POINT_HELPER_BLOCKS * 4u / FW;
    static_assert(NHELP % 8u == 0u, "the main workgroups keep their XCD");
    if (blockIdx.x < NHELP) {
        // helper workgroups (plan.h, TILE_MODE_POINTS): four points at a time, 16 lanes per point, lane = candidate
        const uint32_t n = pv.params->n_points;
        if (n == 0u) return;
        const float q_f = pv.params->q_f;
        const uint32_t hw = blockIdx.x * FW + (uint32_t)wave, nhw = NHELP * FW;
        for (uint32_t qd = hw; qd < n * 16u; qd += nhw)
            forward_points_quad<C, MASK>(pv, sv, pv.ptiles[qd >> 4], qd & 15u, lane, q_f, o0, o1, o2, o3, rz);
        return;
    }
    const uint32_t tile = xcd_block_chunk<PIGS_XCD_CHUNK * 4 / FW>(nmain, blockIdx.x - NHELP) * FW + (uint32_t)wave;
    if (tile >= sv.ntiles) return;
#if PIGS_FWD_STAGGER
    {   // experiment: the workgroups of the first generation start in phases, so that the launch's
        // waves do not load, evaluate and store all at the same time.  Unit: s_sleep 32 = 2 048 cycles.
        const uint32_t b = blockIdx.x - NHELP;
        if (b < PIGS_FWD_STAGGER_FIRST) {
            const uint32_t phase = ((b >> 3) ^ (b >> 8)) % PIGS_FWD_STAGGER_PHASES;
            for (uint32_t k = 0; k < phase * PIGS_FWD_STAGGER; ++k) __builtin_amdgcn_s_sleep(32);
        }
    }
#endif
    forward_tile<C, MASK>(pv, sv, tile, lane, lds_all[wave], o0, o1, o2, o3, rz);
}

// ------------------------------------------------------------------------------------------
// The FIRST forward of a plan in the launch that builds its tile lists (PIGS_BUILD_DEFER_LISTS; round 4).  The
// list build is a chain of dependent loads (a wave issues in 38 % of its cycles), the forward is float32
// arithmetic: in two launches neither hides the other, and the forward's own first loads have nothing to hide
// behind.  Here a wave builds the lists of its four tiles (written out as ever: the backward and every further
// sample_*() of the same preprocess read them) and evaluates those tiles at once -- while it computes, the other
// waves of its SIMD are still walking the grid.  One kernel boundary and the forward's cold start go away.
// A tile in TILE_MODE_POINTS is walked by its own wave here (the helper workgroups of the two-launch path read a
// queue that is complete only when this launch ends): right, and slower for clouds with thin outskirts -- the
// hosts defer the lists for every plan all the same, because a cloud's first step is one of thousands.
// ------------------------------------------------------------------------------------------
template <int C, int MASK>
__global__ __launch_bounds__(256) void plan_lists_forward_kernel(ListArgs a, float* __restrict__ o0, float* __restrict__ o1,
                                                                 float* __restrict__ o2, float* __restrict__ o3, Resid<float> rz) {
    __shared__ ListsLds<LISTS_TPW> lds_all[4];
    static_assert(sizeof(FwdLds) <= sizeof(ListsLds<LISTS_TPW>), "the forward's queues live in the list build's LDS");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lists_strip_cover(a);
    const uint32_t tile0 = lists_tile0<LISTS_TPW>(wave);
    const uint32_t ntiles = a.sv.ntiles;
    if (tile0 >= ntiles) return;
    build_block_lists<LISTS_TPW>(a, lds_all[wave], tile0, lane);
    // what this wave's lanes stored (headers, group lists) is read back by other lanes of it: the stores have
    // reached the L2 before the first load is issued
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    FwdLds& flds = *reinterpret_cast<FwdLds*>(&lds_all[wave]);
    for (int t = 0; t < LISTS_TPW; ++t) {
        const uint32_t tile = tile0 + (uint32_t)t;
        if (tile >= ntiles) break;
        const uint32_t h0 = __hip_atomic_load(a.pv.hdr + (size_t)tile * TILE_HDR_WORDS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((h0 >> TILE_MODE_SHIFT) == TILE_MODE_POINTS) {
            for (uint32_t quad = 0; quad < 16u; ++quad) forward_points_quad<C, MASK>(a.pv, a.sv, tile, quad, lane, a.q_f, o0, o1, o2, o3, rz);
        } else {
            forward_tile<C, MASK>(a.pv, a.sv, tile, lane, flds, o0, o1, o2, o3, rz);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Backward helpers: a step's records go to the wave's LDS once (slot = lane); the group masks are
// split into four per-row index lists holding LDS byte offsets, padded with the offset of an
// all-zero record to the longest of the four.
// ------------------------------------------------------------------------------------------
#ifndef PIGS_BWD_STEP
#define PIGS_BWD_STEP 64      // entries per step of the backward: its LDS (records, lists, sums table) scales with it
                              // (32 doubles the resident waves and splits C3's 49-entry lists in two steps: 83 vs 81 us)
#endif
constexpr int BWD_STEP = PIGS_BWD_STEP;
static_assert(BWD_STEP == 32 || BWD_STEP == 64, "entries per step");
constexpr int LIST_PAD = 8;
struct TileLds {
    float4 rec[BWD_STEP + 1][2];            // slot BWD_STEP: the all-zero record (v = 0: contributes nothing)
    uint16_t list[4][BWD_STEP + LIST_PAD];  // byte offsets into rec (< 2 112)
};
constexpr uint32_t ZERO_REC_OFF = BWD_STEP * 32u;

// splits the step's masks; returns the padded row count (a multiple of UNROLL); rank[g] = position of
// this lane's entry in group g's list (meaningful where its mask bit is set)
template <int UNROLL>
__device__ __forceinline__ int split_step(TileLds& lds, uint32_t gm, int lane, int* rank) {
    int cnt[4], rows = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const bool bit = gm >> g & 1u;
        const uint64_t m = __ballot(bit);
        rank[g] = lanes_below(m);
        if (bit) lds.list[g][rank[g]] = (uint16_t)(lane * 32);
        cnt[g] = __builtin_popcountll(m);
        rows = cnt[g] > rows ? cnt[g] : rows;
    }
    rows = (rows + UNROLL - 1) / UNROLL * UNROLL;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (cnt[g] + lane < rows + UNROLL) lds.list[g][cnt[g] + lane] = (uint16_t)ZERO_REC_OFF;   // + UNROLL: the prefetch
    return rows;
}

// ------------------------------------------------------------------------------------------
// Backward: the same tile / list structure.  Every (row, Gaussian) pair of a step yields
// NV = 5 + c per-lane contributions that must be summed over the row's 16 points.  The row sums
// are written -- plain stores, no read-modify-write: a (group, list position) pair is met once --
// into an LDS table indexed by group and list position; at the end of the step every lane, which
// knows the positions of its own entry in the (up to four) group lists from the split, adds its
// rows of the table and flushes ONE atomic per entry and value into gacc[k][j] (entries follow the
// sorted order, so consecutive lanes hit near-consecutive addresses); plan_unpermute_kernel writes
// the caller's layout.  (LDS float atomics into a per-entry table took 44 LDS cycles per
// instruction here: the kernel ran at the LDS's pace.)
// ------------------------------------------------------------------------------------------
// Round 4: the table holds HALF a step's list positions (BWD_HALF); the rows of a step are taken in two halves and an
// entry's lane collects its rows of the table after each (registers), so that the step's LDS is 6.1 KB per wave instead
// of 9.8 and SIX workgroups fit a CU where four did (the row arithmetic is what bounds the kernel).
constexpr int BWD_HALF = BWD_STEP / 2;
template <int NV>
struct TileLdsBwd {
    static constexpr int S = NV <= 6 ? 6 : 8;        // floats per table row (8-byte aligned)
    TileLds t;
    float sums[4][BWD_HALF + 4][S];                  // [group][list position - first of the half]: reduced contributions
};

// Row sums of FOUR wave-rows at once by a transposing fold.  Input: v[u][k], u = 0..3 (four consecutive
// list rows), k < NV, each to be summed over the 16 lanes of every DPP row.  Two folding levels merge
// the four u of one k into ONE register while they halve the lanes twice: bank_mask lets a DPP add
// write only some of a row's four banks (4 lanes each), so two adds build one merged register --
//   level A (row_ror:8, lanes i <-> i^8):       banks {0,1} <- v[u0] ,  banks {2,3} <- v[u1]
//   level B (row_half_mirror, i <-> 7-i of 8):   banks {0,2} <- first ,  banks {1,3} <- second
// -- then two quad steps finish the sum inside each bank.  6 + 2 instructions per k and four rows
// = 2 NV per row instead of 4 NV.  Afterwards every lane of bank b of a DPP row holds, in z[k], the
// sum over that row's 16 lanes of v[SIGMA(b)][k], SIGMA = {0, 2, 1, 3}.
#define PIGS_FOLD_A(OUT, X, Y)                                                            \
    "v_add_f32_dpp " OUT ", " X ", " X " row_ror:8 row_mask:0xf bank_mask:0x3\n\t"        \
    "v_add_f32_dpp " OUT ", " Y ", " Y " row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
#define PIGS_FOLD_B(OUT, X, Y)                                                            \
    "v_add_f32_dpp " OUT ", " X ", " X " row_half_mirror row_mask:0xf bank_mask:0x5\n\t"  \
    "v_add_f32_dpp " OUT ", " Y ", " Y " row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
// two values k at a time: their chains are interleaved, so that one s_nop at the head covers the two
// wait states a DPP read needs behind the VALU write of its source
__device__ __forceinline__ void fold4_pair(float& z0, float& z1, float a0, float a1, float a2, float a3, float b0,
                                           float b1, float b2, float b3) {
    float t0, t1, t2, t3;
    asm volatile("s_nop 1\n\t"
                 PIGS_FOLD_A("%2", "%6", "%7") PIGS_FOLD_A("%3", "%8", "%9")
                 PIGS_FOLD_A("%4", "%10", "%11") PIGS_FOLD_A("%5", "%12", "%13")
                 PIGS_FOLD_B("%0", "%2", "%3") PIGS_FOLD_B("%1", "%4", "%5")
                 : "=&v"(z0), "=&v"(z1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}
__device__ __forceinline__ void fold4_single(float& z0, float a0, float a1, float a2, float a3) {
    float t0, t1;
    asm volatile("s_nop 1\n\t"
                 PIGS_FOLD_A("%1", "%3", "%4") PIGS_FOLD_A("%2", "%5", "%6")
                 "s_nop 1\n\t"
                 PIGS_FOLD_B("%0", "%1", "%2")
                 : "=&v"(z0), "=&v"(t0), "=&v"(t1)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
}
#define PIGS_QUAD1(MOD, R) "v_add_f32_dpp " R ", " R ", " R " " MOD " row_mask:0xf bank_mask:0xf\n\t"
#define PIGS_QUAD6(MOD) PIGS_QUAD1(MOD, "%0") PIGS_QUAD1(MOD, "%1") PIGS_QUAD1(MOD, "%2") PIGS_QUAD1(MOD, "%3") \
    PIGS_QUAD1(MOD, "%4") PIGS_QUAD1(MOD, "%5")
template <int NV>
__device__ __forceinline__ void quad_sums(float* z) {
    static_assert(NV == 6 || NV == 7, "5 + c values");
    if constexpr (NV == 6)
        asm volatile("s_nop 1\n\t" PIGS_QUAD6("quad_perm:[1,0,3,2]") PIGS_QUAD6("quad_perm:[2,3,0,1]") "s_nop 1"
                     : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]));
    else
        asm volatile("s_nop 1\n\t" PIGS_QUAD6("quad_perm:[1,0,3,2]") PIGS_QUAD1("quad_perm:[1,0,3,2]", "%6")
                     PIGS_QUAD6("quad_perm:[2,3,0,1]") PIGS_QUAD1("quad_perm:[2,3,0,1]", "%6") "s_nop 1"
                     : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]));
}

// rows: a multiple of 4 (split_step<4> pads the lists with the all-zero record; the sums of such rows
// land behind the lists' ends in the table and are never read)
template <int C, int MASK>      // MASK: the mask of the arithmetic (a residual's: ORDR_AS); list rows r0 .. r0 + rows - 1
__device__ __forceinline__ void backward_rows(const float* s, const Gsym<float, 2, C, MASK>& G,
                                              TileLdsBwd<BwdLayout<2, C>::N>& lds, int r0, int rows, int lane) {
    using BL = BwdLayout<2, C>;
    constexpr int NV = BL::N;
    constexpr int S = TileLdsBwd<NV>::S;
    const char* base = (const char*)&lds.t.rec[0][0];
    const int g = lane >> 4;
    const uint16_t* lst = lds.t.list[g];
    const int bank = (lane >> 2) & 3;
    const int sigma = ((bank & 1) << 1) | (bank >> 1);      // {0, 2, 1, 3}: the list row whose sums this lane's bank ends up with
    const bool leader = (lane & 3) == 0;
    for (int k0 = 0; k0 < rows; k0 += 4) {
        float part[4][NV];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t off = lst[r0 + k0 + u];
            const Rec r = make_rec(*(const float4*)(base + off), *(const float4*)(base + off + 16));
#pragma unroll
            for (int q = 0; q < NV; ++q) part[u][q] = 0.f;
            bwd_accumulate<float, 2, C, MASK, (MASK & ORD3) != 0, C == 1>(part[u], s, r.mu, r.con, r.v, G);
        }
        float z[8];
#pragma unroll
        for (int q = 0; q + 1 < NV; q += 2)
            fold4_pair(z[q], z[q + 1], part[0][q], part[1][q], part[2][q], part[3][q], part[0][q + 1], part[1][q + 1],
                       part[2][q + 1], part[3][q + 1]);
        if constexpr (NV & 1) {
            fold4_single(z[NV - 1], part[0][NV - 1], part[1][NV - 1], part[2][NV - 1], part[3][NV - 1]);
            z[NV] = 0.f;
        }
        quad_sums<NV>(z);
        if (leader) {
            float2* dst = (float2*)lds.sums[g][k0 + sigma];
#pragma unroll
            for (int q = 0; q < S; q += 2) dst[q / 2] = make_float2(z[q], z[q + 1]);
        }
    }
}

// Tiles that run at the same time should not be neighbours in the domain: neighbouring tiles share
// most of their Gaussians, their waves start together and move in step, and their atomics then meet
// on the same cache lines at the same moment (same-line atomics retire one every ~10 ns; measured
// 82 -> 67 us at C3).  A multiplicative shuffle by a prime that does not divide the tile count
// (one of five whose product exceeds any tile count) is a bijection on [0, ntiles).
#ifndef PIGS_BWD_SPREAD
#define PIGS_BWD_SPREAD 2     // 0 = tiles in launch order, 1 = shuffled over the whole domain (66.9 us),
                              // 2 = inside the XCD chunks of 1024 tiles (65.5 us; the lines stay in one L2)
#endif
__device__ __forceinline__ uint32_t spread_tile(uint32_t t, uint32_t ntiles) {
    if (t >= ntiles) return t;           // the launch's padding: stays outside
#if PIGS_BWD_SPREAD == 1
    const uint32_t p = ntiles % 37u ? 37u : ntiles % 41u ? 41u : ntiles % 43u ? 43u : ntiles % 47u ? 47u : 53u;
    return (uint32_t)(((uint64_t)t * p) % ntiles);
#elif PIGS_BWD_SPREAD == 2
    const uint32_t base = t & ~1023u;
    if (base + 1024u > ntiles) return t;
    return base + ((t & 1023u) * 37u & 1023u);
#else
    return t;
#endif
}

template <int C, int MASK>
constexpr int bwd_waves() {
    // c = 2: the sums table has 8 floats per row (NV = 7), 47.7 KB of LDS per workgroup -> 3 workgroups per
    // CU whatever the registers allow, so every c = 2 variant asks for 3 waves (168 VGPRs: no spills in
    // the widest gradient sets either); c = 1 with order 3 the same for its registers
    return (C == 2 || MASK == 15) ? 3 : (MASK == 7 || MASK == 19 || MASK == ORDR || MASK == 1 || MASK == 2) ? PIGS_BWD_WAVES : 4;
}
// this lane's point of a tile and the gradients that arrive at it (lanes behind the last point: zero)
template <int C, int MASK>
__device__ __forceinline__ void load_tile_point(const SamplesView& sv, uint32_t tile, int lane, const float* __restrict__ G0p,
                                                const float* __restrict__ G1p, const float* __restrict__ G2p,
                                                const float* __restrict__ G3p, const Resid<float>& rz, SPoint& sp, bool& valid,
                                                Gsym<float, 2, C, (MASK == ORDR ? ORDR_AS : MASK)>& G,
                                                const float4* __restrict__ stage = nullptr) {
    const uint32_t m = tile * TILE_POINTS + (uint32_t)lane;
    valid = m < sv.M;
    sp = tile_point(sv, point_order(sv), tile, (uint32_t)lane);
    if constexpr (C == 1 && (MASK == 7 || MASK == 19)) {
        if (stage) {        // the incoming gradients of this point as one record (gradients_to_stage_kernel)
            const float4 a = stage[2 * (size_t)sp.m];
            G.g0[0] = a.x; G.g1[0][0] = a.y; G.g1[1][0] = a.z;
            if constexpr (MASK == 7) {
                const float4 b = stage[2 * (size_t)sp.m + 1];
                G.g2[0][0] = a.w; G.g2[1][0] = b.x; G.g2[2][0] = b.y;
            } else {
                G.g2[0][0] = a.w; G.g2[1][0] = 0.f; G.g2[2][0] = a.w;
            }
        } else {
            G.load((int64_t)sp.m, G0p, G1p, G2p, G3p);
        }
    } else if constexpr (MASK == ORDR) G.load_residual((int64_t)sp.m, G0p, rz);
    else G.load((int64_t)sp.m, G0p, G1p, G2p, G3p);
    if (!valid) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            G.g0[ch] = 0.f;
            G.g1[0][ch] = G.g1[1][ch] = 0.f;
            G.g2[0][ch] = G.g2[1][ch] = G.g2[2][ch] = 0.f;
            G.g3[0][ch] = G.g3[1][ch] = G.g3[2][ch] = G.g3[3][ch] = 0.f;
        }
    }
}

// After a half of a step's rows: this lane's entry adds its rows of the sums table (those whose list position lies in
// the half that starts at r0).
template <int NV>
__device__ __forceinline__ void collect_rows(const TileLdsBwd<NV>& lds, uint32_t gm, const int* rank, int r0, float* esum) {
    constexpr int S = TileLdsBwd<NV>::S;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int rr = rank[g] - r0;
        if ((gm >> g & 1u) && rr >= 0 && rr < BWD_HALF) {
            const float2* src = (const float2*)lds.sums[g][rr];
#pragma unroll
            for (int q = 0; q < S; q += 2) {
                const float2 v = src[q / 2];
                esum[q] += v.x; esum[q + 1] += v.y;
            }
        }
    }
}
// The end of a step: every entry's sums leave as atomics into gacc[j][8] (one 32-byte row per sorted Gaussian).  Float
// atomics execute at the memory side, one request per 64-byte segment an instruction touches (MI355X_MICROARCH.md,
// Global float atomics: full rate for 256 contiguous bytes, lanes in different rows up to 17x slower), and the entries
// of a step come in runs of consecutive sorted indices (the list build walks contiguous record ranges).  So an
// instruction takes EIGHT consecutive entries, lane = (entry, value): eight 32-byte rows, mostly adjacent -- ~2.4x
// fewer segment requests than one instruction per value over all the entries of the step (which touched every run
// once per value: round 3, gacc[8][N]).  A lane fetches its (entry, value) from the entry's lane by shuffles.
template <int NV>
__device__ __forceinline__ void flush_entries(const PlanView& pv, uint32_t gm, uint32_t idx, const float* esum, int lane) {
    const uint64_t live = __ballot((gm & 15u) != 0u);
    const int q = lane & 7, sub = lane >> 3;
#pragma unroll 2
    for (int j = 0; j < BWD_STEP / 8; ++j) {
        if (((live >> (8 * j)) & 0xffull) == 0ull) continue;           // wave-uniform: none of these eight entries reaches the tile
        const int e = 8 * j + sub;
        const uint32_t ie = (uint32_t)__shfl((int)idx, e);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const float t = __shfl(esum[k], e);
            v = q == k ? t : v;
        }
#ifndef PIGS_BWD_PROBE_NO_ATOMICS
        if ((live >> e & 1ull) && q < NV) atomicAdd(&pv.gacc[(size_t)ie * 8 + q], v);
#else
        if (v == 1.2345e-30f) pv.gacc[(size_t)ie * 8 + q] = v;      // keeps the sums alive, never stores
#endif
    }
}

// one tile through its own lists (tile list / group lists / record ranges)
template <int C, int MASK>
__device__ __forceinline__ void backward_tile(const PlanView& pv, const SamplesView& sv, uint32_t tile,
                                              TileLdsBwd<BwdLayout<2, C>::N>& lds, int lane, const float* __restrict__ G0p,
                                              const float* __restrict__ G1p, const float* __restrict__ G2p,
                                              const float* __restrict__ G3p, const Resid<float>& rz) {
    constexpr int EM = MASK == ORDR ? ORDR_AS : MASK;      // a residual's backward = orders 0, 1, trace
    using BL = BwdLayout<2, C>;
    constexpr int NV = BL::N;
    constexpr int S = TileLdsBwd<NV>::S;
    SPoint sp;
    bool valid;
    Gsym<float, 2, C, EM> G;
    load_tile_point<C, MASK>(sv, tile, lane, G0p, G1p, G2p, G3p, rz, sp, valid, G, pv.stage);
    const float s[2] = {sp.x, sp.y};
    if (lane < 2) lds.t.rec[BWD_STEP][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    // A tile that fell back to record ranges has no masks: they are found entry by entry against the
    // boxes of its four groups (a range holds every Gaussian NEAR the tile; few reach a given group when
    // the tile's points are scattered, which is when lists overflow).  Built only when the walk meets
    // such a tile, and outside `step`.
    // gradients that arrive at second / third derivatives (or the trace) use the plan's wide cut-off (plan.h)
    constexpr bool WIDE = (EM & (ORD2 | ORD3 | ORD2T)) != 0;
    auto ranges_mask = [&]() {
        const float INF = __builtin_huge_valf();
        const float q_cut = WIDE ? pv.params->q_b : pv.params->q_f;
        float x0 = valid ? sp.x : INF, x1 = valid ? sp.x : -INF, y0 = valid ? sp.y : INF, y1 = valid ? sp.y : -INF;
        row_box_dpp(x0, x1, y0, y1);
        float4 b0 = make_float4(readlane_f(x0, 0), readlane_f(y0, 0), readlane_f(x1, 0), readlane_f(y1, 0));
        float4 b1 = make_float4(readlane_f(x0, 16), readlane_f(y0, 16), readlane_f(x1, 16), readlane_f(y1, 16));
        float4 b2 = make_float4(readlane_f(x0, 32), readlane_f(y0, 32), readlane_f(x1, 32), readlane_f(y1, 32));
        float4 b3 = make_float4(readlane_f(x0, 48), readlane_f(y0, 48), readlane_f(x1, 48), readlane_f(y1, 48));
        return [=, &pv](uint32_t idx, bool have) -> uint32_t {
            const float4 A = pv.rec[2 * idx], B = pv.rec[2 * idx + 1];
            const Ellipse e = ellipse_of(A, B.x);
            const float4 bx[4] = {b0, b1, b2, b3};
            uint32_t gm = 0u;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (have && bx[g].x <= bx[g].z && ellipse_reaches_rect(e, bx[g].x, bx[g].y, bx[g].z, bx[g].w, q_cut)) gm |= 1u << g;
            return gm;
        };
    };
    if ((pv.hdr[(size_t)tile * TILE_HDR_WORDS] >> TILE_MODE_SHIFT) == TILE_MODE_POINTS) return;      // the helper workgroups' (backward_points_helper)
    for_each_step<BWD_STEP, WIDE>(pv, tile, lane, [&](uint32_t idx, uint32_t gm, bool have) {
        const float4 A = pv.rec[2 * idx], B = pv.rec[2 * idx + 1];
        wave_lds_fence();
        if (lane < BWD_STEP) {
            lds.t.rec[lane][0] = A;
            lds.t.rec[lane][1] = B;
        }
        int rank[4];
        const int rows = split_step<4>(lds.t, gm, lane, rank);
        wave_lds_fence();
        float esum[S];                        // this lane's entry: the sums of its rows of the table
#pragma unroll
        for (int q = 0; q < S; ++q) esum[q] = 0.f;
        for (int r0 = 0; r0 < rows; r0 += BWD_HALF) {        // (wave-uniform: at most two halves)
#ifndef PIGS_BWD_PROBE_NO_ROWS            // probes of tools/ab_studies.sh: the kernel without its row arithmetic / its atomics
            backward_rows<C, EM>(s, G, lds, r0, rows - r0 < BWD_HALF ? rows - r0 : BWD_HALF, lane);
#endif
            wave_lds_fence();
            collect_rows<NV>(lds, have ? gm : 0u, rank, r0, esum);
            wave_lds_fence();
        }
        flush_entries<NV>(pv, have ? gm : 0u, idx, esum, lane);
    }, ranges_mask);
}

// helper workgroups of the backward (plan.h, TILE_MODE_POINTS): four points at a time, 16 lanes per point, lane =
// candidate.  No two lanes share a (point, Gaussian) pair, so there is nothing to reduce: NV atomics per pair
// (such tiles are few, their points meet few Gaussians).
template <int C, int MASK>
__device__ __forceinline__ void backward_points_helper(const PlanView& pv, const SamplesView& sv, uint32_t hw, uint32_t nhw, int lane,
                                                       const float* __restrict__ G0p, const float* __restrict__ G1p,
                                                       const float* __restrict__ G2p, const float* __restrict__ G3p,
                                                       const Resid<float>& rz) {
    constexpr int EM = MASK == ORDR ? ORDR_AS : MASK;
    constexpr int NV = BwdLayout<2, C>::N;
    constexpr bool WIDE = (EM & (ORD2 | ORD3 | ORD2T)) != 0;
    const uint32_t n = pv.params->n_points;
    if (n == 0u) return;
    const float q_cut = WIDE ? pv.params->q_b : pv.params->q_f;
    const int row = lane >> 4, i = lane & 15;
    for (uint32_t qd = hw; qd < n * 16u; qd += nhw) {
        const uint32_t m = pv.ptiles[qd >> 4] * TILE_POINTS + (qd & 15u) * 4u + (uint32_t)row;
        const bool valid = m < sv.M;
        const SPoint sp = tile_point(sv, point_order(sv), pv.ptiles[qd >> 4], (qd & 15u) * 4u + (uint32_t)row);
        const float s[2] = {sp.x, sp.y};
        Gsym<float, 2, C, EM> G;
        if constexpr (MASK == ORDR) G.load_residual((int64_t)sp.m, G0p, rz);
        else G.load((int64_t)sp.m, G0p, G1p, G2p, G3p);
        // The walk in step over the whole wave (round 4): every lane meets its own (point, Gaussian) pair, and a pair's
        // NV sums leave as ONE atomic request -- lane = (pair, value), eight pairs per instruction, each a pair's 32-byte
        // row of gacc -- where an instruction per value with 64 different Gaussians in its lanes was 64 requests, NV
        // times over (the memory side takes one request per 64-byte segment: a clamped-normal cloud's backward spent
        // ~150 of its 240 us in these tiles).
        walk_point_instep(pv, sp.x, sp.y, i, [&](bool have, uint32_t j, const float4 A, const float4 B) {
            const bool hit = have && valid && !(pair_q(A, B, sp.x, sp.y) > q_cut);
            float part[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) part[q] = 0.f;
            if (hit) {
                const Rec r = make_rec(A, B);
                bwd_accumulate<float, 2, C, EM, (EM & ORD3) != 0, C == 1>(part, s, r.mu, r.con, r.v, G);
            }
            const uint64_t hm = __ballot(hit);
            const int q = lane & 7, sub = lane >> 3;
            for (int it = 0; it < 8; ++it) {
                if (((hm >> (8 * it)) & 0xffull) == 0ull) continue;          // wave-uniform
                const int src = 8 * it + sub;
                const uint32_t js = (uint32_t)__shfl((int)j, src);
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const float t = __shfl(part[k], src);
                    v = q == k ? t : v;
                }
                if ((hm >> src & 1ull) && q < NV) atomicAdd(&pv.gacc[(size_t)js * 8 + q], v);
            }
        });
    }
}

template <int C, int MASK>
__global__ __launch_bounds__(256, (bwd_waves<C, MASK>())) void tile_backward_kernel(
    PlanView pv, SamplesView sv, const float* __restrict__ G0p, const float* __restrict__ G1p,
    const float* __restrict__ G2p, const float* __restrict__ G3p, Resid<float> rz) {
    __shared__ TileLdsBwd<BwdLayout<2, C>::N> lds_all[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nmain = (sv.ntiles + 3u) / 4u;
    static_assert(POINT_HELPER_BLOCKS % 8u == 0u, "the main workgroups keep their XCD");
    if (blockIdx.x < POINT_HELPER_BLOCKS) {        // the helpers come first in the launch (tile_forward_kernel)
        backward_points_helper<C, MASK>(pv, sv, blockIdx.x * 4u + (uint32_t)wave, POINT_HELPER_BLOCKS * 4u, lane, G0p, G1p, G2p, G3p, rz);
        return;
    }
    const uint32_t tile = spread_tile(xcd_block(nmain, blockIdx.x - POINT_HELPER_BLOCKS) * 4 + (uint32_t)wave, sv.ntiles);
    if (tile >= sv.ntiles) return;
    backward_tile<C, MASK>(pv, sv, tile, lds_all[wave], lane, G0p, G1p, G2p, G3p, rz);
}

// ------------------------------------------------------------------------------------------
// Backward over BLOCK lists: one wave = one block of four consecutive tiles (256 consecutive sorted points:
// a 16 x 16 patch of a grid), the unit the list build walks the grid for.  Neighbouring tiles share most of
// their Gaussians (a block's list holds ~1.9x what one tile's does, not 4x), and what bounds the tile-by-
// tile backward beside its arithmetic is the L2's atomic rate (one cache-line operation per ~40 ps
// chip-wide; a tile's entries touch ~12 lines per value): here an entry's sums over all four tiles are
// added up in the lane's registers and leave as ONE atomic per entry and value.  A step = 64 entries of the
// block list (records to LDS once); for each of the four tiles: its point and incoming gradients, the
// entries' masks for this tile split into the four per-row lists, the rows (as in backward_tile), the
// lane's rows of the sums table added to its registers.
// ------------------------------------------------------------------------------------------
template <int C, int MASK>
__global__ __launch_bounds__(256, (bwd_waves<C, MASK>())) void block_backward_kernel(
    PlanView pv, SamplesView sv, const float* __restrict__ G0p, const float* __restrict__ G1p,
    const float* __restrict__ G2p, const float* __restrict__ G3p, Resid<float> rz) {
    constexpr int EM = MASK == ORDR ? ORDR_AS : MASK;
    using BL = BwdLayout<2, C>;
    constexpr int NV = BL::N;
    constexpr int S = TileLdsBwd<NV>::S;
    constexpr bool WIDE = (EM & (ORD2 | ORD3 | ORD2T)) != 0;
    __shared__ TileLdsBwd<NV> lds_all[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nblocks = (sv.ntiles + 3u) / 4u;
    if (blockIdx.x >= (nblocks + 3u) / 4u) {
        backward_points_helper<C, MASK>(pv, sv, (blockIdx.x - (nblocks + 3u) / 4u) * 4u + (uint32_t)wave,
                                        (gridDim.x - (nblocks + 3u) / 4u) * 4u, lane, G0p, G1p, G2p, G3p, rz);
        return;
    }
    // the same strips of the domain on the same XCD as the list build's; inside an XCD's chunk of 256 blocks the
    // blocks that run together are dealt apart (neighbours flush the same lines at the same moment: spread_tile)
    uint32_t block = xcd_block_chunk<PIGS_XCD_CHUNK / 4>((nblocks + 3u) / 4u) * 4 + (uint32_t)wave;
    if (block >= nblocks) return;
    {
        const uint32_t base = block & ~255u;
        if (PIGS_BWD_SPREAD != 0 && base + 256u <= nblocks) block = base + ((block & 255u) * 37u & 255u);
    }
    TileLdsBwd<NV>& lds = lds_all[wave];
    const uint32_t tile0 = block * 4;
    const uint32_t bh = pv.hdr[(size_t)tile0 * TILE_HDR_WORDS + 5];
    if (!(bh >> 31)) {                       // no block list (very wide Gaussians, scattered points): tile by tile
        for (uint32_t t = 0; t < 4 && tile0 + t < sv.ntiles; ++t)
            backward_tile<C, MASK>(pv, sv, tile0 + t, lds, lane, G0p, G1p, G2p, G3p, rz);
        return;
    }
    const uint32_t nb = bh & 0x7fffffffu;
    const uint2* blist = (const uint2*)(pv.tlist + (size_t)tile0 * pv.list_cap);
    if (lane < 2) lds.t.rec[BWD_STEP][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    // the four tiles' points and incoming gradients stay in registers for the whole block (4 x 10 VGPRs; the
    // kernel's occupancy is set by its LDS): loaded once, all in flight together
    // -- where they are few (one channel, orders up to 2); the wide gradient sets (two channels, third
    // derivatives: up to 20 values per tile) are fetched again for every (step, tile) instead of spilling
    constexpr bool KEEP = FwdLayout<2, C, EM>::N <= 7;
    constexpr int NK = KEEP ? 4 : 1;
    SPoint sp[NK];
    bool valid[NK];
    Gsym<float, 2, C, EM> G[NK];
    if constexpr (KEEP) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t tt = tile0 + (uint32_t)t < sv.ntiles ? tile0 + (uint32_t)t : tile0;      // a ragged last block repeats tile 0 (its masks are zero)
            load_tile_point<C, MASK>(sv, tt, lane, G0p, G1p, G2p, G3p, rz, sp[t], valid[t], G[t]);
        }
    }
    for (uint32_t e0 = 0; e0 < nb; e0 += BWD_STEP) {
        const bool have = lane < BWD_STEP && e0 + (uint32_t)lane < nb;
        const uint2 ent = have ? blist[e0 + lane] : make_uint2(0u, 0u);
        const uint32_t idx = ent.x;
        const float4 A = pv.rec[2 * idx], B = pv.rec[2 * idx + 1];
        wave_lds_fence();
        if (lane < BWD_STEP) {
            lds.t.rec[lane][0] = A;
            lds.t.rec[lane][1] = B;
        }
        float sum[S];
#pragma unroll
        for (int q = 0; q < S; ++q) sum[q] = 0.f;
        uint32_t any = 0u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t byte = ent.y >> (8 * t) & 0xffu;
            const uint32_t gm = WIDE ? byte >> 4 : byte & 15u;
            if (__ballot(gm != 0u) == 0ull) continue;         // no entry of this step reaches this tile
            const int tk = KEEP ? t : 0;
            if constexpr (!KEEP) load_tile_point<C, MASK>(sv, tile0 + (uint32_t)t, lane, G0p, G1p, G2p, G3p, rz, sp[0], valid[0], G[0]);
            const float s[2] = {sp[tk].x, sp[tk].y};
            int rank[4];
            wave_lds_fence();
            const int rows = split_step<4>(lds.t, gm, lane, rank);
            wave_lds_fence();
            any |= gm;
            for (int r0 = 0; r0 < rows; r0 += BWD_HALF) {
                backward_rows<C, EM>(s, G[tk], lds, r0, rows - r0 < BWD_HALF ? rows - r0 : BWD_HALF, lane);
                wave_lds_fence();
                collect_rows<NV>(lds, gm, rank, r0, sum);
                wave_lds_fence();
            }
        }
        if (have && any != 0u) {
#pragma unroll
            for (int q = 0; q < NV; ++q) atomicAdd(&pv.gacc[(size_t)idx * 8 + q], sum[q]);
        }
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
    float v[8];
    {
        float4* row = (float4*)(pv.gacc + (size_t)j * 8);
        const float4 lo = row[0], hi = row[1];
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        row[0] = make_float4(0.f, 0.f, 0.f, 0.f);      // leave the scratch zeroed for the next backward
        row[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (C == 1) {
        // the backward accumulates the factored sums (pair_math.h, FACTORED): finish them with the
        // Gaussian's own conic and value
        const float4 A = pv.rec[2 * j], B = pv.rec[2 * j + 1];
        const float a = A.z, b = A.w, c = B.x, val = B.y;
        const float sx = v[BL::MU + 0], sy = v[BL::MU + 1];
        v[BL::MU + 0] = val * (a * sx + b * sy);
        v[BL::MU + 1] = val * (b * sx + c * sy);
#pragma unroll
        for (int k = 0; k < 3; ++k) v[BL::CON + k] *= val;
    }
    g_means[2 * n] = v[BL::MU + 0];
    g_means[2 * n + 1] = v[BL::MU + 1];
#pragma unroll
    for (int k = 0; k < 3; ++k) g_conics[3 * n + k] = v[BL::CON + k];
#pragma unroll
    for (int k = 0; k < C; ++k) g_values[(size_t)C * n + k] = v[BL::VAL + k];
}

// ------------------------------------------------------------------------------------------
// staging launches (PlanView::stage): thread = point in the CALLER's order, everything coalesced
// ------------------------------------------------------------------------------------------
template <int MASK>
__global__ __launch_bounds__(256) void stage_to_outputs_kernel(const float4* __restrict__ stage, uint32_t M, float* __restrict__ o0,
                                                               float* __restrict__ o1, float* __restrict__ o2) {
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float4 a = stage[2 * (size_t)m];
    if (o0) __builtin_nontemporal_store(a.x, &o0[m]);
    if (o1) { __builtin_nontemporal_store(a.y, &o1[2 * (size_t)m]); __builtin_nontemporal_store(a.z, &o1[2 * (size_t)m + 1]); }
    if constexpr (MASK == 7) {
        const float4 b = stage[2 * (size_t)m + 1];
        if (o2) {
            float* h = o2 + 4 * (size_t)m;
            __builtin_nontemporal_store(a.w, h); __builtin_nontemporal_store(b.x, h + 1);
            __builtin_nontemporal_store(b.y, h + 2); __builtin_nontemporal_store(b.z, h + 3);
        }
    } else {
        if (o2) __builtin_nontemporal_store(a.w, &o2[m]);
    }
}
// the incoming gradients as Gsym holds them: {g0, g1x, g1y, g2_xx}, {g2_xy + g2_yx, g2_yy, 0, 0} (null arrays: zero)
template <int MASK>
__global__ __launch_bounds__(256) void gradients_to_stage_kernel(float4* __restrict__ stage, uint32_t M, const float* __restrict__ G0,
                                                                 const float* __restrict__ G1, const float* __restrict__ G2) {
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float g0 = G0 ? G0[m] : 0.f;
    const float2 g1 = G1 ? ((const float2*)G1)[m] : make_float2(0.f, 0.f);
    if constexpr (MASK == 7) {
        const float4 g2 = G2 ? ((const float4*)G2)[m] : make_float4(0.f, 0.f, 0.f, 0.f);
        stage[2 * (size_t)m] = make_float4(g0, g1.x, g1.y, g2.x);
        stage[2 * (size_t)m + 1] = make_float4(0.f + g2.y + g2.z, g2.w, 0.f, 0.f);
    } else {
        stage[2 * (size_t)m] = make_float4(g0, g1.x, g1.y, G2 ? G2[m] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static bool samples_supported(int64_t M) { return M >= 1 && M < (1LL << 31) - 64; }
static bool plan_supported(int64_t N, int64_t M, int c) {
    return samples_supported(M) && N >= 1 && c >= 1 && c <= 2 && N < (1LL << LIST_IDX_BITS);
}

static SamplesView make_samples_view(const SamplesLayout& p, const void* sws) {
    const char* b = (const char*)sws;
    SamplesView v{};
    v.params = (const SampleParams*)(b + p.off_params);
    v.spts = (const SPoint*)(b + p.off_spts);
    v.M = (uint32_t)p.M;
    v.ntiles = p.ntiles;
    return v;
}

static PlanView make_view(const PlanLayout& p, void* ws, float q_max) {
    char* b = (char*)ws;
    PlanView v{};
    v.params = (const PlanParams*)(b + p.off_params);
    v.starts = (const uint32_t*)(b + p.off_starts);
    v.rec = (const float4*)(b + p.off_rec);
    v.gbox = (const float4*)(b + p.off_box);
    v.g2o = (const uint32_t*)(b + p.off_g2o);
    v.hdr = (const uint32_t*)(b + p.off_hdr);
    v.tlist = (const uint32_t*)(b + p.off_tlist);
    v.glist = (const uint32_t*)(b + p.off_glist);
    v.ptiles = (const uint32_t*)(b + p.off_ptiles);
    v.N = (uint32_t)p.N;
    v.list_cap = p.list_cap;
    v.G0 = p.G0; v.L = p.L;
    for (int l = 0; l <= PLAN_MAX_LEVELS; ++l) v.level_off[l] = p.level_off[l];
    v.q_max = q_max;
    v.gacc = (float*)(b + p.off_gacc);
    v.pbox = (const float4*)(b + p.off_pbox);
    v.sbox = (const float4*)(b + p.off_sbox);
    v.stage = (float4*)(b + p.off_stage);
    return v;
}

size_t samples_error_offset() { return offsetof(SampleParams, scan_error); }
size_t plan_error_offset() { return offsetof(PlanParams, scan_error); }
size_t samples_lattice_offset() { return offsetof(SampleParams, lat); }
size_t plan_strips_offset() { return offsetof(PlanParams, strips); }      // (the parameters open the workspace)

int plan_layout_info(int64_t N, int64_t M, int c, int64_t* info) {
    if (!plan_supported(N, M, c)) return PIGS_ERR_UNSUPPORTED;
    const PlanLayout p = make_plan_layout(N, M, c);
    info[0] = p.ntiles; info[1] = p.list_cap; info[2] = (int64_t)p.off_hdr; info[3] = (int64_t)p.off_tlist;
    info[4] = (int64_t)p.off_g2o; info[5] = (int64_t)p.off_glist;
    return PIGS_OK;
}

size_t samples_workspace_bytes(int64_t M) {
    if (!samples_supported(M)) return 0;
    return make_samples_layout(M).total_bytes;
}

size_t plan_workspace_bytes(int64_t N, int64_t M, int c) {
    if (!plan_supported(N, M, c)) return 0;
    return make_plan_layout(N, M, c).total_bytes;
}

constexpr int64_t LATTICE_MIN_POINTS = 1 << 12;
static void fill_samples_args(BuildArgs& a, const SamplesLayout& s, void* sws, const void* samples, bool coarse) {
    char* b = (char*)sws;
    a.sparams = (SampleParams*)(b + s.off_params);
    a.sboxes = (float4*)(b + s.off_boxes);
    a.slat = (float4*)(b + s.off_lat);
    {   // index-tiled order: from LATTICE_MIN_POINTS points on.  (Until the Gaussians ran one launch ahead -- BuildArgs::ahead
        // -- the bar stood at 2^18 points: detecting a lattice adds ~2.5 us to the first launch, sorting one costs ~1 us of
        // the count and scatter launches per 131 072 points.  A lattice that is expected now saves a whole launch: 128^2 ...
        // 384^2 grids, cold step -4 ... -5 us; BASELINE configs[1] 34.1 -> 31.5 us.)  PIGS_LATTICE=1 / 0: always / never
        const char* e = getenv("PIGS_LATTICE");
        a.no_lattice = e ? e[0] == '0' : s.M < LATTICE_MIN_POINTS;
    }
    a.skey = (uint2*)(b + s.off_skey);
    a.spts = (SPoint*)(b + s.off_spts);
    a.samples = (const float*)samples;
    a.M = (uint32_t)s.M;
    a.scells_cap = s.scells_cap;
    a.coarse = coarse;
    a.cells_per_bin = s.cells_per_bin; a.h_chunk = s.h_chunk; a.h_wgs = s.h_wgs;
    a.tmp = (STmp*)(b + s.off_tmp);
    if (coarse) {      // the scan runs over the (bin, workgroup) count matrix, which every build overwrites whole
        a.scounts = (uint32_t*)(b + s.off_hist);
        a.sagg = (unsigned long long*)(b + s.off_hagg);
        a.sstarts = (uint32_t*)(b + s.off_hstarts);
        a.s_scan_blocks = s.h_scan_blocks;
        a.szero = (uint32_t*)(b + s.off_hagg);
        a.s_zero_words = (uint32_t)((s.off_hstarts - s.off_hagg) / 4);    // the scan's aggregates
    } else {
        a.scounts = (uint32_t*)(b + s.off_counts);
        a.sagg = (unsigned long long*)(b + s.off_agg);
        a.sstarts = (uint32_t*)(b + s.off_starts);
        a.s_scan_blocks = s.scan_blocks;
        a.szero = a.scounts;
        a.s_zero_words = (uint32_t)((s.off_starts - s.off_counts) / 4);     // counters + aggregates
    }
}

// ---- which way a samples build sorts its points (plan.h, SamplesLayout) ----
// Unordered points want the coarse-bin path, points in runs the one-pass build, and the host cannot look at
// the points without a synchronisation.  So every build of a large point set leaves {runs, points} of a sample
// of its waves in the workspace header, the library copies that pair to pinned memory on the build's stream
// behind the build (no wait), and the NEXT build of a point set of the same size on the same device takes what
// the last completed copy says: a training loop that draws new random collocation points every step
// (main_pn.py:103, test_no_mlp.py:86) switches after its first step or two, a lattice never does.  A capture
// neither asks nor copies (it keeps the mode of the moment).  PIGS_SAMPLES_ORDER = ordered | unordered in the
// environment, or the PIGS_BUILD_POINTS_* flags, overrule the memory; the result is the same either way
// (order inside a fine cell aside), only the time differs.
// Where the two mechanisms start to pay, measured on uniform random points with one Gaussian per 16 points (one box,
// cold step / forward launches / backward launches, us): 256^2 points: one-pass 49.1 / 18.7 / 19.1, coarse-bin 51.1,
// staged 23.0 / 22.4; 384^2: 61.1 vs 58.9 cold, staged forward 20.8 vs 14.3; 512^2: 70.8 vs 66.9 cold, staged
// 32.2 / 45.2 vs 30.4 / 39.5; 1024^2: 162 vs 122 cold, staged 49 / 104 vs 57 / 110.  Below ~100 k points the extra
// launch of the coarse-bin build costs more than the atomics it saves, and the staging launch more than the
// scattered sectors up to ~500 k.
constexpr int64_t COARSE_MIN_POINTS = 1 << 17;
constexpr int64_t STAGE_MIN_POINTS = 1 << 19;
struct OrderHint {
    int device = -1;
    int64_t M = 0;
    bool coarse = false, pending = false;
    uint32_t builds = 0;           // samples builds of this (device, M) so far: the statistic is asked for after the first
                                   // two and after every 16th (the 8-byte copy is a 4 us blit in the build's stream)
    hipEvent_t ev = nullptr;
    uint32_t* host = nullptr;      // pinned: the first 64 bytes of SampleParams (box, grid, order_stat, lat_cand, lat)
    uint32_t rf = 0;               // the row length of the last completed build when it took the index-tiled order (else 0)
    float box[4] = {0.f, 0.f, 0.f, 0.f};   // the samples' bounding box of the last completed build
    bool box_seen = false;         // ... and whether the completed build before it had the same one: a box to build the
    bool box_stable = false;       //     Gaussians' grid on before this build's own is known (BuildArgs::ahead)
    uint32_t same = 0;             // completed copies in a row that said the same (row length, box): the copies get rarer
    uint64_t stamp = 0;
};
constexpr uint32_t HINT_WORDS = 16;
static_assert(offsetof(SampleParams, box) == 0 && offsetof(SampleParams, order_stat) == 40 && offsetof(SampleParams, lat) == 56,
              "the hint copy: the first 64 bytes of SampleParams");
static std::mutex g_hint_mu;
static OrderHint g_hints[16];
static uint64_t g_hint_clock = 0;

static void hint_poll(OrderHint& h);
static OrderHint* hint_entry(int device, int64_t M, bool create) {      // g_hint_mu held; create: never inside a capture
    OrderHint* lru = nullptr;
    for (auto& h : g_hints) {
        if (h.device == device && h.M == M) { h.stamp = ++g_hint_clock; return &h; }
        if (!h.pending && (!lru || h.stamp < lru->stamp)) lru = &h;
    }
    if (!create) return nullptr;
    if (!lru) {
        // every slot waits for a copy (sizes that were built once and never came back): the copies have long
        // landed -- take note of them and give the least recently used slot away
        for (auto& h : g_hints) {
            hint_poll(h);
            if (!h.pending && (!lru || h.stamp < lru->stamp)) lru = &h;
        }
        if (!lru) return nullptr;
    }
    if (lru->ev && lru->device != device) {      // an event belongs to the device it was created on
        (void)hipEventDestroy(lru->ev);
        lru->ev = nullptr;
    }
    if (!lru->ev && hipEventCreateWithFlags(&lru->ev, hipEventDisableTiming) != hipSuccess) { lru->ev = nullptr; (void)hipGetLastError(); return nullptr; }
    if (!lru->host && hipHostMalloc((void**)&lru->host, HINT_WORDS * sizeof(uint32_t), hipHostMallocPortable) != hipSuccess) { lru->host = nullptr; (void)hipGetLastError(); return nullptr; }
    lru->device = device; lru->M = M; lru->coarse = false; lru->pending = false; lru->builds = 0; lru->rf = 0; lru->stamp = ++g_hint_clock;
    lru->box_seen = lru->box_stable = false; lru->same = 0;
    return lru;
}

static bool stream_capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

static void hint_poll(OrderHint& h) {           // g_hint_mu held
    if (!h.pending) return;
    const hipError_t q = hipEventQuery(h.ev);
    (void)hipGetLastError();          // hipErrorNotReady is an answer, not a failure
    if (q == hipSuccess) {
        h.pending = false;
        const uint32_t rf_before = h.rf;
        h.rf = h.host[14];         // SampleParams::lat[0]
        if (h.rf != 0u) h.coarse = false;      // index-tiled: the points arrived in order (and left no run statistic)
        else if (h.host[11] > 0u) h.coarse = (uint64_t)h.host[10] * 100u > (uint64_t)h.host[11] * 55u;
        float b[4];
        memcpy(b, h.host, sizeof(b));
        const bool finite = fabsf(b[0]) < 3.0e38f && fabsf(b[1]) < 3.0e38f && fabsf(b[2]) < 3.0e38f && fabsf(b[3]) < 3.0e38f && b[2] > b[0] && b[3] > b[1];
        h.box_stable = finite && h.box_seen && memcmp(b, h.box, sizeof(b)) == 0;
        h.same = h.box_stable && h.rf == rf_before ? h.same + 1u : 0u;
        h.box_seen = finite;
        memcpy(h.box, b, sizeof(b));
    }
}
// order: 0 = ask the memory, 1 = one pass, 2 = coarse bins
static bool samples_take_coarse(const SamplesLayout& s, int order, hipStream_t stream) {
    if (s.cells_per_bin > SAMPLES_MAX_CELLS_PER_BIN) return false;
    if (const char* e = getenv("PIGS_SAMPLES_ORDER")) {
        if (!strcmp(e, "ordered")) order = 1;
        else if (!strcmp(e, "unordered")) order = 2;
    }
    if (order) return order == 2;
    if (s.M < COARSE_MIN_POINTS) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lock(g_hint_mu);
    OrderHint* h = hint_entry(dev, s.M, false);
    if (!h) return false;
    if (!stream_capturing(stream)) hint_poll(*h);
    return h->coarse;
}

// the row length the last completed build of M points found (index-tiled order), or 0
// *box_ok / box: the same box in the last two completed builds
static uint32_t samples_rf_hint(const SamplesLayout& s, hipStream_t stream, bool* box_ok = nullptr, float* box = nullptr) {
    if (box_ok) *box_ok = false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0u; }
    std::lock_guard<std::mutex> lock(g_hint_mu);
    OrderHint* h = hint_entry(dev, s.M, false);
    if (!h) return 0u;
    if (!stream_capturing(stream)) hint_poll(*h);
    if (box_ok && box && h->box_stable) { *box_ok = true; memcpy(box, h->box, 4 * sizeof(float)); }
    return h->rf;
}

// behind a samples build: ask for its {runs, points}
static void samples_note_order(const SamplesLayout& s, void* sws, hipStream_t stream) {
    if (s.M < 64) return;          // (the same record carries the lattice row length: wanted for every size)
    if (stream_capturing(stream)) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lock(g_hint_mu);
    OrderHint* h = hint_entry(dev, s.M, true);
    if (!h) return;
    const uint32_t nth = h->builds++;
    // (with a row length in the memory the build runs on expectations -- an eighth of the count's workgroups, the
    // Gaussians one launch ahead -- and a point set that stopped meeting them should not be met 15 more times: every 8th
    // build then; the copy is a ~4 us blit in the build's stream)
    // ... every 32nd once three copies in a row have said the same)
    if (h->pending || (nth >= 2u && (nth & (h->rf ? (h->same >= 2u ? 31u : 7u) : 15u)) != 0u)) return;
    const SampleParams* sp = (const SampleParams*)((const char*)sws + s.off_params);
    if (hipMemcpyAsync(h->host, sp, HINT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
        hipEventRecord(h->ev, stream) == hipSuccess)
        h->pending = true;
    (void)hipGetLastError();
}

// The sampling launches' question (staging, PlanView::stage): did the point set of this size arrive in no order?
// A pending statistic is looked at here too -- a point set that is built once and sampled many times (fixed random
// collocation points) never comes back to samples_take_coarse -- unless the stream is being captured.
static bool points_unordered(int64_t M, hipStream_t stream) {
    if (const char* e = getenv("PIGS_STAGE")) return e[0] == '1';          // tests / A-B runs: staging on or off whatever the memory and the size
    if (M < STAGE_MIN_POINTS) return false;
    if (const char* e = getenv("PIGS_SAMPLES_ORDER")) return !strcmp(e, "unordered");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    const bool cap = stream_capturing(stream);
    std::lock_guard<std::mutex> lock(g_hint_mu);
    for (auto& h : g_hints)
        if (h.device == dev && h.M == M) {
            if (!cap) hint_poll(h);
            return h.coarse;
        }
    return false;
}

int samples_order_hint(int64_t M) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    std::lock_guard<std::mutex> lock(g_hint_mu);
    OrderHint* h = hint_entry(dev, M, false);
    if (!h) return -1;
    hint_poll(*h);
    return h->coarse ? 1 : 0;
}

static void fill_plan_args(BuildArgs& a, const PlanLayout& p, void* ws, float q_f, float q_b, const void* means,
                           const void* conics, const void* values) {
    char* b = (char*)ws;
    a.params = (PlanParams*)(b + p.off_params);
    a.counts = (uint32_t*)(b + p.off_counts);
    a.agg = (unsigned long long*)(b + p.off_agg);
    a.starts = (uint32_t*)(b + p.off_starts);
    a.gkey = (uint2*)(b + p.off_gkey);
    a.rec = (float4*)(b + p.off_rec);
    a.gbox = (float4*)(b + p.off_box);
    a.gacc = (float*)(b + p.off_gacc);
    a.g2o = (uint32_t*)(b + p.off_g2o);
    a.pbox = (float4*)(b + p.off_pbox);
    a.sbox = (float4*)(b + p.off_sbox);
    a.parea = (float*)(b + p.off_parea);
    a.means = (const float*)means; a.conics = (const float*)conics; a.values = (const float*)values;
    a.N = (uint32_t)p.N; a.c = p.c; a.G0 = p.G0; a.L = p.L;
    a.scan_blocks = p.scan_blocks;
    a.zero_words = (uint32_t)((p.off_starts - p.off_counts) / 4);
    for (int l = 0; l <= PLAN_MAX_LEVELS; ++l) a.level_off[l] = p.level_off[l];
    a.q_f = q_f; a.q_b = q_b;
    a.q_max = q_b > q_f ? q_b : q_f;
}

// ---- deferred tile lists (PIGS_BUILD_DEFER_LISTS) ----
// A plan built with the flag has everything but its tile lists; the first pigs_plan_forward / pigs_plan_backward /
// pigs_residual_* call on that workspace builds them -- a forward in the SAME launch (plan_lists_forward_kernel).
// Which workspaces are waiting is the library's to remember (the sampling entry points carry no flags): keyed by
// the workspace's address, set or cleared by every build into it, cleared by the first sampling call.
struct DeferredLists { float q_f, q_wide; };
static std::mutex g_defer_mu;
static std::unordered_map<const void*, DeferredLists> g_deferred;
static void defer_set(const void* ws, bool on, float q_f, float q_wide) {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    if (on) g_deferred[ws] = DeferredLists{q_f, q_wide};
    else g_deferred.erase(ws);
}
static bool defer_take(const void* ws, DeferredLists& d) {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    auto it = g_deferred.find(ws);
    if (it == g_deferred.end()) return false;
    d = it->second;
    g_deferred.erase(it);
    return true;
}
// ---- does a plan of these sizes hold tiles in TILE_MODE_POINTS? ----
// The fused first forward walks such tiles in their own list wave (right, and slow when there are hundreds: the
// thin outskirts of a clustered cloud), the two-launch path spreads them over helper workgroups.  Which one a plan
// wants is known on the device only, so -- like the order of the points (OrderHint above) -- the library remembers,
// per device and (N, M): behind the list build of the first two plans of a size and of every 16th, PlanParams::
// n_points is copied to pinned memory on the build's stream (nobody waits); a first forward takes the fused launch
// unless the last completed copy for its sizes showed such tiles.  Never inside a capture.
constexpr float STRIP_MAX_COVER = 64.f;      // (plan_takes_strips below)
struct PointsHint {
    int device = -1;
    int64_t N = 0, M = 0;
    bool has_points = false, pending = false;
    float cover = -1.f;            // PlanParams::strip_cover of the last completed build (< 0: none yet)
    bool strips = false;           // ... and whether that build kept the caller's order
    uint32_t same = 0;             // completed copies in a row that led to the same choice: the copies get rarer
    uint32_t builds = 0;
    hipEvent_t ev = nullptr;
    uint32_t* host = nullptr;      // pinned {n_points, strip_cover, strips, points_wanted}
    uint64_t stamp = 0;
};
static_assert(offsetof(PlanParams, strip_cover) == offsetof(PlanParams, n_points) + 4 && offsetof(PlanParams, strips) == offsetof(PlanParams, n_points) + 8 &&
              offsetof(PlanParams, points_wanted) == offsetof(PlanParams, n_points) + 12, "one copy: n_points, strip_cover, strips, points_wanted");
static PointsHint g_phints[16];
static void phint_poll(PointsHint& h) {          // g_hint_mu held
    if (!h.pending) return;
    const hipError_t q = hipEventQuery(h.ev);
    (void)hipGetLastError();
    if (q == hipSuccess) {
        h.pending = false;
        const bool took_before = h.cover >= 0.f && h.cover <= STRIP_MAX_COVER && !h.has_points;
        h.has_points = h.host[0] != 0u || h.host[3] != 0u;
        memcpy(&h.cover, &h.host[1], sizeof(float));
        if (!(h.cover >= 0.f)) h.cover = 3.0e38f;      // NaN: as bad as it gets
        h.strips = h.host[2] != 0u;
        h.same = (h.cover <= STRIP_MAX_COVER && !h.has_points) == took_before ? h.same + 1u : 0u;
    }
}
static PointsHint* phint_entry(int device, int64_t N, int64_t M, bool create) {      // g_hint_mu held
    PointsHint* lru = nullptr;
    for (auto& h : g_phints) {
        if (h.device == device && h.N == N && h.M == M) { h.stamp = ++g_hint_clock; return &h; }
        if (!h.pending && (!lru || h.stamp < lru->stamp)) lru = &h;
    }
    if (!create) return nullptr;
    if (!lru) {
        for (auto& h : g_phints) {
            phint_poll(h);
            if (!h.pending && (!lru || h.stamp < lru->stamp)) lru = &h;
        }
        if (!lru) return nullptr;
    }
    if (lru->ev && lru->device != device) { (void)hipEventDestroy(lru->ev); lru->ev = nullptr; }
    if (!lru->ev && hipEventCreateWithFlags(&lru->ev, hipEventDisableTiming) != hipSuccess) { lru->ev = nullptr; (void)hipGetLastError(); return nullptr; }
    if (!lru->host && hipHostMalloc((void**)&lru->host, 4 * sizeof(uint32_t), hipHostMallocPortable) != hipSuccess) { lru->host = nullptr; (void)hipGetLastError(); return nullptr; }
    lru->device = device; lru->N = N; lru->M = M; lru->has_points = false; lru->pending = false; lru->builds = 0; lru->stamp = ++g_hint_clock;
    lru->cover = -1.f; lru->strips = false; lru->same = 0;
    return lru;
}
static bool plan_expects_points(int64_t N, int64_t M, hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    const bool cap = stream_capturing(stream);
    std::lock_guard<std::mutex> lock(g_hint_mu);
    PointsHint* h = phint_entry(dev, N, M, false);
    if (!h) return false;
    if (!cap) phint_poll(*h);
    return h->has_points;
}
// Does the next build of these sizes keep the caller's order (PlanParams::strips)?  Yes when the last completed build's
// strips covered the samples' domain at most STRIP_MAX_COVER times -- that number is about how many strips a block of
// tiles meets (a lattice in row order, a strip of 1 x 16 widened by its ellipses' reach on every side: ~10 times at
// kappa = 0.5, ~17 at 0.8, ~33 at 1.3; Gaussians in no order: every strip covers the domain, N / 16 times).
// PIGS_GAUSS_STRIPS=0 / 1: never / always.
constexpr int64_t STRIP_MIN_GAUSSIANS = 1024;      // (below, the chain it replaces is not what a step waits for)
static bool plan_takes_strips(int64_t N, int64_t M, hipStream_t stream) {
    if (const char* e = getenv("PIGS_GAUSS_STRIPS")) return e[0] == '1';
    // (a build that also sorts or looks at the samples saves no launch by it -- the Gaussians' count, scan and scatter ride
    // along in launches that exist anyway -- but the lists from strips are the faster ones since the survivors wait for a
    // full buffer: 20.0 against 21.3 us at C3, 55 against 60 at kappa 1.3, and the scan / scatter launches carry less)
    if (N < STRIP_MIN_GAUSSIANS) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    const bool cap = stream_capturing(stream);
    std::lock_guard<std::mutex> lock(g_hint_mu);
    PointsHint* h = phint_entry(dev, N, M, false);
    if (!h) return false;
    if (!cap) phint_poll(*h);
    // (tiles of far-apart points -- a cloud's thin outskirts -- are walked point by point through the GRID at sampling
    // time, TILE_MODE_POINTS: a plan that had such tiles keeps the cells.  Measured without this line: clamped normals
    // sigma = 0.15 over lattice Gaussians, warm step 111 -> 1 126 us.)
    return h->cover >= 0.f && h->cover <= STRIP_MAX_COVER && !h->has_points;
}
static void plan_note_points(const PlanLayout& p, const void* ws, hipStream_t stream) {      // behind a list build
    if (stream_capturing(stream)) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lock(g_hint_mu);
    PointsHint* h = phint_entry(dev, p.N, p.M, true);
    if (!h) return;
    const uint32_t nth = h->builds++;
    // (builds that keep the caller's order run on an expectation -- strips that cover the domain a few times over -- and
    // Gaussians that stopped meeting it should not be met 15 more times)
    if (h->pending || (nth >= 2u && (nth & (h->strips ? (h->same >= 2u ? 31u : 7u) : 15u)) != 0u)) return;
    const PlanParams* pp = (const PlanParams*)((const char*)ws + p.off_params);
    if (hipMemcpyAsync(h->host, &pp->n_points, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
        hipEventRecord(h->ev, stream) == hipSuccess)
        h->pending = true;
    (void)hipGetLastError();
}

static PlanView make_view(const PlanLayout& p, void* ws, float q_max);
static SamplesView make_samples_view(const SamplesLayout& p, const void* sws);
static ListArgs make_list_args(const PlanLayout& p, const SamplesLayout& s, void* ws, const void* sws, float q_f, float q_wide) {
    ListArgs la{};
    la.pv = make_view(p, ws, q_wide);
    la.q_f = q_f;
    la.sv = make_samples_view(s, sws);
    la.hdr = (uint32_t*)((char*)ws + p.off_hdr);
    la.tlist = (uint32_t*)((char*)ws + p.off_tlist);
    la.glist = (uint32_t*)((char*)ws + p.off_glist);
    la.ptiles = (uint32_t*)((char*)ws + p.off_ptiles);
    la.n_points = &((PlanParams*)((char*)ws + p.off_params))->n_points;
    la.points_wanted = &((PlanParams*)((char*)ws + p.off_params))->points_wanted;
    la.parea = (const float*)((char*)ws + p.off_parea);
    la.strip_cover = &((PlanParams*)((char*)ws + p.off_params))->strip_cover;
    return la;
}
// Small point sets (a list launch of fewer tiles than this is one sparse generation of waves): one tile per wave.
// PIGS_BWD_BLOCK builds keep four (the block lists are per four tiles).
constexpr uint32_t LISTS_SMALL_TILES = 4096;
static void launch_lists(uint32_t ntiles, const ListArgs& la, hipStream_t stream, bool strips = false) {
    const bool small = ntiles <= LISTS_SMALL_TILES && !PIGS_BWD_BLOCK && LISTS_TPW != 1;
    const dim3 grid(small ? (ntiles + 3) / 4 : (ntiles + 4 * LISTS_TPW - 1) / (4 * LISTS_TPW));
    if (small && strips) hipLaunchKernelGGL((plan_lists_kernel<1, true>), grid, dim3(256), 0, stream, la);
    else if (small) hipLaunchKernelGGL((plan_lists_kernel<1, false>), grid, dim3(256), 0, stream, la);
    else if (strips) hipLaunchKernelGGL((plan_lists_kernel<LISTS_TPW, true>), grid, dim3(256), 0, stream, la);
    else hipLaunchKernelGGL((plan_lists_kernel<LISTS_TPW, false>), grid, dim3(256), 0, stream, la);
}

// The chain bbox -> count -> scan -> scatter for the samples (build_samples), the Gaussians
// (build_plan) or both in the same four launches, then the tile lists.
static int run_build(bool do_samples, bool do_plan, bool plan_ws_clean, bool no_lookback, void* sws, size_t sws_bytes, void* ws, size_t ws_bytes, int64_t N,
                     int64_t M, int c, float q_max, float q_max_b, const void* means, const void* conics, const void* values,
                     const void* samples, hipStream_t stream, bool build_lists = true, int order = 0, bool defer_lists = false) {
    if (!samples_supported(M)) return PIGS_ERR_UNSUPPORTED;
    const SamplesLayout s = make_samples_layout(M);
    if (!sws || sws_bytes < s.total_bytes) return PIGS_ERR_WORKSPACE;
    BuildArgs a{};
    a.do_samples = do_samples; a.do_plan = do_plan; a.no_lookback = no_lookback;
    a.zero_gacc = !plan_ws_clean;
    const bool coarse = do_samples && samples_take_coarse(s, order, stream);
    fill_samples_args(a, s, sws, samples, coarse);
    bool box_ok = false;
    a.rf_hint = do_samples ? samples_rf_hint(s, stream, &box_ok, a.hint_box) : 0u;
    {
        static const char* e = getenv("PIGS_BBOX_BLOCKS");      // (A/B: 256 | 512)
        // same box, 1024^2 points: the plain pass 6.6 -> 5.9 us with 512 workgroups, the pass with a row length 7.5 -> 7.8
        a.bbox_blocks = e ? (uint32_t)atoi(e) : (M >= (int64_t)BBOX_WIDE_POINTS && a.rf_hint == 0u ? 512u : 256u);
        if (a.bbox_blocks != 512u) a.bbox_blocks = 256u;
    }
    PlanLayout p{};
    if (do_plan) {
        if (!plan_supported(N, M, c)) return PIGS_ERR_UNSUPPORTED;
        if (!(q_max > 0.f)) return PIGS_ERR_INVALID;
        if (!(q_max_b >= q_max)) q_max_b = q_max;      // <= 0 / NaN: one cut-off
        p = make_plan_layout(N, M, c);
        if (!ws || ws_bytes < p.total_bytes) return PIGS_ERR_WORKSPACE;
        fill_plan_args(a, p, ws, q_max, q_max_b, means, conics, values);
    }
    clear_hip_error();
    const uint32_t gb = do_plan ? (uint32_t)((N + 255) / 256) : 0u;
    // The Gaussians one launch ahead (BuildArgs::ahead): a lattice is expected, the samples' box was the same in the last
    // two completed builds of this size, the plan workspace's counters are zero.
    static const bool no_ahead = getenv("PIGS_NO_AHEAD") != nullptr;
    const bool ahead = do_samples && do_plan && plan_ws_clean && !coarse && !no_lookback && !PIGS_FUSED_BUILD && !no_ahead &&
                       a.rf_hint != 0u && !a.no_lattice && box_ok && s.scan_blocks <= 1024u;
    a.ahead = ahead; a.s_scan_in_scatter = ahead;
    // Gaussians whose order in the caller's array is already spatial keep it (PlanParams::strips): one pass instead of
    // count, scan and scatter
    const bool strips = do_plan && build_lists && !defer_lists && !no_lookback && !PIGS_FUSED_BUILD && !PIGS_BWD_BLOCK &&
                        plan_takes_strips(N, M, stream);
    a.strips = strips;
    if (ahead && strips) {
        // nothing of the Gaussians is left for launches 2 and 3: the samples' fall-back moves into launch 2 whole
        // (samples_sort_in_count; at most 256 workgroups: they meet at device-wide barriers) -- FOUR launches up to the forward
        a.sort_in_count = 1; a.s_scan_in_scatter = 0;
        a.s_blocks = (uint32_t)((M + 1023) / 1024);
        const uint32_t swgs = (a.s_blocks + 7u) / 8u < 256u ? (a.s_blocks + 7u) / 8u : 256u;
        hipLaunchKernelGGL(samples_bbox_kernel, dim3(a.bbox_blocks + gb), dim3(BBOX_THREADS), 0, stream, a);
        hipLaunchKernelGGL(plan_count_kernel, dim3(swgs), dim3(256), 0, stream, a);
    } else if (ahead) {
        a.s_blocks = (uint32_t)((M + 1023) / 1024);
        hipLaunchKernelGGL(samples_bbox_kernel, dim3(a.bbox_blocks + gb), dim3(BBOX_THREADS), 0, stream, a);
        hipLaunchKernelGGL(plan_count_kernel, dim3(p.scan_blocks + (a.s_blocks + 7u) / 8u), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(plan_scatter_kernel, dim3(gb + (uint32_t)((M + 255) / 256)), dim3(256), 0, stream, a);
    } else
    if (do_samples) hipLaunchKernelGGL(samples_bbox_kernel, dim3(a.bbox_blocks), dim3(BBOX_THREADS), 0, stream, a);
    else if (!plan_ws_clean) hipLaunchKernelGGL(plan_zero_kernel, dim3(64), dim3(256), 0, stream, a);
    const uint32_t fused_blocks = gb > p.scan_blocks ? gb : p.scan_blocks;
    if (ahead) {
        // (launched above)
    } else if (do_plan && !do_samples && !no_lookback && fused_blocks <= FUSED_BUILD_MAX_BLOCKS && PIGS_FUSED_BUILD) {
        hipLaunchKernelGGL(plan_gauss_build_kernel, dim3(fused_blocks), dim3(256), 0, stream, a);
    } else {
        // a lattice expected (the row length of the last build of this size): an eighth of the one-pass count's workgroups
        // -- they leave at once when the points are index-tiled, and stride over the blocks when they are not
        a.s_blocks = (uint32_t)((M + 1023) / 1024);
        const uint32_t count_wgs = coarse ? s.h_wgs : (a.rf_hint && !a.no_lattice ? (a.s_blocks + 7u) / 8u : a.s_blocks);
        hipLaunchKernelGGL(plan_count_kernel, dim3(gb + (do_samples ? count_wgs : 0u)), dim3(256), 0, stream, a);
        const uint32_t scan_wgs = (do_plan && !strips ? p.scan_blocks : 0u) + (do_samples ? a.s_scan_blocks : 0u);
        if (scan_wgs) hipLaunchKernelGGL(plan_scan_kernel, dim3(scan_wgs), dim3(256), 0, stream, a);
        const bool staged = coarse && s.h_chunk <= SCATTER_STAGE_MAX;
        const uint32_t scatter_wgs = (strips ? 0u : gb) + (do_samples ? (staged ? s.h_wgs : (uint32_t)((M + 255) / 256)) : 0u);
        if (scatter_wgs)
            hipLaunchKernelGGL(plan_scatter_kernel, dim3(scatter_wgs), dim3(256),
                               staged ? s.h_chunk * sizeof(uint4) + 2 * SAMPLES_COARSE_BINS * sizeof(uint32_t) : 0, stream, a);
        if (coarse) {
            if (s.cells_per_bin * 16u <= SAMPLES_MAX_CELLS_PER_BIN)      // the LDS holds 16 sub-cell counters per cell
                hipLaunchKernelGGL(samples_binsort_kernel<16>, dim3(SAMPLES_COARSE_BINS), dim3(1024),
                                   s.cells_per_bin * 16u * sizeof(uint32_t), stream, a);
            else
                hipLaunchKernelGGL(samples_binsort_kernel<1>, dim3(SAMPLES_COARSE_BINS), dim3(1024),
                                   s.cells_per_bin * sizeof(uint32_t), stream, a);
        }
    }
    if (do_plan && build_lists && !defer_lists)
        launch_lists(s.ntiles, make_list_args(p, s, ws, sws, q_max, a.q_max), stream, strips);
    const int rc = launch_status();
    if (do_plan) defer_set(ws, build_lists && defer_lists && rc == PIGS_OK, q_max, a.q_max);
    if (do_plan && build_lists && !defer_lists && rc == PIGS_OK) plan_note_points(p, ws, stream);
    if (do_samples && rc == PIGS_OK && order == 0) samples_note_order(s, sws, stream);
    return rc;
}

// ---- the Gaussian grid alone, for the neighbour lists of aggregate_neighbors (aggregate.hip): the
// Gaussians' own centres play the sample points (they only size the grid's domain), no tile lists.
// Workspace = samples workspace (M = N) followed by a plan workspace (N, M = N, c = 1).
size_t aggregate_grid_bytes(int64_t N) {
    if (!plan_supported(N, N, 1)) return 0;
    return align_up(make_samples_layout(N).total_bytes, 256) + make_plan_layout(N, N, 1).total_bytes;
}

int aggregate_grid_build(void* ws, size_t ws_bytes, int64_t N, float q_grid, const float* means, const float* conics,
                         hipStream_t stream) {
    if (!plan_supported(N, N, 1)) return PIGS_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < aggregate_grid_bytes(N)) return PIGS_ERR_WORKSPACE;
    const size_t sb = align_up(make_samples_layout(N).total_bytes, 256);
    // values: any N readable floats (the records' value slot is never read by the neighbour lists)
    return run_build(true, true, false, false, ws, sb, (char*)ws + sb, ws_bytes - sb, N, N, 1, q_grid, q_grid, means, conics,
                     means, means, stream, false);
}

PlanView aggregate_grid_view(void* ws, int64_t N, float q_grid) {
    const size_t sb = align_up(make_samples_layout(N).total_bytes, 256);
    return make_view(make_plan_layout(N, N, 1), (char*)ws + sb, q_grid);
}

int samples_build(void* sws, size_t sws_bytes, int64_t M, const void* samples, hipStream_t stream) {
    return run_build(true, false, false, false, sws, sws_bytes, nullptr, 0, 0, M, 1, 1.f, 1.f, nullptr, nullptr, nullptr, samples, stream);
}

int plan_build(void* ws, size_t ws_bytes, void* sws, size_t sws_bytes, int flags, int64_t N, int64_t M, int c,
               float q_max, float q_max_backward, const void* means, const void* conics, const void* values, const void* samples,
               hipStream_t stream) {
    return run_build((flags & 1) != 0, true, (flags & 2) != 0, (flags & 4) != 0, sws, sws_bytes, ws, ws_bytes, N, M, c, q_max, q_max_backward, means, conics, values,
                     samples, stream, true, (flags & 8) ? 1 : (flags & 16) ? 2 : 0, (flags & PIGS_BUILD_DEFER_LISTS) != 0);
}

// the masks the fused first forward is compiled for (the rest: the list launch, then the forward launch)
template <int C> static bool fused_first_compiled(int mask) { return C == 1 ? (mask == 1 || mask == 7 || mask == 19 || mask == 32) : mask == 7; }

template <int C>
static int plan_forward_c(const PlanView& pv_in, const SamplesView& sv, int mask, float* const* out, hipStream_t stream,
                          const Resid<float>& rz, const ListArgs* first = nullptr) {
    // + the helper workgroups of the TILE_MODE_POINTS tiles (they leave at once when the plan queued none)
    const dim3 grid((sv.ntiles + PIGS_FWD_WG_WAVES - 1) / PIGS_FWD_WG_WAVES + POINT_HELPER_BLOCKS * 4 / PIGS_FWD_WG_WAVES),
        block(64 * PIGS_FWD_WG_WAVES);
    // points that arrive in no order send their outputs through the staging records (PlanView::stage)
    PlanView pv = pv_in;
    const bool staged = C == 1 && (mask == 7 || mask == 19) && points_unordered(sv.M, stream);
    if (!staged) pv.stage = nullptr;
    clear_hip_error();
    bool done = false;
    if (first) {
        // the plan's tile lists are still to be built (PIGS_BUILD_DEFER_LISTS): in this launch where a fused kernel
        // is compiled for the mask, in a launch of their own in front of the forward otherwise
        ListArgs la = *first;
        la.pv.stage = pv.stage;
        const dim3 lgrid((sv.ntiles + 4 * LISTS_TPW - 1) / (4 * LISTS_TPW));
#define PIGS_FUSED(MK)                                                                                                  \
    case MK:                                                                                                            \
        hipLaunchKernelGGL((plan_lists_forward_kernel<C, MK>), lgrid, dim3(256), 0, stream, la, out[0], out[1], out[2], \
                           out[3], rz);                                                                                 \
        done = true;                                                                                                    \
        break;
        if (fused_first_compiled<C>(mask) && !getenv("PIGS_NO_FUSED_FIRST") && !plan_expects_points(pv.N, sv.M, stream)) {
            if constexpr (C == 1) {
                switch (mask) { PIGS_FUSED(1) PIGS_FUSED(7) PIGS_FUSED(19) PIGS_FUSED(32) }
            } else {
                switch (mask) { PIGS_FUSED(7) }
            }
        }
#undef PIGS_FUSED
        if (!done) launch_lists(sv.ntiles, la, stream);
    }
#define PIGS_CASE(MK)                                                                                          \
    case MK:                                                                                                   \
        hipLaunchKernelGGL((tile_forward_kernel<C, MK>), grid, block, 0, stream, pv, sv, out[0], out[1], out[2], \
                           out[3], rz);                                                                        \
        break;
    if (!done) switch (mask) {
        PIGS_CASE(1) PIGS_CASE(2) PIGS_CASE(4) PIGS_CASE(8) PIGS_CASE(7) PIGS_CASE(15) PIGS_CASE(16) PIGS_CASE(19) PIGS_CASE(32)
        default: return PIGS_ERR_UNSUPPORTED;
    }
#undef PIGS_CASE
    if (staged) {
        const dim3 g2((sv.M + 255) / 256), b2(256);
        if (mask == 7) hipLaunchKernelGGL(stage_to_outputs_kernel<7>, g2, b2, 0, stream, pv.stage, sv.M, out[0], out[1], out[2]);
        else hipLaunchKernelGGL(stage_to_outputs_kernel<19>, g2, b2, 0, stream, pv.stage, sv.M, out[0], out[1], out[2]);
    }
    return launch_status();
}

template <int C>
static int plan_backward_c(const PlanView& pv_in, const SamplesView& sv, int mask, const float* const* g, float* gm,
                           float* gc, float* gv, hipStream_t stream, const Resid<float>& rz) {
    const dim3 grid((sv.ntiles + 3) / 4 + POINT_HELPER_BLOCKS), block(256);
    // points that arrive in no order fetch their incoming gradients from the staging records (PlanView::stage)
    PlanView pv = pv_in;
    const bool staged = C == 1 && (mask == 7 || mask == 19) && !PIGS_BWD_BLOCK && points_unordered(sv.M, stream);
    if (!staged) pv.stage = nullptr;
    clear_hip_error();
    if (staged) {
        const dim3 g2((sv.M + 255) / 256), b2(256);
        if (mask == 7) hipLaunchKernelGGL(gradients_to_stage_kernel<7>, g2, b2, 0, stream, pv.stage, sv.M, g[0], g[1], g[2]);
        else hipLaunchKernelGGL(gradients_to_stage_kernel<19>, g2, b2, 0, stream, pv.stage, sv.M, g[0], g[1], g[2]);
    }
#if PIGS_BWD_BLOCK
    const dim3 bgrid((((sv.ntiles + 3) / 4) + 3) / 4 + POINT_HELPER_BLOCKS);
#define PIGS_CASE(MK)                                                                                             \
    case MK:                                                                                                      \
        hipLaunchKernelGGL((block_backward_kernel<C, MK>), bgrid, block, 0, stream, pv, sv, g[0], g[1], g[2], g[3], rz); \
        break;
#else
#define PIGS_CASE(MK)                                                                                             \
    case MK:                                                                                                      \
        hipLaunchKernelGGL((tile_backward_kernel<C, MK>), grid, block, 0, stream, pv, sv, g[0], g[1], g[2], g[3], rz); \
        break;
#endif
    switch (mask) {
        PIGS_CASE(1) PIGS_CASE(2) PIGS_CASE(4) PIGS_CASE(8) PIGS_CASE(7) PIGS_CASE(15) PIGS_CASE(16) PIGS_CASE(19) PIGS_CASE(32)
        default: return PIGS_ERR_UNSUPPORTED;
    }
#undef PIGS_CASE
    hipLaunchKernelGGL((plan_unpermute_kernel<C>), dim3((pv.N + 255) / 256), dim3(256), 0, stream, pv, gm, gc, gv);
    return launch_status();
}

static Resid<float> resid_of(const double* r, const void* target) {
    Resid<float> rz{};
    if (r) { rz.a0 = (float)r[0]; rz.a1[0] = (float)r[1]; rz.a1[1] = (float)r[2]; rz.aL = (float)r[3]; }
    rz.target = (const float*)target;
    return rz;
}

int plan_forward(void* ws, size_t ws_bytes, const void* sws, size_t sws_bytes, int64_t N, int64_t M, int c,
                 float q_max, int mask, void* const* out, hipStream_t stream, const double* resid, const void* target) {
    if (!plan_supported(N, M, c)) return PIGS_ERR_UNSUPPORTED;
    const PlanLayout p = make_plan_layout(N, M, c);
    const SamplesLayout s = make_samples_layout(M);
    if (!ws || ws_bytes < p.total_bytes || !sws || sws_bytes < s.total_bytes) return PIGS_ERR_WORKSPACE;
    const PlanView pv = make_view(p, ws, q_max);
    const SamplesView sv = make_samples_view(s, sws);
    float* o[4];
    for (int k = 0; k < 4; ++k) o[k] = mask_uses_slot(mask, k) ? (float*)out[k] : nullptr;
    const int cm = covering_mask_of(mask);
    if (c != 1 && c != 2) return PIGS_ERR_UNSUPPORTED;
    DeferredLists d{};
    ListArgs la{};
    const bool first = defer_take(ws, d);
    if (first) la = make_list_args(p, s, ws, sws, d.q_f, d.q_wide);
    const int rc = c == 1 ? plan_forward_c<1>(pv, sv, cm, o, stream, resid_of(resid, target), first ? &la : nullptr)
                          : plan_forward_c<2>(pv, sv, cm, o, stream, resid_of(resid, target), first ? &la : nullptr);
    if (first && rc == PIGS_OK) plan_note_points(p, ws, stream);
    return rc;
}

int plan_backward(void* ws, size_t ws_bytes, const void* sws, size_t sws_bytes, int64_t N, int64_t M, int c,
                  float q_max, int mask, const void* const* gout, void* g_means, void* g_conics, void* g_values,
                  hipStream_t stream, const double* resid) {
    if (!plan_supported(N, M, c)) return PIGS_ERR_UNSUPPORTED;
    const PlanLayout p = make_plan_layout(N, M, c);
    const SamplesLayout s = make_samples_layout(M);
    if (!ws || ws_bytes < p.total_bytes || !sws || sws_bytes < s.total_bytes) return PIGS_ERR_WORKSPACE;
    const PlanView pv = make_view(p, ws, q_max);
    const SamplesView sv = make_samples_view(s, sws);
    {   // a backward as the first sampling call on a plan with deferred lists: the list launch first
        DeferredLists d{};
        if (defer_take(ws, d)) {
            clear_hip_error();
            launch_lists(s.ntiles, make_list_args(p, s, ws, sws, d.q_f, d.q_wide), stream);
            const int rc = launch_status();
            if (rc != PIGS_OK) return rc;
            plan_note_points(p, ws, stream);
        }
    }
    const float* g[4];
    for (int k = 0; k < 4; ++k) g[k] = mask_uses_slot(mask, k) ? (const float*)gout[k] : nullptr;
    const int cm = covering_mask_of(mask);
    switch (c) {
        case 1: return plan_backward_c<1>(pv, sv, cm, g, (float*)g_means, (float*)g_conics, (float*)g_values, stream, resid_of(resid, nullptr));
        case 2: return plan_backward_c<2>(pv, sv, cm, g, (float*)g_means, (float*)g_conics, (float*)g_values, stream, resid_of(resid, nullptr));
    }
    return PIGS_ERR_UNSUPPORTED;
}

}  // namespace pigs

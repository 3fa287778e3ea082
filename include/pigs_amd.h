/*
 * pigs_amd -- C ABI of the MI355X-native differentiable Gaussian sampler.
 *
 * This is the drop-in boundary for the hot path of kr4b/pigs: the native half of
 * `diff_gaussian_sampling.GaussianSampler`.  The reference's own native extension is an
 * un-vendored submodule (/root/reference/.gitmodules:1-3), so its C++ interface cannot be
 * cited; each entry point below replaces the native work behind one Python-visible method of
 * that class, cited by its call sites:
 *
 *   pigs_sample_forward   GaussianSampler.sample_gaussians()                (model_pn.py:650,770; test_gaussian_sampling.py:57)
 *                         .sample_gaussians_derivative()                    (model_pn.py:651,771; test_derivatives.py:124)
 *                         .sample_gaussians_laplacian()  [full Hessian]     (model_pn.py:652,772; test_derivatives.py:220)
 *                         .sample_gaussians_third_derivative()              (model_pn.py:654,778; test_pde.py:53)
 *   pigs_sample_backward  autograd backward of those outputs wrt (means, values, conics)
 *                         (test_derivatives.py:123,214-215,349-352; main_pn.py:220; test_no_mlp.py:146)
 *   pigs_samples_build,
 *   pigs_plan_*           GaussianSampler.preprocess(means, values, covariances, conics, samples)
 *                         (model_pn.py:648,768,784; test_gaussian_sampling.py:56; test_1d.py:30)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP), row-major contiguous; nothing here takes or
 *     returns a torch type.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - all calls are asynchronous on `stream`; none allocates, frees or synchronises, so they
 *     can be captured into a hipGraph.
 *   - layouts: means[N][d], conics[N][d(d+1)/2] (upper triangle row-major: d=2 -> xx,xy,yy,
 *     gaussians.py:186-189), values[N][c], samples[M][d];
 *     out0[M][c], out1[M][d][c], out2[M][d][d][c], out3[M][d][d][d][c]  (model_pn.py:650-654).
 *   - orders_mask: bit k (k = 0..3) set = derivative order k is requested (outputs) / has an incoming
 *     gradient (backward).  Pointers of orders outside the mask may be NULL.
 *     Bit 4 (value 16) = the TRACE of the order-2 output, u_xx + u_yy, as [M][c] -- the Laplacian
 *     the PDE residuals consume (model_pn.py:614-617) -- written to / read from the out2 / gout2
 *     slot in place of the full Hessian; bits 2 and 4 exclude each other (PIGS_ERR_INVALID), the
 *     trace together with order 3 has no fused kernel (PIGS_ERR_UNSUPPORTED: two calls).
 *   - supported: d in {1,2}, c in {1..4}, dtype f32/f64 (binned plan: d=2, f32).
 *   - return value: PIGS_OK or an error code; pigs_status_string() names it.
 */
#ifndef PIGS_AMD_H
#define PIGS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIGS_ABI_VERSION 8

enum pigs_status {
    PIGS_OK = 0,
    PIGS_ERR_INVALID = 1,      /* bad argument (negative size, null required pointer, ...) */
    PIGS_ERR_UNSUPPORTED = 2,  /* d / c / dtype / mask combination not compiled            */
    PIGS_ERR_LAUNCH = 3,       /* HIP reported a launch error                              */
    PIGS_ERR_WORKSPACE = 4     /* workspace too small for the plan                         */
};

enum pigs_dtype { PIGS_F32 = 0, PIGS_F64 = 1 };

int pigs_abi_version(void);
const char* pigs_status_string(int status);
/* text of the HIP error behind the calling thread's last PIGS_ERR_LAUNCH */
const char* pigs_last_hip_error(void);

/* Dense forward: out_k = sum over ALL N Gaussians (exact reference semantics, no culling). */
int pigs_sample_forward(int dtype, int d, int c, int orders_mask, int64_t N, int64_t M,
                        const void* means, const void* conics, const void* values, const void* samples,
                        void* out0, void* out1, void* out2, void* out3, void* stream);

/* Dense backward: gradients of sum_k <gout_k, out_k> wrt means [N][d], flat conics
 * [N][d(d+1)/2] and values [N][c].  The three gradient buffers are overwritten. */
int pigs_sample_backward(int dtype, int d, int c, int orders_mask, int64_t N, int64_t M,
                         const void* means, const void* conics, const void* values, const void* samples,
                         const void* gout0, const void* gout1, const void* gout2, const void* gout3,
                         void* g_means, void* g_conics, void* g_values, void* stream);

/*
 * Fused covariance builder -- the caller-side step in front of preprocess():
 * gaussians.build_covariances(scaling, transform) (gaussians.py:163-193; model_pn.py:499-502,
 * 538-541, 607-610, 695-698; test_gaussian_sampling.py:36; test_derivatives.py:55-58).  d = 2.
 *   scaling [N][2] (variances, > 0), transform [N] (raw correlation, squashed by tanh)
 *   -> covariances [N][3] and conics [N][3], flat (xx, xy, yy); either output may be NULL.
 * The backward returns the gradients of <g_covariances, cov> + <g_conics, conic> wrt scaling
 * [N][2] and transform [N]; a NULL incoming gradient reads as zero.
 */
int pigs_build_covariances(int dtype, int64_t N, const void* scaling, const void* transform,
                           void* covariances, void* conics, void* stream);
int pigs_build_covariances_backward(int dtype, int64_t N, const void* scaling, const void* transform,
                                    const void* g_covariances, const void* g_conics,
                                    void* g_scaling, void* g_transform, void* stream);

/*
 * Binned ("plan") path -- float32, d = 2, c <= 2.  preprocess() builds, in caller-owned device
 * memory, two position independent workspaces:
 *
 *   SAMPLES workspace (pigs_samples_workspace_bytes(M) bytes, 256-byte aligned): the sample points
 *     sorted into ~16-point cells; tiles of 64 / groups of 16 consecutive sorted points are the
 *     units of work.  Built from `samples` alone and immutable afterwards, so it may be shared by
 *     any number of plans: the reference re-binds new Gaussians to an unchanged sample set on every
 *     step of a roll-out (main_pn.py:317-324), and so does any fixed collocation grid.
 *   PLAN workspace (pigs_plan_workspace_bytes(N, M, c) bytes): the Gaussians binned by centre into a
 *     multi-level cell grid (packed, sorted 32-byte records) and, for every tile, the list of
 *     Gaussians whose q <= q_max ellipse reaches the tile (with the 16-point groups each one
 *     reaches).  Forward, backward and every further sample_*() of the same preprocess() read the
 *     lists; none walks the grid again.
 *
 * The sampling entry points evaluate, for every point, only the Gaussians whose q <= q_max ellipse
 * reaches the bounding box of the point's group (dropped terms are below exp(-q_max/2) of a term's
 * scale; q_max = 36 -> 1.5e-8).  A plan holds TWO cut-offs (ABI 6): `q_max` for the forward and for
 * the backward of gradients that arrive at orders 0 and 1, and `q_max_backward` >= q_max for the
 * backward of gradients that arrive at order 2, order 3 or the trace (a value <= q_max, 0 included,
 * means one cut-off).  Why: the conic gradient of a second-derivative term carries a q^2 prefactor and
 * its sum over the points nearly cancels, so with thousands of points per Gaussian and one-signed
 * incoming gradients the tail beyond q = 36 was 2.5e-5 of the largest entry; at 40 (the Python host's
 * default) it is 3.5e-6, the dense kernel's own float32 error level (DESIGN.md "Cut-off").  The same (N, M, c) and the same samples workspace must be passed to every
 * call on a plan workspace; the plan remembers its cut-offs (the q_max argument of pigs_plan_forward /
 * pigs_plan_backward is kept for ABI shape and ignored since ABI 6).  pigs_plan_backward uses scratch
 * inside the plan workspace: calls sharing one must be stream ordered.
 */
size_t pigs_samples_workspace_bytes(int64_t M);                  /* 0 = unsupported size */
size_t pigs_plan_workspace_bytes(int64_t N, int64_t M, int c);   /* 0 = unsupported sizes */

/* samples workspace alone (4 launches; 5 on the coarse-bin path).
 * Two ways to sort the points, same result (the order inside a 16-point cell aside): ONE PASS -- one returning
 * atomic per run of consecutive points that share a cell, right for lattices in row order -- and COARSE BINS --
 * per-workgroup LDS ranking inside 256 coarse bins, a scan of the (bin, workgroup) counts, a scatter into bin
 * segments and a per-bin LDS sort; right for points in no order (torch.rand collocation points, main_pn.py:103),
 * which otherwise pay one global atomic and one 12-byte scattered write each.  The host cannot see which it
 * has without a synchronisation, so the library remembers, per device and M: every build of M >= 131 072
 * points leaves {runs, points} of a sample of its waves in the workspace, copied to pinned memory on `stream`
 * behind the build (nobody waits); the next build of the same M takes the path the last completed copy
 * recommends (more than 0.55 runs per point: coarse bins).  Captured builds neither ask nor copy.
 * PIGS_SAMPLES_ORDER=ordered|unordered in the environment overrules the memory (tests), as do the
 * PIGS_BUILD_POINTS_* flags of pigs_plan_build.
 * The same memory decides how pigs_plan_forward / pigs_plan_backward move the outputs / incoming gradients of
 * c = 1, orders (0, 1, 2) or (0, 1, trace) launches: directly through the points' original indices (lattices:
 * runs of consecutive indices), or -- points in no order -- through one 32-byte record per point in the plan
 * workspace and a streaming launch that deals the records out / gathers them (from 524 288 points, where it
 * starts to pay; PIGS_STAGE=0|1 overrules). */
int pigs_samples_build(void* samples_ws, size_t samples_ws_bytes, int64_t M, const void* samples, void* stream);
/* what the library currently remembers for builds of M points on the current device: 1 = coarse bins,
 * 0 = one pass, -1 = nothing yet (introspection for tools and tests) */
int pigs_samples_order_hint(int64_t M);

/* plan workspace.  `flags`:
 *   PIGS_BUILD_SAMPLES       also (re)builds the samples workspace from `samples` in the same launches
 *                            (5 in all); without it the samples workspace must be built already, or be
 *                            being built earlier on the same stream (`samples` is not read then).
 *   PIGS_BUILD_PLAN_WS_CLEAN the plan workspace's counters are known to be zero: it was the target of
 *                            an earlier pigs_plan_build with the same (N, M, c) that has completed or
 *                            precedes this call on the same stream (every build leaves them zeroed),
 *                            or the caller zero-filled it.  Saves the zeroing launch of a build on an
 *                            existing samples workspace (4 launches instead of 5); ignored together
 *                            with PIGS_BUILD_SAMPLES, whose first launch zeroes anyway.
 * (ABI 4 called this parameter build_samples: 0 / 1 keep their meaning.)
 *   PIGS_BUILD_DEBUG_NO_LOOKBACK  test hook: the in-kernel scans never use their workgroup-to-workgroup
 *                            hand-over and take the recompute path everywhere (see pigs_*_error_offset);
 *                            results are the same, the build is slower.
 *   PIGS_BUILD_POINTS_ORDERED / PIGS_BUILD_POINTS_UNORDERED  (with PIGS_BUILD_SAMPLES) the caller knows how its
 *                            points arrive: take the one-pass / the coarse-bin samples build whatever the
 *                            library remembers (see pigs_samples_build); neither flag: the library decides. */
#define PIGS_BUILD_SAMPLES 1
#define PIGS_BUILD_PLAN_WS_CLEAN 2
#define PIGS_BUILD_DEBUG_NO_LOOKBACK 4
#define PIGS_BUILD_POINTS_ORDERED 8
#define PIGS_BUILD_POINTS_UNORDERED 16
/* ABI 7.  PIGS_BUILD_DEFER_LISTS: the build stops in front of its last launch, the tile lists; the FIRST
 * pigs_plan_forward / pigs_plan_backward / pigs_residual_* call on this plan workspace builds them -- a forward
 * (orders 0..2, orders 0, 1 + trace, order 0, the residual; c = 1, and orders 0..2 for c = 2) in the SAME launch
 * as its own evaluation: a wave builds the lists of its four tiles and samples them at once, so the latency-bound
 * list build hides behind the arithmetic of the other waves and one kernel boundary goes away (the reference's
 * pattern: preprocess, then sample_*(), model_pn.py:768-772).  The lists are written out as ever; every further
 * call reads them.  The library remembers which workspaces are waiting by their address (every build into a
 * workspace sets or clears the mark, the first sampling call clears it); the first sampling call must be stream
 * ordered behind the build, like any use of the plan.  PIGS_NO_FUSED_FIRST in the environment keeps the list
 * build in a launch of its own (A/B runs). */
#define PIGS_BUILD_DEFER_LISTS 32
int pigs_plan_build(void* workspace, size_t workspace_bytes, void* samples_ws, size_t samples_ws_bytes,
                    int flags, int64_t N, int64_t M, int c, float q_max, float q_max_backward,
                    const void* means, const void* conics, const void* values, const void* samples, void* stream);

int pigs_plan_forward(void* workspace, size_t workspace_bytes, const void* samples_ws, size_t samples_ws_bytes,
                      int64_t N, int64_t M, int c, float q_max,
                      int orders_mask, void* out0, void* out1, void* out2, void* out3, void* stream);

int pigs_plan_backward(void* workspace, size_t workspace_bytes, const void* samples_ws, size_t samples_ws_bytes,
                       int64_t N, int64_t M, int c, float q_max, int orders_mask,
                       const void* gout0, const void* gout1, const void* gout2, const void* gout3,
                       void* g_means, void* g_conics, void* g_values, void* stream);

/*
 * Linear residual of the sampled field in ONE launch (extension): the diffusion / wave
 * residuals of the reference's losses (model_pn.py:612-617, 834-849; test_no_mlp.py:127-144) are
 *     r[m][c] = a0 u + a1x du/dx + a1y du/dy + aL (u_xx + u_yy) - target[m][c]
 * with constant coefficients `coeffs` = {a0, a1x, a1y, aL} (HOST doubles) and an optional `target`
 * [M][c] (device; e.g. u_prev / dt): 4 bytes per point and channel leave the kernel instead of the 28 of
 * u, grad u and the Hessian, and the loss is one elementwise + reduction on r.  The backward takes the
 * gradient gout [M][c] that arrives at r and returns the gradients wrt means, conics, values (the three
 * buffers are overwritten; d r / d target = -1 is the caller's).  plan_ws == NULL: dense (d in {1,2},
 * f32 / f64); else through a built plan (d = 2, f32; backward with the plan's wide cut-off).
 */
int pigs_residual_forward(int dtype, int d, int c, int64_t N, int64_t M,
                          const void* means, const void* conics, const void* values, const void* samples,
                          const double coeffs[4], const void* target, void* out,
                          void* plan_ws, size_t plan_ws_bytes, const void* samples_ws, size_t samples_ws_bytes, void* stream);
int pigs_residual_backward(int dtype, int d, int c, int64_t N, int64_t M,
                           const void* means, const void* conics, const void* values, const void* samples,
                           const double coeffs[4], const void* gout, void* g_means, void* g_conics, void* g_values,
                           void* plan_ws, size_t plan_ws_bytes, const void* samples_ws, size_t samples_ws_bytes, void* stream);

/* Byte offset, inside a samples / plan workspace, of a uint32 DIAGNOSTIC that a build leaves at 0 and
 * sets to non-zero when a workgroup of its in-kernel scan did not receive a predecessor's total within
 * the bounded wait and summed that predecessor's counters itself.  The result is valid either way
 * (the counters are final before the scan starts); the flag only says that the slow path ran -- never
 * observed with the hardware's in-order dispatch, always with PIGS_BUILD_DEBUG_NO_LOOKBACK.
 * (ABI <= 5: the scan gave up instead and this word meant "workspace invalid".) */
size_t pigs_samples_error_offset(void);
size_t pigs_plan_error_offset(void);

/* ABI 7.  Byte offset, inside a samples workspace, of two uint32 {rf, rs} that a samples build leaves behind:
 * non-zero when the points were taken in INDEX-TILED order -- they arrived as an rf x rs lattice in row order
 * (rf points along the fastest axis; both multiples of 8: meshgrid(indexing="xy").reshape(-1, 2),
 * test_gaussian_sampling.py:43-46, main_pn.py:317-324), so a point's tile (an 8 x 8 index patch) and group (4 x 4)
 * are index arithmetic: the build neither keys, counts, scans, scatters NOR COPIES the points -- and {0, 0} when
 * they were sorted into cells.  The decision is the build's own, on the device (the first backward step of the
 * fastest coordinate gives rf; the largest steps between index neighbours along and across rows bound every index
 * tile, which must stay within twice its share of the bounding box) for point sets of 4 096 points and more;
 * results never depend on it; PIGS_LATTICE=0 / 1 in the environment: never / at every size.
 * When the library expects a lattice (the last completed build of this size was one) and that build and the one before it
 * had the same bounding box, pigs_plan_build with PIGS_BUILD_SAMPLES | PIGS_BUILD_PLAN_WS_CLEAN bins the Gaussians on
 * the REMEMBERED box beside its first look at the points (one launch fewer, two with Gaussians in strips, below; a grid's
 * domain steers the quality of the binning, never a result; PIGS_NO_AHEAD in the environment switches it off).
 * CONTRACT that comes with it: a samples workspace in index-tiled order holds the ADDRESS of `samples`, not the
 * points; pigs_plan_build / pigs_plan_forward / pigs_plan_backward / pigs_residual_* read the caller's array
 * through it.  `samples` must therefore stay allocated and unmodified for as long as the samples workspace is
 * used (the sorted order has no such requirement; a caller that cannot promise it sets PIGS_LATTICE=0). */
size_t pigs_samples_lattice_offset(void);

/* Byte offset, inside a PLAN workspace, of one uint32: non-zero when the build kept the Gaussians in the CALLER's
 * order (ABI 8).  Gaussians whose order in the arrays is already spatial -- the reference lays them out on a meshgrid
 * (model_pn.py:338-342) and training moves them by fractions of a spacing -- are not binned into grid cells: every
 * 16 consecutive ones are a strip with a bounding box (16 strips a super-strip), and the tile lists are built from
 * those boxes; no count, scan or scatter launch.  The library measures in every build how many times over the strips cover
 * the samples' domain and decides the next build of the same sizes from the last completed measurement (at most 64 times: strips);
 * always correct whatever the order, results never depend on it beyond the order of a list's entries;
 * PIGS_GAUSS_STRIPS=0 / 1 in the environment: never / always. */
size_t pigs_plan_strips_offset(void);

/* Introspection for tools and tests (never needed to use a plan): where the tile lists sit inside a
 * plan workspace.  info[0] = tiles, info[1] = entries per list slab, info[2] = byte offset of the
 * tile headers (8 uint32 each: [0] = count | mode << 30; mode 0 = list of `count` entries `sorted
 * Gaussian index | wide group mask << 24 | narrow group mask << 28`, mode 1 = `count` record ranges
 * {first, length}, mode 2 = group lists only; [1..4] = the lengths of the four group lists), info[3] = byte offset of the tile-list slabs (uint32[tiles][slab]),
 * info[4] = byte offset of the sorted -> caller Gaussian index table (uint32[N]), info[5] = byte
 * offset of the group-list slabs (uint32[tiles][4][slab], sorted Gaussian indices).
 * Returns PIGS_ERR_UNSUPPORTED for sizes the binned path does not take (N >= 2^24 among them). */
int pigs_plan_layout_info(int64_t N, int64_t M, int c, int64_t info[6]);

/*
 * preprocess_aggregate() / aggregate_neighbors() -- GaussianSampler methods of the reference
 * (model_pn.py:257-264; test_neighbor_aggregation.py:75-98).  PARITY UNPINNED: their arithmetic exists
 * only in the reference's absent CUDA source; these entry points implement this repository's own
 * definition (DESIGN.md "aggregate_neighbors"; pigs_amd/csrc/aggregate.hip), d = 2, float32 / float64:
 *   neighbours of i = { j : (mu_i - mu_j)^T C_j (mu_i - mu_j) <= q_max };  a_ij = softmax_j <queries_i, keys_j> / sqrt(K);
 *   out_i = sum_j a_ij (transform features_j + distance_transform [e_ij ; g_ij e_ij]),  e_ij = Fourier embedding
 *   of mu_j - mu_i with `frequencies` (E = 4F + 1 entries), g_ij = exp(-q_ij / 2).
 * The neighbour relation is kept as index lists -- `cap` int32 slots per Gaussian, by rows (the j of an
 * i) and by columns (the i that hold a j).
 *
 * pigs_aggregate_lists: counts [N] and lists [N][cap] by rows and by columns.  Up to N = 2048 every pair
 *   is tested (two launches, lists ascending, `workspace` and `flags` unused: cap = N can never overflow and
 *   needs no counting pass); beyond, through the sampler's multi-level Gaussian grid in `workspace`
 *   (pigs_aggregate_workspace_bytes(dtype, N) bytes, 256-byte aligned; 0 = unsupported N): `flags` & PIGS_AGGREGATE_BUILD_GRID (re)builds the grid from
 *   `means` / `conics` first (4 launches; float64 inputs are binned through float32 copies with a
 *   widened cut-off -- the grid only nominates candidates, every pair is tested in the caller's dtype;
 *   this presumes ellipses far larger than the float32 spacing of the coordinates), without it the
 *   workspace must hold the grid of the same Gaussians.  With row_lists == col_lists == NULL only the
 *   counts are written (the FULL list lengths: the caller sizes `cap` from their maximum, then calls
 *   again with the lists).  *overflow (int32, zeroed by the caller) is set when a list did not fit
 *   `cap` (it is then truncated).  List order is the grid's (not ascending; may differ between builds).
 * pigs_aggregate_forward: out [N][L], and for the backward lse [N] (log-sum-exp of the scaled scores)
 *   and acc [N][L + 2E] = (sum_j a_ij features_j ; sum_j a_ij [e_ij ; g_ij e_ij]).
 * pigs_aggregate_backward: the whole backward from gout [N][L] (4 launches; 5 for N > 2048): all six
 *   gradients -- g_features [N][L], g_transform [L][L], g_queries [N][K], g_keys [N][K], g_frequencies [F],
 *   g_distance_transform [L][2E].  `scratch` (pigs_aggregate_backward_scratch_bytes(dtype, N, L, F) bytes)
 *   holds dacc = gout [transform | distance_transform], D_i = <dacc_i, acc_i> and the per-row shares of the
 *   frequency gradient between the launches.  The per-Gaussian gradients are gathers (no atomics); the
 *   three sums over the Gaussians are plain sums up to N = 2048 and atomic sums of 2048-Gaussian
 *   partials beyond.
 */
#define PIGS_AGGREGATE_BUILD_GRID 1
size_t pigs_aggregate_workspace_bytes(int dtype, int64_t N);
int pigs_aggregate_lists(int dtype, int64_t N, int64_t cap, const void* means, const void* conics, double q_max,
                         void* workspace, size_t workspace_bytes, int flags,
                         int32_t* row_counts, int32_t* row_lists, int32_t* col_counts, int32_t* col_lists,
                         int32_t* overflow, void* stream);

int pigs_aggregate_forward(int dtype, int64_t N, int64_t cap, int L, int K, int F,
                           const void* means, const void* conics, const int32_t* row_counts, const int32_t* row_lists,
                           const void* features, const void* transform, const void* queries, const void* keys,
                           const void* frequencies, const void* distance_transform,
                           void* out, void* lse, void* acc, void* stream);

size_t pigs_aggregate_backward_scratch_bytes(int dtype, int64_t N, int L, int F);
int pigs_aggregate_backward(int dtype, int64_t N, int64_t cap, int L, int K, int F,
                            const void* means, const void* conics, const int32_t* row_counts, const int32_t* row_lists,
                            const int32_t* col_counts, const int32_t* col_lists,
                            const void* features, const void* transform, const void* queries, const void* keys,
                            const void* frequencies, const void* distance_transform,
                            const void* lse, const void* acc, const void* gout, void* scratch, size_t scratch_bytes,
                            void* g_features, void* g_transform, void* g_queries, void* g_keys, void* g_frequencies,
                            void* g_distance_transform, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIGS_AMD_H */

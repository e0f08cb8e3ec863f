/*
 * Internal measurement / tuning switches of libsycl_points_amd.so — NOT part of the C ABI (include/sycl_points_amd.h).
 *
 * They exist for tests/ (bit-identity of every shortcut against the path without it), bench.py (timing one launch of a
 * pair) and scratch/. Every switch lives in the handle it acts on: nothing here is process-global, so a caller that never
 * includes this header can never be affected by one that does. Results are identical under every setting except
 * SP_INTERNAL_FUSED_STAGE_MASK, which drops launches.
 */
#ifndef SP_INTERNAL_H
#define SP_INTERNAL_H

#include "../../include/sycl_points_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    /* sp_gicp_source: launches issued by sp_gicp_iteration_fused / sp_gicp_align_* for this source
     * (bit 0 = per-iteration kernels, bit 1 = final reduce + solve / finish kernel). Default 3. */
    SP_INTERNAL_FUSED_STAGE_MASK = 0,
    /* sp_gicp_source: 2 (default) carry a correspondence to the next iteration when either certificate proves it unchanged,
     * 1 first certificate only, 0 always search. Takes effect at once (the cache is dropped). */
    SP_INTERNAL_FUSED_REUSE = 1,
    /* sp_gicp_source: NN walk inside the fused kernel: -1 (default) 2x2x2 fast path iff the source is cell-sorted,
     * 0 ring walk, 1 fast path. */
    SP_INTERNAL_FUSED_FAST_NN = 2,
    /* sp_grid: self-kNN kernel: 0 (default) chosen by k, 1 LDS-tile kernel (k <= 10), 2 wave-cooperative kernel. */
    SP_INTERNAL_SELF_KNN_MODE = 3,
    /* sp_gicp_source: sp_gicp_align_fused, when the convergence criteria can be met, runs the TAIL of the alignment as one
     * launch that loops on the device (1, default; taken when every workgroup of the grid is resident) or every iteration
     * as a launch of its own (0). Same bits either way. */
    SP_INTERNAL_FUSED_PERSISTENT = 4,
    /* sp_gicp_source: first iteration of that tail (default 4; 0: the whole alignment as one launch). */
    SP_INTERNAL_FUSED_PERSISTENT_FROM = 5,
    /* sp_bvh: searches for 2 <= k <= 21 with the lane's k best in a heap (1, default) or by the sorted-insertion kernel
     * that serves every other search (0). Same lists either way. */
    SP_INTERNAL_BVH_SELF_HEAP = 6,
    /* sp_bvh: external queries (400 k or more) are searched in the order of the tree's Morton curve (1, default) or as given (0).
     * Same lists either way. */
    SP_INTERNAL_BVH_SORT_QUERIES = 7,
    /* sp_grid: the same for sp_grid_search / sp_grid_radius_search: in cell order (1, default) or as given (0). */
    SP_INTERNAL_GRID_SORT_QUERIES = 8,
    /* sp_gicp_source: sp_gicp_align_optimize on a source of up to 2048 points: one wave per point in the linearisation steps
     * (1, default) or one lane per point (0). Same correspondences either way; the sums are grouped differently. */
    SP_INTERNAL_OPT_WAVE_QUERY = 9,
    /* sp_gicp_source: sp_gicp_align_optimize, wave-per-point launches of up to 2048 points: an LM / dog-leg trial step also
     * linearises at the trial pose (1, default: an accepted trial's next linearisation is then already there) or not (0). The
     * same sequence of optimiser decisions either way. */
    SP_INTERNAL_OPT_FUSE_TRIALS = 10
};

int sp_internal_source_option(sp_gicp_source* source, int option, int value);
int sp_internal_grid_option(sp_grid* grid, int option, int value);
int sp_internal_bvh_option(sp_bvh* bvh, int option, int value);
/* Process-wide: grids of up to 8192 points and fewer than 32768 cells are built by one launch of one workgroup (1, default) or by the
 * general chain of launches (0); < 0 only asks. Returns the previous setting. Same structure bit for bit. */
int sp_internal_grid_small_build(int enable);
/* Device pointer to the per-launch log of an sp_gicp_align_* workspace: entry k = number of source points launch k had to
 * search for (the others reused their previous correspondence by certificate). *n_entries_out = entries kept (64). */
const uint32_t* sp_internal_align_searched_log(void* workspace, size_t* n_entries_out);
/* The library's own stable radix sort of (u32 key, u32 value) pairs on the low `bits` key bits (csrc/radix_sort.hip), for its
 * test: all pointers are device pointers, the four arrays are overwritten, *result_in_b_out (host) says which pair holds
 * the sorted data. workspace: sp_internal_radix_sort_workspace_bytes(n). */
size_t sp_internal_radix_sort_workspace_bytes(size_t n);
int sp_internal_radix_sort_u32(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, size_t n,
                               unsigned bits, void* workspace, size_t workspace_bytes, int* result_in_b_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SP_INTERNAL_H */

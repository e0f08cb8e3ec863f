/*
 * sycl_points_amd — C ABI of the MI355X (gfx950) hot path.
 *
 * The reference (fateshelled/sycl_points) is a header-only C++/SYCL library with no C ABI: its device code is
 * reached through C++ templates. This header is the boundary a maintainer would bind instead of
 * those SYCL kernels; every entry point cites the reference interface it replaces
 * (paths relative to /root/reference/cpp/include/sycl_points/).
 *
 * Conventions
 *  - all array pointers are DEVICE pointers (HBM) unless the parameter name ends in `_host`;
 *  - points / normals are float[4] (x,y,z,w), covariances are float[16] column-major 4x4 with the 3x3 block used
 *    (Eigen::Vector4f / Eigen::Matrix4f storage, points/types.hpp:11-18); entries outside the 3x3 block are
 *    written as 0 and ignored on input; transforms are float[16] column-major (Eigen::Matrix4f::data());
 *  - every call only ENQUEUES work on `stream` (a hipStream_t, may be NULL for the default stream) and returns
 *    immediately; no call allocates or synchronises unless its comment says so, so a sequence of calls can be
 *    captured into a hipGraph;
 *  - return value: SP_OK, or an error code whose meaning mirrors the C++ exception the reference would throw;
 *    sp_last_error() returns the message (thread local).
 */
#ifndef SYCL_POINTS_AMD_H
#define SYCL_POINTS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_ERR_INVALID_ARGUMENT 1 /* reference: std::invalid_argument */
#define SP_ERR_RUNTIME 2          /* reference: std::runtime_error    */
#define SP_ERR_HIP 3              /* reference: sycl::exception from wait_and_throw */

#define SP_ABI_VERSION 6
int sp_abi_version(void);
/* The library keeps device buffers it no longer needs (temporaries of a build, the arrays of a destroyed grid / tree / target) in
 * a pool, tagged with the stream whose work may still use them: a later call on the SAME stream takes them without waiting (stream
 * order), nobody else does. A caller that is about to destroy a stream it has passed to the library calls this first (the
 * facade's DeviceQueue does): it waits for the stream and clears its tags. Without it such buffers stay out of use until the
 * pool's sweep (a device-wide wait once 32 of them have piled up). */
void sp_stream_retired(void* stream);
const char* sp_last_error(void);

/* Number of devices visible / select the device for the calling thread (utils/sycl_utils.hpp:398-465). */
int sp_device_count(void);
int sp_set_device(int device);

/* ------------------------------------------------------------------------------------------------ KNN */

/* Brute-force kNN (algorithms/knn/bruteforce.hpp:24-96, kernel K1).
 * Exact; k <= 20 (SP_ERR_INVALID_ARGUMENT otherwise — the reference has no check and overruns its arrays);
 * strict '<' so the lowest target index wins ties; rows ascending by squared distance, padded with -1 / FLT_MAX.
 * workspace: sp_knn_bruteforce_workspace_bytes(nq, nt, k) bytes. */
size_t sp_knn_bruteforce_workspace_bytes(size_t nq, size_t nt, size_t k);
int sp_knn_bruteforce(const float* queries, size_t nq, const float* targets, size_t nt, size_t k, int32_t* idx_out,
                      float* d2_out, void* workspace, size_t workspace_bytes, void* stream);
/* From 16 K targets on, the search bounds every query's k-th distance first (chunk minima of an approximate distance with a
 * proven error term) and evaluates the reference's expression only where a neighbour can be. That first pass runs on the
 * matrix cores (bf16-split operands, v_mfma_f32_32x32x16_bf16); valu != 0 selects its packed-fp32 VALU form instead, for
 * measurement (process-wide; the results do not depend on it).
 * Target clouds of 256 .. 12032 points (a voxel-downsampled scan) are answered by ONE launch that keeps the whole cloud in LDS
 * (a wave per four queries: lane minima -> bound of the k-th distance -> the targets within it -> 64-lane sort); valu == 2
 * switches that path off (the general ones then serve every size), valu == 3 on again (default). Same lists. */
int sp_knn_bruteforce_set_pass_a(int valu);

/* KD-tree (algorithms/knn/kdtree.hpp:142-766).
 * sp_kdtree_create: KDTree::build (kdtree.hpp:165-178, 292-413) — host median-split build from HOST points
 *   (same split rule, same std::nth_element, so the same topology as the reference), then upload. Allocates and
 *   synchronises. leaf_threshold default in the reference is 16.
 * sp_kdtree_search: KDTree::knn_search_async (kdtree.hpp:203-224, 424-562, kernel K2): neighbours of transT*q,
 *   traversal order, tie rule (first visited wins) and the 16-entry far stack of the reference;
 *   k <= 100 else SP_ERR_RUNTIME ("`k` is too large", kdtree.hpp:221-223).
 * sp_kdtree_radius_search: KDTree::radius_search_async (kdtree.hpp:251-280, 574-719, kernel K3).
 * sp_kdtree_remove_by_flags: KDTree::remove_nodes_by_flags (kdtree.hpp:282-284, 721-765, kernel K4);
 *   flags 1 = keep, 0 = remove; new_indices[p] must be >= 0 for kept points (filter_by_flags.hpp:97-99). */
typedef struct sp_kdtree sp_kdtree;
int sp_kdtree_create(const float* points_host, size_t n, size_t leaf_threshold, void* stream, sp_kdtree** out);
void sp_kdtree_destroy(sp_kdtree* tree);
size_t sp_kdtree_size(const sp_kdtree* tree); /* number of points */
int sp_kdtree_search(const sp_kdtree* tree, const float* queries, size_t nq, size_t k, const float* transT,
                     int transT_on_device, int32_t* idx_out, float* d2_out, void* stream);
int sp_kdtree_radius_search(const sp_kdtree* tree, const float* queries, size_t nq, size_t max_k, float radius,
                            const float* transT, int transT_on_device, int32_t* idx_out, float* d2_out, void* stream);
int sp_kdtree_remove_by_flags(sp_kdtree* tree, const uint8_t* flags, const int32_t* new_indices, size_t n_flags,
                              void* stream);

/* GridKNN — an MI355X-native KNNBase implementation (no counterpart file in the reference; it plugs into the
 * KNNBase::knn_search_async seam, algorithms/knn/knn.hpp:14-61, exactly as KDTree / Octree do).
 * Exact kNN on a uniform cell grid built ON THE DEVICE from device points (bounding box, cell ids, radix sort,
 * cell_start table): a handful of independent loads per query instead of the KD-tree's chain of dependent node loads.
 * Results are bit-identical to sp_knn_bruteforce (same distance arithmetic, ties to the lowest index), k <= 20.
 * sp_grid_create allocates and synchronises. cell_size <= 0: chosen so that a cell holds `points_per_cell` points on
 * average (<= 0: 2). */
typedef struct sp_grid sp_grid;
int sp_grid_create(const float* points, size_t n, float cell_size, float points_per_cell, void* stream, sp_grid** out);
/* The same with the cell size steered by what the build measures instead of by the bounding-box volume alone: while the OCCUPIED
 * cells hold well more points than those of a uniform cloud at `points_per_cell` would (a cloud of surfaces — a voxel-downsampled
 * LiDAR scan fills 2 % of its box), the cell shrinks by the dimension the measurements imply and the grid is built again (at
 * most three more sorts; none for a uniform cloud; the table stays below 32 M cells). What Registration::align's in-loop search
 * uses when it stands in for the caller's KDTree (knn/kdtree.hpp:463-553 is density-agnostic; a fixed-volume grid is not). */
int sp_grid_create_adaptive(const float* points, size_t n, float points_per_cell, void* stream, sp_grid** out);
/* sp_grid_create for a caller that already knows a box holding every finite point (min x, y, z, max x, y, z): the bounding-box
 * kernel, its read-back and the wait for it are skipped — the build then never waits for its stream. What voxel downsampling
 * leaves is such a cloud: its voxels' key box is known on the host (the facade's VoxelGrid hands it on with the cloud). The box
 * is vouched for: a finite point outside the grid it implies raises the library's device error word (reported by a later call)
 * and that grid's searches are not exact. */
int sp_grid_create_bounded(const float* points, size_t n, const float* bounds_min_max6, float cell_size, float points_per_cell,
                           void* stream, sp_grid** out);
void sp_grid_destroy(sp_grid* grid);
size_t sp_grid_size(const sp_grid* grid);
float sp_grid_cell_size(const sp_grid* grid);
/* Points in the fullest cell (measured on the first call: one small kernel and a blocking read-back; cached). The grid's searches assume near-uniform density (a query scans its 27
 * cells): a value far above the points-per-cell target says that a hierarchy (sp_bvh_*) serves this cloud better. */
uint32_t sp_grid_max_cell_points(const sp_grid* grid);
/* idx_out[i] = original index of the i-th point in the grid's cell order (z-major, then y, then x; stable inside a cell).
 * A cloud stored in this order (it is also the order VoxelGrid::downsampling produces, voxel_downsampling.hpp:146-288:
 * sorted by voxel key) keeps neighbouring lanes on neighbouring cells of ANY grid it is later searched against, so
 * sp_gicp_source_prepare can skip its per-alignment sort (SP_SOURCE_PRESORTED). */
int sp_grid_order(const sp_grid* grid, uint32_t* idx_out, void* stream);
/* KNNBase::knn_search_async on the grid (k <= 20). For k > 10 the call takes nq + 1 words of scratch from the library's
 * buffer pool for its duration (the list of queries a first, 27-cell pass could not prove): inside a stream capture that
 * works once the pool holds such a buffer, i.e. after one eager call of the same size. Queries in the cell order of any grid
 * (sp_grid_order) are served three times faster than in random order (their candidates share cache lines); 400 k queries or
 * more are therefore sorted by cell first (a key per query and the library's radix sort, scratch from the pool as above; rows
 * still by query number). */
int sp_grid_search(const sp_grid* grid, const float* queries, size_t nq, size_t k, const float* transT,
                   int transT_on_device, int32_t* idx_out, float* d2_out, void* stream);
/* The grid's counterpart of KDTree::radius_search_async (knn/kdtree.hpp:251-280, 574-719): the max_k nearest target
 * points within `radius` of transT*q, ascending, padded with -1 / FLT_MAX; max_k <= 20 (SP_ERR_RUNTIME otherwise).
 * Bit-identical to sp_kdtree_radius_search on tie-free data. */
int sp_grid_radius_search(const sp_grid* grid, const float* queries, size_t nq, size_t max_k, float radius,
                          const float* transT, int transT_on_device, int32_t* idx_out, float* d2_out, void* stream);
/* The grid's counterpart of KDTree::remove_nodes_by_flags (knn/kdtree.hpp:282-284, 721-765) with the same arguments as
 * sp_kdtree_remove_by_flags: flags 1 = keep, 0 = remove; a kept point p is relabelled new_indices[p]; points whose
 * index is >= n_flags are left alone. The cell order survives removal, so the kept points are compacted without a
 * sort and the cell table is rebased (about 0.1 ms per 1M points). Allocates and synchronises; sp_grid_size shrinks.
 * Objects that borrow the grid (sp_gicp_target) must be re-created afterwards. */
int sp_grid_remove_by_flags(sp_grid* grid, const uint8_t* flags, const int32_t* new_indices, size_t n_flags,
                            void* stream);

/* Self-kNN of the cloud a grid was built on, with the covariance / normal estimation optionally fused in
 * (covariance::estimate_async(knn, points, k), feature/covariance.hpp:305-311, and estimate_normals_async(knn, ...),
 * :451-459, when the KNNBase is a GridKNN over the same cloud). Any of idx_out+d2_out / covs_out / normals_out may be
 * NULL; rows are written at the points' ORIGINAL indices. Neighbour lists are bit-identical to sp_knn_bruteforce of the
 * cloud against itself, covariances bit-identical to sp_cov_estimate on those lists. k <= 20.
 * workspace: sp_grid_self_workspace_bytes(grid). */
size_t sp_grid_self_workspace_bytes(const sp_grid* grid);
int sp_grid_self_knn(const sp_grid* grid, size_t k, int32_t* idx_out, float* d2_out, float* covs_out, float* normals_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The same for the queries at grid positions [pos_first, pos_first + pos_count) only (position = index in the grid's cell
 * order, sp_grid_order); the other rows of the outputs are left untouched. This is how the pre-loop (k = 20 neighbours +
 * covariances of the target) is sharded by query over the ranks of a multi-GPU run: every rank holds the
 * whole grid, searches 1/N of the positions, and the covariance rows are exchanged with one all-gather:
 *   sp_grid_gather_rows   rows of this rank's positions, caller's order -> position order (a contiguous chunk to send)
 *   sp_allgather          (multi-GPU section)
 *   sp_grid_scatter_rows  position order -> caller's order, for all positions
 * row_bytes must be a multiple of 16 (a covariance row is 64, a point / normal 16). */
int sp_grid_self_knn_range(const sp_grid* grid, size_t k, size_t pos_first, size_t pos_count, int32_t* idx_out,
                           float* d2_out, float* covs_out, float* normals_out, void* workspace, size_t workspace_bytes,
                           void* stream);
int sp_grid_gather_rows(const sp_grid* grid, const void* rows, size_t row_bytes, size_t pos_first, size_t pos_count,
                        void* out_by_position, void* stream);
int sp_grid_scatter_rows(const sp_grid* grid, const void* in_by_position, size_t row_bytes, size_t pos_first,
                         size_t pos_count, void* rows, void* stream);

/* ------------------------------------------------------------------------------ covariance / normals */

/* covariance::estimate_async (algorithms/feature/covariance.hpp:16-47, 260-311, kernel K5). */
int sp_cov_estimate(const float* points, size_t n, const int32_t* knn_idx, size_t k, float* covs_out, void* stream);
/* covariance::estimate_robust_async (feature/covariance.hpp:97-250, 323-381): M-estimated neighbourhood covariance —
 * weighted estimate, Mahalanobis distances to it, their median (x mad_scale, floored at min_robust_scale) as the scale of
 * the IRLS weights, robust_max_iterations times. k <= 64 ("neighbor K is too large. MAX_K is 64" -> SP_ERR_RUNTIME);
 * robust_type SP_LOSS_NONE is sp_cov_estimate. */
int sp_cov_estimate_robust(const float* points, size_t n, const int32_t* knn_idx, size_t k, int robust_type,
                           float mad_scale, float min_robust_scale, size_t robust_max_iterations, float* covs_out,
                           void* stream);
/* covariance::kernel::normalize_covariance (feature/covariance.hpp:76-95) over a covariance array (in place allowed). */
int sp_cov_normalize(const float* covs, size_t n, float* covs_out, void* stream);
/* covariance::estimate_normals_async (covariance.hpp:49-65, 417-459, kernel K6). */
int sp_normals_from_knn(const float* points, size_t n, const int32_t* knn_idx, size_t k, float* normals_out,
                        void* stream);
/* covariance::extract_normals_async (covariance.hpp:465-503, kernel K7). */
int sp_normals_from_cov(const float* points, const float* covs, size_t n, float* normals_out, void* stream);
/* covariance::kernel::update_covariance_plane applied to a whole array (covariance.hpp:67-74); in place allowed. */
int sp_cov_update_plane(const float* covs, size_t n, float* covs_out, void* stream);

/* ------------------------------------------------------------------- device-built tree (any density profile) */

/* KDTree::build + knn_search_async (algorithms/knn/kdtree.hpp:292-413, 424-562) for callers that rebuild the structure every
 * frame (pipeline/submapping.hpp:197, pipeline/pointcloud_processing.hpp:64) on clouds whose density varies by orders of
 * magnitude (raw LiDAR scans): a bounding-volume hierarchy over the Morton-sorted points, leaves of 16 points, built ENTIRELY on
 * the device (bounding box, 30-bit Morton keys, radix sort, boxes level by level: a fraction of a millisecond per 1M points
 * against 30 ms for the host's recursive nth_element) and balanced by count, so it is indifferent to where the points are
 * (GridKNN is the faster structure on near-uniform clouds, sp_grid_*). sp_bvh_create allocates and synchronises.
 *   sp_bvh_search    exact kNN, 1 <= k <= 32, queries searched at transT * q (NULL: identity; host or device matrix as for
 *                    sp_kdtree_search); rows as KNNResult (knn/result.hpp:12-34): ascending, -1 / FLT_MAX padded; ties to the
 *                    lowest index — bit-identical to sp_knn_bruteforce. The queries may come in any order (400 k or more are
 *                    searched along the tree's own curve, rows still by query number; scratch from the library's pool).
 *   sp_bvh_self_knn  the cloud's own points as queries (row i = neighbours of point i, itself first), walked in tree order. */
typedef struct sp_bvh sp_bvh;
int sp_bvh_create(const float* points, size_t n, void* stream, sp_bvh** out);
void sp_bvh_destroy(sp_bvh* bvh);
size_t sp_bvh_size(const sp_bvh* bvh);
int sp_bvh_search(const sp_bvh* bvh, const float* queries, size_t nq, size_t k, const float* transT, int transT_on_device,
                  int32_t* idx_out, float* d2_out, void* stream);
int sp_bvh_self_knn(const sp_bvh* bvh, size_t k, int32_t* idx_out, float* d2_out, void* stream);
/* KDTree::radius_search_async (knn/kdtree.hpp:574-719) on the device-built hierarchy: per query the max_k (<= 32) nearest of the
 * points within `radius` (squared distance <= radius^2, inclusive as in the reference), ascending, -1 / FLT_MAX padded; ties to
 * the lowest index. Same arguments as sp_kdtree_radius_search. */
int sp_bvh_radius_search(const sp_bvh* bvh, const float* queries, size_t nq, size_t max_k, float radius, const float* transT,
                         int transT_on_device, int32_t* idx_out, float* d2_out, void* stream);
/* KDTree::remove_nodes_by_flags (knn/kdtree.hpp:282-284, 721-765), same arguments as sp_kdtree_remove_by_flags: flags 1 = keep,
 * 0 = remove; a kept point p is relabelled new_indices[p] (an order-preserving relabelling, as FilterByFlags::calculate_indices
 * gives, common/filter_by_flags.hpp:30-57); points whose index is >= n_flags are left alone. Lazy, as in the reference: the
 * removed points stay in their leaves but can no longer be found (no rebuild, no compaction; sp_bvh_size does not change);
 * sp_bvh_self_knn afterwards writes the rows of the kept points at their new indices. Allocates scratch and synchronises. */
int sp_bvh_remove_by_flags(sp_bvh* bvh, const uint8_t* flags, const int32_t* new_indices, size_t n_flags, void* stream);
/* The points the tree was built on, back in their original order (x, y, z, 1): the tree keeps its own copy, like the nodes of
 * the reference's KDTree, so a caller that needs them again later (the facade builds the reference-topology KD-tree lazily,
 * for radius search / lazy delete) does not depend on the source cloud still being there. */
int sp_bvh_export_points(const sp_bvh* bvh, float* points_out, void* stream);

/* --------------------------------------------------------------------------------------- voxel grid */

/* filter::kernel::compute_voxel_bit (algorithms/common/voxel_constants.hpp:36-62, kernel K9).
 * inv_voxel_size is 1.0f / voxel_size computed once on the host (filter/voxel_downsampling.hpp:27). */
int sp_voxel_keys(const float* points, size_t n, float inv_voxel_size, uint64_t* keys_out, void* stream);

/* VoxelGrid::downsampling (filter/voxel_downsampling.hpp:50-79, 146-288): per-voxel mean of the points (and of rgb
 * and timestamps, median of intensities, when those attribute pointers are non-NULL), voxels with
 * point_sum.w < min_voxel_count dropped, output in ascending key order. The host std::sort + sequential
 * run-length mean of the reference is replaced by a device radix sort of (key, index) (64-bit keys here: 8 passes; see the
 * boxed form below for the fast path) and a segmented reduction;
 * within a voxel points are summed in ascending index order (the reference's order is unspecified: its sort is
 * unstable). *n_out_dev (a device uint32) receives the voxel count; out arrays must hold n entries and must not overlap
 * the inputs.
 * workspace: sp_voxel_downsample_workspace_bytes(n) bytes. */
size_t sp_voxel_downsample_workspace_bytes(size_t n);
int sp_voxel_downsample(const float* points, size_t n, float inv_voxel_size, size_t min_voxel_count,
                        const float* rgb, const float* intensities, const float* timestamps, float* points_out,
                        float* rgb_out, float* intensities_out, float* timestamps_out, uint64_t* keys_out_opt,
                        uint32_t* n_out_dev, void* workspace, size_t workspace_bytes, void* stream);

/* The same operation with the bounding box of the cloud's voxel coordinates known to the HOST (box6 = min x,y,z, max x,y,z
 * of the 21-bit key fields; sp_voxel_key_box computes it on the device: read it back, or keep the previous scan's box and
 * check *status_dev_opt). The keys are then compressed to the voxel's position in the box — same order — and sorted by
 * exactly the bits the box needs (3 passes of the hand-written radix sort for a 200^3 box; the unboxed call sorts the 64-bit
 * keys in 8 passes of the same sort: the sort is most of the run time). Results are identical to sp_voxel_downsample.
 * *status_dev_opt receives the number of valid points whose voxel lies outside the box: non-zero means the box did not
 * cover the cloud and the outputs must be discarded. A NULL, empty or too large box (>= 2^32 - 1 cells) falls back to the
 * 64-bit path.
 * box_shards_dev_opt (SP_VOXEL_BOX_SHARDS shards of SP_VOXEL_BOX_SHARD_STRIDE ints — one 128-byte line each — of which the
 * first 6 are used): the key kernel also records THIS cloud's coordinate box on the way — fold the shards (min of
 * [s][0..2], max of [s][3..5]; an empty cloud leaves INT32_MAX / INT32_MIN) for the next call's guess, or for the exact box
 * of a redo when *status_dev_opt != 0. No separate pass over the points. */
#define SP_VOXEL_BOX_SHARDS 16
#define SP_VOXEL_BOX_SHARD_STRIDE 32
int sp_voxel_key_box(const float* points, size_t n, float inv_voxel_size, int32_t* box6_dev, void* stream);
int sp_voxel_downsample_boxed(const float* points, size_t n, float inv_voxel_size, size_t min_voxel_count,
                              const float* rgb, const float* intensities, const float* timestamps, float* points_out,
                              float* rgb_out, float* intensities_out, float* timestamps_out, uint64_t* keys_out_opt,
                              uint32_t* n_out_dev, const int32_t* box6_host, uint32_t* status_dev_opt,
                              int32_t* box_shards_dev_opt, void* workspace, size_t workspace_bytes, void* stream);
/* sp_voxel_downsample_boxed with ONE record in place of the status word and the sharded box (VoxelGrid::downsampling,
 * voxel_downsampling.hpp:146-288, frame after frame):
 *   report8 = {voxels written, valid points outside box6 (non-zero: discard the outputs and call again with the box below),
 *              THIS cloud's key box min x, y, z, max x, y, z (INT32_MAX / INT32_MIN when no point is valid)}
 * stored once by the call's last kernel, word 0 LAST and behind a system-scope release: report8 may be the device pointer of
 * host-mapped pinned memory whose word 0 the caller armed with a value no count takes and spins on — no copy, no
 * synchronisation. (The other outputs may still be in flight when word 0 lands; work enqueued on the stream stays ordered
 * behind them.) The key kernel's workgroups leave one record each in the workspace and the last kernel folds them, so there are
 * no atomics to initialise: one launch fewer than the boxed call, two with its read-back. n_out_dev_opt may be NULL. */
int sp_voxel_downsample_report(const float* points, size_t n, float inv_voxel_size, size_t min_voxel_count,
                               const float* rgb, const float* intensities, const float* timestamps, float* points_out,
                               float* rgb_out, float* intensities_out, float* timestamps_out, uint64_t* keys_out_opt,
                               uint32_t* n_out_dev_opt, const int32_t* box6_host, uint32_t* report8, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------- transform */

/* transform::transform_async (algorithms/common/transform.hpp:14-37, 45-94, kernel K14); in place allowed;
 * covs / normals may be NULL. */
int sp_transform(const float* points, const float* covs, const float* normals, size_t n, const float* transT_host,
                 float* points_out, float* covs_out, float* normals_out, void* stream);

/* BoxFilterOperator kernel (filter/preprocess_operator/box_filter_operator.hpp:36-44, common.hpp:15-25, K10):
 * flags_out[i] = 1 keep / 0 remove. */
int sp_box_filter_flags(const float* points, size_t n, float min_distance, float max_distance, uint8_t* flags_out,
                        void* stream);
/* FilterByFlags (algorithms/common/filter_by_flags.hpp:30-57, 87-99) on the device: stable compaction of rows of
 * `row_bytes` bytes; new_indices_out_opt[i] = new index or -1; *n_out_dev = kept count.
 * workspace: sp_compact_workspace_bytes(n). */
size_t sp_compact_workspace_bytes(size_t n);
int sp_compact_by_flags(const void* rows, size_t n, size_t row_bytes, const uint8_t* flags, void* rows_out,
                        int32_t* new_indices_out_opt, uint32_t* n_out_dev, void* workspace, size_t workspace_bytes,
                        void* stream);
/* The same for several attribute arrays of one cloud at once (FilterByFlags is applied to points, covariances, normals, ...
 * in turn, preprocess_operator_base / filter_by_flags.hpp:87-99): ONE scan of the flags, one compaction launch per array,
 * one count. n_arrays <= 16; rows[a] has rows of row_bytes[a] bytes and goes to rows_out[a] (host arrays of device pointers). */
int sp_compact_by_flags_multi(const void* const* rows, const size_t* row_bytes, void* const* rows_out, int n_arrays, size_t n,
                              const uint8_t* flags, int32_t* new_indices_out_opt, uint32_t* n_out_dev, void* workspace,
                              size_t workspace_bytes, void* stream);

/* BoxFilter + FilterByFlags in one launch (+ one that zeroes the scan's state): the box test of sp_box_filter_flags on `points`,
 * the exclusive scan of its flags by decoupled look-back and the stable move of the kept rows of every attribute array — what
 * PreprocessFilter::box_filter does with a cloud (preprocess_operator/box_filter_operator.hpp:24-54 + common/filter_by_flags.hpp:
 * 30-57: a flags kernel, then one filter_by_flags per attribute). flags_out_opt receives the flags (1 keep, 0 remove), the other
 * arguments are sp_compact_by_flags_multi's (rows_out[a] == NULL: that array is not moved). Only enqueues. */
int sp_box_filter_compact_multi(const float* points, size_t n, float min_distance, float max_distance, const void* const* rows,
                                const size_t* row_bytes, void* const* rows_out, int n_arrays, uint8_t* flags_out_opt,
                                int32_t* new_indices_out_opt, uint32_t* n_out_dev, void* workspace, size_t workspace_bytes,
                                void* stream);

/* Rows picked by index, every attribute of a cloud in ONE launch: rows_out[a][j] = rows[a][indices[j]] for j < m (indices:
 * device memory, uint32). What random sampling is once the host has drawn its sample (preprocess_operator/
 * random_sampling_operator.hpp:24-51 draws on the host, then filters by flags, common/filter_by_flags.hpp:30-57: with the indices
 * in ascending order the result is that compaction's, without scanning the flags of the whole cloud per attribute).
 * n_arrays <= 16, row_bytes[a] a multiple of 4. Only enqueues. */
int sp_gather_rows_multi(const void* const* rows, const size_t* row_bytes, void* const* rows_out, int n_arrays,
                         const uint32_t* indices, size_t m, void* stream);

/* ------------------------------------------------------------------------------------- registration */

/* RegType (algorithms/registration/factor.hpp:18-32) and RobustLossType (algorithms/robust/robust.hpp:13-19). */
enum { SP_REG_POINT_TO_POINT = 0, SP_REG_POINT_TO_PLANE = 1, SP_REG_POINT_TO_DISTRIBUTION = 2, SP_REG_GICP = 3,
       SP_REG_GENZ = 4 };
enum { SP_LOSS_NONE = 0, SP_LOSS_HUBER = 1, SP_LOSS_TUKEY = 2, SP_LOSS_CAUCHY = 3, SP_LOSS_GEMAN_MCCLURE = 4 };

/* The reduced linear system, the 176 bytes the reference keeps in LinearizedDevice
 * (algorithms/registration/registration.hpp:26-84): H row-major, b, error, inlier count.
 * inlier_lo/inlier_hi carry the same count as two exactly-representable floats (count = hi*4096 + lo) so that a
 * float sum all-reduce over ranks stays exact. */
typedef struct sp_linearized {
    float H[36];
    float b[6];
    float error;
    uint32_t inlier;
    float inlier_lo;
    float inlier_hi;
    float pad[2];
} sp_linearized; /* 192 bytes */

typedef struct sp_factor_params { /* RegistrationFactorParams, registration_params.hpp:46-71 */
    int reg_type;
    int robust_type;
    float max_correspondence_distance;
    float robust_scale;
    float genz_alpha;               /* Registration::genz_alpha_ (registration.hpp:370,519) */
    float genz_planarity_threshold; /* registration_params.hpp:52-54 */
    /* RotationConstraint (registration_params.hpp:56-64; rotation_constraint.hpp:15-128): when enabled every inlier
     * correspondence adds the Jensen-Bregman LogDet divergence of (R Cs R^T, Ct), weighted by
     * rotation_constraint_weight and the robust kernel at rotation_robust_scale, to H, b and the error
     * (registration.hpp:630-650, 758-766). Needs source and target covariances. Only the unfused entry points
     * (sp_gicp_linearize / sp_gicp_error) carry the term; the prepared / one-call paths reject it. */
    int rotation_constraint_enable;
    float rotation_constraint_weight;
    float rotation_robust_scale;
} sp_factor_params;

/* Registration::linearize_parallel_reduction_async (registration.hpp:513-664, kernel K11): for every source point
 * whose neighbour distance nn_d2 <= max_corr^2, linearise the factor (factor.hpp:69-449), apply the robust weight
 * (robust.hpp:56-114) and sum H, b, error, inlier count into *out. src_covs / tgt_covs / tgt_normals may be NULL
 * where the reg_type does not need them (registration.hpp:539-543: Identity / Zero are substituted).
 * transT: current pose, host or device (transT_on_device). workspace: sp_gicp_workspace_bytes(n).
 * The reduction is a fixed tree: two runs on the same inputs give bit-identical sums. */
size_t sp_gicp_workspace_bytes(size_t n);
int sp_gicp_linearize(const float* src_points, const float* src_covs, size_t n, const float* tgt_points,
                      const float* tgt_covs, const float* tgt_normals, const int32_t* nn_idx, const float* nn_d2,
                      const float* transT, int transT_on_device, const sp_factor_params* params, sp_linearized* out,
                      void* workspace, size_t workspace_bytes, void* stream);
/* Registration::compute_error_parallel_reduction (registration.hpp:678-777, kernel K12): out->error, out->inlier. */
int sp_gicp_error(const float* src_points, const float* src_covs, size_t n, const float* tgt_points,
                  const float* tgt_covs, const float* tgt_normals, const int32_t* nn_idx, const float* nn_d2,
                  const float* transT, int transT_on_device, const sp_factor_params* params, sp_linearized* out,
                  void* workspace, size_t workspace_bytes, void* stream);
/* Registration::compute_icp_robust_weights_async (registration.hpp:412-462, kernel K13). */
int sp_icp_robust_weights(const float* src_points, const float* src_covs, size_t n, const float* tgt_points,
                          const float* tgt_covs, const float* tgt_normals, const int32_t* nn_idx, const float* nn_d2,
                          const float* transT, int transT_on_device, const sp_factor_params* params,
                          float* weights_out, void* stream);
/* Registration::compute_genz_alpha (registration.hpp:464-511): writes {inlier, planar} counts to counts_out[2]. */
int sp_genz_counts(const float* tgt_covs, const int32_t* nn_idx, const float* nn_d2, size_t n, float max_corr,
                   float planarity_threshold, uint32_t* counts_out, void* stream);

/* Prepared / fused GICP iteration (MI355X-native form of registration.hpp:229-234 = NN search + K11 in one pass).
 * The reference recomputes covariance::kernel::update_covariance_plane for both covariances of every correspondence in
 * every iteration (factor.hpp:249-255). The result depends on the covariance alone, so here it is computed once:
 *   sp_gicp_target_create : plane-regularised target covariances, packed 8 floats per point
 *                           (xx,xy,xz,yy | yz,zz,rho^2,0) and stored in the cell order of `grid` (which must have been built
 *                           on the target points and must outlive the object); sp_gicp_target_update recomputes in place.
 *                           rho = half the distance from the point to its nearest other target point (one k = 2
 *                           self-search on the grid; create allocates and synchronises): the iteration kernel keeps a
 *                           source point's previous correspondence t without searching when |T p - t| < rho_t, which
 *                           proves t is still the exact nearest neighbour.
 *   sp_gicp_source_create : buffers for a prepared source of up to n_max points (allocates).
 *   sp_gicp_source_prepare: (enqueue only) packed plane-regularised source covariances. sort_by_cell selects how
 *                           neighbouring lanes get neighbouring cells (their loads then share cache lines):
 *                             SP_SOURCE_ORDER_UNKNOWN (0) keep the caller's order, assume nothing (ring-walk search);
 *                             SP_SOURCE_SORT (1) reorder by the target-grid cell that transT*p falls into
 *                               (two radix-sort passes per alignment);
 *                             SP_SOURCE_PRESORTED (2) keep the caller's order, which is already spatially coherent —
 *                               sp_grid_order of any grid on the source, or the output order of voxel downsampling —
 *                               no sort, block-walk search: the fastest combination.
 *                           The order only changes which lane handles which point, never a result.
 *   sp_gicp_iteration_fused: per source point q = T p -> exact NN on the grid -> linearise with the packed
 *                           covariances -> reduce to *out. If nn_idx_out/nn_d2_out are non-NULL the correspondences
 *                           are also written, in ORIGINAL source order (for sp_gicp_error / compute_error_frozen); with
 *                           NULL outputs (here and in sp_gicp_align_*) the search is bounded by
 *                           max_correspondence_distance — a neighbour beyond it would be rejected anyway, so a source
 *                           point without a correspondence costs one block of cells instead of a walk out to wherever its
 *                           nearest target point is (clouds that only partly overlap). If
 *                           `gn` is non-NULL (single-GPU loops) the same launch also solves (H + lambda I) delta = -b
 *                           and updates the DEVICE pose transT in place, writing delta_out8 as sp_gn_update does; with
 *                           gn == NULL the caller all-reduces *out over ranks and calls sp_gn_update.
 * Same mathematics as sp_grid_search + sp_gicp_linearize; rounding differs (symmetric packing, upper-triangle H,
 * summation order). reg_type must be SP_REG_GICP or SP_REG_POINT_TO_DISTRIBUTION (the one the target was prepared for,
 * sp_gicp_target_prepare); every robust loss is supported. */
typedef struct sp_gicp_target sp_gicp_target;
typedef struct sp_gicp_source sp_gicp_source;
typedef struct sp_gn_params { float lambda, crit_rotation, crit_translation; } sp_gn_params;
enum { SP_SOURCE_ORDER_UNKNOWN = 0, SP_SOURCE_SORT = 1, SP_SOURCE_PRESORTED = 2 };
int sp_gicp_target_create(const sp_grid* grid, const float* tgt_covs, size_t n, void* stream, sp_gicp_target** out);
int sp_gicp_target_update(sp_gicp_target* target, const float* tgt_covs, void* stream);
/* The same without the reuse certificates (no k = 3 self-search: a fifth of the cost at 6 k points, where the search is
 * 0.1 ms of launch-bound work): every linearisation then searches every point, starting from its previous winner. What a
 * caller wants for a target it aligns ONE small source against — the reference's example builds a new target per frame and
 * aligns a 1000-point sample to it. sp_gicp_target_certify adds the certificates later (a target that turns out to be reused);
 * it synchronises, rewrites the rows and invalidates the correspondence caches of prepared sources. */
int sp_gicp_target_create_plain(const sp_grid* grid, const float* tgt_covs, size_t n, void* stream, sp_gicp_target** out);
int sp_gicp_target_certify(sp_gicp_target* target, const float* tgt_covs, void* stream);
int sp_gicp_target_has_certificates(const sp_gicp_target* target);
/* The same, choosing the factor the rows serve: SP_REG_GICP (what create makes: V diag(1e-3,1,1) V^T of the covariance) or
 * SP_REG_POINT_TO_DISTRIBUTION (linearize_point_to_distribution, factor.hpp:311-373: the information matrix itself,
 * inverse(Ct) of the RAW covariance, Zero when |det| < 1e-6, so the iteration inverts nothing per point and reads no source
 * covariance — sp_gicp_source_prepare then accepts src_covs == NULL). sp_gicp_target_update keeps the current choice. The
 * iteration / align / error entry points require params->reg_type to match (SP_ERR_INVALID_ARGUMENT otherwise). */
int sp_gicp_target_prepare(sp_gicp_target* target, const float* tgt_covs, int reg_type, void* stream);
void sp_gicp_target_destroy(sp_gicp_target* target);
int sp_gicp_source_create(size_t n_max, sp_gicp_source** out);
int sp_gicp_source_prepare(sp_gicp_source* source, const sp_gicp_target* target, const float* src_points,
                           const float* src_covs, size_t n, const float* transT, int transT_on_device, int sort_by_cell,
                           void* stream);
void sp_gicp_source_destroy(sp_gicp_source* source);
int sp_gicp_iteration_fused(const sp_gicp_target* target, const sp_gicp_source* source, float* transT,
                            int transT_on_device, const sp_factor_params* params, const sp_gn_params* gn,
                            int32_t* nn_idx_out, float* nn_d2_out, sp_linearized* out, float* delta_out8,
                            void* workspace, size_t workspace_bytes, void* stream);
/* Registration::compute_error_parallel_reduction (registration.hpp:678-777, kernel K12) on the prepared path: the error of
 * the factor at transT_trial with the correspondences FROZEN at those of the last linearisation of `source` against
 * `target` (sp_gicp_iteration_fused / sp_gicp_align_*), read from the source's correspondence cache — what the trial steps
 * of optimize_levenberg_marquardt (:830-895) and optimize_powell_dogleg (:897-965) call per step. transT_lin_host is the
 * pose of that linearisation (host, column-major): the inlier gate nn_d2 <= max_corr^2 is evaluated there, as the
 * reference's frozen distances are. out->error / out->inlier (H, b are not written). SP_ERR_RUNTIME when nothing has
 * been linearised since sp_gicp_source_prepare. workspace: sp_gicp_workspace_bytes(n). */
int sp_gicp_error_prepared(const sp_gicp_target* target, const sp_gicp_source* source, const float* transT_lin_host,
                           const float* transT_trial, int trial_on_device, const sp_factor_params* params,
                           sp_linearized* out, void* workspace, size_t workspace_bytes, void* stream);
/* Registration::align's whole Gauss-Newton loop (registration.hpp:229-276) enqueued by ONE call; pose, convergence flag and
 * iteration count stay in a state block of the workspace, nothing is read back by the host. Per iteration:
 *   streaming launch its first act finishes the PREVIOUS iteration, in every workgroup for itself: the previous launch's
 *                    partial rows summed in a fixed order, (H + lambda I) delta = -b solved, T <- T * se3_exp(delta)
 *                    (workgroup 0 publishes the state); then every point: cached correspondence when its reuse certificate
 *                    holds, else exact NN on the grid -> linearise -> one partial row per workgroup.
 *   after the last   a one-workgroup launch finishes the last iteration the same way and writes the outputs.
 * Once is_converged() (registration.hpp:407-410) holds, the remaining launches return at once, as the reference breaks out
 * of its loop. All pointers are device memory:
 *   transT_device  in: initial guess, out: final pose (column-major 4x4)
 *   lin_out        system of the last executed iteration (optional)
 *   delta_out8     its delta[6], converged flag, solve-ok flag (optional)
 *   iterations_out number of Gauss-Newton steps applied (optional); the reference's result.iterations is this - 1
 *   nn_idx_out / nn_d2_out: neighbours of the last linearisation, in original source order (optional, both or none)
 * Workspace: sp_gicp_workspace_bytes(n). Graph-capturable. */
int sp_gicp_align_fused(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                        const sp_factor_params* params, const sp_gn_params* gn, int max_iterations,
                        int32_t* nn_idx_out, float* nn_d2_out, sp_linearized* lin_out, float* delta_out8,
                        uint32_t* iterations_out, void* workspace, size_t workspace_bytes, void* stream);
/* The same loop one iteration at a time, for callers that put something between the iterations — on several GPUs (source
 * tile-sharded, target replicated) an all-reduce:
 *   for k in 0 .. max_iterations-1:  sp_gicp_align_step(k)                       (enqueue iteration k)
 *                                    all-reduce(sum) sp_gicp_align_rows(ws, k)   (32 KB of float32, in place, same stream order)
 *   sp_gicp_align_finish(last_k = max_iterations-1)
 * rows_all_reduced == 0: one GPU, the loop of sp_gicp_align_fused (iteration k is finished by step k + 1, or by finish:
 * lin_out and the state block describe iteration k only from then on). With rows_all_reduced != 0 the streaming launch leaves
 * its sums for the collective, and launch k + 1 (after the last one: finish) finishes iteration k from the all-reduced sums
 * in its prologue — the same sums and the same solve on every rank, hence the identical pose without a broadcast.
 * rows_all_reduced == 1: all partial rows travel (inlier counts as float VALUES, exact: < 2^24 per row); every rank must
 * all-reduce all of sp_gicp_align_rows' floats (rows a rank does not use are zeroed by step 0). All ranks must pass the same
 * rows_all_reduced. transT_device must not be written between step 0 and finish. */
int sp_gicp_align_step(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                       const sp_factor_params* params, const sp_gn_params* gn, int k, int rows_all_reduced,
                       int32_t* nn_idx_out, float* nn_d2_out, sp_linearized* lin_out, void* workspace,
                       size_t workspace_bytes, void* stream);
float* sp_gicp_align_rows(void* workspace, int k, size_t* n_floats_out);
/* rows_all_reduced == 2 (the form sp_gicp_align_sharded uses): the launch reduces its own partial rows to ONE 128-byte row
 * inside the kernel — every workgroup stores its row write-through and takes an agent-scope ticket; the last arriver sums
 * the rows in the fixed order of the single-GPU loop (bit-identical to it) — and the caller all-reduces just
 * sp_gicp_align_row(ws, k): 28 sums, the inlier count as two exactly-summable floats (hi * 4096 + lo), the searched-point
 * count. */
float* sp_gicp_align_row(void* workspace, int k, size_t* n_floats_out);
int sp_gicp_align_finish(const sp_gicp_source* source, float* transT_device, const sp_gn_params* gn, int last_k,
                         int rows_all_reduced, sp_linearized* lin_out, float* delta_out8, uint32_t* iterations_out,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Pose of the LAST LINEARISATION of an alignment enqueued through sp_gicp_align_* with last iteration index last_k (the
 * pose before the final update; the pose at which convergence was detected when the loop stopped early): what the
 * reference's neighbors_ are frozen at after align() (registration.hpp:229-234), i.e. the transT_lin of a following
 * sp_gicp_error_prepared / Registration::compute_error_frozen. Copies 16 floats (column-major) to transT_lin_out (host or
 * device memory) in stream order. */
int sp_gicp_align_linearization_pose(const void* workspace, int last_k, float* transT_lin_out, void* stream);
/* Registration::align's optimiser loop for EVERY OptimizationMethod (registration.hpp:201-276 with optimize_gauss_newton
 * :803-828, optimize_levenberg_marquardt :830-895, optimize_powell_dogleg :897-964 and dogleg_step.hpp:35-101) and, in the same
 * launch, the robust-scale annealing of pipeline::RobustAligner around it (pipeline/robust.hpp:78-111: one align() per level,
 * each starting at the pose the previous level ended on) — device-resident: ONE launch whose workgroups loop on the device,
 * ONE read-back (sp_align_result) per alignment. The reference crosses host <-> device twice per Gauss-Newton iteration and
 * 2 + inner tries times per LM iteration (:674-675, :685-686); its own example (LM + Geman-McClure + 3 levels on a 1000-point
 * sample, example_registration.cpp:29-55) is ~60 such round trips.
 *   step "linearise"  every source point: cached correspondence when its reuse certificate holds, else exact NN on the grid
 *                     -> K11 sums (28 floats + inlier count) -> one partial row per workgroup
 *   step "trial"      K12 of the trial pose over the correspondence-cache rows (frozen correspondences, sp_gicp_error_prepared's
 *                     arithmetic) -> one partial row per workgroup
 *   between steps     every workgroup waits for the rows of all workgroups (arrival counter, bounded wait), sums them in a fixed
 *                     order and runs the optimiser's state machine for itself — the same decisions in every workgroup:
 *                       GN      delta = LDLT(H + lambda I).solve(-b); T <- T exp(delta)
 *                       LM      up to lm_max_inner_iterations trials T exp(delta(lambda)): accept on new_error <= current_error
 *                               (lambda /= factor), stop on |new_error - last_error| <= 1e-6, else lambda *= factor
 *                       DOGLEG  compute_dogleg_step; predicted <= 0 -> shrink; one trial; rho < eta1 -> shrink, else accept
 *                               (rho > eta2 and a full step -> grow)
 *                     an outer iteration ends -> is_converged() or max_iterations ends the level -> next robust scale or done.
 * Workgroups have 256 lanes up to 64 K source points (the reference pipeline's default 1000-point random sample: 4 workgroups
 * on 4 compute units) and 1024 beyond; a launch of one workgroup (<= 256 points) needs no counter at all.
 * robust_scales[n_levels] (host, 1 <= n_levels <= SP_OPT_MAX_LEVELS): the robust scale of each level; lambda / trust radius /
 * result fields restart at every level as a fresh align() would. transT_device: in the initial guess, out the final pose.
 * result_device: sp_align_result in device memory, written once at the end (status 0 ok; 2 a wait ran out — not every workgroup
 * of the launch was resident, the pose is NaN: run again after sp_gicp_source_set_persistent(source, 0), which makes this
 * entry point return SP_ERR_RUNTIME "not available" so that the caller takes its per-step loop).
 * Only enqueues. SP_ERR_RUNTIME (nothing enqueued) when the launch cannot be resident now: the grid exceeds the device's compute
 * units, `stream` is capturing, or another stream's persistent launch may still be running. reg_type GICP or
 * POINT_TO_DISTRIBUTION as the target was prepared; no rotation constraint. Workspace: sp_gicp_workspace_bytes(n). */
enum { SP_OPT_GAUSS_NEWTON = 0, SP_OPT_LEVENBERG_MARQUARDT = 1, SP_OPT_POWELL_DOGLEG = 2 }; /* OptimizationMethod, registration_params.hpp:17-21 */
enum { SP_OPT_MAX_LEVELS = 8, SP_OPT_LOG_ENTRIES = 64 };
typedef struct sp_opt_params { /* RegistrationParams: optimization_method, max_iterations, criteria, gn, lm, dogleg (registration_params.hpp:74-114) */
    int method;
    int max_iterations;
    float crit_rotation, crit_translation;
    float gn_lambda;
    int lm_max_inner_iterations;
    float lm_lambda_factor, lm_init_lambda, lm_max_lambda, lm_min_lambda;
    float dl_initial_radius, dl_min_radius, dl_max_radius, dl_eta1, dl_eta2, dl_gamma_decrease, dl_gamma_increase;
} sp_opt_params;
typedef struct sp_opt_log_entry { /* one outer iteration */
    uint16_t level, iteration;
    uint16_t trials;     /* K12 evaluations of this iteration */
    uint16_t accepted;   /* 1 the pose moved (LM: new_error <= current_error; dog-leg: rho >= eta1; GN: always), 2 LM's
                            stagnation exit (|new_error - last_error| <= 1e-6: pose taken, `updated` false), 0 rejected */
    float damping;       /* lambda (LM) / trust-region radius (dog-leg) AFTER the iteration */
    float error;         /* RegistrationResult::error after the iteration */
} sp_opt_log_entry;
#define SP_ALIGN_RESULT_DONE 0x600DF00Du
typedef struct sp_align_result { /* RegistrationResult (result.hpp:12-28) of the LAST level + what the facade needs beside it */
    float T[16];         /* final pose, column-major */
    float T_lin[16];     /* pose of the last linearisation: the correspondence cache is frozen at it (compute_error_frozen) */
    float H[36];         /* row-major */
    float b[6];
    float error;         /* RegistrationResult::error (GN / dog-leg: of the last linearisation unless a trial was accepted) */
    float error_raw;     /* error of the last linearisation (RegistrationResult::error_raw; H_raw = H, b_raw = b on this path) */
    uint32_t inlier;
    uint32_t iterations; /* index of the last outer iteration of the last level (RegistrationResult::iterations) */
    uint32_t converged;
    uint32_t status;     /* 0 ok, 2 a wait between steps ran out (pose NaN) */
    uint32_t linearizations, trials; /* steps executed, all levels */
    uint32_t searched;   /* source points searched for, all linearisations (the rest reused their correspondence) */
    float damping;       /* final lambda / trust-region radius */
    uint32_t log_entries;
    uint32_t pad[3];     /* pad[0] = SP_ALIGN_RESULT_DONE, stored LAST (system-scope release) when the block is complete: result_device
                          * may be the device pointer of host-mapped pinned memory whose pad[0] the caller cleared and spins on — no
                          * read-back copy, no synchronisation (transT_device may be host-mapped too: 64 bytes each way) */
    sp_opt_log_entry log[SP_OPT_LOG_ENTRIES]; /* the first outer iterations, all levels in order */
} sp_align_result;
int sp_gicp_align_optimize(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                           const sp_factor_params* params, const sp_opt_params* opt, const float* robust_scales, int n_levels,
                           sp_align_result* result_device, void* workspace, size_t workspace_bytes, void* stream);
/* 0: sp_gicp_align_fused runs every iteration as a launch of its own and sp_gicp_align_optimize reports "not available" — no
 * launch whose workgroups wait for each other is started for this source. Default 1. */
int sp_gicp_source_set_persistent(sp_gicp_source* source, int enable);
/* How sp_gicp_align_optimize's linearisation steps deal the source points: 0 a lane per point; 1 (default) a wave per point for
 * sources of up to 2048 points (more SIMDs than points: the step costs the longest chain of dependent loads, and 64 lanes scanning
 * one query's ball make that chain a handful of round trips); 2 a wave per point for sources of up to 131072 points — for a
 * target whose cells are crowded (sp_grid_max_cell_points in the hundreds or thousands: a raw LiDAR scan), where a lane's walk
 * through its block of cells is thousands of candidates long. Same correspondences in every mode. */
int sp_gicp_source_set_wave_per_point(sp_gicp_source* source, int mode);
/* Registration::optimize_gauss_newton (registration.hpp:791-828) as ONE device thread, so a whole fixed-length
 * iteration loop can stay on the stream with no host round trip:
 *   delta = LDLT(H + lambda*I).solve(-b);  T <- T * se3_exp(delta);  delta_out[0..5] = delta,
 *   delta_out[6] = 1.0f if converged (|rot| < crit_rot && |trans| < crit_trans) else 0, delta_out[7] = solve ok.
 * lin->inlier_lo/hi (possibly summed over ranks) are folded back into lin->inlier. T_dev is updated in place. */
int sp_gn_update(sp_linearized* lin, float* T_dev, float lambda, float crit_rotation, float crit_translation,
                 float* delta_out8, void* stream);
/* Host twin of the same arithmetic, for a host-driven loop (reads/writes HOST memory, no stream). */
int sp_gn_update_host(const sp_linearized* lin_host, float* T_host, float lambda, float crit_rotation,
                      float crit_translation, float* delta_out8_host);
/* lie::se3_exp (utils/eigen_utils.hpp:909-943) and the Isometry3f product used for T <- T*exp(delta), on the host. */
void sp_se3_exp_host(const float* twist6, float* T_out16);
void sp_rigid_mul_host(const float* A16, const float* B16, float* out16);
int sp_ldlt6_solve_host(const float* H36_rowmajor, const float* rhs6, float* x6);
/* compute_dogleg_step<6> (algorithms/registration/dogleg_step.hpp:35-101), the step geometry of
 * Registration::optimize_powell_dogleg (registration.hpp:897-965); trust-region bookkeeping stays with the caller. */
void sp_dogleg_step_host(const float* H36_rowmajor, const float* g6, float trust_region_radius, float* p_out6,
                         float* step_norm_out, float* predicted_reduction_out);

/* ---- pose-space terms applied to the reduced system on the host, between the device reduction and the solve
 * (Registration::align, registration.hpp:236-253). All matrices here are HOST memory; H row-major as in sp_linearized,
 * poses column-major 4x4. */
/* lie::se3_log (utils/eigen_utils.hpp:991-1034): rotation-first twist of a rigid transform. */
void sp_se3_log_host(const float* T16, float* twist6);
/* DegenerateRegularization::regularize (algorithms/registration/degenerate_regularization.hpp:41-110, NL-Reg): for the
 * rotation and the translation 3x3 block of H, every eigen-direction whose eigenvalue / inlier is below its threshold
 * gets a Tikhonov penalty base_factor * inlier * v v^T; H += P, b += P * se3_log(T_initial^-1 * T_current).
 * Nothing happens for type NONE or inlier == 0. */
enum { SP_DEGENERATE_REG_NONE = 0, SP_DEGENERATE_REG_NL_REG = 1 };
typedef struct sp_degenerate_reg_params { /* DegenerateRegularizationParams, :41-46 */
    int type;
    float rot_eigenvalue_threshold;   /* 10.0 */
    float trans_eigenvalue_threshold; /* 1.0 */
    float base_factor;                /* 1.0 */
} sp_degenerate_reg_params;
int sp_degenerate_regularize_host(const sp_degenerate_reg_params* params, float* H36_rowmajor, float* b6,
                                  uint32_t inlier, const float* T_current16, const float* T_initial16);
/* MapPrior (algorithms/registration/map_prior.hpp:14-213). update (:97-174): from the previous frame's raw Hessian,
 * raw error, inlier count and optimised pose, and the predicted pose of this frame, compute
 * Omega = R - R (Ad^T (H_raw / s^2) Ad + R)^-1 R with R = Q^-1 the motion-adaptive process noise; state->has_prior
 * stays 0 when the prior is disabled or cannot be formed. apply (:181-201): e = se3_log(T_pred^-1 T_est);
 * returns the prior cost e^T Omega e / 2 (0 without a prior) and, when H/b are given, adds Omega, Omega e and the cost
 * to H, b and *error. */
typedef struct sp_map_prior_params { /* MapPriorParams, :14-35 */
    int enabled;
    float rot_vel_sigma;    /* 1.0 */
    float trans_vel_sigma;  /* 1.0 */
    float rot_base_sigma;   /* 3.16e-2 */
    float trans_base_sigma; /* 1e-2 */
} sp_map_prior_params;
typedef struct sp_map_prior_state {
    int has_prior;
    float omega[36];      /* row-major */
    float T_pred_inv[16]; /* column-major */
} sp_map_prior_state;
int sp_map_prior_update_host(const sp_map_prior_params* params, const float* H_raw36_rowmajor, float error_raw,
                             uint32_t inlier, const float* T_prev16, const float* T_pred16, sp_map_prior_state* state);
float sp_map_prior_apply_host(const sp_map_prior_state* state, const float* T_est16, float* H36_rowmajor, float* b6,
                              float* error);

/* ----------------------------------------------------------------------------------------- multi-GPU */

/* One process per GPU; the source cloud is sharded over the ranks, the target (points, covariances, grid, prepared rows)
 * is replicated. The exchange is RCCL over xGMI, bound at run time (dlopen librccl.so.1): on a machine
 * without RCCL these entry points return SP_ERR_RUNTIME and everything else still works.
 *   sp_comm_unique_id   ncclGetUniqueId: SP_COMM_ID_BYTES bytes, made by ONE rank and handed to the others out of band
 *                       (MPI_Bcast, a torch.distributed broadcast, a file)
 *   sp_comm_create      ncclCommInitRank on the calling thread's current device; collective over the `world` callers
 *   sp_allreduce_rows   sum over the ranks, in place, of launch k's fan-in row (sp_gicp_align_row): 128 bytes
 *   sp_allreduce_f32    the same for any float buffer (e.g. the 192-byte system of the generic loop, 2 scalars of an LM step)
 *   sp_allgather        ncclAllGather of bytes_per_rank bytes per rank (target covariances of a pre-loop sharded by query)
 *   sp_gicp_align_sharded  sp_gicp_align_fused with the source sharded over `comm`: per iteration one launch
 *                       (sp_gicp_align_step, rows_all_reduced = 2) + one sp_allreduce_rows, then sp_gicp_align_finish; every
 *                       rank ends with the identical pose. Only enqueues (hipGraph-capturable). All ranks must pass the same
 *                       params, gn and max_iterations; a rank with an empty shard still calls it.
 * An error return on one rank (bad arguments, a HIP error) leaves the other ranks inside that iteration's collective: the
 * communicator cannot be used again — destroy it (sp_comm_destroy) on every rank and create a new one. Launches after
 * convergence contribute a zero row. */
#define SP_COMM_ID_BYTES 128
typedef struct sp_comm sp_comm;
int sp_comm_unique_id(void* id_out);
int sp_comm_create(const void* id, int rank, int world, sp_comm** out);
void sp_comm_destroy(sp_comm* comm);
int sp_comm_rank(const sp_comm* comm);
int sp_comm_world(const sp_comm* comm);
int sp_allreduce_rows(sp_comm* comm, void* workspace, int k, void* stream);
int sp_allreduce_f32(sp_comm* comm, float* buf, size_t n_floats, void* stream);
int sp_allgather(sp_comm* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int sp_gicp_align_sharded(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                          const sp_factor_params* params, const sp_gn_params* gn, int max_iterations, sp_comm* comm,
                          int32_t* nn_idx_out, float* nn_d2_out, sp_linearized* lin_out, float* delta_out8,
                          uint32_t* iterations_out, void* workspace, size_t workspace_bytes, void* stream);

/* The same sharded loop WITHOUT a collective per iteration (direct exchange). An all-reduce of 128 bytes is pure latency — a
 * library launch between two kernels, 15-30 us on an 8-GPU node against ~30 us of compute per iteration. Here every rank owns
 * a small slot buffer in uncached device memory and maps its peers' (hipIpc handles, exchanged ONCE through the caller's own
 * channel); the streaming launch's last-arriving workgroup stores the rank's 128-byte row, tagged with the iteration's
 * sequence number, straight into every rank's buffer over xGMI, and the NEXT streaming launch's prologue (every workgroup
 * for itself) waits until all `world` rows carry the tag, adds them in rank order (identical sums, identical pose on every
 * rank) and solves: one launch per iteration, nothing in between.
 *   sp_xchg_create / sp_xchg_handle / sp_xchg_connect   every rank creates, all ranks gather the SP_XCHG_HANDLE_BYTES-byte
 *                       handles in rank order (MPI_Allgather, torch.distributed, a file ...) and connect; world <= 8
 *   sp_gicp_align_direct   arguments as sp_gicp_align_sharded; only enqueues; every rank must call it the same number of
 *                       times (the sequence numbers are counted per alignment); max_iterations <= 255
 *   sp_gicp_align_status   the wait is BOUNDED (sp_xchg_set_timeout_ms, default 2000): a row that does not arrive stops the
 *                       alignment and raises a flag in its state block instead of hanging the queue; this call reads the
 *                       flag back (it synchronises the stream) -> SP_OK, or SP_ERR_RUNTIME when a peer was missing.
 * Summation order differs from one GPU (per-rank sums first), so poses agree to rounding, as with sp_gicp_align_sharded. */
#define SP_XCHG_HANDLE_BYTES 64
typedef struct sp_xchg sp_xchg;
int sp_xchg_create(int rank, int world, sp_xchg** out);
int sp_xchg_handle(const sp_xchg* xchg, void* handle_out);
int sp_xchg_connect(sp_xchg* xchg, const void* handles_of_all_ranks);
int sp_xchg_set_timeout_ms(sp_xchg* xchg, unsigned milliseconds);
int sp_xchg_rank(const sp_xchg* xchg);
int sp_xchg_world(const sp_xchg* xchg);
void sp_xchg_destroy(sp_xchg* xchg);
int sp_gicp_align_direct(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                         const sp_factor_params* params, const sp_gn_params* gn, int max_iterations, sp_xchg* xchg,
                         int32_t* nn_idx_out, float* nn_d2_out, sp_linearized* lin_out, float* delta_out8,
                         uint32_t* iterations_out, void* workspace, size_t workspace_bytes, void* stream);
int sp_gicp_align_status(const void* workspace, int last_k, void* stream);

/* ------------------------------------------------------------------------------------- voxel hash map */

/* VoxelHashMap (algorithms/mapping/voxel_hash_map.hpp:22-1072): submap accumulation keyed by compute_voxel_bit
 * (sp_voxel_keys' key). The table (double hashing, 100 probes, capacities from the reference's prime list, rehash when
 * voxel_num / capacity exceeds the threshold) lives in HBM and is owned by the object; per voxel it keeps the sums of the
 * map-frame points, of log(R C R^T) (log-Euclidean covariance mean), of rgb and of intensity, a count and the time stamp of
 * its last update.
 *   sp_vhm_create        constructor (:28-36); voxel_size <= 0 -> SP_ERR_INVALID_ARGUMENT (std::invalid_argument, :41-43)
 *   sp_vhm_set / _get    set_voxel_size / set_max_staleness / set_remove_old_data_cycle / set_rehash_threshold /
 *                        set_min_num_point and their getters (:38-80)
 *   sp_vhm_clear         clear (:83-113)
 *   sp_vhm_add_point_cloud  add_point_cloud (:117-141): [rehash] -> integrate the cloud (device points in the SENSOR frame,
 *                        optional covariances / rgb / intensities; sensor_pose_host16 = column-major 4x4 in the map frame)
 *                        -> [remove_old_data every remove_old_data_cycle calls] -> ++staleness counter. Accumulation uses
 *                        relaxed device-scope atomics, as the reference does: counts are exact, float sums agree to rounding.
 *   sp_vhm_downsampling  downsampling (:146-190): voxels with count >= min_num_point whose centroid lies in the axis-aligned
 *                        box center +- distance -> mean point (w = 1), exp of the mean log-covariance, mean rgb / intensity,
 *                        in table-slot order (deterministic for a given table; the reference's order is that of an atomic
 *                        counter). Attribute outputs are written only when the map holds that attribute (sp_vhm_info); out
 *                        arrays must hold sp_vhm_info(SP_VHM_INFO_VOXEL_NUM) entries. keys_out_opt: the voxel keys.
 *   sp_vhm_overlap_ratio compute_overlap_ratio (:196-246)
 *   sp_vhm_remove_old_data remove_old_data (:248, 788-843)
 * Like the reference's methods these calls wait for their kernels (voxel counts are read back by the host). */
typedef struct sp_voxel_hash_map sp_voxel_hash_map;
enum { SP_VHM_VOXEL_SIZE = 0, SP_VHM_MAX_STALENESS = 1, SP_VHM_REMOVE_OLD_DATA_CYCLE = 2, SP_VHM_REHASH_THRESHOLD = 3,
       SP_VHM_MIN_NUM_POINT = 4 };
enum { SP_VHM_INFO_VOXEL_NUM = 0, SP_VHM_INFO_CAPACITY = 1, SP_VHM_INFO_STALENESS_COUNTER = 2, SP_VHM_INFO_HAS_COV = 3,
       SP_VHM_INFO_HAS_RGB = 4, SP_VHM_INFO_HAS_INTENSITY = 5 };
int sp_vhm_create(float voxel_size, void* stream, sp_voxel_hash_map** out);
void sp_vhm_destroy(sp_voxel_hash_map* map);
int sp_vhm_set(sp_voxel_hash_map* map, int param, float value);
float sp_vhm_get(const sp_voxel_hash_map* map, int param);
size_t sp_vhm_info(const sp_voxel_hash_map* map, int what);
int sp_vhm_clear(sp_voxel_hash_map* map, void* stream);
int sp_vhm_add_point_cloud(sp_voxel_hash_map* map, const float* points, const float* covs, const float* rgb,
                           const float* intensities, size_t n, const float* sensor_pose_host16, void* stream);
int sp_vhm_downsampling(sp_voxel_hash_map* map, const float* center_host3, float distance, float* points_out,
                        float* covs_out, float* rgb_out, float* intensities_out, uint64_t* keys_out_opt,
                        size_t out_capacity, size_t* n_out_host, void* stream);
int sp_vhm_overlap_ratio(const sp_voxel_hash_map* map, const float* points, size_t n, const float* sensor_pose_host16,
                         float* ratio_out_host, void* stream);
int sp_vhm_remove_old_data(sp_voxel_hash_map* map, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SYCL_POINTS_AMD_H */

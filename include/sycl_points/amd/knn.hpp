// sycl_points facade for MI355X — KNN layer.
//   algorithms/knn/result.hpp     : KNNResult
//   algorithms/knn/knn.hpp        : KNNBase (the operator boundary Registration::align sits on)
//   algorithms/knn/bruteforce.hpp : knn_search_bruteforce
//   algorithms/knn/kdtree.hpp     : KDTree (host build with the reference's split rule, device search)
//   + GridKNN: an MI355X-native KNNBase (device-built uniform grid), no counterpart file in the reference.
#pragma once
#include <atomic>
#include "core.hpp"

namespace sycl_points {
namespace algorithms {

namespace filter {
constexpr uint8_t REMOVE_FLAG = 0;   // common/filter_by_flags.hpp:11-12
constexpr uint8_t INCLUDE_FLAG = 1;
}  // namespace filter

namespace knn {

/// algorithms/knn/result.hpp:12-34
struct KNNResult {
    using Ptr = std::shared_ptr<KNNResult>;
    shared_vector_ptr<int32_t> indices = nullptr;
    shared_vector_ptr<float> distances = nullptr;
    size_t query_size = 0;
    size_t k = 0;

    /// result.hpp:20-27: query_size x k entries of -1 / FLT_MAX. They are put there by a device fill, not by a host vector
    /// (shared_vector::resize_on_device); `fill = false` when a search that writes every entry follows at once.
    void allocate(const sycl_utils::DeviceQueue& queue, size_t query_size_ = 0, size_t k_ = 0, bool fill = true) {
        query_size = query_size_;
        k = k_;
        indices = std::make_shared<shared_vector<int32_t>>(queue);
        distances = std::make_shared<shared_vector<float>>(queue);
        const uint32_t minus_one = 0xffffffffu, flt_max = 0x7f7fffffu;
        indices->resize_on_device(query_size * k, fill ? &minus_one : nullptr);
        distances->resize_on_device(query_size * k, fill ? &flt_max : nullptr);
    }
    void resize(size_t query_size_ = 0, size_t k_ = 0) {
        query_size = query_size_;
        k = k_;
        indices->resize(query_size * k);
        distances->resize(query_size * k);
    }
};

/// algorithms/knn/knn.hpp:14-61 — same virtual interface; `depends` is kept for source compatibility (all work is
/// enqueued in order on the cloud's stream, so dependencies are implicit).
class KNNBase {
public:
    virtual ~KNNBase() = default;
    virtual sycl_utils::events knn_search_async(const PointCloudShared& queries, const size_t k, KNNResult& result,
                                                const std::vector<sycl_utils::event>& depends = {},
                                                const TransformMatrix& transT = TransformMatrix::Identity()) const = 0;

    KNNResult knn_search(const PointCloudShared& queries, const size_t k, const std::vector<sycl_utils::event>& depends = {},
                         const TransformMatrix& transT = TransformMatrix::Identity()) const {
        KNNResult result;
        knn_search_async(queries, k, result, depends, transT).wait_and_throw();
        return result;
    }
    sycl_utils::events nearest_neighbor_search_async(const PointCloudShared& queries, KNNResult& result,
                                                     const std::vector<sycl_utils::event>& depends = {},
                                                     const TransformMatrix& transT = TransformMatrix::Identity()) const {
        return knn_search_async(queries, 1, result, depends, transT);
    }
    void nearest_neighbor_search(const PointCloudShared& queries, KNNResult& result,
                                 const std::vector<sycl_utils::event>& depends = {},
                                 const TransformMatrix& transT = TransformMatrix::Identity()) const {
        nearest_neighbor_search_async(queries, result, depends, transT).wait_and_throw();
    }
};

namespace detail {
inline void prepare_result(const sycl_utils::DeviceQueue& q, KNNResult& r, size_t nq, size_t k) {
    // kdtree.hpp:446-450; every search kernel writes all nq x k entries (padding included): no fill, no host storage
    if (r.indices == nullptr || r.distances == nullptr) {
        r.allocate(q, nq, k, false);
    } else {
        r.query_size = nq;
        r.k = k;
        r.indices->resize_on_device(nq * k);
        r.distances->resize_on_device(nq * k);
    }
}
}  // namespace detail

/// algorithms/knn/bruteforce.hpp:24-96 (synchronous, like the reference)
inline KNNResult knn_search_bruteforce(const sycl_utils::DeviceQueue& queue, const PointCloudShared& queries,
                                       const PointCloudShared& targets, const size_t k) {
    const size_t nq = queries.size(), nt = targets.size();
    KNNResult result;
    result.allocate(queue, nq, k);
    if (nq == 0) return result;
    const size_t ws_bytes = sp_knn_bruteforce_workspace_bytes(nq, nt, k);
    void* ws = nullptr;
    size_t ws_got = 0;
    if (ws_bytes) ws = sycl_points::detail::DeviceBufferCache::acquire(ws_bytes, &ws_got, queue.stream());  // (no hipMalloc / hipFree per call)
    const int rc = sp_knn_bruteforce(queries.points_device(), nq, targets.points_device(), nt, k,
                                     result.indices->device_data_for_write(nq * k),
                                     result.distances->device_data_for_write(nq * k), ws, ws_bytes, queue.stream());
    if (rc == SP_OK) queue.wait();
    if (ws) sycl_points::detail::DeviceBufferCache::release(ws, ws_got, queue.stream(), rc == SP_OK);
    throw_on_error(rc);
    return result;
}

/// MI355X-native KNNBase for clouds of any density profile: a bounding-volume hierarchy over the Morton-sorted points, built
/// entirely on the device (sp_bvh_*, csrc/bvh.hip). Exact kNN, k <= 32, bit-identical to knn_search_bruteforce.
class BVH : public KNNBase {
public:
    using Ptr = std::shared_ptr<BVH>;
    sycl_utils::DeviceQueue queue;

    explicit BVH(const sycl_utils::DeviceQueue& q) : queue(q) {}
    ~BVH() override { if (bvh_) sp_bvh_destroy(bvh_); }
    BVH(const BVH&) = delete;
    BVH& operator=(const BVH&) = delete;

    static Ptr build(const sycl_utils::DeviceQueue& q, const PointContainerShared& points) {
        auto t = std::make_shared<BVH>(q);
        throw_on_error(sp_bvh_create(reinterpret_cast<const float*>(points.device_data()), points.size(), q.stream(), &t->bvh_));
        return t;
    }
    static Ptr build(const sycl_utils::DeviceQueue& q, const PointCloudShared& cloud) { return build(q, *cloud.points); }
    const sp_bvh* handle() const { return bvh_; }
    size_t size() const { return sp_bvh_size(bvh_); }

    sycl_utils::events knn_search_async(const PointCloudShared& queries, const size_t k, KNNResult& result,
                                        const std::vector<sycl_utils::event>& = {},
                                        const TransformMatrix& transT = TransformMatrix::Identity()) const override {
        const size_t nq = queries.size();
        if (k > 32) throw std::runtime_error("[BVH::knn_search_async] `k` is too large (max 32).");
        detail::prepare_result(queue, result, nq, nq ? k : 0);
        if (nq == 0) return sycl_utils::events();
        throw_on_error(sp_bvh_search(bvh_, queries.points_device(), nq, k, transT.data(), 0,
                                     result.indices->device_data_for_write(nq * k),
                                     result.distances->device_data_for_write(nq * k), queue.stream()));
        return sycl_utils::events(queue.stream());
    }
    /// The cloud's own points as queries, walked in tree order (row i = neighbours of point i, itself first).
    KNNResult self_knn(const size_t k) const {
        const size_t n = size();
        if (k > 32) throw std::runtime_error("[BVH::self_knn] `k` is too large (max 32).");
        KNNResult result;
        result.allocate(queue, n, n ? k : 0);
        if (n == 0) return result;
        throw_on_error(sp_bvh_self_knn(bvh_, k, result.indices->device_data_for_write(n * k),
                                       result.distances->device_data_for_write(n * k), queue.stream()));
        queue.wait();
        return result;
    }

private:
    sp_bvh* bvh_ = nullptr;
};

/// algorithms/knn/kdtree.hpp:142-766.
///
/// The reference builds its tree on the host (recursive nth_element: 30 ms per 1M points here, on 16 threads) and its callers
/// rebuild it every frame (pipeline/submapping.hpp:197, pipeline/pointcloud_processing.hpp:64). build() here builds the
/// device's own hierarchy instead (BVH above: 0.2-0.5 ms) and knn_search answers from it — the same exact neighbours; only
/// the order inside a group of exactly equal distances differs (lowest index first, the brute-force rule, instead of the
/// first visited). The reference-topology tree is built the first time something needs it: radius search, lazy delete, k > 32,
/// or set_reference_tie_order(true).
class KDTree : public KNNBase {
public:
    using Ptr = std::shared_ptr<KDTree>;
    sycl_utils::DeviceQueue queue;

    explicit KDTree(const sycl_utils::DeviceQueue& q) : queue(q) {}
    ~KDTree() override {
        if (tree_) sp_kdtree_destroy(tree_);
        if (bvh_) sp_bvh_destroy(bvh_);
        if (self_grid_) sp_grid_destroy(self_grid_);
        if (self_grid_ws_) sycl_points::detail::DeviceBufferCache::release(self_grid_ws_, self_grid_ws_bytes_, queue.stream());
        if (dev_points_) sycl_points::detail::DeviceBufferCache::release(dev_points_, dev_points_bytes_, queue.stream());
    }
    KDTree(const KDTree&) = delete;
    KDTree& operator=(const KDTree&) = delete;

    static Ptr build(const sycl_utils::DeviceQueue& q, const PointContainerShared& points, size_t leaf_threshold = 16) {
        auto t = std::make_shared<KDTree>(q);
        // (a cloud of a few hundred points: the host build is microseconds and the reference's balanced tree is the shallower one.
        // From a few thousand points on the device build wins even where the search is a little slower — the reference's
        // example on its 6 k-point downsampled scans, same box: build 1.40 -> 0.33 ms, search 0.51 -> 0.78 ms per loop.)
        if (points.size() < kDeviceBuildMinPoints) {
            throw_on_error(sp_kdtree_create(reinterpret_cast<const float*>(points.data()), points.size(), leaf_threshold, q.stream(),
                                            &t->tree_));
        } else {
            // The tree keeps its own copy of the points (like the nodes of the reference's tree: the source cloud may be gone or
            // changed by the time it is searched) — 16 MB per million points, a few microseconds on the device — and builds the
            // hierarchy from it when something first needs it. The commonest life of a tree — searched with its own cloud for the
            // covariances (a grid answers that on a near-uniform cloud), then handed to Registration::align (which searches on a
            // grid of its own) — never does: half a millisecond per million points and tree saved.
            t->hierarchy_ = true;
            size_t got = 0;
            t->dev_points_ = sycl_points::detail::DeviceBufferCache::acquire(points.size() * 16, &got, q.stream());
            t->dev_points_bytes_ = got;
            hip_check(hipMemcpyAsync(t->dev_points_, points.device_data(), points.size() * 16, hipMemcpyDeviceToDevice, q.stream()), "D2D");
        }
        static std::atomic<uint64_t> next_id{1};
        t->id_ = next_id.fetch_add(1);
        t->size_ = points.size();
        t->leaf_threshold_ = leaf_threshold;
        return t;
    }
    static Ptr build(const sycl_utils::DeviceQueue& q, const PointCloudShared& cloud, size_t leaf_threshold = 16) {
        auto t = build(q, *cloud.points, leaf_threshold);
        t->built_on_ = cloud.points;  // (shared ownership: the address cannot be handed to another container meanwhile)
        t->built_generation_ = cloud.points->generation();
        return t;
    }
    /// MI355X extension: answer knn_search from the reference-topology tree (first-visited tie order, host build) always.
    void set_reference_tie_order(bool v) { reference_order_ = v; }
    /// MI355X extension: which structure knn_search_async(queries, k, ..., transT) answers from — the decision itself, without
    /// searching (the Python mirror sycl_points_amd.api.KDTree takes the same one; tests hold the two to each other).
    enum class Backend { HostTree, Hierarchy, Grid, BruteForce };
    Backend backend_for(const PointCloudShared& queries, size_t k, const TransformMatrix& transT = TransformMatrix::Identity()) const {
        if (!(on_hierarchy() && k <= 32)) return Backend::HostTree;
        const bool own_cloud = built_on_ != nullptr && queries.points == built_on_ && queries.points->generation() == built_generation_ &&
                               queries.size() == size_ && transT == TransformMatrix::Identity();
        if (own_cloud && k >= 8 && k <= 20 && uniform_grid() != nullptr) return Backend::Grid;
        // A small cloud — the reference example searches its 6 k-point downsampled scans with k = 10 — is answered by the exact
        // brute-force search (sp_knn_bruteforce: bounding pass, then the reference's expression where a neighbour can be) in
        // tens of microseconds; building the hierarchy alone takes 0.17 ms whatever the size, its k = 10 search 0.2 ms. Same
        // lists: both break distance ties by the lowest index.
        // (up to 12 k targets and 8 * 10^7 pairs sp_knn_bruteforce is ONE launch with the cloud in LDS: 30 us for 6 k x 6 k)
        const bool small = size_ <= 12032 && queries.size() * size_ <= size_t(80) * 1000 * 1000;
        if (pristine_ && k <= 20 && transT == TransformMatrix::Identity() && queries.size() <= kBruteForceMaxQueries &&
            (small || (size_ >= 2048 && size_ <= kBruteForceMaxTargets && size_ >= 256 * k)))
            return Backend::BruteForce;
        return Backend::Hierarchy;
    }

    sycl_utils::events knn_search_async(const PointCloudShared& queries, const size_t k, KNNResult& result,
                                        const std::vector<sycl_utils::event>& = {},
                                        const TransformMatrix& transT = TransformMatrix::Identity()) const override {
        const size_t nq = queries.size();
        if (k > 100) throw std::runtime_error("[KDTree::knn_search_async] `k` is too large. not support.");
        detail::prepare_result(queue, result, nq, nq ? k : 0);
        if (nq == 0) return sycl_utils::events();
        if (backend_for(queries, k, transT) == Backend::BruteForce) {
            const size_t ws_bytes = sp_knn_bruteforce_workspace_bytes(nq, size_, k);
            void* ws = nullptr;
            size_t ws_got = 0;
            if (ws_bytes) ws = sycl_points::detail::DeviceBufferCache::acquire(ws_bytes, &ws_got, queue.stream());
            const int rc = sp_knn_bruteforce(queries.points_device(), nq, device_points(), size_, k,
                                             result.indices->device_data_for_write(nq * k),
                                             result.distances->device_data_for_write(nq * k), ws, ws_bytes, queue.stream());
            // (back to the cache tagged with this stream: the next user waits for the search's event, nobody for the device)
            if (ws) sycl_points::detail::DeviceBufferCache::release(ws, ws_got, queue.stream());
            throw_on_error(rc);
            return sycl_utils::events(queue.stream());
        }
        if (on_hierarchy() && k <= 32) {
            // the tree's own cloud, untouched since build() and searched in place (the covariance pre-step of every pipeline):
            // its points are walked in tree order, neighbouring lanes share their path (1.5x faster than in query order)
            if (built_on_ != nullptr && queries.points == built_on_ && queries.points->generation() == built_generation_ && nq == size_ &&
                transT == TransformMatrix::Identity()) {
                // A large cloud of near-uniform density (a voxel-downsampled scan, a submap): the uniform grid's lane-per-query
                // selection answers 8 <= k <= 20 five to nine times faster than the hierarchy (1 M points, k = 20: 0.6 against
                // 5.5 ms) with the same lists. Built once per tree (0.2 ms), kept only if no cell is overfull.
                if (k >= 8 && k <= 20 && uniform_grid() != nullptr) {
                    throw_on_error(sp_grid_self_knn(self_grid_, k, result.indices->device_data_for_write(nq * k),
                                                    result.distances->device_data_for_write(nq * k), nullptr, nullptr, self_grid_ws_,
                                                    sp_grid_self_workspace_bytes(self_grid_), queue.stream()));
                    return sycl_utils::events(queue.stream());
                }
                throw_on_error(sp_bvh_self_knn(hierarchy(), k, result.indices->device_data_for_write(nq * k),
                                               result.distances->device_data_for_write(nq * k), queue.stream()));
                return sycl_utils::events(queue.stream());
            }
            throw_on_error(sp_bvh_search(hierarchy(), queries.points_device(), nq, k, transT.data(), 0,
                                         result.indices->device_data_for_write(nq * k),
                                         result.distances->device_data_for_write(nq * k), queue.stream()));
            return sycl_utils::events(queue.stream());
        }
        throw_on_error(sp_kdtree_search(host_tree(), queries.points_device(), nq, k, transT.data(), 0,
                                        result.indices->device_data_for_write(nq * k),
                                        result.distances->device_data_for_write(nq * k), queue.stream()));
        return sycl_utils::events(queue.stream());
    }
    sycl_utils::events radius_search_async(const PointCloudShared& queries, const size_t max_k, const float radius,
                                           KNNResult& result, const std::vector<sycl_utils::event>& = {},
                                           const TransformMatrix& transT = TransformMatrix::Identity()) const {
        const size_t nq = queries.size();
        if (max_k > 100) throw std::runtime_error("[KDTree::radius_search_async] `max_k` is too large. not support.");
        if (nq == 0 || max_k == 0) {
            detail::prepare_result(queue, result, 0, 0);
            return sycl_utils::events();
        }
        detail::prepare_result(queue, result, nq, max_k);
        if (on_hierarchy() && max_k <= 32) {
            throw_on_error(sp_bvh_radius_search(hierarchy(), queries.points_device(), nq, max_k, radius, transT.data(), 0,
                                                result.indices->device_data_for_write(nq * max_k),
                                                result.distances->device_data_for_write(nq * max_k), queue.stream()));
            return sycl_utils::events(queue.stream());
        }
        throw_on_error(sp_kdtree_radius_search(host_tree(), queries.points_device(), nq, max_k, radius, transT.data(), 0,
                                               result.indices->device_data_for_write(nq * max_k),
                                               result.distances->device_data_for_write(nq * max_k), queue.stream()));
        return sycl_utils::events(queue.stream());
    }
    void remove_nodes_by_flags(const shared_vector<uint8_t>& flags, const shared_vector<int32_t>& indices) {
        if (flags.size() != indices.size())
            throw std::runtime_error("[KDTree::remove_nodes_by_flags_impl] flags and indices must have the same size.");
        // lazy delete in whichever structures exist (both must agree from now on); the grid on the tree's own cloud is dropped,
        // and so is the shortcut for searches of that cloud (its points carry other indices now)
        if (hierarchy_)
            throw_on_error(sp_bvh_remove_by_flags(hierarchy(), flags.device_data(), indices.device_data(), flags.size(), queue.stream()));
        if (tree_ != nullptr || !hierarchy_)
            throw_on_error(sp_kdtree_remove_by_flags(host_tree(), flags.device_data(), indices.device_data(), flags.size(), queue.stream()));
        queue.wait();
        pristine_ = false;
        // only the hierarchy exists: a reference-topology tree built later (k > 32, set_reference_tie_order) starts from the
        // ORIGINAL points, so it has to see the same removals in the same order — kept here, replayed by host_tree()
        if (tree_ == nullptr) {
            Removal r;
            r.flags.assign(flags.host().begin(), flags.host().end());
            r.indices.assign(indices.host().begin(), indices.host().end());
            removals_.push_back(std::move(r));
        }
        if (self_grid_) { sp_grid_destroy(self_grid_); self_grid_ = nullptr; }
        self_grid_tried_ = true;
        built_on_ = nullptr;
    }
    /// Identity of the built tree (unique per build), its point count, and whether no node was ever removed — what
    /// Registration::align needs to decide that a GridKNN on the same cloud answers the same nearest-neighbour queries.
    uint64_t id() const { return id_; }
    size_t size() const { return size_; }
    bool pristine() const { return pristine_; }

private:
    static constexpr size_t kDeviceBuildMinPoints = 1024;
    static constexpr size_t kBruteForceMaxTargets = 16384, kBruteForceMaxQueries = 65536;
    /// The device-built hierarchy answers: kNN (k <= 32), radius search and — since round 4 — after a lazy delete too
    /// (sp_bvh_radius_search / sp_bvh_remove_by_flags); the reference's tree only for its own tie order and k > 32.
    bool on_hierarchy() const { return hierarchy_ && !reference_order_; }
    /// The device-built hierarchy, built on first use from the tree's copy of the points.
    sp_bvh* hierarchy() const {
        if (bvh_ == nullptr) throw_on_error(sp_bvh_create(device_points(), size_, queue.stream(), &bvh_));
        return bvh_;
    }
    /// The points the tree was built on, in their original order, on the device (the tree's own copy, taken at build()).
    const float* device_points() const { return static_cast<const float*>(dev_points_); }
    /// The reference's tree (host build with its rule, kdtree.hpp:292-413).
    sp_kdtree* host_tree() const {
        if (tree_ == nullptr) {
            std::vector<float> host(4 * std::max<size_t>(size_, 1));
            if (size_) {
                hip_check(hipMemcpyAsync(host.data(), device_points(), size_ * 16, hipMemcpyDeviceToHost, queue.stream()), "D2H");
                hip_check(hipStreamSynchronize(queue.stream()), "sync");
            }
            throw_on_error(sp_kdtree_create(host.data(), size_, leaf_threshold_, queue.stream(), &tree_));
            // nodes removed while only the hierarchy existed: the same lazy deletes, in their order (kdtree.hpp:721-765)
            for (const Removal& r : removals_) {
                const size_t n = r.flags.size();
                size_t got_f = 0, got_i = 0;
                hipStream_t st = queue.stream();
                void* df = sycl_points::detail::DeviceBufferCache::acquire(std::max<size_t>(n, 1), &got_f, st);
                void* di = sycl_points::detail::DeviceBufferCache::acquire(std::max<size_t>(n, 1) * 4, &got_i, st);
                hipError_t e = hipMemcpyAsync(df, r.flags.data(), n, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipMemcpyAsync(di, r.indices.data(), n * 4, hipMemcpyHostToDevice, st);
                int rc = SP_OK;
                if (e == hipSuccess)
                    rc = sp_kdtree_remove_by_flags(tree_, static_cast<const uint8_t*>(df), static_cast<const int32_t*>(di), n, st);
                (void)hipStreamSynchronize(st);
                sycl_points::detail::DeviceBufferCache::release(df, got_f, st, true);
                sycl_points::detail::DeviceBufferCache::release(di, got_i, st, true);
                hip_check(e, "H2D");
                throw_on_error(rc);
            }
            removals_.clear();
            removals_.shrink_to_fit();
        }
        return tree_;
    }
    /// A grid on the tree's own cloud for the self-kNN of large clouds, when its density allows (see knn_search_async).
    sp_grid* uniform_grid() const {
        if (!self_grid_tried_) {
            self_grid_tried_ = true;
            if (size_ >= kGridSelfMinPoints) {
                constexpr float kPointsPerCell = 6.0f;
                throw_on_error(sp_grid_create(device_points(), size_, 0.0f, kPointsPerCell, queue.stream(), &self_grid_));
                if (sp_grid_max_cell_points(self_grid_) > kGridSelfMaxCell) {  // surfaces, clusters: the hierarchy's case
                    sp_grid_destroy(self_grid_);
                    self_grid_ = nullptr;
                } else {
                    // (from the facade's buffer cache: hipMalloc / hipFree cost 0.1-0.2 ms apiece, per tree and frame)
                    self_grid_ws_ = sycl_points::detail::DeviceBufferCache::acquire(sp_grid_self_workspace_bytes(self_grid_), &self_grid_ws_bytes_, queue.stream());
                }
            }
        }
        return self_grid_;
    }
    static constexpr size_t kGridSelfMinPoints = 32768;
    static constexpr uint32_t kGridSelfMaxCell = 48;
    mutable sp_grid* self_grid_ = nullptr;
    mutable void* self_grid_ws_ = nullptr;
    mutable size_t self_grid_ws_bytes_ = 0;
    mutable bool self_grid_tried_ = false;
    mutable sp_kdtree* tree_ = nullptr;
    mutable sp_bvh* bvh_ = nullptr;   // built by hierarchy()
    bool hierarchy_ = false;          // the tree answers from the device-built hierarchy (>= kDeviceBuildMinPoints points)
    void* dev_points_ = nullptr;      // its copy of the points (hierarchy_ only)
    size_t dev_points_bytes_ = 0;
    uint64_t id_ = 0;
    size_t size_ = 0, leaf_threshold_ = 16;
    bool pristine_ = true, reference_order_ = false;
    struct Removal { std::vector<uint8_t> flags; std::vector<int32_t> indices; };
    mutable std::vector<Removal> removals_;  // lazy deletes the reference-topology tree has not seen yet (it does not exist)
    std::shared_ptr<PointContainerShared> built_on_;  // the cloud's point container at build(), and its generation then
    uint64_t built_generation_ = 0;
};

/// MI355X-native KNNBase: exact kNN on a device-built uniform grid (sp_grid_*). Bit-identical to
/// knn_search_bruteforce. Registration::align recognises it and takes the fused NN + linearise path.
class GridKNN : public KNNBase {
public:
    using Ptr = std::shared_ptr<GridKNN>;
    sycl_utils::DeviceQueue queue;

    explicit GridKNN(const sycl_utils::DeviceQueue& q) : queue(q) {}
    ~GridKNN() override { if (grid_) sp_grid_destroy(grid_); }
    GridKNN(const GridKNN&) = delete;
    GridKNN& operator=(const GridKNN&) = delete;

    static Ptr build(const sycl_utils::DeviceQueue& q, const PointCloudShared& cloud, float points_per_cell = 0.5f,
                     float cell_size = 0.0f) {
        auto g = std::make_shared<GridKNN>(q);
        // (a cloud whose producer left a bounding box behind — voxel downsampling does: the build then needs no box of its own,
        // i.e. no kernel, no read-back and no wait before it can size its cell table)
        float box[6];
        const float* const pts = cloud.points_device();  // (before the look-up: an upload does not change the generation)
        if (sycl_points::detail::BoundsHints::get(cloud.points->generation(), box))
            throw_on_error(sp_grid_create_bounded(pts, cloud.size(), box, cell_size, points_per_cell, q.stream(), &g->grid_));
        else
            throw_on_error(sp_grid_create(pts, cloud.size(), cell_size, points_per_cell, q.stream(), &g->grid_));
        g->id_ = next_grid_id();
        return g;
    }
    static uint64_t next_grid_id() {
        static std::atomic<uint64_t> next_id{1};
        return next_id.fetch_add(1);
    }
    /// Cell size steered by the measured occupancy instead of the bounding-box volume (sp_grid_create_adaptive): for clouds of
    /// surfaces — what Registration::align builds when it stands in for the caller's KDTree.
    static Ptr build_adaptive(const sycl_utils::DeviceQueue& q, const PointCloudShared& cloud, float points_per_cell = 0.5f) {
        auto g = std::make_shared<GridKNN>(q);
        throw_on_error(sp_grid_create_adaptive(cloud.points_device(), cloud.size(), points_per_cell, q.stream(), &g->grid_));
        g->id_ = next_grid_id();
        return g;
    }
    const sp_grid* handle() const { return grid_; }
    uint64_t id() const { return id_; }  ///< unique per build (a handle address can be reused after destruction)
    size_t size() const { return sp_grid_size(grid_); }
    float cell_size() const { return sp_grid_cell_size(grid_); }

    sycl_utils::events knn_search_async(const PointCloudShared& queries, const size_t k, KNNResult& result,
                                        const std::vector<sycl_utils::event>& = {},
                                        const TransformMatrix& transT = TransformMatrix::Identity()) const override {
        const size_t nq = queries.size();
        if (k > 20) throw std::runtime_error("[GridKNN::knn_search_async] `k` is too large (max 20).");
        detail::prepare_result(queue, result, nq, nq ? k : 0);
        if (nq == 0) return sycl_utils::events();
        throw_on_error(sp_grid_search(grid_, queries.points_device(), nq, k, transT.data(), 0,
                                      result.indices->device_data_for_write(nq * k),
                                      result.distances->device_data_for_write(nq * k), queue.stream()));
        return sycl_utils::events(queue.stream());
    }

    /// The grid's counterpart of KDTree::radius_search_async (kdtree.hpp:251-280): the max_k nearest within `radius`.
    sycl_utils::events radius_search_async(const PointCloudShared& queries, const size_t max_k, const float radius,
                                           KNNResult& result, const std::vector<sycl_utils::event>& = {},
                                           const TransformMatrix& transT = TransformMatrix::Identity()) const {
        const size_t nq = queries.size();
        if (max_k > 20) throw std::runtime_error("[GridKNN::radius_search_async] `max_k` is too large (max 20).");
        detail::prepare_result(queue, result, (nq && max_k) ? nq : 0, (nq && max_k) ? max_k : 0);
        if (nq == 0 || max_k == 0) return sycl_utils::events();
        throw_on_error(sp_grid_radius_search(grid_, queries.points_device(), nq, max_k, radius, transT.data(), 0,
                                             result.indices->device_data_for_write(nq * max_k),
                                             result.distances->device_data_for_write(nq * max_k), queue.stream()));
        return sycl_utils::events(queue.stream());
    }

    /// The grid's counterpart of KDTree::remove_nodes_by_flags (kdtree.hpp:282-284): flags 1 = keep, kept point p is
    /// relabelled indices[p]. The grid gets a new identity (Registration re-prepares its target on the next align()).
    void remove_nodes_by_flags(const shared_vector<uint8_t>& flags, const shared_vector<int32_t>& indices) {
        if (flags.size() != indices.size())
            throw std::runtime_error("[GridKNN::remove_nodes_by_flags] flags and indices must have the same size.");
        throw_on_error(sp_grid_remove_by_flags(grid_, flags.device_data(), indices.device_data(), flags.size(), queue.stream()));
        static std::atomic<uint64_t> next_removed_id{1ull << 40};
        id_ = next_removed_id.fetch_add(1);
    }

private:
    sp_grid* grid_ = nullptr;
    uint64_t id_ = 0;
};

}  // namespace knn
}  // namespace algorithms
}  // namespace sycl_points

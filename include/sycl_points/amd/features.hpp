// sycl_points facade for MI355X — covariance / normals, voxel grid, pre-filters, transform.
//   algorithms/feature/covariance.hpp            : covariance::estimate_async, estimate_normals_async, extract_normals(_async)
//   algorithms/filter/voxel_downsampling.hpp     : filter::VoxelGrid
//   algorithms/filter/preprocess_filter.hpp      : filter::PreprocessFilter (box_filter, random_sampling)
//   algorithms/common/filter_by_flags.hpp        : filter::FilterByFlags
//   algorithms/common/transform.hpp              : transform::transform, transform_copy
#pragma once
#include <exception>
#include <mutex>
#include <utility>
#include <vector>
#include <cctype>
#include <numeric>
#include <random>

#include "knn.hpp"

namespace sycl_points {
namespace algorithms {

namespace detail {
/// Scratch device memory that lives for one call (the reference allocates shared_vectors per call the same way,
/// e.g. registration.hpp:685-686).
// Device scratch of one call (workspaces, counters), from the cache the containers use (core.hpp: keyed by device, no
// hipMalloc / hipFree per call). Every user synchronises its stream before the scratch leaves scope, so the buffer goes
// back idle; when the scope is left by an exception that may not have happened yet, and the device is drained first.
struct DeviceScratch {
    void* p = nullptr;
    size_t bytes = 0;
    /// st: the stream the scratch will be used on (a buffer last used on the same stream is taken without waiting)
    explicit DeviceScratch(size_t n, hipStream_t st = nullptr) : st_(st) {
        if (n) p = ::sycl_points::detail::DeviceBufferCache::acquire(n, &bytes, st);
    }
    ~DeviceScratch() {
        if (!p) return;
        if (std::uncaught_exceptions() > 0) (void)hipDeviceSynchronize();
        if (stream_ordered) ::sycl_points::detail::DeviceBufferCache::release(p, bytes, st_, /*idle=*/false);
        else ::sycl_points::detail::DeviceBufferCache::release(p, bytes, nullptr, /*idle=*/true);
    }
    /// set by a user that did NOT synchronise: everything that touched the scratch was enqueued on the stream it was taken for
    bool stream_ordered = false;
    hipStream_t st_ = nullptr;
    DeviceScratch(const DeviceScratch&) = delete;
    DeviceScratch& operator=(const DeviceScratch&) = delete;
};
// One 4-byte read-back through pinned memory (a copy into pageable memory is staged and blocks inside the runtime).
inline void* pinned_word() {
    static void* p = [] { void* q = nullptr; hip_check(hipHostMalloc(&q, 64), "hipHostMalloc"); return q; }();
    return p;
}
/// 4 KB of pinned host memory per host thread for read-backs of a few hundred bytes (nullptr: none to be had)
inline void* pinned_block_4k() {
    thread_local void* p = [] {
        void* q = nullptr;
        if (hipHostMalloc(&q, 4096, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); q = nullptr; }
        return q;
    }();
    return p;
}
/// A count a kernel hands to the host WITHOUT a copy and a stream synchronisation: the kernel stores it through the device
/// pointer of a host-mapped pinned word, the host spins on the word (armed with a value no count takes). A D2H copy + a
/// hipStreamSynchronize is 15-20 us of runtime calls and wake-up for four bytes; the spin sees the store a microsecond or two
/// after it lands. Later work on the stream stays ordered behind the kernel as ever; the kernel's OTHER outputs may still be in
/// flight when the count arrives (they stay on the device). One word per host thread.
struct MappedWord {
    volatile uint32_t* host = nullptr;
    uint32_t* dev = nullptr;
    static constexpr uint32_t kArmed = 0xffffffffu;
    static MappedWord& mine() {
        thread_local MappedWord w = [] {
            MappedWord m;
            void* h = nullptr;
            if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return m; }  // (every device of the process may store to it)
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return m; }
            m.host = static_cast<volatile uint32_t*>(h);
            m.dev = static_cast<uint32_t*>(d);
            return m;
        }();
        return w;
    }
    bool usable() const { return host != nullptr; }
    void arm() const { *host = kArmed; }
    /// the value once the kernel has stored it; a stream error (the kernel never ran) ends the wait through the synchronisation
    uint32_t wait(hipStream_t st) const {
        for (unsigned spins = 0;; ++spins) {
            const uint32_t v = *host;
            if (v != kArmed) return v;
            if (spins > (1u << 22)) {  // (tens of milliseconds: something is wrong or the device is very busy — fall back to waiting)
                hip_check(hipStreamSynchronize(st), "sync");
                return *host;
            }
        }
    }
};
inline uint32_t read_u32(const void* dev, hipStream_t st) {
    static std::mutex m;  // one pinned word for the process: read-backs are short and rare
    std::lock_guard<std::mutex> lock(m);
    uint32_t* const v = static_cast<uint32_t*>(pinned_word());
    hip_check(hipMemcpyAsync(v, dev, 4, hipMemcpyDeviceToHost, st), "D2H");
    hip_check(hipStreamSynchronize(st), "sync");
    return *v;
}
}  // namespace detail

// ================================================================================================ robust loss tags
namespace robust {
enum class RobustLossType { NONE, HUBER, TUKEY, CAUCHY, GEMAN_MCCLURE };  // robust/robust.hpp:13-19
inline RobustLossType RobustLossType_from_string(const std::string& str) {
    std::string u = str;
    for (auto& c : u) c = (char)std::toupper((unsigned char)c);
    if (u == "NONE") return RobustLossType::NONE;
    if (u == "HUBER") return RobustLossType::HUBER;
    if (u == "TUKEY") return RobustLossType::TUKEY;
    if (u == "CAUCHY") return RobustLossType::CAUCHY;
    if (u == "GEMAN_MCCLURE") return RobustLossType::GEMAN_MCCLURE;
    throw std::runtime_error("[RobustLossType_from_string] Invalid RobustLossType str '" + str + "'");
}
}  // namespace robust

// ================================================================================================ covariance
namespace covariance {

/// covariance.hpp:260-295
inline sycl_utils::events estimate_async(const sycl_utils::DeviceQueue& queue, const knn::KNNResult& neighbors,
                                         const PointContainerShared& points, CovarianceContainerShared& covs,
                                         const std::vector<sycl_utils::event>& = {}) {
    const size_t N = points.size();
    if (N == 0) { covs.resize(0); return sycl_utils::events(); }
    throw_on_error(sp_cov_estimate(reinterpret_cast<const float*>(points.device_data()), N, neighbors.indices->device_data(),
                                   neighbors.k, reinterpret_cast<float*>(covs.device_data_for_write(N)), queue.stream()));
    return sycl_utils::events(queue.stream());
}
/// covariance.hpp:297-303
inline sycl_utils::events estimate_async(const knn::KNNResult& neighbors, const PointCloudShared& points,
                                         const std::vector<sycl_utils::event>& depends = {}) {
    return estimate_async(points.queue, neighbors, *points.points, *points.covs, depends);
}
/// covariance.hpp:305-311 — search + estimate. With a GridKNN built on `points` the neighbour lists are never written:
/// the self-kNN kernel accumulates the covariance directly (sp_grid_self_knn).
inline sycl_utils::events estimate_async(const knn::KNNBase& knn, const PointCloudShared& points, const size_t k,
                                         const std::vector<sycl_utils::event>& depends = {}) {
    if (const auto* grid = dynamic_cast<const knn::GridKNN*>(&knn)) {
        if (grid->size() == points.size() && k <= 20 && points.size() > 0) {
            detail::DeviceScratch ws(sp_grid_self_workspace_bytes(grid->handle()));
            throw_on_error(sp_grid_self_knn(grid->handle(), k, nullptr, nullptr,
                                            reinterpret_cast<float*>(points.covs->device_data_for_write(points.size())),
                                            nullptr, ws.p, sp_grid_self_workspace_bytes(grid->handle()), points.queue.stream()));
            points.queue.wait();  // the scratch dies with this scope
            return sycl_utils::events(points.queue.stream());
        }
    }
    knn::KNNResult neighbors;
    auto ev = knn.knn_search_async(points, k, neighbors, depends);
    return estimate_async(neighbors, points, ev.evs);
}
/// covariance.hpp:323-381 — M-estimated covariances (K8)
inline sycl_utils::events estimate_robust_async(const sycl_utils::DeviceQueue& queue, const knn::KNNResult& neighbors,
                                                const PointContainerShared& points, CovarianceContainerShared& covs,
                                                robust::RobustLossType robust_type = robust::RobustLossType::CAUCHY,
                                                float mad_scale = 1.0f, float min_robust_scale = 1.0f,
                                                size_t robust_max_iterations = 1,
                                                const std::vector<sycl_utils::event>& = {}) {
    if (neighbors.k > 64) throw std::runtime_error("[covariance::estimate_robust_async] neighbor K is too large. MAX_K is 64");
    const size_t N = points.size();
    if (N == 0) { covs.resize(0); return sycl_utils::events(); }
    throw_on_error(sp_cov_estimate_robust(reinterpret_cast<const float*>(points.device_data()), N,
                                          neighbors.indices->device_data(), neighbors.k, int(robust_type), mad_scale,
                                          min_robust_scale, robust_max_iterations,
                                          reinterpret_cast<float*>(covs.device_data_for_write(N)), queue.stream()));
    return sycl_utils::events(queue.stream());
}
/// covariance.hpp:383-390
inline sycl_utils::events estimate_robust_async(const knn::KNNResult& neighbors, const PointCloudShared& points,
                                                robust::RobustLossType robust_type = robust::RobustLossType::CAUCHY,
                                                float mad_scale = 1.0f, float min_robust_scale = 1.0f,
                                                size_t robust_max_iterations = 1,
                                                const std::vector<sycl_utils::event>& depends = {}) {
    return estimate_robust_async(points.queue, neighbors, *points.points, *points.covs, robust_type, mad_scale,
                                 min_robust_scale, robust_max_iterations, depends);
}
/// covariance.hpp:400-411
inline sycl_utils::events estimate_robust_async(const knn::KNNBase& knn, const PointCloudShared& points,
                                                const size_t k_correspondences,
                                                robust::RobustLossType robust_type = robust::RobustLossType::CAUCHY,
                                                float mad_scale = 1.0f, float min_robust_scale = 1.0f,
                                                size_t robust_max_iterations = 1,
                                                const std::vector<sycl_utils::event>& depends = {}) {
    knn::KNNResult neighbors;
    auto ev = knn.knn_search_async(points, k_correspondences, neighbors, depends);
    return estimate_robust_async(neighbors, points, robust_type, mad_scale, min_robust_scale, robust_max_iterations, ev.evs);
}
/// covariance.hpp:417-443
inline sycl_utils::events estimate_normals_async(const knn::KNNResult& neighbors, const PointCloudShared& points,
                                                 const std::vector<sycl_utils::event>& = {}) {
    const size_t N = points.size();
    if (N == 0) { points.normals->resize(0); return sycl_utils::events(); }
    throw_on_error(sp_normals_from_knn(points.points_device(), N, neighbors.indices->device_data(), neighbors.k,
                                       reinterpret_cast<float*>(points.normals->device_data_for_write(N)), points.queue.stream()));
    return sycl_utils::events(points.queue.stream());
}
/// covariance.hpp:451-459
inline sycl_utils::events estimate_normals_async(const knn::KNNBase& knn, const PointCloudShared& points, const size_t k,
                                                 const std::vector<sycl_utils::event>& depends = {}) {
    knn::KNNResult neighbors;
    auto ev = knn.knn_search_async(points, k, neighbors, depends);
    return estimate_normals_async(neighbors, points, ev.evs);
}
/// covariance.hpp:465-495
inline sycl_utils::events extract_normals_async(const PointCloudShared& points, const std::vector<sycl_utils::event>& = {}) {
    if (!points.has_cov()) throw std::runtime_error("[covariance::extract_normals_async] covariances not computed");
    const size_t N = points.size();
    throw_on_error(sp_normals_from_cov(points.points_device(), points.covs_device(), N,
                                       reinterpret_cast<float*>(points.normals->device_data_for_write(N)), points.queue.stream()));
    return sycl_utils::events(points.queue.stream());
}
inline void extract_normals(const PointCloudShared& points, const std::vector<sycl_utils::event>& depends = {}) {
    extract_normals_async(points, depends).wait_and_throw();
}

}  // namespace covariance

// ================================================================================================ filters
namespace filter {

/// filter/voxel_downsampling.hpp:14-289. Keys, sort and aggregation all run on the device (sp_voxel_downsample).
class VoxelGrid {
public:
    using Ptr = std::shared_ptr<VoxelGrid>;
    VoxelGrid(const sycl_utils::DeviceQueue& queue, const float voxel_size) : queue_(queue) { set_voxel_size(voxel_size); }
    void set_voxel_size(const float voxel_size) {
        if (voxel_size <= 0.0f) throw std::invalid_argument("voxel_size must be positive");
        voxel_size_ = voxel_size;
        voxel_size_inv_ = 1.0f / voxel_size_;
        have_key_box_ = false;  // the remembered key box is in units of the old voxel size
    }
    float get_voxel_size() const { return voxel_size_; }
    void set_min_voxel_count(const size_t n) { min_voxel_count_ = n; }

    void downsampling(const PointContainerShared& points, PointContainerShared& result) {
        const size_t N = points.size();
        if (N == 0) { result.resize(0); return; }
        run(reinterpret_cast<const float*>(points.device_data()), N, nullptr, nullptr, nullptr, result, nullptr, nullptr, nullptr);
    }
    void downsampling(const PointCloudShared& cloud, PointCloudShared& result) {
        const size_t N = cloud.size();
        if (N == 0) { result.resize_points(0); return; }
        const bool in_place = (&cloud == &result) || (cloud.points == result.points);
        PointCloudShared tmp(queue_);
        PointCloudShared& out = in_place ? tmp : result;
        run(cloud.points_device(), N, cloud.has_rgb() ? cloud.rgb.get() : nullptr,
            cloud.has_intensity() ? cloud.intensities.get() : nullptr,
            cloud.has_timestamps() ? cloud.timestamp_offsets.get() : nullptr, *out.points, out.rgb.get(),
            out.intensities.get(), out.timestamp_offsets.get());
        if (!cloud.has_rgb()) out.rgb->clear();
        if (!cloud.has_intensity()) out.intensities->clear();
        if (!cloud.has_timestamps()) out.timestamp_offsets->clear();
        out.covs->clear();
        out.normals->clear();
        const double t0 = cloud.start_time_ms, t1 = cloud.end_time_ms;
        const bool ts = cloud.has_timestamps();
        if (in_place) {
            result.points = tmp.points; result.rgb = tmp.rgb; result.intensities = tmp.intensities;
            result.timestamp_offsets = tmp.timestamp_offsets; result.covs = tmp.covs; result.normals = tmp.normals;
        }
        if (ts) { result.start_time_ms = t0; result.end_time_ms = t1; }
    }

private:
    void run(const float* pts, size_t N, const RGBContainerShared* rgb, const IntensityContainerShared* inten,
             const TimestampContainerShared* ts, PointContainerShared& out_pts, RGBContainerShared* out_rgb,
             IntensityContainerShared* out_inten, TimestampContainerShared* out_ts) {
        const size_t ws_bytes = sp_voxel_downsample_workspace_bytes(N);
        // voxel count | boxed-path status | key box of sp_voxel_key_box (6 ints) | this cloud's key box, sharded
        constexpr int kInfoInts = 32 + SP_VOXEL_BOX_SHARD_STRIDE * SP_VOXEL_BOX_SHARDS;
        hipStream_t st = queue_.stream();
        detail::DeviceScratch ws(ws_bytes, st), info(kInfoInts * 4, st);
        uint32_t* info_dev = static_cast<uint32_t*>(info.p);
        static_assert(kInfoInts * 4 <= 4096, "pinned block");
        int32_t h_own[kInfoInts];
        int32_t* const h = detail::pinned_block_4k() ? static_cast<int32_t*>(detail::pinned_block_4k()) : h_own;  // (read-backs by DMA)
        // The sort runs on keys compressed to the (widened) key box of the PREVIOUS cloud — scans of one sensor have similar
        // extents; the device verifies that this cloud fits, and the key kernel finds this cloud's own box on the way (next
        // call's guess; the exact box of the redo when the cloud did not fit). The first call has no guess and computes the
        // box first: every call sorts compressed keys (the 64-bit sort is left for boxes of >= 2^32 cells).
        // The call reports {voxels, points outside the box, this cloud's key box} in ONE record (sp_voxel_downsample_report): through
        // host-mapped words the last kernel stores to and this thread spins on when they are to be had — no copy, no
        // synchronisation: the scratch then goes back behind the stream's work — else through a copy and a wait.
        const detail::MappedWord& mapped = detail::MappedWord::mine();
        const bool poll = mapped.usable();
        auto launch = [&](const int32_t* box) {
            uint32_t* const report = poll ? mapped.dev : info_dev;
            if (poll) mapped.arm();
            throw_on_error(sp_voxel_downsample_report(
                pts, N, voxel_size_inv_, min_voxel_count_, rgb ? reinterpret_cast<const float*>(rgb->device_data()) : nullptr,
                inten ? inten->device_data() : nullptr, ts ? ts->device_data() : nullptr,
                reinterpret_cast<float*>(out_pts.device_data_for_write(N)),
                rgb ? reinterpret_cast<float*>(out_rgb->device_data_for_write(N)) : nullptr,
                inten ? out_inten->device_data_for_write(N) : nullptr, ts ? out_ts->device_data_for_write(N) : nullptr, nullptr,
                nullptr, box, report, ws.p, ws_bytes, st));
            if (poll) {
                h[0] = static_cast<int32_t>(mapped.wait(st));
                std::atomic_thread_fence(std::memory_order_acquire);
                for (int j = 1; j < 8; ++j) h[j] = static_cast<int32_t>(mapped.host[j]);
                ws.stream_ordered = true;  // (kernels of the call may still be running: not idle, but ordered on st)
                info.stream_ordered = true;
            } else {
                hip_check(hipMemcpyAsync(h, info.p, 32, hipMemcpyDeviceToHost, st), "D2H");
                hip_check(hipStreamSynchronize(st), "sync");
            }
        };
        if (!have_key_box_) {
            throw_on_error(sp_voxel_key_box(pts, N, voxel_size_inv_, reinterpret_cast<int32_t*>(info_dev + 2), st));
            hip_check(hipMemcpyAsync(h, info.p, 32, hipMemcpyDeviceToHost, st), "D2H");
            hip_check(hipStreamSynchronize(st), "sync");
            have_key_box_ = h[2] <= h[5] && h[3] <= h[6] && h[4] <= h[7];
            for (int a = 0; a < 6; ++a) key_box_[a] = h[2 + a];
        }
        launch(have_key_box_ ? key_box_ : nullptr);
        if (h[1] != 0) {  // the cloud left the remembered box: again, with its own
            int32_t exact[6] = {h[2], h[3], h[4], h[5], h[6], h[7]};
            launch(exact);
        }
        const bool had_box = have_key_box_;
        have_key_box_ = h[2] <= h[5] && h[3] <= h[6] && h[4] <= h[7];
        if (have_key_box_) {
            // Keep what earlier clouds needed as well (one VoxelGrid usually serves several scans in turn — source and target of
            // a registration —, and a guess that forgets the other scan is redone every call), unless that has grown to more
            // than 8x the cells this cloud needs.
            int32_t lo[3], hi[3], ulo[3], uhi[3];
            double cells = 1.0, ucells = 1.0;
            for (int a = 0; a < 3; ++a) {
                const int32_t margin = std::max<int32_t>(2, (h[5 + a] - h[2 + a] + 1) / 8);
                lo[a] = std::max<int32_t>(h[2 + a] - margin, 0);
                hi[a] = std::min<int32_t>(h[5 + a] + margin, (1 << 21) - 1);
                ulo[a] = had_box ? std::min(lo[a], key_box_[a]) : lo[a];
                uhi[a] = had_box ? std::max(hi[a], key_box_[3 + a]) : hi[a];
                cells *= double(hi[a] - lo[a] + 1);
                ucells *= double(uhi[a] - ulo[a] + 1);
            }
            const bool keep = ucells <= 8.0 * cells;
            for (int a = 0; a < 3; ++a) {
                key_box_[a] = keep ? ulo[a] : lo[a];
                key_box_[3 + a] = keep ? uhi[a] : hi[a];
            }
        }
        const size_t V = static_cast<uint32_t>(h[0]);
        out_pts.set_device_size(V);
        if (V > 0 && h[2] <= h[5] && h[3] <= h[6] && h[4] <= h[7]) {
            // every output point is the mean of points of one voxel, so it lies in that voxel; the voxels' key box is h[2..7]
            // (coordinates offset by 2^20, compute_voxel_bit): the cloud's bounding box, a voxel wider on every side against the
            // rounding of floor(p / size), for whoever builds a grid on the cloud next
            float box[6];
            for (int a = 0; a < 3; ++a) {
                box[a] = float(h[2 + a] - (1 << 20) - 1) * voxel_size_;
                box[3 + a] = float(h[5 + a] - (1 << 20) + 2) * voxel_size_;
            }
            ::sycl_points::detail::BoundsHints::put(out_pts.generation(), box);
        }
        if (rgb) out_rgb->set_device_size(V);
        if (inten) out_inten->set_device_size(V);
        if (ts) out_ts->set_device_size(V);
    }
    sycl_utils::DeviceQueue queue_;
    float voxel_size_ = 1.0f, voxel_size_inv_ = 1.0f;
    size_t min_voxel_count_ = 1;
    int32_t key_box_[6] = {0, 0, 0, 0, 0, 0};  // widened key box of the previous cloud (the compressed sort's guess)
    bool have_key_box_ = false;
};

/// common/filter_by_flags.hpp:15-99 — stable compaction on the device (sp_compact_by_flags).
class FilterByFlags {
public:
    using Ptr = std::shared_ptr<FilterByFlags>;
    explicit FilterByFlags(const sycl_utils::DeviceQueue& queue) : queue_(queue) {}

    template <typename T>
    void filter_by_flags(const shared_vector<T>& source, shared_vector<T>& output, const shared_vector<uint8_t>& flags) const {
        const size_t N = source.size();
        if (N == 0) return;
        const size_t ws_bytes = sp_compact_workspace_bytes(N);
        hipStream_t st = queue_.stream();
        detail::DeviceScratch ws(ws_bytes, st), count(4, st), tmp(N * sizeof(T), st);
        throw_on_error(sp_compact_by_flags(source.device_data(), N, sizeof(T), flags.device_data(), tmp.p, nullptr,
                                           static_cast<uint32_t*>(count.p), ws.p, ws_bytes, st));
        const size_t M = detail::read_u32(count.p, st);
        T* dst = output.device_data_for_write(std::max<size_t>(M, 1));
        if (M) hip_check(hipMemcpyAsync(dst, tmp.p, M * sizeof(T), hipMemcpyDeviceToDevice, st), "D2D");
        hip_check(hipStreamSynchronize(st), "sync");
        output.set_device_size(M);
    }
    template <typename T>
    void filter_by_flags(shared_vector<T>& data, const shared_vector<uint8_t>& flags) const { filter_by_flags(data, data, flags); }

    void calculate_indices(const shared_vector<uint8_t>& flags, shared_vector<int32_t>& indices) const {
        const size_t N = flags.size();
        if (N == 0) return;
        const size_t ws_bytes = sp_compact_workspace_bytes(N);
        detail::DeviceScratch ws(ws_bytes, queue_.stream()), count(4, queue_.stream());
        throw_on_error(sp_compact_by_flags(flags.device_data(), N, 0 + 4 * 0 + 4, flags.device_data(), nullptr,
                                           indices.device_data_for_write(N), static_cast<uint32_t*>(count.p), ws.p, ws_bytes,
                                           queue_.stream()));
        queue_.wait();
    }

private:
    sycl_utils::DeviceQueue queue_;
};

/// filter/preprocess_filter.hpp — the two operators the hot path's callers use (box filter, random sampling).
class PreprocessFilter {
public:
    using Ptr = std::shared_ptr<PreprocessFilter>;
    explicit PreprocessFilter(const sycl_utils::DeviceQueue& queue) : queue_(queue), by_flags_(queue), mt_(1234) {
        flags_ = std::make_shared<shared_vector<uint8_t>>(queue);
    }
    void set_random_seed(uint_fast32_t seed) { mt_.seed(seed); }

    /// preprocess_operator/box_filter_operator.hpp:24-54 (K10 on the device, compaction on the device)
    void box_filter(const PointCloudShared& source, PointCloudShared& output, float min_distance = 1.0f,
                    float max_distance = std::numeric_limits<float>::max()) {
        const size_t N = source.size();
        if (N == 0) return;
        // the box test, the scan of its flags and the move of every attribute's kept rows in one launch (sp_box_filter_compact_multi)
        box_ = BoxArgs{true, min_distance, max_distance};
        apply_flags(source, output);
        box_.on = false;
    }
    void box_filter(PointCloudShared& data, float min_distance = 1.0f, float max_distance = std::numeric_limits<float>::max()) {
        box_filter(data, data, min_distance, max_distance);
    }
    /// preprocess_operator/random_sampling_operator.hpp:24-51 — host partial Fisher-Yates with std::mt19937
    void random_sampling(const PointCloudShared& source, PointCloudShared& output, size_t sampling_num) {
        const size_t N = source.size();
        if (N <= sampling_num) {
            if (&source != &output) output = PointCloudShared(source);
            return;
        }
        std::vector<size_t> indices(N);
        std::iota(indices.begin(), indices.end(), 0);
        for (size_t i = 0; i < sampling_num; ++i) {
            std::uniform_int_distribution<size_t> dist(i, N - 1);
            std::swap(indices[i], indices[dist(mt_)]);
        }
        // The reference sets a flag per drawn index and filters every attribute by the flags: the sample in the cloud's own order.
        // The same rows by their (sorted) indices: 4 KB up, ONE gather launch for all attributes — no scan of the whole cloud's
        // flags, no launch per attribute, no count to read back.
        std::vector<uint32_t> picked(sampling_num);
        for (size_t i = 0; i < sampling_num; ++i) picked[i] = static_cast<uint32_t>(indices[i]);
        std::sort(picked.begin(), picked.end());
        gather_rows(source, output, picked);
    }
    void random_sampling(PointCloudShared& data, size_t sampling_num) { random_sampling(data, data, sampling_num); }

private:
    /// output = the rows `picked` of every attribute of source (sp_gather_rows_multi)
    void gather_rows(const PointCloudShared& source, PointCloudShared& output, const std::vector<uint32_t>& picked) {
        const size_t M = picked.size();
        PointCloudShared out(queue_);
        const void* rows[6];
        void* dst[6];
        size_t bytes[6];
        int na = 0;
        auto add = [&](auto& src_vec, auto& dst_vec) {
            using T = typename std::remove_reference_t<decltype(src_vec)>::value_type;
            rows[na] = src_vec.device_data();
            dst[na] = dst_vec.device_data_for_write(M);
            bytes[na] = sizeof(T);
            ++na;
        };
        add(*source.points, *out.points);
        if (source.has_cov()) add(*source.covs, *out.covs);
        if (source.has_normal()) add(*source.normals, *out.normals);
        if (source.has_rgb()) add(*source.rgb, *out.rgb);
        if (source.has_intensity()) add(*source.intensities, *out.intensities);
        if (source.has_timestamps()) add(*source.timestamp_offsets, *out.timestamp_offsets);
        hipStream_t st = queue_.stream();
        size_t got = 0;
        void* idx = ::sycl_points::detail::DeviceBufferCache::acquire(std::max<size_t>(M, 1) * 4, &got, st);
        hipError_t e = hipMemcpyAsync(idx, picked.data(), M * 4, hipMemcpyHostToDevice, st);  // (pageable: staged before the call returns)
        int rc = SP_OK;
        if (e == hipSuccess) rc = sp_gather_rows_multi(rows, bytes, dst, na, static_cast<const uint32_t*>(idx), M, st);
        ::sycl_points::detail::DeviceBufferCache::release(idx, got, st);
        hip_check(e, "H2D");
        throw_on_error(rc);
        const double t0 = source.start_time_ms, t1 = source.end_time_ms;
        output.points = out.points; output.covs = out.covs; output.normals = out.normals; output.rgb = out.rgb;
        output.intensities = out.intensities; output.timestamp_offsets = out.timestamp_offsets;
        output.start_time_ms = t0; output.end_time_ms = t1;
    }
    void apply_flags(const PointCloudShared& source, PointCloudShared& output, size_t known_count = SIZE_MAX) {
        // FilterByFlags over every attribute the cloud carries (preprocess_operator_base): one scan of the flags, one
        // compaction launch per attribute, written straight into the new containers (no staging copy), ONE count read-back.
        const size_t N = source.size();
        PointCloudShared out(queue_);
        const void* rows[6];
        void* dst[6];
        size_t bytes[6];
        int na = 0;
        auto add = [&](auto& src_vec, auto& dst_vec) {
            using T = typename std::remove_reference_t<decltype(src_vec)>::value_type;
            rows[na] = src_vec.device_data();
            dst[na] = dst_vec.device_data_for_write(N);
            bytes[na] = sizeof(T);
            ++na;
        };
        // the box filter reads the points once: straight out of the pinned host copy when that is the current one (a fresh scan)
        bool pts_in_place = false;
        const PointType* const pts = box_.on ? source.points->device_readable_once(&pts_in_place) : source.points->device_data();
        rows[na] = pts;
        dst[na] = out.points->device_data_for_write(N);
        bytes[na] = sizeof(PointType);
        ++na;
        if (source.has_cov()) add(*source.covs, *out.covs);
        if (source.has_normal()) add(*source.normals, *out.normals);
        if (source.has_rgb()) add(*source.rgb, *out.rgb);
        if (source.has_intensity()) add(*source.intensities, *out.intensities);
        if (source.has_timestamps()) add(*source.timestamp_offsets, *out.timestamp_offsets);
        const size_t ws_bytes = sp_compact_workspace_bytes(N);
        hipStream_t st = queue_.stream();
        hipStream_t ws_release_stream = nullptr;
        // (with a known count nothing synchronises here: the scratch goes back tagged with the stream's event instead of idle)
        struct Scratch {
            void* p = nullptr; size_t bytes = 0; hipStream_t* tag;
            Scratch(size_t n, hipStream_t* t, hipStream_t use) : tag(t) { if (n) p = ::sycl_points::detail::DeviceBufferCache::acquire(n, &bytes, use); }
            ~Scratch() {
                if (!p) return;
                if (std::uncaught_exceptions() > 0) (void)hipDeviceSynchronize();  // (left by an exception: nothing was waited for)
                ::sycl_points::detail::DeviceBufferCache::release(p, bytes, *tag, *tag == nullptr);
            }
        } ws(ws_bytes, &ws_release_stream, st), count(4, &ws_release_stream, st);
        // the count comes through a host-mapped word the kernel stores to (no copy, no synchronisation) when one is to be had
        const detail::MappedWord& mapped = detail::MappedWord::mine();
        const bool poll = known_count == SIZE_MAX && mapped.usable();
        uint32_t* const count_dev = poll ? mapped.dev : static_cast<uint32_t*>(count.p);
        if (poll) mapped.arm();
        if (box_.on) {
            const int rc = sp_box_filter_compact_multi(reinterpret_cast<const float*>(pts), N, box_.min_distance, box_.max_distance, rows,
                                                       bytes, dst, na, flags_->device_data_for_write(N), nullptr, count_dev, ws.p,
                                                       ws_bytes, st);
            if (pts_in_place) source.points->host_read_enqueued(st);
            throw_on_error(rc);
        } else
            throw_on_error(sp_compact_by_flags_multi(rows, bytes, dst, na, N, flags_->device_data(), nullptr, count_dev, ws.p, ws_bytes,
                                                     st));
        size_t M = known_count;
        if (poll) { M = mapped.wait(st); ws_release_stream = st; }  // (nothing synchronised: the scratch goes back behind the stream's work)
        else if (known_count == SIZE_MAX) M = detail::read_u32(count.p, st);  // (synchronises: the scratch is idle when it leaves scope)
        else ws_release_stream = st;
        out.points->set_device_size(M);
        if (source.has_cov()) out.covs->set_device_size(M);
        if (source.has_normal()) out.normals->set_device_size(M);
        if (source.has_rgb()) out.rgb->set_device_size(M);
        if (source.has_intensity()) out.intensities->set_device_size(M);
        if (source.has_timestamps()) out.timestamp_offsets->set_device_size(M);
        const double t0 = source.start_time_ms, t1 = source.end_time_ms;
        output.points = out.points; output.covs = out.covs; output.normals = out.normals; output.rgb = out.rgb;
        output.intensities = out.intensities; output.timestamp_offsets = out.timestamp_offsets;
        output.start_time_ms = t0; output.end_time_ms = t1;
    }
    struct BoxArgs { bool on = false; float min_distance = 0.0f, max_distance = 0.0f; } box_;  // apply_flags makes the flags itself
    sycl_utils::DeviceQueue queue_;
    FilterByFlags by_flags_;
    shared_vector_ptr<uint8_t> flags_;
    std::mt19937 mt_;
};

}  // namespace filter

// ================================================================================================ transform
namespace transform {

/// common/transform.hpp:45-94
inline sycl_utils::events transform_async(PointCloudShared& cloud, const TransformMatrix& trans) {
    const size_t N = cloud.size();
    if (N == 0) return sycl_utils::events();
    const bool c = cloud.has_cov(), n = cloud.has_normal();
    float* cv = c ? reinterpret_cast<float*>(cloud.covs->device_data_rw()) : nullptr;
    float* nr = n ? reinterpret_cast<float*>(cloud.normals->device_data_rw()) : nullptr;
    float* pt = reinterpret_cast<float*>(cloud.points->device_data_rw());
    throw_on_error(sp_transform(pt, cv, nr, N, trans.data(), pt, cv, nr, cloud.queue.stream()));
    return sycl_utils::events(cloud.queue.stream());
}
inline void transform(PointCloudShared& cloud, const TransformMatrix& trans) { transform_async(cloud, trans).wait_and_throw(); }
/// common/transform.hpp:107-147
inline PointCloudShared transform_copy(const PointCloudShared& cloud, const TransformMatrix& trans) {
    PointCloudShared ret(cloud);
    transform(ret, trans);
    return ret;
}

}  // namespace transform
}  // namespace algorithms
}  // namespace sycl_points

// sycl_points facade for MI355X — algorithms/mapping/voxel_hash_map.hpp:22-250 (VoxelHashMap) over the sp_vhm_* entry
// points. Same public surface as the reference: constructor (queue, voxel_size), the five setters / getters, clear,
// add_point_cloud(cloud, sensor_pose), downsampling(result, center, distance), compute_overlap_ratio, remove_old_data.
// The table lives in HBM inside the library object; clouds go in and come out through their device mirrors.
#pragma once
#include "core.hpp"

namespace sycl_points {
namespace algorithms {
namespace mapping {

class VoxelHashMap {
public:
    using Ptr = std::shared_ptr<VoxelHashMap>;

    VoxelHashMap(const sycl_utils::DeviceQueue& queue, const float voxel_size) : queue_(queue) {
        throw_on_error(sp_vhm_create(voxel_size, queue_.stream(), &h_));  // voxel_size <= 0: std::invalid_argument
    }
    ~VoxelHashMap() { sp_vhm_destroy(h_); }
    VoxelHashMap(const VoxelHashMap&) = delete;
    VoxelHashMap& operator=(const VoxelHashMap&) = delete;

    void set_voxel_size(const float voxel_size) { throw_on_error(sp_vhm_set(h_, SP_VHM_VOXEL_SIZE, voxel_size)); }
    float get_voxel_size() const { return sp_vhm_get(h_, SP_VHM_VOXEL_SIZE); }
    void set_max_staleness(const uint32_t v) { throw_on_error(sp_vhm_set(h_, SP_VHM_MAX_STALENESS, (float)v)); }
    uint32_t get_max_staleness() const { return (uint32_t)sp_vhm_get(h_, SP_VHM_MAX_STALENESS); }
    void set_remove_old_data_cycle(const uint32_t v) { throw_on_error(sp_vhm_set(h_, SP_VHM_REMOVE_OLD_DATA_CYCLE, (float)v)); }
    uint32_t get_remove_old_data_cycle() const { return (uint32_t)sp_vhm_get(h_, SP_VHM_REMOVE_OLD_DATA_CYCLE); }
    void set_rehash_threshold(const float v) { throw_on_error(sp_vhm_set(h_, SP_VHM_REHASH_THRESHOLD, v)); }
    float get_rehash_threshold() const { return sp_vhm_get(h_, SP_VHM_REHASH_THRESHOLD); }
    void set_min_num_point(const uint32_t v) { throw_on_error(sp_vhm_set(h_, SP_VHM_MIN_NUM_POINT, (float)v)); }
    uint32_t get_min_num_point() const { return (uint32_t)sp_vhm_get(h_, SP_VHM_MIN_NUM_POINT); }

    /// voxel_hash_map.hpp:83-113
    void clear() { throw_on_error(sp_vhm_clear(h_, queue_.stream())); }

    /// voxel_hash_map.hpp:117-141 — cloud in the sensor frame, sensor_pose in the map frame.
    void add_point_cloud(const PointCloudShared& cloud, const Eigen::Isometry3f& sensor_pose) {
        const size_t N = cloud.size();
        throw_on_error(sp_vhm_add_point_cloud(
            h_, N ? cloud.points_device() : nullptr, N ? cloud.covs_device() : nullptr,
            (N && cloud.has_rgb()) ? reinterpret_cast<const float*>(cloud.rgb->device_data()) : nullptr,
            (N && cloud.has_intensity()) ? cloud.intensities->device_data() : nullptr, N, sensor_pose.matrix().data(),
            queue_.stream()));
    }

    /// voxel_hash_map.hpp:146-190 — voxel means whose centroid lies in the box center +- distance.
    void downsampling(PointCloudShared& result, const Eigen::Vector3f& center, const float distance = 100.0f) {
        const size_t cap = sp_vhm_info(h_, SP_VHM_INFO_VOXEL_NUM);
        if (cap == 0) { result.clear(); return; }
        const bool has_cov = sp_vhm_info(h_, SP_VHM_INFO_HAS_COV), has_rgb = sp_vhm_info(h_, SP_VHM_INFO_HAS_RGB),
                   has_int = sp_vhm_info(h_, SP_VHM_INFO_HAS_INTENSITY);
        const float c[3] = {center.x(), center.y(), center.z()};
        size_t n = 0;
        throw_on_error(sp_vhm_downsampling(
            h_, c, distance, reinterpret_cast<float*>(result.points->device_data_for_write(cap)),
            has_cov ? reinterpret_cast<float*>(result.covs->device_data_for_write(cap)) : nullptr,
            has_rgb ? reinterpret_cast<float*>(result.rgb->device_data_for_write(cap)) : nullptr,
            has_int ? result.intensities->device_data_for_write(cap) : nullptr, nullptr, cap, &n, queue_.stream()));
        result.points->set_device_size(n);
        if (has_cov) result.covs->set_device_size(n); else result.covs->clear();
        if (has_rgb) result.rgb->set_device_size(n); else result.rgb->clear();
        if (has_int) result.intensities->set_device_size(n); else result.intensities->clear();
        result.normals->clear();
        result.timestamp_offsets->clear();
    }

    /// voxel_hash_map.hpp:196-246
    float compute_overlap_ratio(const PointCloudShared& cloud, const Eigen::Isometry3f& sensor_pose) const {
        if (!cloud.points || cloud.points->empty()) return 0.0f;
        float r = 0.0f;
        throw_on_error(sp_vhm_overlap_ratio(h_, cloud.points_device(), cloud.size(), sensor_pose.matrix().data(), &r,
                                            queue_.stream()));
        return r;
    }

    void remove_old_data() { throw_on_error(sp_vhm_remove_old_data(h_, queue_.stream())); }

private:
    sycl_utils::DeviceQueue queue_;
    sp_voxel_hash_map* h_ = nullptr;
};

}  // namespace mapping
}  // namespace algorithms
}  // namespace sycl_points

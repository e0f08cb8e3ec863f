// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
// CPU restatement of covariance / normal estimation, voxel-grid downsampling, transforms and the box filter:
//   covariance       /root/reference/cpp/include/sycl_points/algorithms/feature/covariance.hpp:16-95
//   voxel key        .../common/voxel_constants.hpp:36-62
//   voxel aggregate  .../filter/voxel_downsampling.hpp:82-98,146-288
//   transforms       .../common/transform.hpp:14-37
//   box filter       .../filter/preprocess_operator/common.hpp:15-25, box_filter_operator.hpp:36-44
#pragma once
#include <algorithm>
#include <vector>

#include "oracle_math.hpp"

namespace oracle {

inline Mat3 block3(const float* cov_colmajor4) {
    Mat3 r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) r(i, j) = cov_colmajor4[j * 4 + i];
    return r;
}
inline void set_block3(float* cov_colmajor4, const Mat3& m) {
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) cov_colmajor4[j * 4 + i] = m(i, j);
}

// feature/covariance.hpp:16-47.  cov_out is a column-major 4x4 (Eigen::Matrix4f), 3x3 block used.
inline void cov_estimate_one(float* cov_out, const float* points, size_t k, const int32_t* index_ptr, size_t i,
                             size_t min_num = 4) {
    for (int e = 0; e < 16; ++e) cov_out[e] = 0.0f;
    Vec3 sum_points = Vec3::Zero();
    Mat3 sum_outer = Mat3::Zero();
    size_t n = 0;
    for (size_t j = 0; j < k; ++j) {
        const int32_t idx = index_ptr[i * k + j];
        if (idx < 0) continue;
        Vec3 pt;
        pt[0] = points[4 * (size_t)idx + 0];
        pt[1] = points[4 * (size_t)idx + 1];
        pt[2] = points[4 * (size_t)idx + 2];
        add_inplace<3, 1>(sum_points, pt);
        add_inplace<3, 3>(sum_outer, outer<3>(pt, pt));
        ++n;
    }
    min_num = std::max(min_num, (size_t)4);
    if (n < min_num) {
        cov_out[0] = 1.0f;
        cov_out[5] = 1.0f;
        cov_out[10] = 1.0f;
        return;
    }
    const Vec3 mean = scale<3, 1>(sum_points, 1.0f / n);
    const Mat3 c = ensure_symmetric<3>(subtract<3, 3>(scale<3, 3>(sum_outer, 1.0f / n), outer<3>(mean, mean)));
    set_block3(cov_out, c);
}

// feature/covariance.hpp:49-65
inline void extract_normal(const float* point, const float* cov, float* normal) {
    Vec3 vals;
    Mat3 vecs;
    symmetric_eigen_decomposition_3x3(block3(cov), vals, vecs);
    Vec3 n3, p3;
    for (int i = 0; i < 3; ++i) { n3[i] = vecs(i, 0); p3[i] = point[i]; }
    if (dot<3>(n3, p3) <= 1.0) {
        normal[0] = n3[0]; normal[1] = n3[1]; normal[2] = n3[2]; normal[3] = 0.0f;
    } else {
        normal[0] = -n3[0]; normal[1] = -n3[1]; normal[2] = -n3[2]; normal[3] = 0.0f;
    }
}

// feature/covariance.hpp:67-74
inline void update_covariance_plane(float* cov) {
    Vec3 vals;
    Mat3 vecs;
    symmetric_eigen_decomposition_3x3(block3(cov), vals, vecs);
    Vec3 dv; dv[0] = 1e-3f; dv[1] = 1.0f; dv[2] = 1.0f;
    const Mat3 diag = as_diagonal<3>(dv);
    set_block3(cov, matmul<3, 3, 3>(matmul<3, 3, 3>(vecs, diag), transpose<3, 3>(vecs)));
}

// feature/covariance.hpp:76-95
inline void normalize_covariance(float* cov) {
    Vec3 vals;
    Mat3 vecs;
    symmetric_eigen_decomposition_3x3(scale<3, 3>(block3(cov), 1e3f), vals, vecs);
    const float mx = vals[2];
    if (mx < std::numeric_limits<float>::min()) {
        set_block3(cov, Mat3::Identity());
        return;
    }
    vals[0] = std::clamp(vals[0] / mx, 1e-3f, 1.0f);
    vals[1] = std::clamp(vals[1] / mx, 1e-3f, 1.0f);
    vals[2] = 1.0f;
    set_block3(cov, matmul<3, 3, 3>(matmul<3, 3, 3>(vecs, as_diagonal<3>(vals)), transpose<3, 3>(vecs)));
}

// common/transform.hpp:14-22   result = T * cov * T^T (two 4x4x4 fma products)
inline void transform_cov(const float* cov, float* out, const float* T_colmajor) {
    Mat4 T, C;
    std::memcpy(T.d, T_colmajor, 64);
    std::memcpy(C.d, cov, 64);
    const Mat4 r = matmul<4, 4, 4>(T, matmul<4, 4, 4>(C, transpose<4, 4>(T)));
    std::memcpy(out, r.d, 64);
}
// common/transform.hpp:24-29.  The reference calls eigen_utils::normalize<4>(ret) and DISCARDS its return
// value, so the stored normal is T*n un-normalised; restated literally.
inline void transform_normal(const float* n, float* out, const float* T_colmajor) {
    Mat4 T;
    std::memcpy(T.d, T_colmajor, 64);
    Vec4 v;
    std::memcpy(v.d, n, 16);
    const Vec4 r = matvec<4, 4>(T, v);
    std::memcpy(out, r.d, 16);
}

// common/voxel_constants.hpp:36-62
constexpr uint64_t VOXEL_INVALID = std::numeric_limits<uint64_t>::max();
inline uint64_t compute_voxel_bit(const float* p, float inv) {
    constexpr int64_t mask = (1 << 21) - 1;
    constexpr int64_t offset = 1 << 20;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) return VOXEL_INVALID;
    const int64_t c0 = (int64_t)std::floor(p[0] * inv) + offset;
    const int64_t c1 = (int64_t)std::floor(p[1] * inv) + offset;
    const int64_t c2 = (int64_t)std::floor(p[2] * inv) + offset;
    if (c0 < 0 || mask < c0 || c1 < 0 || mask < c1 || c2 < 0 || mask < c2) return VOXEL_INVALID;
    return ((uint64_t)(c0 & mask) << 0) | ((uint64_t)(c1 & mask) << 21) | ((uint64_t)(c2 & mask) << 42);
}

// filter/voxel_downsampling.hpp:82-98
inline float voxel_median(std::vector<float>& v) {
    if (v.empty()) return 0.0f;
    const size_t mid = v.size() / 2;
    std::nth_element(v.begin(), v.begin() + mid, v.end());
    const float upper = v[mid];
    if ((v.size() % 2) != 0U) return upper;
    std::nth_element(v.begin(), v.begin() + (mid - 1), v.begin() + mid);
    return 0.5f * (v[mid - 1] + upper);
}

struct VoxelOut {
    std::vector<float> points;       // 4 per voxel
    std::vector<float> rgb;          // 4 per voxel
    std::vector<float> intensities;  // 1 per voxel
    std::vector<float> timestamps;   // 1 per voxel
    std::vector<uint64_t> keys;      // key of each emitted voxel (ascending)
};

// filter/voxel_downsampling.hpp:146-288.  `stable`=false uses std::sort exactly as the reference does
// (unstable: the intra-voxel summation order is whatever libstdc++'s introsort leaves); `stable`=true orders
// equal keys by ascending point index — the order the device path defines — so sums can be compared bit for bit.
inline VoxelOut voxel_downsample(const float* points, size_t N, float voxel_size, size_t min_voxel_count,
                                 const float* rgb, const float* intensity, const float* timestamps, bool stable) {
    VoxelOut out;
    const float inv = 1.0f / voxel_size;  // voxel_downsampling.hpp:27
    std::vector<uint64_t> bits(N);
    for (size_t i = 0; i < N; ++i) bits[i] = compute_voxel_bit(points + 4 * i, inv);
    std::vector<size_t> order;
    order.reserve(N);
    for (size_t i = 0; i < N; ++i)
        if (bits[i] != VOXEL_INVALID) order.push_back(i);
    auto cmp = [&](size_t l, size_t r) { return bits[l] < bits[r]; };
    if (stable)
        std::stable_sort(order.begin(), order.end(), cmp);
    else
        std::sort(order.begin(), order.end(), cmp);
    const size_t M = order.size();
    const float min_count = (float)min_voxel_count;
    std::vector<float> ivals;
    size_t b = 0;
    while (b < M) {
        const uint64_t key = bits[order[b]];
        float psum[4] = {0, 0, 0, 0}, csum[4] = {0, 0, 0, 0};
        float tsum = 0.0f;
        ivals.clear();
        size_t e = b;
        while (e < M && bits[order[e]] == key) {
            const size_t idx = order[e];
            for (int c = 0; c < 4; ++c) psum[c] += points[4 * idx + c];
            if (rgb)
                for (int c = 0; c < 4; ++c) csum[c] += rgb[4 * idx + c];
            if (intensity) ivals.push_back(intensity[idx]);
            if (timestamps) tsum += timestamps[idx];
            ++e;
        }
        const float cnt = psum[3];
        if (cnt >= min_count) {
            for (int c = 0; c < 4; ++c) out.points.push_back(psum[c] / cnt);
            if (rgb)
                for (int c = 0; c < 4; ++c) out.rgb.push_back(csum[c] / cnt);
            if (intensity) out.intensities.push_back(voxel_median(ivals));
            if (timestamps) out.timestamps.push_back(tsum / cnt);
            out.keys.push_back(key);
        }
        b = e;
    }
    return out;
}

// preprocess_operator/common.hpp:15-25 + box_filter_operator.hpp:36-44  (flag 1 = keep, 0 = remove)
inline uint8_t box_filter_flag(const float* p, float min_d, float max_d) {
    if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]) && std::isfinite(p[3]))) return 0;
    const float linf = sycl_max(std::fabs(p[0]), sycl_max(std::fabs(p[1]), std::fabs(p[2])));
    if (linf < min_d || linf > max_d) return 0;
    return 1;
}

}  // namespace oracle

// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
// CPU restatement of the M-estimated neighbourhood covariance (K8):
//   /root/reference/cpp/include/sycl_points/algorithms/feature/covariance.hpp:97-250 (kernel), 323-381 (host entry)
// Pins: none in the reference's tests ("parity unpinned"); with robust_max_iterations = 0 the result must equal
// covariance::kernel::estimate bit for bit (weights 1.0f), which tests/test_oracle_pins.py checks.
#pragma once
#include <limits>

#include "oracle_features.hpp"
#include "oracle_registration.hpp"

namespace oracle {

// covariance.hpp:97-134
inline bool cov_estimate_weighted(Mat3& cov, Vec3& mean, const float* points, size_t k, const int32_t* index_ptr,
                                  const float* weights, size_t i, size_t min_num = 4) {
    cov = Mat3::Zero();
    Vec3 sum_points = Vec3::Zero();
    Mat3 sum_outer = Mat3::Zero();
    size_t n = 0;
    float total_weight = 0.0f;
    for (size_t j = 0; j < k; ++j) {
        const int32_t idx = index_ptr[i * k + j];
        if (idx < 0) continue;
        Vec3 pt;
        pt[0] = points[4 * (size_t)idx + 0];
        pt[1] = points[4 * (size_t)idx + 1];
        pt[2] = points[4 * (size_t)idx + 2];
        add_inplace<3, 1>(sum_points, scale<3, 1>(pt, weights[j]));
        add_inplace<3, 3>(sum_outer, scale<3, 3>(outer<3>(pt, pt), weights[j]));
        ++n;
        total_weight += weights[j];
    }
    min_num = std::max(min_num, (size_t)4);
    if (n < min_num || total_weight < std::numeric_limits<float>::epsilon()) {
        cov = Mat3::Identity();
        return false;
    }
    mean = scale<3, 1>(sum_points, 1.0f / total_weight);
    cov = ensure_symmetric<3>(subtract<3, 3>(scale<3, 3>(sum_outer, 1.0f / total_weight), outer<3>(mean, mean)));
    return true;
}

// covariance.hpp:143-173 (insertion sort of a copy, then the middle element / mean of the two middle elements)
inline float cov_median(const float* data, float* buffer, size_t n) {
    if (n == 0) return 0.0f;
    for (size_t i = 0; i < n; ++i) buffer[i] = data[i];
    for (size_t i = 1; i < n; ++i) {
        const float key = buffer[i];
        size_t j = i;
        while (j > 0 && buffer[j - 1] > key) {
            buffer[j] = buffer[j - 1];
            --j;
        }
        buffer[j] = key;
    }
    const size_t mid = n / 2;
    return (n % 2 == 0) ? (buffer[mid - 1] + buffer[mid]) * 0.5f : buffer[mid];
}

// covariance.hpp:175-180: 4-wide products whose fourth terms are zero (cov_inv row/column 3 and diff.w are 0)
inline float cov_mahalanobis2(const Mat3& cov_inv, const Vec3& mean, const float* pt) {
    Vec<4> diff;
    diff[0] = pt[0] - mean[0]; diff[1] = pt[1] - mean[1]; diff[2] = pt[2] - mean[2]; diff[3] = 0.0f;
    Mat<4, 4> M = Mat<4, 4>::Zero();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M(i, j) = cov_inv(i, j);
    return dot<4>(diff, matvec<4, 4>(M, diff));
}

// covariance.hpp:182-250. cov_out: column-major 4x4.
inline void cov_estimate_robust_one(float* cov_out, const float* points, size_t k, const int32_t* index_ptr, size_t i,
                                    int robust_type, float mad_scale, float min_robust_scale, size_t max_iter) {
    if (robust_type == LOSS_NONE) {
        cov_estimate_one(cov_out, points, k, index_ptr, i);
        return;
    }
    constexpr size_t MAX_K = 64;
    float weights[MAX_K], dist2[MAX_K];
    for (size_t j = 0; j < MAX_K; ++j) { weights[j] = 1.0f; dist2[j] = 0.0f; }
    Mat3 cov;
    Vec3 mean = Vec3::Zero();
    bool ok = cov_estimate_weighted(cov, mean, points, k, index_ptr, weights, i);
    if (ok) {
        for (size_t it = 0; it < max_iter; ++it) {
            const Mat3 cov_inv = inverse(cov);
            for (size_t j = 0; j < k; ++j) {
                const int32_t idx = index_ptr[i * k + j];
                if (idx < 0) continue;
                dist2[j] = cov_mahalanobis2(cov_inv, mean, points + 4 * (size_t)idx);
            }
            const float median = cov_median(dist2, weights, k);  // weights doubles as the sort buffer
            float robust_scale = mad_scale * median;
            if (robust_scale < min_robust_scale) robust_scale = min_robust_scale;
            for (size_t j = 0; j < k; ++j) weights[j] = robust_weight(robust_type, dist2[j], robust_scale);
            ok = cov_estimate_weighted(cov, mean, points, k, index_ptr, weights, i);
            if (!ok) break;
        }
    }
    for (int e = 0; e < 16; ++e) cov_out[e] = 0.0f;
    set_block3(cov_out, cov);
}

}  // namespace oracle

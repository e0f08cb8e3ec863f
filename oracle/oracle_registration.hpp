// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
// CPU restatement of the registration path:
//   per-point factors   /root/reference/cpp/include/sycl_points/algorithms/registration/factor.hpp:69-482
//   robust kernels      .../robust/robust.hpp:56-114
//   linearize + reduce  .../registration/registration.hpp:513-664 (K11), error-only :678-777 (K12)
//   GenZ alpha          .../registration/registration.hpp:464-511
//   optimisers          .../registration/registration.hpp:201-276, 407-410, 791-895
//   annealing wrapper   .../registration/pipeline/robust.hpp:42-114
// The reduction order of sycl::reduction is unspecified (parity unpinned); this restatement sums
// fixed blocks of 1024 points sequentially in fp32 and then the block partials sequentially in fp32,
// which is deterministic and independent of the OpenMP thread count.
#pragma once
#include <functional>
#include <vector>

#include "oracle_features.hpp"
#include "oracle_knn.hpp"
#include "oracle_math.hpp"
#include "oracle_reg_terms.hpp"

namespace oracle {

enum RegType { POINT_TO_POINT = 0, POINT_TO_PLANE = 1, POINT_TO_DISTRIBUTION = 2, GICP = 3, GENZ = 4 };  // factor.hpp:18-32
enum RobustLossType { LOSS_NONE = 0, HUBER = 1, TUKEY = 2, CAUCHY = 3, GEMAN_MCCLURE = 4 };             // robust.hpp:13-19
enum OptimizationMethod { GAUSS_NEWTON = 0, LEVENBERG_MARQUARDT = 1, POWELL_DOGLEG = 2 };

// robust.hpp:56-90
inline float robust_weight(int loss, float residual_norm, float scale) {
    if (loss == LOSS_NONE) return 1.0f;
    if (residual_norm <= 1e-8f) return 1.0f;
    const float nr = residual_norm / scale;
    switch (loss) {
        case HUBER: return sycl_min(1.0f, 1.0f / nr);
        case TUKEY: {
            if (nr >= 1.0f) return 0.0f;
            const float x = nr * nr;
            const float f = 1.0f - x;
            return f * f;
        }
        case CAUCHY: {
            const float x = nr * nr;
            return 1.0f / (1.0f + x);
        }
        case GEMAN_MCCLURE: {
            const float x = nr * nr;
            const float den = 1.0f + x;
            return 1.0f / (den * den);
        }
    }
    return 1.0f;
}
// robust.hpp:96-114
inline float robust_error(int loss, float r, float s) {
    switch (loss) {
        case LOSS_NONE: return 0.5f * r * r;
        case HUBER: return r <= s ? 0.5f * r * r : s * (r - 0.5f * s);
        case TUKEY:
            return r <= s ? (s * s / 6.0f) * (1.0f - std::pow(1.0f - ((r * r) / (s * s)), 3.0f)) : s * s / 6.0f;
        case CAUCHY: return 0.5f * s * s * std::log(1.0f + ((r * r) / (s * s)));
        case GEMAN_MCCLURE: return 0.5f * (s * s * r * r) / (s * s + r * r);
    }
    return 0.5f * r * r;
}

struct KernelResult {  // linearized_result.hpp:27-38
    Mat6 H = Mat6::Zero();
    Vec6 b = Vec6::Zero();
    float squared_error = std::numeric_limits<float>::max();
    uint32_t inlier = 0;
};

inline Mat4 to_mat4(const float* colmajor) {
    Mat4 m;
    std::memcpy(m.d, colmajor, 64);
    return m;
}
inline Vec4 to_vec4(const float* p) {
    Vec4 v;
    std::memcpy(v.d, p, 16);
    return v;
}

// factor.hpp:69-84
inline Mat<4, 6> compute_se3_jacobian(const Mat4& T, const Vec4& source_pt) {
    Mat<4, 6> J = Mat<4, 6>::Zero();
    const Mat3 skewed = skew(source_pt);
    Mat3 R;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) R(i, j) = T(i, j);
    const Mat3 T_skewed = matmul<3, 3, 3>(R, skewed);
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            J(i, j) = T_skewed(i, j);
            J(i, 3 + j) = -R(i, j);
        }
    return J;
}
// factor.hpp:90-104 (weight matrix is Identity at every call site: an fma product with I)
inline Mat<4, 6> compute_weighted_se3_jacobian(const Mat4& T, const Vec4& source_pt, const Mat4& W) {
    return matmul<4, 4, 6>(W, compute_se3_jacobian(T, source_pt));
}
// factor.hpp:111-123
inline Mat4 compute_mahalanobis_covariance(const Mat4& source_cov, const Mat4& target_cov, const Mat4& T) {
    Mat4 mah = Mat4::Zero();
    Mat4 tsc;
    transform_cov(source_cov.d, tsc.d, T.d);
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) mah(i, j) = tsc(i, j) + target_cov(i, j);
    return mah;
}
// feature/covariance.hpp:143-148
inline Mat4 cov_inverse(const Mat4& cov) {
    Mat4 r = Mat4::Zero();
    const Mat3 inv = inverse(block3(cov.d));
    set_block3(r.d, inv);
    return r;
}
inline Vec4 residual_of(const Mat4& T, const Vec4& s, const Vec4& t) {
    Vec4 ts;
    transform_point(s.d, ts.d, T.d);
    Vec4 r;
    r[0] = t[0] - ts[0]; r[1] = t[1] - ts[1]; r[2] = t[2] - ts[2]; r[3] = 0.0f;
    return r;
}

// factor.hpp:130-149
inline KernelResult linearize_point_to_point(const Mat4& T, const Vec4& s, const Vec4& t, float& residual_norm) {
    const Vec4 residual = residual_of(T, s, t);
    const Mat<4, 6> J = compute_weighted_se3_jacobian(T, s, Mat4::Identity());
    KernelResult ret;
    const Mat<6, 4> JT = transpose<4, 6>(J);
    ret.H = ensure_symmetric<6>(matmul<6, 4, 6>(JT, J));
    ret.b = matvec<6, 4>(JT, residual);
    const float sq = norm_squared<4>(residual);
    residual_norm = std::sqrt(sq);
    ret.squared_error = sq;
    ret.inlier = 1;
    return ret;
}
inline float error_point_to_point(const Mat4& T, const Vec4& s, const Vec4& t) {  // factor.hpp:156-164
    return norm_squared<4>(residual_of(T, s, t));
}
// factor.hpp:172-210
inline KernelResult linearize_point_to_plane(const Mat4& T, const Vec4& s, const Vec4& t, const Vec4& tn,
                                             float& residual_norm) {
    const Vec4 residual = residual_of(T, s, t);
    Vec3 normal, r3;
    for (int i = 0; i < 3; ++i) { normal[i] = tn[i]; r3[i] = residual[i]; }
    const float proj = dot<3>(normal, r3);
    Vec4 plane_error = Vec4::Zero();
    for (int i = 0; i < 3; ++i) plane_error[i] = normal[i] * proj;
    const Mat<4, 6> se3J = compute_se3_jacobian(T, s);
    Mat<3, 6> J36;
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 3; ++i) J36(i, j) = se3J(i, j);
    const Mat<1, 3> nT = transpose<3, 1>(normal);
    const Mat<1, 6> row = matmul<1, 3, 6>(nT, J36);
    const Mat<3, 6> Jp = matmul<3, 1, 6>(normal, row);
    Mat<4, 6> J = Mat<4, 6>::Zero();
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 3; ++i) J(i, j) = Jp(i, j);
    KernelResult ret;
    const Mat<6, 4> JT = transpose<4, 6>(J);
    ret.H = ensure_symmetric<6>(matmul<6, 4, 6>(JT, J));
    ret.b = matvec<6, 4>(JT, plane_error);
    residual_norm = std::fabs(proj);
    ret.squared_error = proj * proj;
    ret.inlier = 1;
    return ret;
}
inline float error_point_to_plane(const Mat4& T, const Vec4& s, const Vec4& t, const Vec4& tn) {  // factor.hpp:218-230
    const Vec4 residual = residual_of(T, s, t);
    Vec3 normal, r3;
    for (int i = 0; i < 3; ++i) { normal[i] = tn[i]; r3[i] = residual[i]; }
    const float proj = dot<3>(normal, r3);
    return proj * proj;
}
// factor.hpp:239-278
inline KernelResult linearize_gicp(const Mat4& T, const Vec4& s, const Mat4& scov, const Vec4& t, const Mat4& tcov,
                                   float& residual_norm) {
    const Vec4 residual = residual_of(T, s, t);
    Mat4 ns = scov, nt = tcov;
    update_covariance_plane(ns.d);
    update_covariance_plane(nt.d);
    const Mat4 mah_inv = cov_inverse(compute_mahalanobis_covariance(ns, nt, T));
    const Mat<4, 6> J = compute_weighted_se3_jacobian(T, s, Mat4::Identity());
    const Mat<6, 4> JTm = matmul<6, 4, 4>(transpose<4, 6>(J), mah_inv);
    KernelResult ret;
    ret.H = ensure_symmetric<6>(matmul<6, 4, 6>(JTm, J));
    ret.b = matvec<6, 4>(JTm, residual);
    const float sq = dot<4>(residual, matvec<4, 4>(mah_inv, residual));
    residual_norm = std::sqrt(sq);
    ret.squared_error = sq;
    ret.inlier = 1;
    return ret;
}
inline float error_gicp(const Mat4& T, const Vec4& s, const Mat4& scov, const Vec4& t, const Mat4& tcov) {  // :287-306
    const Vec4 residual = residual_of(T, s, t);
    Mat4 ns = scov, nt = tcov;
    update_covariance_plane(ns.d);
    update_covariance_plane(nt.d);
    const Mat4 mah_inv = cov_inverse(compute_mahalanobis_covariance(ns, nt, T));
    return dot<4>(residual, matvec<4, 4>(mah_inv, residual));
}
// factor.hpp:311-354
inline KernelResult linearize_point_to_distribution(const Mat4& T, const Vec4& s, const Vec4& t, const Mat4& tcov,
                                                    float& residual_norm) {
    const Vec4 residual = residual_of(T, s, t);
    const Mat<4, 6> J = compute_weighted_se3_jacobian(T, s, Mat4::Identity());
    const Mat4 mah = cov_inverse(tcov);
    const Mat<6, 4> JTm = matmul<6, 4, 4>(transpose<4, 6>(J), mah);
    KernelResult ret;
    ret.H = ensure_symmetric<6>(matmul<6, 4, 6>(JTm, J));
    ret.b = matvec<6, 4>(JTm, residual);
    const float sq = dot<4>(residual, matvec<4, 4>(mah, residual));
    residual_norm = std::sqrt(sq);
    ret.squared_error = sq;
    ret.inlier = 1;
    return ret;
}
inline float error_point_to_distribution(const Mat4& T, const Vec4& s, const Vec4& t, const Mat4& tcov) {  // :362-373
    const Mat4 mah = cov_inverse(tcov);
    const Vec4 residual = residual_of(T, s, t);
    return dot<4>(residual, matvec<4, 4>(mah, residual));
}
// factor.hpp:378-392
inline bool genz_is_planar(const Mat4& tcov, float thr) {
    Vec3 vals;
    Mat3 vecs;
    symmetric_eigen_decomposition_3x3(block3(tcov.d), vals, vecs);
    const float sum = vals[0] + vals[1] + vals[2];
    const float curv = (sum > 1e-12f) ? vals[0] / sum : 1.0f;
    return curv < thr;
}
// factor.hpp:413-449
inline KernelResult linearize_geometry(int reg, const Mat4& T, const Vec4& s, const Mat4& scov, const Vec4& t,
                                       const Mat4& tcov, const Vec4& tn, float& residual_norm, float genz_alpha,
                                       float& genz_weight, float genz_thr) {
    switch (reg) {
        case POINT_TO_POINT: return linearize_point_to_point(T, s, t, residual_norm);
        case POINT_TO_PLANE: return linearize_point_to_plane(T, s, t, tn, residual_norm);
        case GICP: return linearize_gicp(T, s, scov, t, tcov, residual_norm);
        case POINT_TO_DISTRIBUTION: return linearize_point_to_distribution(T, s, t, tcov, residual_norm);
        case GENZ: {
            const bool planar = genz_is_planar(tcov, genz_thr);
            genz_weight = planar ? genz_alpha : (1.0f - genz_alpha);
            float sel_norm = 0.0f;
            const KernelResult sel =
                planar ? linearize_point_to_plane(T, s, t, tn, sel_norm) : linearize_point_to_point(T, s, t, sel_norm);
            residual_norm = sel_norm;
            KernelResult r;
            r.H = scale<6, 6>(sel.H, genz_weight);
            r.b = scale<6, 1>(sel.b, genz_weight);
            r.squared_error = sel.squared_error * genz_weight;
            r.inlier = 1;
            return r;
        }
    }
    return KernelResult();
}
// factor.hpp:459-482
inline float geometry_error(int reg, const Mat4& T, const Vec4& s, const Mat4& scov, const Vec4& t, const Mat4& tcov,
                            const Vec4& tn, float genz_alpha, float& genz_weight, float genz_thr) {
    switch (reg) {
        case POINT_TO_POINT: return error_point_to_point(T, s, t);
        case POINT_TO_PLANE: return error_point_to_plane(T, s, t, tn);
        case GICP: return error_gicp(T, s, scov, t, tcov);
        case POINT_TO_DISTRIBUTION: return error_point_to_distribution(T, s, t, tcov);
        case GENZ: {
            const bool planar = genz_is_planar(tcov, genz_thr);
            genz_weight = planar ? genz_alpha : (1.0f - genz_alpha);
            return planar ? error_point_to_plane(T, s, t, tn) : error_point_to_point(T, s, t);
        }
    }
    return 0.0f;
}

// ------------------------------------------------------------------ clouds + reductions
struct Cloud {  // borrowed host pointers; covs/normals may be null (registration.hpp:539-543)
    const float* points = nullptr;   // N x 4
    const float* covs = nullptr;     // N x 16 column-major
    const float* normals = nullptr;  // N x 4
    size_t n = 0;
};

struct Linearized {  // linearized_result.hpp:12-24
    Mat6 H = Mat6::Zero();
    Vec6 b = Vec6::Zero();
    float error = std::numeric_limits<float>::max();
    uint32_t inlier = 0;
};

struct FactorParams {  // registration_params.hpp:46-71
    int reg_type = GICP;
    float max_correspondence_distance = 2.0f;
    int robust_type = LOSS_NONE;
    float robust_default_scale = 10.0f;
    float genz_planarity_threshold = 0.2f;
    // RotationConstraint (registration_params.hpp:56-64)
    bool rot_enable = false;
    float rot_weight = 1.0f;
    float rot_robust_default_scale = 10.0f;
};

constexpr size_t REDUCE_BLOCK = 1024;

// registration.hpp:464-511
inline float compute_genz_alpha(const Cloud& target, const int32_t* nn_idx, const float* nn_d2, size_t N, float max_corr,
                                float thr) {
    uint32_t inl = 0, plane = 0;
    const float max_d2 = max_corr * max_corr;
    for (size_t i = 0; i < N; ++i) {
        if (nn_d2[i] > max_d2) continue;
        if (genz_is_planar(to_mat4(target.covs + 16 * (size_t)nn_idx[i]), thr)) ++plane;
        ++inl;
    }
    if (inl == 0) return 1.0f;
    return (float)plane / (float)inl;
}

// registration.hpp:513-664 (K11).  Optional per-point outputs (H 36 row-major, b 6, err 1, flag 1 = 44 floats)
inline Linearized linearize_reduce(const FactorParams& fp, const Cloud& source, const Cloud& target, const int32_t* nn_idx,
                                   const float* nn_d2, const float* T_colmajor, float robust_scale, float genz_alpha,
                                   float* per_point = nullptr, float rotation_robust_scale = 10.0f) {
    const size_t N = source.n;
    const Mat4 T = to_mat4(T_colmajor);
    const float max_d2 = fp.max_correspondence_distance * fp.max_correspondence_distance;
    const size_t nblocks = (N + REDUCE_BLOCK - 1) / REDUCE_BLOCK;
    std::vector<float> partial(nblocks * 43, 0.0f);
    std::vector<uint32_t> partial_inl(nblocks, 0);
#pragma omp parallel for schedule(static)
    for (long long blk = 0; blk < (long long)nblocks; ++blk) {
        float acc[43];
        for (int e = 0; e < 43; ++e) acc[e] = 0.0f;
        uint32_t inl = 0;
        const size_t lo = blk * REDUCE_BLOCK, hi = std::min(N, lo + REDUCE_BLOCK);
        for (size_t i = lo; i < hi; ++i) {
            if (per_point)
                for (int e = 0; e < 44; ++e) per_point[i * 44 + e] = 0.0f;
            if (nn_d2[i] > max_d2) continue;
            const size_t ti = (size_t)nn_idx[i];
            const Mat4 scov = source.covs ? to_mat4(source.covs + 16 * i) : Mat4::Identity();
            const Mat4 tcov = target.covs ? to_mat4(target.covs + 16 * ti) : Mat4::Identity();
            const Vec4 tn = target.normals ? to_vec4(target.normals + 4 * ti) : Vec4::Zero();
            float residual_norm = 0.0f, genz_weight = 1.0f;
            const KernelResult lin =
                linearize_geometry(fp.reg_type, T, to_vec4(source.points + 4 * i), scov, to_vec4(target.points + 4 * ti),
                                   tcov, tn, residual_norm, genz_alpha, genz_weight, fp.genz_planarity_threshold);
            const float w = robust_weight(fp.robust_type, residual_norm, robust_scale);
            float e = robust_error(fp.robust_type, residual_norm, robust_scale);
            if (fp.reg_type == GENZ) e = genz_weight * e;
            // rotation constraint term (registration.hpp:630-650): added to this point's totals before the reduction
            RotTerm rot;
            float rot_scale_w = 0.0f;
            if (fp.rot_enable) {
                rot = linearize_rotation_constraint(scov, tcov, T);
                const float rn_rot = std::sqrt(rot.squared_error);
                rot_scale_w = fp.rot_weight * robust_weight(fp.robust_type, rn_rot, rotation_robust_scale);
                e += fp.rot_weight * robust_error(fp.robust_type, rn_rot, rotation_robust_scale);
            }
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) {
                    float v = w * lin.H(r, c);
                    if (fp.rot_enable && r < 3 && c < 3) v += rot_scale_w * rot.H[r][c];
                    acc[r * 6 + c] += v;
                    if (per_point) per_point[i * 44 + r * 6 + c] = v;
                }
            for (int r = 0; r < 6; ++r) {
                float v = w * lin.b[r];
                if (fp.rot_enable && r < 3) v += rot_scale_w * rot.b[r];
                acc[36 + r] += v;
                if (per_point) per_point[i * 44 + 36 + r] = v;
            }
            acc[42] += e;
            if (per_point) { per_point[i * 44 + 42] = e; per_point[i * 44 + 43] = 1.0f; }
            ++inl;
        }
        for (int e = 0; e < 43; ++e) partial[blk * 43 + e] = acc[e];
        partial_inl[blk] = inl;
    }
    float tot[43];
    for (int e = 0; e < 43; ++e) tot[e] = 0.0f;
    uint32_t inl = 0;
    for (size_t blk = 0; blk < nblocks; ++blk) {
        for (int e = 0; e < 43; ++e) tot[e] += partial[blk * 43 + e];
        inl += partial_inl[blk];
    }
    Linearized out;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) out.H(r, c) = tot[r * 6 + c];
    for (int r = 0; r < 6; ++r) out.b[r] = tot[36 + r];
    out.error = tot[42];
    out.inlier = inl;
    return out;
}

// registration.hpp:678-777 (K12)
inline void error_reduce(const FactorParams& fp, const Cloud& source, const Cloud& target, const int32_t* nn_idx,
                         const float* nn_d2, const float* T_colmajor, float robust_scale, float genz_alpha, float& error,
                         uint32_t& inlier, float rotation_robust_scale = 10.0f) {
    const size_t N = source.n;
    const Mat4 T = to_mat4(T_colmajor);
    const float max_d2 = fp.max_correspondence_distance * fp.max_correspondence_distance;
    const size_t nblocks = (N + REDUCE_BLOCK - 1) / REDUCE_BLOCK;
    std::vector<float> partial(nblocks, 0.0f);
    std::vector<uint32_t> partial_inl(nblocks, 0);
#pragma omp parallel for schedule(static)
    for (long long blk = 0; blk < (long long)nblocks; ++blk) {
        float acc = 0.0f;
        uint32_t inl = 0;
        const size_t lo = blk * REDUCE_BLOCK, hi = std::min(N, lo + REDUCE_BLOCK);
        for (size_t i = lo; i < hi; ++i) {
            if (nn_d2[i] > max_d2) continue;
            const size_t ti = (size_t)nn_idx[i];
            const Mat4 scov = source.covs ? to_mat4(source.covs + 16 * i) : Mat4::Identity();
            const Mat4 tcov = target.covs ? to_mat4(target.covs + 16 * ti) : Mat4::Identity();
            const Vec4 tn = target.normals ? to_vec4(target.normals + 4 * ti) : Vec4::Zero();
            float genz_weight = 1.0f;
            const float sq = geometry_error(fp.reg_type, T, to_vec4(source.points + 4 * i), scov,
                                            to_vec4(target.points + 4 * ti), tcov, tn, genz_alpha, genz_weight,
                                            fp.genz_planarity_threshold);
            const float rn = std::sqrt(sq);
            float e = robust_error(fp.robust_type, rn, robust_scale);
            if (fp.reg_type == GENZ) e = genz_weight * e;
            if (fp.rot_enable) {  // registration.hpp:758-766
                const float rn_rot = std::sqrt(rotation_constraint_error(scov, tcov, T));
                e += fp.rot_weight * robust_error(fp.robust_type, rn_rot, rotation_robust_scale);
            }
            acc += e;
            ++inl;
        }
        partial[blk] = acc;
        partial_inl[blk] = inl;
    }
    float tot = 0.0f;
    uint32_t inl = 0;
    for (size_t blk = 0; blk < nblocks; ++blk) { tot += partial[blk]; inl += partial_inl[blk]; }
    error = tot;
    inlier = inl;
}

// registration.hpp:439-459 (K13)
inline void icp_robust_weights(const FactorParams& fp, const Cloud& source, const Cloud& target, const int32_t* nn_idx,
                               const float* nn_d2, const float* T_colmajor, float robust_scale, float genz_alpha,
                               float* out) {
    const Mat4 T = to_mat4(T_colmajor);
    const float max_d2 = fp.max_correspondence_distance * fp.max_correspondence_distance;
    for (size_t i = 0; i < source.n; ++i) {
        float w = 0.0f;
        if (nn_d2[i] <= max_d2) {
            const size_t ti = (size_t)nn_idx[i];
            const Mat4 scov = source.covs ? to_mat4(source.covs + 16 * i) : Mat4::Identity();
            const Mat4 tcov = target.covs ? to_mat4(target.covs + 16 * ti) : Mat4::Identity();
            const Vec4 tn = target.normals ? to_vec4(target.normals + 4 * ti) : Vec4::Zero();
            float gw = 1.0f;
            const float sq = geometry_error(fp.reg_type, T, to_vec4(source.points + 4 * i), scov,
                                            to_vec4(target.points + 4 * ti), tcov, tn, genz_alpha, gw,
                                            fp.genz_planarity_threshold);
            w = robust_weight(fp.robust_type, std::sqrt(sq), robust_scale);
        }
        out[i] = w;
    }
}

// ------------------------------------------------------------------ outer loop
struct RegParams : FactorParams {  // registration_params.hpp:74-114
    float gn_lambda = 1.0f;
    size_t lm_max_inner_iterations = 10;
    float lm_lambda_factor = 2.0f, lm_init_lambda = 1.0f, lm_max_lambda = 1e3f, lm_min_lambda = 1e-6f;
    int optimization_method = GAUSS_NEWTON;
    size_t max_iterations = 20;
    float crit_translation = 1e-3f, crit_rotation = 1e-3f;
    // Dogleg (registration_params.hpp:84-92)
    float dl_initial_radius = 1.0f, dl_min_radius = 1e-4f, dl_max_radius = 10.0f;
    float dl_eta1 = 0.25f, dl_eta2 = 0.75f, dl_gamma_decrease = 0.25f, dl_gamma_increase = 2.0f;
    DegenerateRegParams degenerate_reg;  // registration_params.hpp:111
};

struct RegResult {  // result.hpp:12-28
    Mat4 T = Mat4::Identity();
    bool converged = false;
    size_t iterations = 0;
    Mat6 H = Mat6::Zero();
    Vec6 b = Vec6::Zero();
    float error = std::numeric_limits<float>::max();
    uint32_t inlier = 0;
    Mat6 H_raw = Mat6::Zero();  // result.hpp: linearisation before regularisation / prior (registration.hpp:236-246)
    Vec6 b_raw = Vec6::Zero();
    float error_raw = std::numeric_limits<float>::max();
};

// The KNNBase seam (knn/knn.hpp:14-61): nearest neighbour of T*q for every source point.
using NearestFn = std::function<void(const float* queries, size_t nq, const float* T_colmajor, int32_t* idx, float* d2)>;

inline bool is_converged(const RegParams& p, const Vec6& d) {  // registration.hpp:407-410 (Eigen .norm(): plain sum)
    const float nr = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float nt = std::sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    return nr < p.crit_rotation && nt < p.crit_translation;
}
inline bool solve_linear_system(const Mat6& H, const Vec6& b, Vec6& x) {  // registration.hpp:791-801
    const Vec6 nb = scale<6, 1>(b, -1.0f);
    return ldlt6_solve(H, nb, x);
}
inline Mat6 add_lambda(const Mat6& H, float lambda) {
    Mat6 r = H;
    for (int i = 0; i < 6; ++i) r(i, i) = H(i, i) + lambda * 1.0f;
    return r;
}

// dogleg_step.hpp:35-101 (Eigen's .norm() / .dot() / H*g restated as plain ascending sums: unpinned third-party order)
struct DoglegStep {
    Vec6 p = Vec6::Zero();
    float step_norm = 0.0f, predicted_reduction = 0.0f;
};
inline float dot6(const Vec6& a, const Vec6& b) {
    float s = 0.0f;
    for (int i = 0; i < 6; ++i) s += a[i] * b[i];
    return s;
}
inline Vec6 matvec6(const Mat6& H, const Vec6& v) {
    Vec6 r;
    for (int i = 0; i < 6; ++i) {
        float s = 0.0f;
        for (int j = 0; j < 6; ++j) s += H(i, j) * v[j];
        r[i] = s;
    }
    return r;
}
inline DoglegStep compute_dogleg_step(const Mat6& H, const Vec6& g, float radius) {
    DoglegStep r;
    Vec6 p_gn = Vec6::Zero();
    float norm_gn = 0.0f, dmin = 0.0f;
    bool has_gn = false;
    {
        Vec6 sol;
        if (ldlt6_solve(H, scale<6, 1>(g, -1.0f), sol, &dmin) && dmin > 0.0f) {
            p_gn = sol;
            norm_gn = std::sqrt(dot6(p_gn, p_gn));
            has_gn = std::isfinite(norm_gn);
        }
    }
    const float g2 = dot6(g, g);
    const float gHg = dot6(g, matvec6(H, g));
    Vec6 p_sd = scale<6, 1>(g, -1.0f);
    if (gHg > std::numeric_limits<float>::epsilon()) {
        const float alpha = g2 / gHg;
        if (std::isfinite(alpha))
            for (int i = 0; i < 6; ++i) p_sd[i] = -alpha * g[i];
    }
    const float norm_sd = std::sqrt(dot6(p_sd, p_sd));
    if (has_gn && norm_gn <= radius) {
        r.p = p_gn;
        r.step_norm = norm_gn;
    } else if (norm_sd >= radius) {
        if (norm_sd > std::numeric_limits<float>::epsilon()) {
            const float sc = radius / norm_sd;
            for (int i = 0; i < 6; ++i) r.p[i] = sc * p_sd[i];
        }
        r.step_norm = radius;
    } else if (has_gn) {
        Vec6 diff;
        for (int i = 0; i < 6; ++i) diff[i] = p_gn[i] - p_sd[i];
        const float a = dot6(diff, diff);
        const float b = 2.0f * dot6(p_sd, diff);
        const float c = dot6(p_sd, p_sd) - radius * radius;
        float disc = b * b - 4.0f * a * c;
        disc = std::max(disc, 0.0f);
        float tau = 0.0f;
        if (a > std::numeric_limits<float>::epsilon()) tau = (-b + std::sqrt(disc)) / (2.0f * a);
        tau = std::clamp(tau, 0.0f, 1.0f);
        for (int i = 0; i < 6; ++i) r.p[i] = p_sd[i] + tau * diff[i];
        r.step_norm = std::sqrt(dot6(r.p, r.p));
    } else {
        r.p = p_sd;
        if (norm_sd > radius && norm_sd > std::numeric_limits<float>::epsilon()) {
            const float sc = radius / norm_sd;
            for (int i = 0; i < 6; ++i) r.p[i] *= sc;
            r.step_norm = radius;
        } else {
            r.step_norm = norm_sd;
        }
    }
    r.predicted_reduction = -(dot6(g, r.p) + 0.5f * dot6(r.p, matvec6(H, r.p)));
    return r;
}

// registration.hpp:201-276 with GN (:803-828), LM (:830-895) and Powell dogleg (:897-965).  Degenerate regularisation and MAP prior are
// default-off no-ops in the reference (degenerate_regularization.hpp:40, map_prior.hpp:15) and are not restated.
inline RegResult align(const RegParams& params, const Cloud& source, const Cloud& target, const NearestFn& nearest,
                       const float* init_T_colmajor, float opt_robust_scale = -1.0f,
                       std::vector<float>* trace_T = nullptr, const MapPrior* map_prior = nullptr,
                       float opt_rotation_robust_scale = -1.0f, std::vector<float>* trace_steps = nullptr) {
    // trace_steps (test instrumentation, not in the reference): per outer iteration {trial evaluations, accepted (1 pose moved,
    // 2 LM's stagnation exit, 0 rejected), lambda / trust radius after the iteration, result.error after it, margin}; margin =
    // how far the iteration's closest decision was from its threshold (LM: min |new_error - current_error| / |current_error|;
    // dog-leg: min(|rho - eta1|, |rho - eta2|) * predicted / |current_error|, i.e. both as the relative change of an error
    // that would flip the branch; FLT_MAX when no float comparison decided anything): a test comparing another
    // implementation's decisions with these must allow a different branch where the margin is within rounding
    RegResult result;
    result.T = to_mat4(init_T_colmajor);
    const Mat4 T_initial = result.T;
    const size_t N = source.n;
    if (N == 0) return result;
    RegParams p = params;
    if (p.robust_type != LOSS_NONE && p.robust_default_scale <= 0.0f) p.robust_type = LOSS_NONE;  // :186-192
    const float robust_scale = opt_robust_scale > 0.0f ? opt_robust_scale : p.robust_default_scale;
    const float rot_scale = opt_rotation_robust_scale > 0.0f ? opt_rotation_robust_scale : p.rot_robust_default_scale;
    const auto prior_error = [&](const Mat4& T) { return map_prior ? map_prior->prior_error(T) : 0.0f; };
    float lm_lambda = p.lm_init_lambda;
    float trust_region_radius = p.dl_initial_radius;
    std::vector<int32_t> nn_idx(N);
    std::vector<float> nn_d2(N);
    float genz_alpha = 1.0f;
    for (size_t iter = 0; iter < p.max_iterations; ++iter) {
        nearest(source.points, N, result.T.d, nn_idx.data(), nn_d2.data());
        if (p.reg_type == GENZ)
            genz_alpha = compute_genz_alpha(target, nn_idx.data(), nn_d2.data(), N, p.max_correspondence_distance,
                                            p.genz_planarity_threshold);
        Linearized lin = linearize_reduce(p, source, target, nn_idx.data(), nn_d2.data(), result.T.d, robust_scale,
                                          genz_alpha, nullptr, rot_scale);
        result.H_raw = lin.H;  // registration.hpp:244-246
        result.b_raw = lin.b;
        result.error_raw = lin.error;
        degenerate_regularize(p.degenerate_reg, lin.H, lin.b, lin.inlier, result.T, T_initial);  // :249-250
        if (map_prior) map_prior->apply(lin.H, lin.b, lin.error, result.T);                      // :253
        if (p.optimization_method == GAUSS_NEWTON) {
            Vec6 delta;
            const bool ok = solve_linear_system(add_lambda(lin.H, p.gn_lambda), lin.b, delta);
            result.converged = ok ? is_converged(p, delta) : false;
            result.T = isometry_mul(result.T, se3_exp(delta));
            result.iterations = iter;
            result.H = lin.H;
            result.b = lin.b;
            result.error = lin.error;
            result.inlier = lin.inlier;
            if (trace_steps) trace_steps->insert(trace_steps->end(), {0.0f, 1.0f, p.gn_lambda, result.error, std::numeric_limits<float>::max()});
        } else if (p.optimization_method == LEVENBERG_MARQUARDT) {
            float st_trials = 0.0f, st_accepted = 0.0f, st_margin = std::numeric_limits<float>::max();
            const float current_error = lin.error;
            float last_error = std::numeric_limits<float>::max();
            Vec6 delta;
            for (size_t i = 0; i < p.lm_max_inner_iterations; ++i) {
                const bool ok = solve_linear_system(add_lambda(lin.H, lm_lambda), lin.b, delta);
                result.converged = ok ? is_converged(p, delta) : false;
                const Mat4 new_T = isometry_mul(result.T, se3_exp(delta));
                float new_error;
                uint32_t inl;
                error_reduce(p, source, target, nn_idx.data(), nn_d2.data(), new_T.d, robust_scale, genz_alpha, new_error,
                             inl, rot_scale);
                new_error += prior_error(new_T);  // registration.hpp:854
                st_trials += 1.0f;
                st_margin = std::min(st_margin, std::fabs(new_error - current_error) / std::max(std::fabs(current_error), 1e-30f));
                if (new_error <= current_error) {
                    result.converged = is_converged(p, delta);
                    result.T = new_T;
                    result.error = new_error;
                    result.inlier = inl;
                    lm_lambda = std::clamp(lm_lambda / p.lm_lambda_factor, p.lm_min_lambda, p.lm_max_lambda);
                    st_accepted = 1.0f;
                    break;
                } else if (std::fabs(new_error - last_error) <= 1e-6f) {
                    result.converged = is_converged(p, delta);
                    result.T = new_T;
                    result.error = new_error;
                    result.inlier = inl;
                    st_accepted = 2.0f;
                    break;
                } else {
                    lm_lambda = std::clamp(lm_lambda * p.lm_lambda_factor, p.lm_min_lambda, p.lm_max_lambda);
                }
                last_error = new_error;
            }
            result.iterations = iter;
            result.H = lin.H;
            result.b = lin.b;
            if (trace_steps) trace_steps->insert(trace_steps->end(), {st_trials, st_accepted, lm_lambda, result.error, st_margin});
        } else if (p.optimization_method == POWELL_DOGLEG) {  // registration.hpp:897-965
            float st_trials = 0.0f, st_accepted = 0.0f, st_margin = std::numeric_limits<float>::max();
            result.H = lin.H;
            result.b = lin.b;
            result.error = lin.error;
            result.inlier = lin.inlier;
            result.iterations = iter;
            const auto clamp_radius = [&](float r) { return std::clamp(r, p.dl_min_radius, p.dl_max_radius); };
            trust_region_radius = clamp_radius(trust_region_radius);
            const DoglegStep dl = compute_dogleg_step(lin.H, lin.b, trust_region_radius);
            if (dl.predicted_reduction <= 0.0f) {
                trust_region_radius = clamp_radius(trust_region_radius * p.dl_gamma_decrease);
            } else {
                const Mat4 new_T = isometry_mul(result.T, se3_exp(dl.p));
                float new_error;
                uint32_t inl;
                error_reduce(p, source, target, nn_idx.data(), nn_d2.data(), new_T.d, robust_scale, genz_alpha, new_error,
                             inl, rot_scale);
                new_error += prior_error(new_T);  // registration.hpp:933
                st_trials = 1.0f;
                const float rho = (lin.error - new_error) / dl.predicted_reduction;
                st_margin = std::min(std::fabs(rho - p.dl_eta1), std::fabs(rho - p.dl_eta2)) * dl.predicted_reduction /
                            std::max(std::fabs(lin.error), 1e-30f);
                if (rho < p.dl_eta1) {
                    trust_region_radius = clamp_radius(trust_region_radius * p.dl_gamma_decrease);
                } else {
                    result.converged = is_converged(p, dl.p);
                    result.T = new_T;
                    result.error = new_error;
                    result.inlier = inl;
                    st_accepted = 1.0f;
                    if (rho > p.dl_eta2 && dl.step_norm >= trust_region_radius * 0.99f)
                        trust_region_radius = clamp_radius(trust_region_radius * p.dl_gamma_increase);
                }
            }
            if (trace_steps) trace_steps->insert(trace_steps->end(), {st_trials, st_accepted, trust_region_radius, result.error, st_margin});
        }
        if (trace_T) trace_T->insert(trace_T->end(), result.T.d, result.T.d + 16);
        if (result.converged) break;
    }
    return result;
}

// pipeline/robust.hpp:52-98 — the scales the annealing wrapper hands to the aligner, level by level: geometric from init_scale
// to min_scale over auto_scaling_iter levels; one level at default_scale when the schedule is off or its bounds are invalid.
inline std::vector<float> robust_annealing_scales(bool loss_is_none, bool auto_scale, float default_scale, float init_scale,
                                                  float min_scale, size_t auto_scaling_iter) {
    bool enable = !loss_is_none && auto_scale;
    if (enable && (min_scale <= 0.0f || min_scale >= init_scale)) enable = false;
    if (enable && auto_scaling_iter == 0) enable = false;
    const size_t levels = enable ? std::max<size_t>(1, auto_scaling_iter) : 1;
    float robust_scale = enable ? init_scale : default_scale;
    const float factor = levels > 1 ? std::pow(min_scale / init_scale, 1.0f / (float)(levels - 1)) : 1.0f;
    std::vector<float> scales;
    for (size_t level = 0; level < levels; ++level) {
        scales.push_back(robust_scale);
        robust_scale *= factor;
    }
    return scales;
}
// pipeline/robust.hpp:42-114 — geometric robust-scale annealing around align().
inline RegResult align_robust_annealing(const RegParams& params, const Cloud& source, const Cloud& target,
                                        const NearestFn& nearest, const float* init_T_colmajor, bool auto_scale,
                                        float init_scale, float min_scale, size_t auto_scaling_iter,
                                        std::vector<float>* trace_steps = nullptr) {
    RegResult result;
    result.T = to_mat4(init_T_colmajor);
    if (source.n == 0) return result;
    for (const float robust_scale : robust_annealing_scales(params.robust_type == LOSS_NONE, auto_scale, params.robust_default_scale,
                                                            init_scale, min_scale, auto_scaling_iter))
        result = align(params, source, target, nearest, result.T.d, robust_scale, nullptr, nullptr, -1.0f, trace_steps);
    return result;
}

}  // namespace oracle

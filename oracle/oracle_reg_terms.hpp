// TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the reference's default-off registration terms.
//   * rotation constraint (Jensen-Bregman LogDet divergence of the two covariances), per correspondence, inside K11/K12:
//       /root/reference/cpp/include/sycl_points/algorithms/registration/rotation_constraint.hpp:15-128
//   * degenerate regularisation (NL-Reg Tikhonov penalty on weakly observed directions), host, per iteration:
//       .../registration/degenerate_regularization.hpp:41-110
//   * MAP prior from the previous frame's Hessian, host: .../registration/map_prior.hpp:97-213
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Third-party arithmetic restated here (Eigen 3, version unpinned by the reference, no reference test pins these
// results => parity unpinned at this boundary): SelfAdjointEigenSolver<Matrix3f> (restated as a cyclic Jacobi iteration
// carried in double and rounded once: the mathematically exact decomposition; the penalty only uses v v^T, so the sign
// of an eigenvector does not matter), AngleAxisf(Matrix3f) (via the quaternion, Eigen/src/Geometry/AngleAxis.h),
// Isometry3f::inverse() (R^T, -R^T t), LDLT<Matrix6f>::solve with a matrix right-hand side (column by column) and the
// dense 6x6 products (plain ascending sums).
#pragma once

#include <cmath>

#include "oracle_math.hpp"

namespace oracle {

// ------------------------------------------------------------------ rotation constraint
inline Mat3 cov_block3(const Mat4& c) {
    Mat3 r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) r(i, j) = c(i, j);
    return r;
}
inline float logdet_floor(const Mat3& m) {  // rotation_constraint.hpp:36-38: sycl::log(sycl::fmax(det, 1e-10f))
    return std::log(std::fmax(determinant(m), 1e-10f));
}
struct RotDivergence {
    float D = 0.0f;
    Vec3 grad = Vec3::Zero();
};
// rotation_constraint.hpp:46-87 (with_grad) and :15-44 (value only: same D)
inline RotDivergence logdet_divergence(const Mat4& source_cov, const Mat4& target_cov, const Mat4& T, bool with_grad) {
    Mat3 R;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) R(i, j) = T(i, j);
    const Mat3 Cs = cov_block3(source_cov), Ct = cov_block3(target_cov);
    const Mat3 Cs_prime = matmul<3, 3, 3>(R, matmul<3, 3, 3>(Cs, transpose<3, 3>(R)));
    const Mat3 M = scale<3, 3>(add<3, 3>(Cs_prime, Ct), 0.5f);
    const float log_det_M = logdet_floor(M);
    const float log_det_ref = 0.5f * (logdet_floor(Cs) + logdet_floor(Ct));
    RotDivergence out;
    out.D = std::fmax(log_det_M - log_det_ref, 0.0f);
    if (with_grad) {
        const Mat3 M_inv = inverse(M);
        const Mat3 comm = subtract<3, 3>(matmul<3, 3, 3>(Cs_prime, M_inv), matmul<3, 3, 3>(M_inv, Cs_prime));
        Vec3 g;
        g[0] = -0.5f * (comm(2, 1) - comm(1, 2));
        g[1] = -0.5f * (comm(0, 2) - comm(2, 0));
        g[2] = -0.5f * (comm(1, 0) - comm(0, 1));
        out.grad = matvec<3, 3>(transpose<3, 3>(R), g);
    }
    return out;
}
// rotation_constraint.hpp:89-111: H (upper-left 3x3) = J J^T, b[0..2] = D J, squared_error = D^2 / 2
struct RotTerm {
    float H[3][3];
    float b[3];
    float squared_error;
};
inline RotTerm linearize_rotation_constraint(const Mat4& scov, const Mat4& tcov, const Mat4& T) {
    const RotDivergence d = logdet_divergence(scov, tcov, T, true);
    RotTerm t;
    for (int i = 0; i < 3; ++i) {
        t.b[i] = d.D * d.grad[i];
        for (int j = 0; j < 3; ++j) t.H[i][j] = d.grad[i] * d.grad[j];
    }
    t.squared_error = 0.5f * d.D * d.D;
    return t;
}
inline float rotation_constraint_error(const Mat4& scov, const Mat4& tcov, const Mat4& T) {  // :113-117
    const RotDivergence d = logdet_divergence(scov, tcov, T, false);
    return 0.5f * d.D * d.D;
}

// ------------------------------------------------------------------ small host helpers (Eigen restated)
// Symmetric 3x3 eigen-decomposition, eigenvalues ascending, unit eigenvectors in the columns of V.
inline void jacobi_eigen3(const float A_in[3][3], float vals[3], float V[3][3]) {
    double A[3][3], U[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = 0.5 * ((double)A_in[i][j] + (double)A_in[j][i]);
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-40 * diag || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double ukp = U[k][p], ukq = U[k][q];
                    U[k][p] = c * ukp - s * ukq;
                    U[k][q] = s * ukp + c * ukq;
                }
            }
    }
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (A[order[j]][order[j]] < A[order[i]][order[i]]) std::swap(order[i], order[j]);
    for (int c = 0; c < 3; ++c) {
        vals[c] = (float)A[order[c]][order[c]];
        for (int r = 0; r < 3; ++r) V[r][c] = (float)U[r][order[c]];
    }
}
inline Mat4 isometry_inverse(const Mat4& T) {  // Eigen Transform::inverse(Isometry): linear^T, -linear^T * t
    Mat4 out = Mat4::Identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out(i, j) = T(j, i);
    for (int i = 0; i < 3; ++i) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += out(i, k) * T(k, 3);
        out(i, 3) = -s;
    }
    return out;
}
inline Vec6 mat6_vec(const Mat6& A, const Vec6& v) {
    Vec6 r;
    for (int i = 0; i < 6; ++i) {
        float s = 0.0f;
        for (int k = 0; k < 6; ++k) s += A(i, k) * v[k];
        r[i] = s;
    }
    return r;
}

// ------------------------------------------------------------------ degenerate regularisation
struct DegenerateRegParams {  // degenerate_regularization.hpp:41-46
    int type = 0;             // 0 none, 1 nl_reg
    float rot_eigenvalue_threshold = 10.0f, trans_eigenvalue_threshold = 1.0f, base_factor = 1.0f;
};
// degenerate_regularization.hpp:60-110: H += P, b += P * se3_log(T_init^-1 * T_cur)
inline void degenerate_regularize(const DegenerateRegParams& p, Mat6& H, Vec6& b, uint32_t inlier, const Mat4& current,
                                  const Mat4& initial) {
    if (inlier == 0 || p.type != 1) return;
    const float lambda = p.base_factor * (float)inlier;
    float blk[3][3], vals_r[3], V_r[3][3], vals_t[3], V_t[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) blk[i][j] = H(i, j);
    jacobi_eigen3(blk, vals_r, V_r);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) blk[i][j] = H(3 + i, 3 + j);
    jacobi_eigen3(blk, vals_t, V_t);
    Mat6 P = Mat6::Zero();
    if (p.rot_eigenvalue_threshold > 0.0f)
        for (int i = 0; i < 3; ++i)
            if (vals_r[i] / (float)inlier < p.rot_eigenvalue_threshold)
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) P(r, c) += lambda * (V_r[r][i] * V_r[c][i]);
    if (p.trans_eigenvalue_threshold > 0.0f)
        for (int i = 0; i < 3; ++i)
            if (vals_t[i] / (float)inlier < p.trans_eigenvalue_threshold)
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) P(3 + r, 3 + c) += lambda * (V_t[r][i] * V_t[c][i]);
    const Vec6 twist = se3_log(isometry_mul(isometry_inverse(initial), current));
    const Vec6 Pt = mat6_vec(P, twist);
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) H(i, j) += P(i, j);
        b[i] += Pt[i];
    }
}

// ------------------------------------------------------------------ MAP prior
struct MapPriorParams {  // map_prior.hpp:14-35
    bool enabled = false;
    float rot_vel_sigma = 1.0f, trans_vel_sigma = 1.0f, rot_base_sigma = 3.16e-2f, trans_base_sigma = 1e-2f;
};
struct MapPrior {
    MapPriorParams params;
    bool has_prior = false;
    Mat6 Omega = Mat6::Zero();
    Mat4 T_pred_inv = Mat4::Identity();
    bool is_active() const { return params.enabled && has_prior; }

    // map_prior.hpp:97-174
    void update(const Mat6& H_raw_prev, float error_raw_prev, uint32_t inlier_prev, const Mat4& T_prev, const Mat4& T_pred) {
        has_prior = false;
        if (!params.enabled) return;
        const float dof = 3.0f * (float)inlier_prev - 6.0f;
        if (dof <= 0.0f) return;
        if (!std::isfinite(error_raw_prev) || error_raw_prev < 0.0f) return;
        const float s_sq = std::max(1.0f, 2.0f * error_raw_prev / dof);
        Mat6 Hc;
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) Hc(i, j) = H_raw_prev(i, j) / s_sq;
        float Rrel[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float s = 0.0f;
                for (int k = 0; k < 3; ++k) s += T_prev(k, i) * T_pred(k, j);
                Rrel[i][j] = s;
            }
        // AngleAxisf(R_rel): through the quaternion (Eigen/src/Geometry/AngleAxis.h operator=(QuaternionBase))
        Mat3 Rm;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rm(i, j) = Rrel[i][j];
        Vec4 q = rotation_matrix_to_quaternion(Rm);  // x, y, z, w
        float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        if (q[3] < 0.0f) n = -n;
        float drot[3] = {0.0f, 0.0f, 0.0f};
        if (n != 0.0f) {
            const float angle = 2.0f * std::atan2(n, std::fabs(q[3]));
            for (int i = 0; i < 3; ++i) drot[i] = (q[i] / n) * angle;
        }
        float dtr[3];
        for (int i = 0; i < 3; ++i) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k) s += T_pred(k, i) * (T_pred(k, 3) - T_prev(k, 3));
            dtr[i] = s;
        }
        const float rv = params.rot_vel_sigma * params.rot_vel_sigma, tv = params.trans_vel_sigma * params.trans_vel_sigma;
        const float rb = params.rot_base_sigma * params.rot_base_sigma, tb = params.trans_base_sigma * params.trans_base_sigma;
        float Rdiag[6];
        for (int i = 0; i < 3; ++i) {
            Rdiag[i] = 1.0f / (std::fabs(drot[i]) * rv + rb);
            Rdiag[3 + i] = 1.0f / (std::fabs(dtr[i]) * tv + tb);
        }
        // H_curr = Ad^T Hc Ad with Ad = diag(R_rel, R_rel)
        Mat6 Ad = Mat6::Zero(), tmp = Mat6::Zero(), Hcur = Mat6::Zero();
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { Ad(i, j) = Rrel[i][j]; Ad(3 + i, 3 + j) = Rrel[i][j]; }
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                float s = 0.0f;
                for (int k = 0; k < 6; ++k) s += Ad(k, i) * Hc(k, j);
                tmp(i, j) = s;
            }
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                float s = 0.0f;
                for (int k = 0; k < 6; ++k) s += tmp(i, k) * Ad(k, j);
                Hcur(i, j) = s;
            }
        Mat6 HR = Hcur;
        for (int i = 0; i < 6; ++i) HR(i, i) += Rdiag[i];
        // Omega = R - R (H + R)^-1 R, one column of R at a time
        Mat6 Om = Mat6::Zero();
        for (int c = 0; c < 6; ++c) {
            Vec6 rhs = Vec6::Zero(), x;
            rhs[c] = Rdiag[c];
            if (!ldlt6_solve(HR, rhs, x)) return;
            for (int r = 0; r < 6; ++r) Om(r, c) = (r == c ? Rdiag[c] : 0.0f) - Rdiag[r] * x[r];
        }
        Omega = Om;
        T_pred_inv = isometry_inverse(T_pred);
        has_prior = true;
    }
    // map_prior.hpp:181-193
    void apply(Mat6& H, Vec6& b, float& error, const Mat4& T_est) const {
        if (!is_active()) return;
        const Vec6 e = se3_log(isometry_mul(T_pred_inv, T_est));
        const Vec6 Oe = mat6_vec(Omega, e);
        float dot = 0.0f;
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) H(i, j) += Omega(i, j);
            b[i] += Oe[i];
            dot += e[i] * Oe[i];
        }
        error += 0.5f * dot;
    }
    // map_prior.hpp:197-201
    float prior_error(const Mat4& T_est) const {
        if (!is_active()) return 0.0f;
        const Vec6 e = se3_log(isometry_mul(T_pred_inv, T_est));
        const Vec6 Oe = mat6_vec(Omega, e);
        float dot = 0.0f;
        for (int i = 0; i < 6; ++i) dot += e[i] * Oe[i];
        return 0.5f * dot;
    }
};

}  // namespace oracle

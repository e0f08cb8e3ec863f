// Host-side pose-space terms of Registration::align that act on the reduced 6x6 system between the device reduction and
// the solve (registration.hpp:236-253): NL-Reg degenerate regularisation and the MAP prior, plus se3_log.
// Plain host arithmetic (the reference runs these on the host with Eigen); H is row-major as in sp_linearized.
#include <cmath>
#include <cstring>

#include "sp_common.h"
#include "sp_math.h"

void sp_set_error(const char* msg);

namespace sp {
namespace {

// rotation_matrix_to_quaternion (eigen_utils.hpp:774-806), quaternion as x,y,z,w
void rot_to_quat(const float R[3][3], float q[4]) {
    const float tr = R[0][0] + R[1][1] + R[2][2];
    if (tr > 0.0f) {
        const float S = sqrtf(tr + 1.0f) * 2.0f;
        q[0] = (R[2][1] - R[1][2]) / S; q[1] = (R[0][2] - R[2][0]) / S; q[2] = (R[1][0] - R[0][1]) / S; q[3] = 0.25f * S;
    } else if (R[0][0] > R[1][1] && R[0][0] > R[2][2]) {
        const float S = sqrtf(1.0f + R[0][0] - R[1][1] - R[2][2]) * 2.0f;
        q[0] = 0.25f * S; q[1] = (R[0][1] + R[1][0]) / S; q[2] = (R[0][2] + R[2][0]) / S; q[3] = (R[2][1] - R[1][2]) / S;
    } else if (R[1][1] > R[2][2]) {
        const float S = sqrtf(1.0f + R[1][1] - R[0][0] - R[2][2]) * 2.0f;
        q[0] = (R[0][1] + R[1][0]) / S; q[1] = 0.25f * S; q[2] = (R[1][2] + R[2][1]) / S; q[3] = (R[0][2] - R[2][0]) / S;
    } else {
        const float S = sqrtf(1.0f + R[2][2] - R[0][0] - R[1][1]) * 2.0f;
        q[0] = (R[0][2] + R[2][0]) / S; q[1] = (R[1][2] + R[2][1]) / S; q[2] = 0.25f * S; q[3] = (R[1][0] - R[0][1]) / S;
    }
}

// so3_log (eigen_utils.hpp:948-986)
void so3_log(const float q_in[4], float w_out[3]) {
    float q[4];
    const float n = sqrtf(fmaf(q_in[3], q_in[3], fmaf(q_in[2], q_in[2], fmaf(q_in[1], q_in[1], q_in[0] * q_in[0]))));
    for (int i = 0; i < 4; ++i) q[i] = (n < 1e-6f) ? 0.0f : q_in[i] * (1.0f / n);
    if (q[3] < 0.0f)
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    const float w = q[3];
    const float vn = sqrtf(chain3(q[0], q[0], q[1], q[1], q[2], q[2]));
    float scale;
    if (vn < 1e-6f) scale = 2.0f / w * (1.0f + vn * vn / (6.0f * w * w));
    else if (fabsf(w) < 1e-6f) scale = kPi / vn;
    else scale = 2.0f * atan2f(vn, fabsf(w)) / vn;
    for (int i = 0; i < 3; ++i) w_out[i] = scale * q[i];
}

// se3_log (eigen_utils.hpp:991-1034), rotation-first twist
void se3_log(const Rigid& T, float a[6]) {
    float q[4], w[3];
    rot_to_quat(T.R, q);
    so3_log(q, w);
    const float theta = sqrtf(chain3(w[0], w[0], w[1], w[1], w[2], w[2]));
    const float O[3][3] = {{0.0f, -w[2], w[1]}, {w[2], 0.0f, -w[0]}, {-w[1], w[0], 0.0f}};
    float Vi[3][3];
    float coeff = 0.0f;
    if (!(theta < 1e-6f)) {
        const float half = 0.5f * theta;
        coeff = (1.0f - theta * cosf(half) / (2.0f * sinf(half))) / (theta * theta);
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float o2 = 0.0f;
            for (int k = 0; k < 3; ++k) o2 += O[i][k] * O[k][j];
            Vi[i][j] = ((i == j) ? 1.0f : 0.0f) - 0.5f * O[i][j];
            if (!(theta < 1e-6f)) Vi[i][j] += coeff * o2;
        }
    for (int i = 0; i < 3; ++i) {
        a[i] = w[i];
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += Vi[i][k] * T.t[k];
        a[3 + i] = s;
    }
}

Rigid rigid_inverse(const Rigid& T) {  // Isometry3f::inverse(): R^T, -R^T t
    Rigid o;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) o.R[i][j] = T.R[j][i];
    }
    for (int i = 0; i < 3; ++i) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += o.R[i][k] * T.t[k];
        o.t[i] = -s;
    }
    return o;
}

// Symmetric 3x3 eigen-pairs by cyclic Jacobi rotations (stands in for Eigen::SelfAdjointEigenSolver<Matrix3f>,
// degenerate_regularization.hpp:71-78): ascending eigenvalues, unit eigenvectors in the columns of V.
void eigen_sym3(const float A_in[3][3], float lam[3], float V[3][3]) {
    float A[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[i][j] = 0.5f * (A_in[i][j] + A_in[j][i]);
            V[i][j] = (i == j) ? 1.0f : 0.0f;
        }
    for (int sweep = 0; sweep < 32; ++sweep) {
        const float off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const float dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (!(off > 1e-18f * dg)) break;
        static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
        for (int e = 0; e < 3; ++e) {
            const int p = P[e], q = Q[e];
            if (A[p][q] == 0.0f) continue;
            const float tau = (A[q][q] - A[p][p]) / (2.0f * A[p][q]);
            const float t = copysignf(1.0f, tau) / (fabsf(tau) + sqrtf(fmaf(tau, tau, 1.0f)));
            const float c = 1.0f / sqrtf(fmaf(t, t, 1.0f)), s = t * c;
            for (int k = 0; k < 3; ++k) {
                const float x = A[k][p], y = A[k][q];
                A[k][p] = c * x - s * y;
                A[k][q] = s * x + c * y;
            }
            for (int k = 0; k < 3; ++k) {
                const float x = A[p][k], y = A[q][k];
                A[p][k] = c * x - s * y;
                A[q][k] = s * x + c * y;
            }
            for (int k = 0; k < 3; ++k) {
                const float x = V[k][p], y = V[k][q];
                V[k][p] = c * x - s * y;
                V[k][q] = s * x + c * y;
            }
        }
    }
    int idx[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (A[idx[j]][idx[j]] < A[idx[i]][idx[i]]) { const int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    float Vs[3][3];
    for (int c = 0; c < 3; ++c) {
        lam[c] = A[idx[c]][idx[c]];
        for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][idx[c]];
    }
    std::memcpy(V, Vs, sizeof(Vs));
}

void add_weak_directions(const float* H, int o, float inlier, float threshold, float lambda, float* P) {
    if (!(threshold > 0.0f)) return;
    float blk[3][3], lam[3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) blk[i][j] = H[(o + i) * 6 + (o + j)];
    eigen_sym3(blk, lam, V);
    for (int e = 0; e < 3; ++e)
        if (lam[e] / inlier < threshold)
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) P[(o + r) * 6 + (o + c)] += lambda * (V[r][e] * V[c][e]);
}

}  // namespace
}  // namespace sp

extern "C" void sp_se3_log_host(const float* T16, float* twist6) { sp::se3_log(sp::load_rigid_colmajor(T16), twist6); }

extern "C" int sp_degenerate_regularize_host(const sp_degenerate_reg_params* params, float* H36_rowmajor, float* b6,
                                             uint32_t inlier, const float* T_current16, const float* T_initial16) {
    if (!params || !H36_rowmajor || !b6 || !T_current16 || !T_initial16) return SP_ERR_INVALID_ARGUMENT;
    if (inlier == 0 || params->type == SP_DEGENERATE_REG_NONE) return SP_OK;
    if (params->type != SP_DEGENERATE_REG_NL_REG) {
        sp_set_error("[DegenerateRegularization] unknown type");
        return SP_ERR_INVALID_ARGUMENT;
    }
    const float n = (float)inlier, lambda = params->base_factor * n;
    float P[36] = {};
    sp::add_weak_directions(H36_rowmajor, 0, n, params->rot_eigenvalue_threshold, lambda, P);
    sp::add_weak_directions(H36_rowmajor, 3, n, params->trans_eigenvalue_threshold, lambda, P);
    float twist[6];
    sp::se3_log(sp::rigid_mul(sp::rigid_inverse(sp::load_rigid_colmajor(T_initial16)), sp::load_rigid_colmajor(T_current16)),
                twist);
    for (int i = 0; i < 6; ++i) {
        float s = 0.0f;
        for (int k = 0; k < 6; ++k) s += P[i * 6 + k] * twist[k];
        b6[i] += s;
        for (int j = 0; j < 6; ++j) H36_rowmajor[i * 6 + j] += P[i * 6 + j];
    }
    return SP_OK;
}

extern "C" int sp_map_prior_update_host(const sp_map_prior_params* params, const float* H_raw36_rowmajor, float error_raw,
                                        uint32_t inlier, const float* T_prev16, const float* T_pred16,
                                        sp_map_prior_state* state) {
    if (!params || !H_raw36_rowmajor || !T_prev16 || !T_pred16 || !state) return SP_ERR_INVALID_ARGUMENT;
    state->has_prior = 0;
    if (!params->enabled) return SP_OK;
    const float dof = 3.0f * (float)inlier - 6.0f;
    if (dof <= 0.0f) return SP_OK;
    if (!(fabsf(error_raw) <= FLT_MAX) || error_raw < 0.0f) return SP_OK;
    const float s_sq = fmaxf(1.0f, 2.0f * error_raw / dof);
    const sp::Rigid prev = sp::load_rigid_colmajor(T_prev16), pred = sp::load_rigid_colmajor(T_pred16);
    float Rrel[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k) s += prev.R[k][i] * pred.R[k][j];
            Rrel[i][j] = s;
        }
    // predicted inter-frame motion in the body frame: AngleAxisf(R_rel) via its quaternion, and R_pred^T (t_pred - t_prev)
    float q[4], drot[3] = {0.0f, 0.0f, 0.0f}, dtr[3];
    sp::rot_to_quat(Rrel, q);
    float vn = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (q[3] < 0.0f) vn = -vn;
    if (vn != 0.0f) {
        const float angle = 2.0f * atan2f(vn, fabsf(q[3]));
        for (int i = 0; i < 3; ++i) drot[i] = (q[i] / vn) * angle;
    }
    for (int i = 0; i < 3; ++i) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += pred.R[k][i] * (pred.t[k] - prev.t[k]);
        dtr[i] = s;
    }
    float Rd[6];  // R = Q^-1, diagonal
    for (int i = 0; i < 3; ++i) {
        Rd[i] = 1.0f / (fabsf(drot[i]) * (params->rot_vel_sigma * params->rot_vel_sigma) +
                        params->rot_base_sigma * params->rot_base_sigma);
        Rd[3 + i] = 1.0f / (fabsf(dtr[i]) * (params->trans_vel_sigma * params->trans_vel_sigma) +
                            params->trans_base_sigma * params->trans_base_sigma);
    }
    // H_curr = Ad^T (H_raw / s^2) Ad, Ad = diag(R_rel, R_rel): block by block
    float Hc[36], tmp[36];
    auto Ad = [&](int r, int c) { return ((r < 3) == (c < 3)) ? Rrel[r % 3][c % 3] : 0.0f; };
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 6; ++k) s += Ad(k, i) * (H_raw36_rowmajor[k * 6 + j] / s_sq);
            tmp[i * 6 + j] = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 6; ++k) s += tmp[i * 6 + k] * Ad(k, j);
            Hc[i * 6 + j] = s;
        }
    for (int i = 0; i < 6; ++i) Hc[i * 6 + i] += Rd[i];
    // Omega = R - R (H + R)^-1 R (matrix inversion lemma; H + R is positive definite)
    sp::LdltScratch w;
    for (int c = 0; c < 6; ++c) {
        float rhs[6] = {}, x[6];
        rhs[c] = Rd[c];
        if (!sp::ldlt6_solve(Hc, rhs, x, w)) return SP_OK;  // no prior this frame (map_prior.hpp:166)
        for (int r = 0; r < 6; ++r) state->omega[r * 6 + c] = (r == c ? Rd[c] : 0.0f) - Rd[r] * x[r];
    }
    sp::store_rigid_colmajor(sp::rigid_inverse(pred), state->T_pred_inv);
    state->has_prior = 1;
    return SP_OK;
}

extern "C" float sp_map_prior_apply_host(const sp_map_prior_state* state, const float* T_est16, float* H36_rowmajor,
                                         float* b6, float* error) {
    if (!state || !state->has_prior || !T_est16) return 0.0f;
    float e[6], Oe[6];
    sp::se3_log(sp::rigid_mul(sp::load_rigid_colmajor(state->T_pred_inv), sp::load_rigid_colmajor(T_est16)), e);
    float cost = 0.0f;
    for (int i = 0; i < 6; ++i) {
        float s = 0.0f;
        for (int k = 0; k < 6; ++k) s += state->omega[i * 6 + k] * e[k];
        Oe[i] = s;
    }
    for (int i = 0; i < 6; ++i) cost += e[i] * Oe[i];
    cost *= 0.5f;
    if (H36_rowmajor && b6) {
        for (int i = 0; i < 6; ++i) {
            b6[i] += Oe[i];
            for (int j = 0; j < 6; ++j) H36_rowmajor[i * 6 + j] += state->omega[i * 6 + j];
        }
        if (error) *error += cost;
    }
    return cost;
}

// Small fixed-size float math for the gfx950 kernels and the host-side solver.
//
// These are hand-specialised forms of the reference's device math
// (sycl_points/utils/eigen_utils.hpp): 3x3 blocks only, structural zeros of the padded 4x4 types removed,
// everything in registers. Where the reference chains sycl::fma the same chain (same operand order) is kept
// so that results are bit-identical to an IEEE evaluation of the reference expression; the file must be
// compiled with -ffp-contract=off so that no other multiply-add is fused.
// Matrices handed in from the API are column-major (Eigen .data() order): element (i,j) = m[j*4+i].
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>

#define SP_HD __host__ __device__ __forceinline__

namespace sp {

constexpr float kPi = 3.14159265358979323846f;  // eigen_utils.hpp:24

// 3-term fma chain starting from 0, the shape of every eigen_utils::multiply / dot inner loop
// (eigen_utils.hpp:88-127,245-253): fma(a2,b2, fma(a1,b1, fma(a0,b0, 0))).
SP_HD float chain3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return fmaf(a2, b2, fmaf(a1, b1, fmaf(a0, b0, 0.0f)));
}

// sycl::min / sycl::max (SYCL 2020 4.17.7): (y < x) ? y : x  /  (x < y) ? y : x
SP_HD float sycl_min(float x, float y) { return (y < x) ? y : x; }
SP_HD float sycl_max(float x, float y) { return (x < y) ? y : x; }

struct Mat3 {  // row-major registers: m[i][j]
    float m[3][3];
};
struct Vec3 {
    float v[3];
};
// Rigid transform held as the 12 meaningful entries of a column-major 4x4 (last row assumed 0 0 0 1).
struct Rigid {
    float R[3][3];
    float t[3];
};

SP_HD Rigid load_rigid_colmajor(const float* T) {
    Rigid r;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) r.R[i][j] = T[j * 4 + i];
        r.t[i] = T[12 + i];
    }
    return r;
}

// transform::kernel::transform_point (common/transform.hpp:31-37) for a point with w == 1:
// row i = fma(T(i,3), 1, fma(T(i,2), z, fma(T(i,1), y, fma(T(i,0), x, 0)))).
SP_HD void transform_point(const Rigid& T, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = fmaf(T.t[0], 1.0f, chain3(T.R[0][0], x, T.R[0][1], y, T.R[0][2], z));
    oy = fmaf(T.t[1], 1.0f, chain3(T.R[1][0], x, T.R[1][1], y, T.R[1][2], z));
    oz = fmaf(T.t[2], 1.0f, chain3(T.R[2][0], x, T.R[2][1], y, T.R[2][2], z));
}

// Squared distance as the KD-tree forms it (knn/kdtree.hpp:509-511): subtract then dot<4>; the w term is 0.
SP_HD float dist2(float qx, float qy, float qz, float tx, float ty, float tz) {
    const float dx = qx - tx, dy = qy - ty, dz = qz - tz;
    return chain3(dx, dx, dy, dy, dz, dz);
}

SP_HD float determinant(const Mat3& A) {  // eigen_utils.hpp:303-307
    return fmaf(A.m[0][0], fmaf(A.m[1][1], A.m[2][2], -A.m[1][2] * A.m[2][1]),
                fmaf(-A.m[0][1], fmaf(A.m[1][0], A.m[2][2], -A.m[1][2] * A.m[2][0]),
                     A.m[0][2] * fmaf(A.m[1][0], A.m[2][1], -A.m[1][1] * A.m[2][0])));
}

SP_HD Mat3 inverse(const Mat3& s) {  // eigen_utils.hpp:403-423 (Zero when |det| < 1e-6)
    Mat3 r;
    const float det = determinant(s);
    if (fabsf(det) < 1e-6f) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) r.m[i][j] = 0.0f;
        return r;
    }
    const float invDet = 1.0f / det;
    r.m[0][0] = fmaf(s.m[1][1], s.m[2][2], -s.m[1][2] * s.m[2][1]) * invDet;
    r.m[1][0] = fmaf(s.m[1][2], s.m[2][0], -s.m[1][0] * s.m[2][2]) * invDet;
    r.m[2][0] = fmaf(s.m[1][0], s.m[2][1], -s.m[1][1] * s.m[2][0]) * invDet;
    r.m[0][1] = fmaf(s.m[0][2], s.m[2][1], -s.m[0][1] * s.m[2][2]) * invDet;
    r.m[1][1] = fmaf(s.m[0][0], s.m[2][2], -s.m[0][2] * s.m[2][0]) * invDet;
    r.m[2][1] = fmaf(s.m[0][1], s.m[2][0], -s.m[0][0] * s.m[2][1]) * invDet;
    r.m[0][2] = fmaf(s.m[0][1], s.m[1][2], -s.m[0][2] * s.m[1][1]) * invDet;
    r.m[1][2] = fmaf(s.m[0][2], s.m[1][0], -s.m[0][0] * s.m[1][2]) * invDet;
    r.m[2][2] = fmaf(s.m[0][0], s.m[1][1], -s.m[0][1] * s.m[1][0]) * invDet;
    return r;
}

// C = A * B with the reference's accumulation (eigen_utils.hpp:88-105): per element a k-ascending fma chain.
SP_HD Mat3 matmul(const Mat3& A, const Mat3& B) {
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[i][j] = chain3(A.m[i][0], B.m[0][j], A.m[i][1], B.m[1][j], A.m[i][2], B.m[2][j]);
    return r;
}
// C = A * B^T
SP_HD Mat3 matmul_bt(const Mat3& A, const Mat3& B) {
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[i][j] = chain3(A.m[i][0], B.m[j][0], A.m[i][1], B.m[j][1], A.m[i][2], B.m[j][2]);
    return r;
}

// Analytic symmetric 3x3 eigen-decomposition, eigenvalues ascending, eigenvectors in columns
// (eigen_utils.hpp:443-562, same operation order). acosf/cosf/cbrtf come from the device math library,
// so results agree with a host evaluation to a few ulp, not bit for bit.
SP_HD void symmetric_eigen3(const Mat3& A, float ev[3], Mat3& V) {
    float max_abs = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) max_abs = fmaxf(max_abs, fabsf(A.m[i][j]));
    if (max_abs < FLT_MIN) {
        ev[0] = ev[1] = ev[2] = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) V.m[i][j] = (i == j) ? 1.0f : 0.0f;
        return;
    }
    const float scale_inv = 1.0f / max_abs;
    Mat3 s;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s.m[i][j] = A.m[i][j] * scale_inv;

    const float c2 = -(((0.0f + s.m[0][0]) + s.m[1][1]) + s.m[2][2]);
    const float c1 = fmaf(s.m[0][0], s.m[1][1], fmaf(s.m[0][0], s.m[2][2], s.m[1][1] * s.m[2][2])) -
                     fmaf(s.m[0][1], s.m[1][0], fmaf(s.m[0][2], s.m[2][0], s.m[1][2] * s.m[2][1]));
    const float c0 = -determinant(s);
    const float p = c1 - c2 * c2 / 3.0f;
    const float q = 2.0f * c2 * c2 * c2 / 27.0f - c2 * c1 / 3.0f + c0;
    const float disc = 4.0f * p * p * p + 27.0f * q * q;

    if (fabsf(disc) <= FLT_EPSILON) {
        const float u = q >= 0 ? -cbrtf(q / 2.0f) : cbrtf(-q / 2.0f);
        ev[0] = 2.0f * u - c2 / 3.0f;
        ev[1] = ev[2] = -u - c2 / 3.0f;
    } else {
        const float sq = sqrtf(-p / 3.0f);
        const float cosv = sycl_max(-1.0f, sycl_min(1.0f, -q / (2.0f * sq * sq * sq)));
        float phi = fabsf(p) < FLT_EPSILON ? 0.0f : acosf(cosv);
        if (phi < 0.0f) phi += kPi;
        ev[0] = fmaf(2.0f * sq, cosf(phi / 3.0f), -c2 / 3.0f);
        ev[2] = fmaf(2.0f * sq, cosf((phi + 4.0f * kPi) / 3.0f), -c2 / 3.0f);
        ev[1] = fmaf(2.0f * sq, cosf((phi + 2.0f * kPi) / 3.0f), -c2 / 3.0f);
    }
    float tmp;
    if (ev[0] > ev[1]) { tmp = ev[0]; ev[0] = ev[1]; ev[1] = tmp; }
    if (ev[1] > ev[2]) { tmp = ev[1]; ev[1] = ev[2]; ev[2] = tmp; }
    if (ev[0] > ev[1]) { tmp = ev[1]; ev[1] = ev[0]; ev[0] = tmp; }

#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // M = s - I*ev[k]: off-diagonals are s(i,j) - 0*ev = s(i,j)
        const float M00 = s.m[0][0] - ev[k], M11 = s.m[1][1] - ev[k], M22 = s.m[2][2] - ev[k];
        const float M01 = s.m[0][1] - 0.0f * ev[k], M02 = s.m[0][2] - 0.0f * ev[k], M10 = s.m[1][0] - 0.0f * ev[k];
        const float M12 = s.m[1][2] - 0.0f * ev[k], M20 = s.m[2][0] - 0.0f * ev[k], M21 = s.m[2][1] - 0.0f * ev[k];
        const float m00 = fmaf(M11, M22, -M12 * M21);
        const float m01 = fmaf(M12, M20, -M10 * M22);
        const float m02 = fmaf(M10, M21, -M11 * M20);
        const float m10 = fmaf(M02, M21, -M01 * M22);
        const float m11 = fmaf(M00, M22, -M02 * M20);
        const float m12 = fmaf(M01, M20, -M00 * M21);
        const float m20 = fmaf(M01, M12, -M02 * M11);
        const float m21 = fmaf(M02, M10, -M00 * M12);
        const float m22 = fmaf(M00, M11, -M01 * M10);
        const float s0 = fmaf(m00, m00, fmaf(m10, m10, m20 * m20));
        const float s1 = fmaf(m01, m01, fmaf(m11, m11, m21 * m21));
        const float s2 = fmaf(m02, m02, fmaf(m12, m12, m22 * m22));
        float v0, v1, v2;
        if (s0 >= s1 && s0 >= s2) {
            v0 = m00; v1 = m10; v2 = m20;
        } else if (s1 >= s0 && s1 >= s2) {
            v0 = m01; v1 = m11; v2 = m21;
        } else {
            v0 = m02; v1 = m12; v2 = m22;
        }
        float norm_sq = fmaf(v2, v2, fmaf(v1, v1, v0 * v0));
        if (norm_sq < FLT_MIN) {
            v0 = 1.0f; v1 = 0.0f; v2 = 0.0f;
            norm_sq = 1.0f;
        }
        const float inv_len = 1.0f / sqrtf(norm_sq);
        V.m[0][k] = v0 * inv_len;
        V.m[1][k] = v1 * inv_len;
        V.m[2][k] = v2 * inv_len;
    }
    ev[0] *= max_abs; ev[1] *= max_abs; ev[2] *= max_abs;
}

// covariance::kernel::update_covariance_plane (feature/covariance.hpp:67-74): V diag(1e-3,1,1) V^T.
// (V*diag) scales column 0 by 1e-3 exactly as the fma product with the diagonal matrix does.
SP_HD Mat3 plane_regularize(const Mat3& C) {
    float ev[3];
    Mat3 V;
    symmetric_eigen3(C, ev, V);
    Mat3 X;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        X.m[i][0] = fmaf(V.m[i][0], 1e-3f, 0.0f);
        X.m[i][1] = V.m[i][1];
        X.m[i][2] = V.m[i][2];
    }
    return matmul_bt(X, V);
}

// covariance::kernel::normalize_covariance (feature/covariance.hpp:76-95): eigenvalues of 1e3*C scaled by the largest,
// the two smaller ones clamped to [1e-3, 1]; identity when the largest is below FLT_MIN.
SP_HD Mat3 normalize_cov(const Mat3& C) {
    Mat3 S;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) S.m[i][j] = C.m[i][j] * 1e3f;
    float ev[3];
    Mat3 V;
    symmetric_eigen3(S, ev, V);
    const float mx = ev[2];
    Mat3 X;
    if (mx < FLT_MIN) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) X.m[i][j] = (i == j) ? 1.0f : 0.0f;
        return X;
    }
    auto clamp01 = [](float v) { return v < 1e-3f ? 1e-3f : (1.0f < v ? 1.0f : v); };  // std::clamp(v, 1e-3f, 1.0f)
    const float e0 = clamp01(ev[0] / mx), e1 = clamp01(ev[1] / mx);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        X.m[i][0] = fmaf(V.m[i][0], e0, 0.0f);
        X.m[i][1] = fmaf(V.m[i][1], e1, 0.0f);
        X.m[i][2] = V.m[i][2];
    }
    return matmul_bt(X, V);
}

// ---------------------------------------------------------------- robust kernels (robust/robust.hpp:56-114)
enum Loss : int { LOSS_NONE = 0, LOSS_HUBER = 1, LOSS_TUKEY = 2, LOSS_CAUCHY = 3, LOSS_GEMAN_MCCLURE = 4 };

template <int LOSS>
SP_HD float robust_weight(float r, float scale) {
    if (LOSS == LOSS_NONE) return 1.0f;
    if (r <= 1e-8f) return 1.0f;
    const float nr = r / scale;
    if (LOSS == LOSS_HUBER) return sycl_min(1.0f, 1.0f / nr);
    if (LOSS == LOSS_TUKEY) {
        if (nr >= 1.0f) return 0.0f;
        const float x = nr * nr;
        const float f = 1.0f - x;
        return f * f;
    }
    if (LOSS == LOSS_CAUCHY) {
        const float x = nr * nr;
        return 1.0f / (1.0f + x);
    }
    const float x = nr * nr;  // GEMAN_MCCLURE
    const float den = 1.0f + x;
    return 1.0f / (den * den);
}
template <int LOSS>
SP_HD float robust_error(float r, float s) {
    if (LOSS == LOSS_HUBER) return r <= s ? 0.5f * r * r : s * (r - 0.5f * s);
    if (LOSS == LOSS_TUKEY)
        return r <= s ? (s * s / 6.0f) * (1.0f - powf(1.0f - ((r * r) / (s * s)), 3.0f)) : s * s / 6.0f;
    if (LOSS == LOSS_CAUCHY) return 0.5f * s * s * logf(1.0f + ((r * r) / (s * s)));
    if (LOSS == LOSS_GEMAN_MCCLURE) return 0.5f * (s * s * r * r) / (s * s + r * r);
    return 0.5f * r * r;
}

// ---------------------------------------------------------------- SE(3) (eigen_utils.hpp:808-943)
SP_HD void so3_exp(const float w[3], float q[4]) {  // quaternion x,y,z,w
    const float theta_sq = chain3(w[0], w[0], w[1], w[1], w[2], w[2]);
    float imag, real;
    if (theta_sq < 1e-6f) {
        const float t4 = theta_sq * theta_sq;
        imag = 0.5f - 1.0f / 48.0f * theta_sq + 1.0f / 3840.0f * t4;
        real = 1.0f - 1.0f / 8.0f * theta_sq + 1.0f / 384.0f * t4;
    } else {
        const float theta = sqrtf(theta_sq);
        const float half = 0.5f * theta;
        imag = sinf(half) / theta;
        real = cosf(half);
    }
    q[0] = imag * w[0]; q[1] = imag * w[1]; q[2] = imag * w[2]; q[3] = real;
}
SP_HD void quat_to_rot(const float qt[4], float R[3][3]) {
    const float x = qt[0], y = qt[1], z = qt[2], w = qt[3];
    const float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0][0] = 1.0f - 2.0f * (y2 + z2); R[0][1] = 2.0f * (xy - wz);        R[0][2] = 2.0f * (xz + wy);
    R[1][0] = 2.0f * (xy + wz);        R[1][1] = 1.0f - 2.0f * (x2 + z2); R[1][2] = 2.0f * (yz - wx);
    R[2][0] = 2.0f * (xz - wy);        R[2][1] = 2.0f * (yz + wx);        R[2][2] = 1.0f - 2.0f * (x2 + y2);
}
// se3_exp, rotation-first twist a = [rx,ry,rz,tx,ty,tz] (eigen_utils.hpp:909-943)
SP_HD Rigid se3_exp(const float a[6]) {
    Rigid out;
    const float w[3] = {a[0], a[1], a[2]};
    const float theta_sq = chain3(w[0], w[0], w[1], w[1], w[2], w[2]);
    const float theta = sqrtf(theta_sq);
    float q[4];
    so3_exp(w, q);
    quat_to_rot(q, out.R);
    if (theta < 1e-6f) {
#pragma unroll
        for (int i = 0; i < 3; ++i) out.t[i] = chain3(out.R[i][0], a[3], out.R[i][1], a[4], out.R[i][2], a[5]);
    } else {
        Mat3 O;
        O.m[0][0] = 0.0f;  O.m[0][1] = -w[2]; O.m[0][2] = w[1];
        O.m[1][0] = w[2];  O.m[1][1] = 0.0f;  O.m[1][2] = -w[0];
        O.m[2][0] = -w[1]; O.m[2][1] = w[0];  O.m[2][2] = 0.0f;
        const Mat3 O2 = matmul(O, O);
        const float A = (1.0f - cosf(theta)) / theta_sq;
        const float B = (theta - sinf(theta)) / (theta_sq * theta);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float Vr[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) Vr[j] = ((i == j) ? 1.0f : 0.0f) + (O.m[i][j] * A + O2.m[i][j] * B);
            out.t[i] = chain3(Vr[0], a[3], Vr[1], a[4], Vr[2], a[5]);
        }
    }
    return out;
}
// Isometry3f product as Eigen forms it (Eigen/src/Geometry/Transform.h): plain multiply-add, k ascending.
SP_HD Rigid rigid_mul(const Rigid& L, const Rigid& Rm) {
    Rigid o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += L.R[i][k] * Rm.R[k][j];
            o.R[i][j] = s;
        }
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) s += L.R[i][k] * Rm.t[k];
        o.t[i] = s + L.t[i];
    }
    return o;
}
SP_HD void store_rigid_colmajor(const Rigid& r, float* T) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int i = 0; i < 3; ++i) T[j * 4 + i] = r.R[i][j];
        T[j * 4 + 3] = 0.0f;
    }
    T[12] = r.t[0]; T[13] = r.t[1]; T[14] = r.t[2]; T[15] = 1.0f;
}

// 6x6 LDL^T with diagonal pivoting, the algorithm of Eigen::LDLT (registration.hpp:791-801): left-looking,
// pivot = largest |diagonal| of the not-yet-factored part. H is row-major and symmetric. Solves H x = rhs.
// Returns false on a zero pivot with a non-zero column (Eigen: NumericalIssue), x = 0 then.
// The run-time indexed working set lives in an LdltScratch the caller provides: a stack object on the host, an LDS
// object on the device (a private array indexed at run time would be placed in scratch memory, i.e. off chip).
struct LdltScratch {
    float m[6][6];
    float y[6];
    float temp[6];
    float rhs[6];
    float x[6];
    float H[36];
    int perm[6];
};
SP_HD bool ldlt6_solve(const float* H, const float* rhs, float* x, LdltScratch&, float* dmin_out = nullptr) {
    // Fully unrolled: every array index below is a compile-time constant after unrolling, so the 6x6 working set stays
    // in registers. The run-time pivot is applied with predicated swaps (`if (piv == p)`), which performs exactly the
    // arithmetic of the textbook loop form, in the same order.
    float m[6][6];
    int perm[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) m[i][j] = H[i * 6 + j];
        perm[i] = i;
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        float best = fabsf(m[k][k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i)
            if (fabsf(m[i][i]) > best) { best = fabsf(m[i][i]); piv = i; }
#if defined(__HIP_DEVICE_COMPILE__)
        // one lane runs this solver: make the pivot a scalar so that `piv == p` is a real (uniform) branch and only the
        // taken swap executes, instead of every candidate swap being evaluated under a select
        piv = __builtin_amdgcn_readfirstlane(piv);
#endif
#pragma unroll
        for (int p = k + 1; p < 6; ++p) {
            if (piv == p) {
#pragma unroll
                for (int j = 0; j < 6; ++j) { const float t = m[k][j]; m[k][j] = m[p][j]; m[p][j] = t; }
#pragma unroll
                for (int i = 0; i < 6; ++i) { const float t = m[i][k]; m[i][k] = m[i][p]; m[i][p] = t; }
                const int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
            }
        }
        float temp[6];
#pragma unroll
        for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
        if (k > 0) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < k; ++j) s += m[k][j] * temp[j];
            m[k][k] -= s;
#pragma unroll
            for (int i = k + 1; i < 6; ++i) {
                float t = 0.0f;
#pragma unroll
                for (int j = 0; j < k; ++j) t += m[i][j] * temp[j];
                m[i][k] -= t;
            }
        }
        const float d = m[k][k];
        if (fabsf(d) > 0.0f) {
#pragma unroll
            for (int i = k + 1; i < 6; ++i) m[i][k] /= d;
        } else {
#pragma unroll
            for (int i = k + 1; i < 6; ++i)
                if (m[i][k] != 0.0f) ok = false;
        }
    }
    if (dmin_out) {  // ldlt.vectorD().minCoeff()
        float dm = m[0][0];
#pragma unroll
        for (int i = 1; i < 6; ++i) dm = m[i][i] < dm ? m[i][i] : dm;
        *dmin_out = dm;
    }
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = 0.0f;
        return false;
    }
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float v = 0.0f;
#pragma unroll
        for (int t = 0; t < 6; ++t) v = (perm[i] == t) ? rhs[t] : v;
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) y[i] -= m[i][j] * y[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = (fabsf(m[i][i]) > FLT_MIN) ? y[i] / m[i][i] : 0.0f;
#pragma unroll
    for (int i = 5; i >= 0; --i)
#pragma unroll
        for (int j = i + 1; j < 6; ++j) y[i] -= m[j][i] * y[j];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) v = (perm[i] == t) ? y[i] : v;
        x[t] = v;
    }
    return true;
}

// compute_dogleg_step<6> (algorithms/registration/dogleg_step.hpp:35-101): Powell dogleg step for H p = -g inside a trust
// region. Eigen's reductions (.norm(), .dot(), H * g) are restated as plain ascending sums — third-party arithmetic the
// reference does not pin.
struct DoglegStep6 {
    float p[6];
    float step_norm, predicted_reduction;
};
SP_HD float dot6(const float* a, const float* b) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += a[i] * b[i];
    return s;
}
SP_HD void matvec6(const float* H, const float* v, float* out) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 6; ++j) s += H[i * 6 + j] * v[j];
        out[i] = s;
    }
}
SP_HD DoglegStep6 dogleg_step6(const float* H, const float* g, float radius, LdltScratch& w) {
    DoglegStep6 r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.p[i] = 0.0f;
    r.step_norm = 0.0f;
    r.predicted_reduction = 0.0f;
    float p_gn[6], ng[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { p_gn[i] = 0.0f; ng[i] = -g[i]; }
    float norm_gn = 0.0f, dmin = 0.0f;
    bool has_gn = false;
    float sol[6];
    if (ldlt6_solve(H, ng, sol, w, &dmin) && dmin > 0.0f) {
#pragma unroll
        for (int i = 0; i < 6; ++i) p_gn[i] = sol[i];
        norm_gn = sqrtf(dot6(p_gn, p_gn));
        has_gn = fabsf(norm_gn) <= FLT_MAX;  // std::isfinite
    }
    const float g2 = dot6(g, g);
    float Hg[6];
    matvec6(H, g, Hg);
    const float gHg = dot6(g, Hg);
    float p_sd[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) p_sd[i] = -g[i];
    if (gHg > FLT_EPSILON) {
        const float alpha = g2 / gHg;
        if (fabsf(alpha) <= FLT_MAX) {  // std::isfinite
#pragma unroll
            for (int i = 0; i < 6; ++i) p_sd[i] = -alpha * g[i];
        }
    }
    const float norm_sd = sqrtf(dot6(p_sd, p_sd));
    if (has_gn && norm_gn <= radius) {
#pragma unroll
        for (int i = 0; i < 6; ++i) r.p[i] = p_gn[i];
        r.step_norm = norm_gn;
    } else if (norm_sd >= radius) {
        if (norm_sd > FLT_EPSILON) {
            const float sc = radius / norm_sd;
#pragma unroll
            for (int i = 0; i < 6; ++i) r.p[i] = sc * p_sd[i];
        }
        r.step_norm = radius;
    } else if (has_gn) {
        float diff[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) diff[i] = p_gn[i] - p_sd[i];
        const float a = dot6(diff, diff);
        const float b = 2.0f * dot6(p_sd, diff);
        const float c = dot6(p_sd, p_sd) - radius * radius;
        float disc = b * b - 4.0f * a * c;
        disc = disc < 0.0f ? 0.0f : disc;
        float tau = 0.0f;
        if (a > FLT_EPSILON) tau = (-b + sqrtf(disc)) / (2.0f * a);
        tau = tau < 0.0f ? 0.0f : (1.0f < tau ? 1.0f : tau);
#pragma unroll
        for (int i = 0; i < 6; ++i) r.p[i] = p_sd[i] + tau * diff[i];
        r.step_norm = sqrtf(dot6(r.p, r.p));
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) r.p[i] = p_sd[i];
        if (norm_sd > radius && norm_sd > FLT_EPSILON) {
            const float sc = radius / norm_sd;
#pragma unroll
            for (int i = 0; i < 6; ++i) r.p[i] *= sc;
            r.step_norm = radius;
        } else {
            r.step_norm = norm_sd;
        }
    }
    float Hp[6];
    matvec6(H, r.p, Hp);
    r.predicted_reduction = -(dot6(g, r.p) + 0.5f * dot6(r.p, Hp));
    return r;
}

}  // namespace sp

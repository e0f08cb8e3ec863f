// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the reference's device math (sycl_points `eigen_utils`).
// Nothing under sycl_points_amd/ or include/ may include, link or call this file;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
//
// Follows /root/reference/cpp/include/sycl_points/utils/eigen_utils.hpp (cited per function as
// eigen_utils.hpp:LINE). Every sycl::fma in the reference is a std::fmaf here, every plain `a*b+c`
// stays an un-contracted multiply followed by an add (build with -ffp-contract=off).
// Implementation-defined pieces of the SYCL runtime are pinned to one concrete choice and
// documented where they occur (sycl::dot, sycl::min/max on NaN, reduction order).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <utility>

namespace oracle {

constexpr float PI = 3.14159265358979323846f;  // eigen_utils.hpp:24

// Column-major fixed-size matrix, the storage order of Eigen::Matrix (eigen_utils.hpp:20).
template <int M, int N>
struct Mat {
    float d[M * N];
    float& operator()(int i, int j) { return d[j * M + i]; }
    const float& operator()(int i, int j) const { return d[j * M + i]; }
    float& operator[](int i) { return d[i]; }
    const float& operator[](int i) const { return d[i]; }
    static Mat Zero() {
        Mat r;
        for (int i = 0; i < M * N; ++i) r.d[i] = 0.0f;
        return r;
    }
    static Mat Identity() {
        Mat r = Zero();
        for (int i = 0; i < (M < N ? M : N); ++i) r(i, i) = 1.0f;
        return r;
    }
};
template <int N>
using Vec = Mat<N, 1>;
using Vec3 = Vec<3>;
using Vec4 = Vec<4>;
using Vec6 = Vec<6>;
using Mat3 = Mat<3, 3>;
using Mat4 = Mat<4, 4>;
using Mat6 = Mat<6, 6>;

// sycl::min / sycl::max on floats: "y < x ? y : x" / "x < y ? y : x" (SYCL 2020 4.17.7).
inline float sycl_min(float x, float y) { return (y < x) ? y : x; }
inline float sycl_max(float x, float y) { return (x < y) ? y : x; }

// eigen_utils.hpp:32-43
template <int M, int N>
Mat<M, N> add(const Mat<M, N>& A, const Mat<M, N>& B) {
    Mat<M, N> r;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) r(i, j) = A(i, j) + B(i, j);
    return r;
}
// eigen_utils.hpp:51-59
template <int M, int N>
void add_inplace(Mat<M, N>& A, const Mat<M, N>& B) {
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) A(i, j) += B(i, j);
}
// eigen_utils.hpp:67-79
template <int M, int N>
Mat<M, N> subtract(const Mat<M, N>& A, const Mat<M, N>& B) {
    Mat<M, N> r;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) r(i, j) = A(i, j) - B(i, j);
    return r;
}
// eigen_utils.hpp:88-105  (matrix x matrix; accumulation over k ascending, one fma per term)
template <int M, int K, int N>
Mat<M, N> matmul(const Mat<M, K>& A, const Mat<K, N>& B) {
    Mat<M, N> r = Mat<M, N>::Zero();
    for (int j = 0; j < N; ++j)
        for (int k = 0; k < K; ++k) {
            const float b_kj = B(k, j);
            for (int i = 0; i < M; ++i) r(i, j) = std::fmaf(A(i, k), b_kj, r(i, j));
        }
    return r;
}
// eigen_utils.hpp:113-127  (matrix x vector; per-row fma chain starting from 0)
template <int M, int N>
Vec<M> matvec(const Mat<M, N>& A, const Vec<N>& v) {
    Vec<M> r = Vec<M>::Zero();
    for (int i = 0; i < M; ++i) {
        float sum = 0.0f;
        for (int j = 0; j < N; ++j) sum = std::fmaf(A(i, j), v[j], sum);
        r[i] = sum;
    }
    return r;
}
// eigen_utils.hpp:135-161
template <int M, int N>
Mat<M, N> scale(const Mat<M, N>& A, float s) {
    Mat<M, N> r;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) r(i, j) = A(i, j) * s;
    return r;
}
// eigen_utils.hpp:208-219
template <int M>
Mat<M, M> ensure_symmetric(const Mat<M, M>& A) {
    Mat<M, M> r;
    for (int j = 0; j < M; ++j)
        for (int i = 0; i < M; ++i) r(i, j) = (i == j) ? A(i, j) : (A(i, j) + A(j, i)) * 0.5f;
    return r;
}
// eigen_utils.hpp:226-237
template <int M, int N>
Mat<N, M> transpose(const Mat<M, N>& A) {
    Mat<N, M> r;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) r(j, i) = A(i, j);
    return r;
}
// eigen_utils.hpp:245-253
template <int N>
float dot(const Vec<N>& u, const Vec<N>& v) {
    float r = 0.0f;
    for (int i = 0; i < N; ++i) r = std::fmaf(u[i], v[i], r);
    return r;
}
// eigen_utils.hpp:272-284
template <int N>
Mat<N, N> outer(const Vec<N>& u, const Vec<N>& v) {
    Mat<N, N> r;
    for (int j = 0; j < N; ++j) {
        const float vj = v[j];
        for (int i = 0; i < N; ++i) r(i, j) = u[i] * vj;
    }
    return r;
}
// eigen_utils.hpp:186-198
template <int M, int N>
Mat<M, N> element_wise_multiply(const Mat<M, N>& A, const Mat<M, N>& B) {
    Mat<M, N> r;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) r(i, j) = A(i, j) * B(i, j);
    return r;
}
// eigen_utils.hpp:259-265  (plain products and one subtraction per component, no fma)
inline Vec<3> cross(const Vec<3>& u, const Vec<3>& v) {
    Vec<3> r;
    r[0] = u[1] * v[2] - u[2] * v[1];
    r[1] = u[2] * v[0] - u[0] * v[2];
    r[2] = u[0] * v[1] - u[1] * v[0];
    return r;
}
// eigen_utils.hpp:315-326, 343-345  (column-major fma chain over all elements)
template <int M, int N>
float frobenius_norm_squared(const Mat<M, N>& A) {
    float r = 0.0f;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) r = std::fmaf(A(i, j), A(i, j), r);
    return r;
}
template <int M, int N>
float frobenius_norm(const Mat<M, N>& A) {
    return std::sqrt(frobenius_norm_squared<M, N>(A));
}
// eigen_utils.hpp:291-298
template <int M>
float trace(const Mat<M, M>& A) {
    float r = 0.0f;
    for (int i = 0; i < M; ++i) r += A(i, i);
    return r;
}
// eigen_utils.hpp:303-307
inline float determinant(const Mat3& A) {
    return std::fmaf(A(0, 0), std::fmaf(A(1, 1), A(2, 2), -A(1, 2) * A(2, 1)),
                     std::fmaf(-A(0, 1), std::fmaf(A(1, 0), A(2, 2), -A(1, 2) * A(2, 0)),
                               A(0, 2) * std::fmaf(A(1, 0), A(2, 1), -A(1, 1) * A(2, 0))));
}
// eigen_utils.hpp:333-335, 352-354
template <int M>
float norm_squared(const Vec<M>& a) {
    return dot<M>(a, a);
}
template <int M>
float norm(const Vec<M>& a) {
    return std::sqrt(norm_squared<M>(a));
}
// eigen_utils.hpp:356-363
template <int M>
Vec<M> normalize(const Vec<M>& a) {
    const float n = norm<M>(a);
    if (n < 1e-6f) return Vec<M>::Zero();
    return scale<M, 1>(a, 1.0f / n);
}
// eigen_utils.hpp:403-423
inline Mat3 inverse(const Mat3& s) {
    const float det = determinant(s);
    if (std::fabs(det) < 1e-6f) return Mat3::Zero();
    const float invDet = 1.0f / det;
    Mat3 r;
    r(0, 0) = std::fmaf(s(1, 1), s(2, 2), -s(1, 2) * s(2, 1)) * invDet;
    r(1, 0) = std::fmaf(s(1, 2), s(2, 0), -s(1, 0) * s(2, 2)) * invDet;
    r(2, 0) = std::fmaf(s(1, 0), s(2, 1), -s(1, 1) * s(2, 0)) * invDet;
    r(0, 1) = std::fmaf(s(0, 2), s(2, 1), -s(0, 1) * s(2, 2)) * invDet;
    r(1, 1) = std::fmaf(s(0, 0), s(2, 2), -s(0, 2) * s(2, 0)) * invDet;
    r(2, 1) = std::fmaf(s(0, 1), s(2, 0), -s(0, 0) * s(2, 1)) * invDet;
    r(0, 2) = std::fmaf(s(0, 1), s(1, 2), -s(0, 2) * s(1, 1)) * invDet;
    r(1, 2) = std::fmaf(s(0, 2), s(1, 0), -s(0, 0) * s(1, 2)) * invDet;
    r(2, 2) = std::fmaf(s(0, 0), s(1, 1), -s(0, 1) * s(1, 0)) * invDet;
    return r;
}
// eigen_utils.hpp:429-437
template <int M>
Mat<M, M> as_diagonal(const Vec<M>& d) {
    Mat<M, M> r = Mat<M, M>::Zero();
    for (int i = 0; i < M; ++i) r(i, i) = d[i];
    return r;
}

// eigen_utils.hpp:443-562.  Analytic (Cardano) eigen-decomposition, eigenvalues ascending.
// sycl::dot(float3,float3) at :548 is implementation-defined; pinned here to the same
// fma chain as eigen_utils::dot (v0*v0, then fma v1, then fma v2).
inline void symmetric_eigen_decomposition_3x3(const Mat3& A, Vec3& eigenvalues, Mat3& eigenvectors) {
    constexpr float EPSILON = std::numeric_limits<float>::epsilon();
    float max_abs_val = 0.0f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) max_abs_val = std::fmax(max_abs_val, std::fabs(A(i, j)));
    if (max_abs_val < std::numeric_limits<float>::min()) {
        eigenvalues = Vec3::Zero();
        eigenvectors = Mat3::Identity();
        return;
    }
    const float scale_inv = 1.0f / max_abs_val;
    const Mat3 sA = scale<3, 3>(A, scale_inv);

    const float c2 = -trace<3>(sA);
    const float c1 = std::fmaf(sA(0, 0), sA(1, 1), std::fmaf(sA(0, 0), sA(2, 2), sA(1, 1) * sA(2, 2))) -
                     std::fmaf(sA(0, 1), sA(1, 0), std::fmaf(sA(0, 2), sA(2, 0), sA(1, 2) * sA(2, 1)));
    const float c0 = -determinant(sA);

    const float p = c1 - c2 * c2 / 3.0f;
    const float q = 2.0f * c2 * c2 * c2 / 27.0f - c2 * c1 / 3.0f + c0;
    const float discriminant = 4.0f * p * p * p + 27.0f * q * q;

    if (std::fabs(discriminant) <= EPSILON) {
        const float u = q >= 0 ? -std::cbrt(q / 2.0f) : std::cbrt(-q / 2.0f);
        eigenvalues[0] = 2.0f * u - c2 / 3.0f;
        eigenvalues[1] = eigenvalues[2] = -u - c2 / 3.0f;
    } else {
        const float s = std::sqrt(-p / 3.0f);
        const float cosv = sycl_max(-1.0f, sycl_min(1.0f, -q / (2.0f * s * s * s)));
        float phi = std::fabs(p) < EPSILON ? 0.0f : std::acos(cosv);
        if (phi < 0.0f) phi += PI;
        eigenvalues[0] = std::fmaf(2.0f * s, std::cos(phi / 3.0f), -c2 / 3.0f);
        eigenvalues[2] = std::fmaf(2.0f * s, std::cos((phi + 4.0f * PI) / 3.0f), -c2 / 3.0f);
        eigenvalues[1] = std::fmaf(2.0f * s, std::cos((phi + 2.0f * PI) / 3.0f), -c2 / 3.0f);
    }
    if (eigenvalues[0] > eigenvalues[1]) std::swap(eigenvalues[0], eigenvalues[1]);
    if (eigenvalues[1] > eigenvalues[2]) std::swap(eigenvalues[1], eigenvalues[2]);
    if (eigenvalues[0] > eigenvalues[1]) std::swap(eigenvalues[1], eigenvalues[0]);

    eigenvectors = Mat3::Zero();
    for (int k = 0; k < 3; ++k) {
        const Mat3 Mm = subtract<3, 3>(sA, scale<3, 3>(Mat3::Identity(), eigenvalues[k]));
        const float m00 = std::fmaf(Mm(1, 1), Mm(2, 2), -Mm(1, 2) * Mm(2, 1));
        const float m01 = std::fmaf(Mm(1, 2), Mm(2, 0), -Mm(1, 0) * Mm(2, 2));
        const float m02 = std::fmaf(Mm(1, 0), Mm(2, 1), -Mm(1, 1) * Mm(2, 0));
        const float m10 = std::fmaf(Mm(0, 2), Mm(2, 1), -Mm(0, 1) * Mm(2, 2));
        const float m11 = std::fmaf(Mm(0, 0), Mm(2, 2), -Mm(0, 2) * Mm(2, 0));
        const float m12 = std::fmaf(Mm(0, 1), Mm(2, 0), -Mm(0, 0) * Mm(2, 1));
        const float m20 = std::fmaf(Mm(0, 1), Mm(1, 2), -Mm(0, 2) * Mm(1, 1));
        const float m21 = std::fmaf(Mm(0, 2), Mm(1, 0), -Mm(0, 0) * Mm(1, 2));
        const float m22 = std::fmaf(Mm(0, 0), Mm(1, 1), -Mm(0, 1) * Mm(1, 0));
        const float s0 = std::fmaf(m00, m00, std::fmaf(m10, m10, m20 * m20));
        const float s1 = std::fmaf(m01, m01, std::fmaf(m11, m11, m21 * m21));
        const float s2 = std::fmaf(m02, m02, std::fmaf(m12, m12, m22 * m22));
        float v[3];
        if (s0 >= s1 && s0 >= s2) {
            v[0] = m00; v[1] = m10; v[2] = m20;
        } else if (s1 >= s0 && s1 >= s2) {
            v[0] = m01; v[1] = m11; v[2] = m21;
        } else {
            v[0] = m02; v[1] = m12; v[2] = m22;
        }
        float norm_sq = std::fmaf(v[2], v[2], std::fmaf(v[1], v[1], v[0] * v[0]));
        if (norm_sq < std::numeric_limits<float>::min()) {
            v[0] = 1.0f; v[1] = 0.0f; v[2] = 0.0f;
            norm_sq = 1.0f;
        }
        const float inv_length = 1.0f / std::sqrt(norm_sq);
        eigenvectors(0, k) = v[0] * inv_length;
        eigenvectors(1, k) = v[1] * inv_length;
        eigenvectors(2, k) = v[2] * inv_length;
    }
    eigenvalues = scale<3, 1>(eigenvalues, max_abs_val);
}

// ---------------------------------------------------------------- geometry / lie
// eigen_utils.hpp:774-803
inline Vec4 rotation_matrix_to_quaternion(const Mat3& R) {
    Vec4 q;
    const float tr = R(0, 0) + R(1, 1) + R(2, 2);
    if (tr > 0.0f) {
        const float S = std::sqrt(tr + 1.0f) * 2.0f;
        q[0] = (R(2, 1) - R(1, 2)) / S; q[1] = (R(0, 2) - R(2, 0)) / S; q[2] = (R(1, 0) - R(0, 1)) / S; q[3] = 0.25f * S;
    } else if ((R(0, 0) > R(1, 1)) && (R(0, 0) > R(2, 2))) {
        const float S = std::sqrt(1.0f + R(0, 0) - R(1, 1) - R(2, 2)) * 2.0f;
        q[0] = 0.25f * S; q[1] = (R(0, 1) + R(1, 0)) / S; q[2] = (R(0, 2) + R(2, 0)) / S; q[3] = (R(2, 1) - R(1, 2)) / S;
    } else if (R(1, 1) > R(2, 2)) {
        const float S = std::sqrt(1.0f + R(1, 1) - R(0, 0) - R(2, 2)) * 2.0f;
        q[0] = (R(0, 1) + R(1, 0)) / S; q[1] = 0.25f * S; q[2] = (R(1, 2) + R(2, 1)) / S; q[3] = (R(0, 2) - R(2, 0)) / S;
    } else {
        const float S = std::sqrt(1.0f + R(2, 2) - R(0, 0) - R(1, 1)) * 2.0f;
        q[2] = 0.25f * S; q[3] = (R(1, 0) - R(0, 1)) / S; q[0] = (R(0, 2) + R(2, 0)) / S; q[1] = (R(1, 2) + R(2, 1)) / S;
    }
    return q;
}
// eigen_utils.hpp:808-836  (quaternion x,y,z,w)
inline Mat3 quaternion_to_rotation_matrix(const Vec4& quat) {
    const float x = quat[0], y = quat[1], z = quat[2], w = quat[3];
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float wx = w * x, wy = w * y, wz = w * z;
    Mat3 R;
    R(0, 0) = 1.0f - 2.0f * (y2 + z2);
    R(0, 1) = 2.0f * (xy - wz);
    R(0, 2) = 2.0f * (xz + wy);
    R(1, 0) = 2.0f * (xy + wz);
    R(1, 1) = 1.0f - 2.0f * (x2 + z2);
    R(1, 2) = 2.0f * (yz - wx);
    R(2, 0) = 2.0f * (xz - wy);
    R(2, 1) = 2.0f * (yz + wx);
    R(2, 2) = 1.0f - 2.0f * (x2 + y2);
    return R;
}
// eigen_utils.hpp:860-880
template <class V>
Mat3 skew(const V& x) {
    Mat3 r;
    r(0, 0) = 0.0f;  r(0, 1) = -x[2]; r(0, 2) = x[1];
    r(1, 0) = x[2];  r(1, 1) = 0.0f;  r(1, 2) = -x[0];
    r(2, 0) = -x[1]; r(2, 1) = x[0];  r(2, 2) = 0.0f;
    return r;
}
// eigen_utils.hpp:886-902
inline Vec4 so3_exp(const Vec3& omega) {
    const float theta_sq = dot<3>(omega, omega);
    float imag_factor, real_factor;
    if (theta_sq < 1e-6f) {
        const float theta_quad = theta_sq * theta_sq;
        imag_factor = 0.5f - 1.0f / 48.0f * theta_sq + 1.0f / 3840.0f * theta_quad;
        real_factor = 1.0f - 1.0f / 8.0f * theta_sq + 1.0f / 384.0f * theta_quad;
    } else {
        const float theta = std::sqrt(theta_sq);
        const float half_theta = 0.5f * theta;
        imag_factor = std::sin(half_theta) / theta;
        real_factor = std::cos(half_theta);
    }
    Vec4 q;
    q[0] = imag_factor * omega[0]; q[1] = imag_factor * omega[1]; q[2] = imag_factor * omega[2]; q[3] = real_factor;
    return q;
}
// eigen_utils.hpp:909-943  (rotation-first twist [rx,ry,rz,tx,ty,tz])
inline Mat4 se3_exp(const Vec6& a) {
    Vec3 omega; omega[0] = a[0]; omega[1] = a[1]; omega[2] = a[2];
    const float theta_sq = dot<3>(omega, omega);
    const float theta = std::sqrt(theta_sq);
    const Mat3 R = quaternion_to_rotation_matrix(so3_exp(omega));
    Mat4 se3 = Mat4::Identity();
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) se3(r, c) = R(r, c);
    Vec3 t; t[0] = a[3]; t[1] = a[4]; t[2] = a[5];
    Vec3 trans;
    if (theta < 1e-6f) {
        trans = matvec<3, 3>(R, t);
    } else {
        const Mat3 Omega = skew(omega);
        const Mat3 Omega_sq = matmul<3, 3, 3>(Omega, Omega);
        const float A = (1.0f - std::cos(theta)) / theta_sq;
        const float B = (theta - std::sin(theta)) / (theta_sq * theta);
        const Mat3 V = add<3, 3>(Mat3::Identity(), add<3, 3>(scale<3, 3>(Omega, A), scale<3, 3>(Omega_sq, B)));
        trans = matvec<3, 3>(V, t);
    }
    se3(0, 3) = trans[0]; se3(1, 3) = trans[1]; se3(2, 3) = trans[2];
    return se3;
}
// eigen_utils.hpp:948-986
inline Vec3 so3_log(const Vec4& quat) {
    Vec4 q = normalize<4>(quat);
    if (q[3] < 0.0f) { q[0] *= -1.0f; q[1] *= -1.0f; q[2] *= -1.0f; q[3] *= -1.0f; }
    const float w = q[3];
    Vec3 xyz; xyz[0] = q[0]; xyz[1] = q[1]; xyz[2] = q[2];
    const float xyz_norm = norm<3>(xyz);
    if (xyz_norm < 1e-6f) {
        const float sc = 2.0f / w * (1.0f + xyz_norm * xyz_norm / (6.0f * w * w));
        return scale<3, 1>(xyz, sc);
    }
    if (std::fabs(w) < 1e-6f) return scale<3, 1>(xyz, PI / xyz_norm);
    const float theta = 2.0f * std::atan2(xyz_norm, std::fabs(w));
    return scale<3, 1>(xyz, theta / xyz_norm);
}
// eigen_utils.hpp:991-1034.  The Eigen expression (I - 0.5*Omega + coeff*Omega*Omega) * t is evaluated
// with plain (non-fma) products in Eigen's natural order; third-party (Eigen) arithmetic, parity unpinned.
inline Vec6 se3_log(const Mat4& T) {
    Mat3 R; Vec3 t;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R(r, c) = T(r, c);
    t[0] = T(0, 3); t[1] = T(1, 3); t[2] = T(2, 3);
    const Vec3 omega = so3_log(rotation_matrix_to_quaternion(R));
    const float theta = norm<3>(omega);
    Vec6 out; out[0] = omega[0]; out[1] = omega[1]; out[2] = omega[2];
    const Mat3 Omega = skew(omega);
    Mat3 Vinv;
    if (theta < 1e-6f) {
        Vinv = subtract<3, 3>(Mat3::Identity(), scale<3, 3>(Omega, 0.5f));
    } else {
        const float half_theta = 0.5f * theta;
        const float coeff = (1.0f - theta * std::cos(half_theta) / (2.0f * std::sin(half_theta))) / (theta * theta);
        Mat3 O2 = Mat3::Zero();
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float s = 0.0f;
                for (int k = 0; k < 3; ++k) s += Omega(i, k) * Omega(k, j);
                O2(i, j) = s;
            }
        Vinv = add<3, 3>(subtract<3, 3>(Mat3::Identity(), scale<3, 3>(Omega, 0.5f)), scale<3, 3>(O2, coeff));
    }
    for (int i = 0; i < 3; ++i) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += Vinv(i, k) * t[k];
        out[3 + i] = s;
    }
    return out;
}

// Isometry3f product as Eigen evaluates it for Transform<float,3,Isometry>:
//   linear = L.linear * R.linear ; translation = L.linear * R.translation + L.translation
// (Eigen/src/Geometry/Transform.h, transform_transform_product_impl). Plain multiply-add, k ascending.
// Third-party (Eigen, unpinned version; registration.hpp:814,850): parity unpinned.
inline Mat4 isometry_mul(const Mat4& L, const Mat4& Rm) {
    Mat4 out = Mat4::Identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k) s += L(i, k) * Rm(k, j);
            out(i, j) = s;
        }
    for (int i = 0; i < 3; ++i) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += L(i, k) * Rm(k, 3);
        out(i, 3) = s + L(i, 3);
    }
    return out;
}

// Eigen::LDLT<Matrix<float,6,6>>::compute + solve (registration.hpp:791-801) restated:
// in-place LDL^T with diagonal pivoting (largest |diag|), then P^T L^-T D^-1 L^-1 P b.
// Eigen is third-party and unpinned in the reference (cpp/CMakeLists.txt:27): parity unpinned;
// H + lambda*I is SPD on this path so pivot choice only changes rounding.
// Returns false when a zero pivot with a non-zero column is met (Eigen: NumericalIssue).
inline bool ldlt6_solve(const Mat6& A, const Vec6& b, Vec6& x, float* dmin_out = nullptr) {
    constexpr int n = 6;
    float m[n][n];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) m[i][j] = A(i, j);
    int perm[n];
    for (int i = 0; i < n; ++i) perm[i] = i;
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        float best = std::fabs(m[k][k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(m[i][i]) > best) { best = std::fabs(m[i][i]); piv = i; }
        if (piv != k) {  // symmetric row/column swap on the lower triangle
            for (int j = 0; j < n; ++j) std::swap(m[k][j], m[piv][j]);
            for (int i = 0; i < n; ++i) std::swap(m[i][k], m[i][piv]);
            std::swap(perm[k], perm[piv]);
        }
        // left-looking update of column k: temp = D(0..k) * L(k,0..k)^T ; A_kk -= L(k,:)*temp ; A21 -= A20*temp
        float temp[n];
        for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
        if (k > 0) {
            float s = 0.0f;
            for (int j = 0; j < k; ++j) s += m[k][j] * temp[j];
            m[k][k] -= s;
            for (int i = k + 1; i < n; ++i) {
                float t = 0.0f;
                for (int j = 0; j < k; ++j) t += m[i][j] * temp[j];
                m[i][k] -= t;
            }
        }
        const float dkk = m[k][k];
        if (std::fabs(dkk) > 0.0f) {
            for (int i = k + 1; i < n; ++i) m[i][k] /= dkk;
        } else {
            for (int i = k + 1; i < n; ++i)
                if (m[i][k] != 0.0f) ok = false;
        }
    }
    if (dmin_out) {  // ldlt.vectorD().minCoeff()
        float dm = m[0][0];
        for (int i = 1; i < n; ++i) dm = std::min(dm, m[i][i]);
        *dmin_out = dm;
    }
    if (!ok) { x = Vec6::Zero(); return false; }
    float y[n];
    for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) y[i] -= m[i][j] * y[j];
    for (int i = 0; i < n; ++i) {
        const float d = m[i][i];
        // Eigen: entries whose |d| <= 1/highest (i.e. denormal-scale) are zeroed, the rest divided
        y[i] = (std::fabs(d) > std::numeric_limits<float>::min()) ? y[i] / d : 0.0f;
    }
    for (int i = n - 1; i >= 0; --i)
        for (int j = i + 1; j < n; ++j) y[i] -= m[j][i] * y[j];
    for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
    return true;
}

}  // namespace oracle

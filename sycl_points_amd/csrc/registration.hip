// K11/K12/K13 + Gauss-Newton update for gfx950
// (replaces algorithms/registration/registration.hpp:412-462, 464-511, 513-664, 678-777, 791-828 and the per-point
//  factors of algorithms/registration/factor.hpp:69-482).
//
// K11: one lane walks source points with a grid stride; per point it gathers the correspondence
// (target point, 64-byte target covariance row, optional normal), linearises the factor in registers and adds the
// weighted 21 unique entries of H, the 6 of b, the error and the inlier count to 29 lane-private accumulators.
// A wave64 butterfly, then a 4-wave LDS step, gives one 32-float partial per workgroup; a second one-workgroup kernel
// sums the <= 1024 partials in a fixed order and writes the 176-byte system. No atomics: the sum is a fixed tree,
// bit-reproducible from run to run. HBM-bound: 168 B per source point.
// The factor arithmetic keeps the reference's fma chains (sp_math.h); structural zeros of the padded 4x4/4x6 types
// are skipped, which leaves every finite result bit-identical.
#include <algorithm>
#include <mutex>

#include "registration_device.h"

namespace sp {
namespace {


struct PointTerm {
    float H[6][6];  // symmetrised, unweighted
    float b[6];
    float sq;             // squared error of the factor
    float residual_norm;  // fed to the robust kernel
    float genz_weight;
};

__device__ __forceinline__ Mat3 load_cov3(const float4* __restrict__ c) {
    const float4 c0 = c[0], c1 = c[1], c2 = c[2];
    Mat3 C;
    C.m[0][0] = c0.x; C.m[1][0] = c0.y; C.m[2][0] = c0.z;
    C.m[0][1] = c1.x; C.m[1][1] = c1.y; C.m[2][1] = c1.z;
    C.m[0][2] = c2.x; C.m[1][2] = c2.y; C.m[2][2] = c2.z;
    return C;
}
__device__ __forceinline__ Mat3 identity3() {
    Mat3 I;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) I.m[i][j] = (i == j) ? 1.0f : 0.0f;
    return I;
}

// compute_se3_jacobian (factor.hpp:69-84): rows 0..2 of [R*skew(p) | -R]; row 3 is zero and never stored.
__device__ __forceinline__ void se3_jacobian(const Rigid& T, float px, float py, float pz, float J[3][6]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float r0 = T.R[i][0], r1 = T.R[i][1], r2 = T.R[i][2];
        J[i][0] = chain3(r0, 0.0f, r1, pz, r2, -py);
        J[i][1] = chain3(r0, -pz, r1, 0.0f, r2, px);
        J[i][2] = chain3(r0, py, r1, -px, r2, 0.0f);
        J[i][3] = -r0;
        J[i][4] = -r1;
        J[i][5] = -r2;
    }
}

// H = sym(J^T M J), b = J^T M r, sq = r^T M r   (factor.hpp:262-275, 339-351; M is a general 3x3)
__device__ __forceinline__ void quadratic_form(const float J[3][6], const Mat3& M, float r0, float r1, float r2,
                                               PointTerm& out) {
    float JTm[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) JTm[a][k] = chain3(J[0][a], M.m[0][k], J[1][a], M.m[1][k], J[2][a], M.m[2][k]);
    float Hf[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) Hf[a][c] = chain3(JTm[a][0], J[0][c], JTm[a][1], J[1][c], JTm[a][2], J[2][c]);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) out.H[a][c] = (a == c) ? Hf[a][c] : (Hf[a][c] + Hf[c][a]) * 0.5f;
#pragma unroll
    for (int a = 0; a < 6; ++a) out.b[a] = chain3(JTm[a][0], r0, JTm[a][1], r1, JTm[a][2], r2);
    const float v0 = chain3(M.m[0][0], r0, M.m[0][1], r1, M.m[0][2], r2);
    const float v1 = chain3(M.m[1][0], r0, M.m[1][1], r1, M.m[1][2], r2);
    const float v2 = chain3(M.m[2][0], r0, M.m[2][1], r1, M.m[2][2], r2);
    out.sq = chain3(r0, v0, r1, v1, r2, v2);
}

__device__ __forceinline__ void point_to_point(const float J[3][6], float r0, float r1, float r2, PointTerm& out) {
    quadratic_form(J, identity3(), r0, r1, r2, out);  // factor.hpp:130-149 with J^T*I == J^T
    out.residual_norm = sqrtf(out.sq);
}

// linearize_point_to_plane (factor.hpp:172-210)
__device__ __forceinline__ void point_to_plane(const float J[3][6], float r0, float r1, float r2, const float4 nrm,
                                               PointTerm& out) {
    const float proj = chain3(nrm.x, r0, nrm.y, r1, nrm.z, r2);
    const float n[3] = {nrm.x, nrm.y, nrm.z};
    float row[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) row[c] = chain3(n[0], J[0][c], n[1], J[1][c], n[2], J[2][c]);
    float Jp[3][6];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 6; ++c) Jp[i][c] = fmaf(n[i], row[c], 0.0f);
    const float pe0 = n[0] * proj, pe1 = n[1] * proj, pe2 = n[2] * proj;
    float Hf[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) Hf[a][c] = chain3(Jp[0][a], Jp[0][c], Jp[1][a], Jp[1][c], Jp[2][a], Jp[2][c]);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) out.H[a][c] = (a == c) ? Hf[a][c] : (Hf[a][c] + Hf[c][a]) * 0.5f;
#pragma unroll
    for (int a = 0; a < 6; ++a) out.b[a] = chain3(Jp[0][a], pe0, Jp[1][a], pe1, Jp[2][a], pe2);
    out.sq = proj * proj;
    out.residual_norm = fabsf(proj);
}

// compute_pca_normalized_curvature / is_genz_planar_correspondence (factor.hpp:378-392)
__device__ __forceinline__ bool genz_planar(const Mat3& tcov, float thr) {
    float ev[3];
    Mat3 V;
    symmetric_eigen3(tcov, ev, V);
    const float sum = ev[0] + ev[1] + ev[2];
    const float curv = (sum > 1e-12f) ? ev[0] / sum : 1.0f;
    return curv < thr;
}

// GICP information matrix: (Ct' + R Cs' R^T)^-1 with both covariances plane-regularised
// (factor.hpp:249-258, 111-123; transform_covs = T*(C*T^T), common/transform.hpp:14-22).
__device__ __forceinline__ Mat3 gicp_information(const Rigid& T, const Mat3& scov, const Mat3& tcov) {
    const Mat3 Cs = plane_regularize(scov);
    const Mat3 Ct = plane_regularize(tcov);
    Mat3 R;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R.m[i][j] = T.R[i][j];
    const Mat3 Y = matmul_bt(Cs, R);  // Cs * R^T
    const Mat3 RCR = matmul(R, Y);
    Mat3 S;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) S.m[i][j] = RCR.m[i][j] + Ct.m[i][j];
    return inverse(S);
}

struct Corr {  // everything gathered for one correspondence
    float4 s, t, tn;
    Mat3 scov, tcov;
};

template <int REG>
__device__ __forceinline__ Corr gather(unsigned i, const float4* __restrict__ src, const float4* __restrict__ scov,
                                       const float4* __restrict__ tgt, const float4* __restrict__ tcov,
                                       const float4* __restrict__ tnrm, const int32_t* __restrict__ nn_idx) {
    Corr c;
    const int ti = nn_idx[i];
    c.s = src[i];
    c.t = tgt[ti];
    if (REG == SP_REG_GICP) c.scov = scov ? load_cov3(scov + 4 * (size_t)i) : identity3();
    if (REG == SP_REG_GICP || REG == SP_REG_POINT_TO_DISTRIBUTION || REG == SP_REG_GENZ)
        c.tcov = tcov ? load_cov3(tcov + 4 * (size_t)ti) : identity3();
    if (REG == SP_REG_POINT_TO_PLANE || REG == SP_REG_GENZ)
        c.tn = tnrm ? tnrm[ti] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return c;
}

// linearize_geometry<reg> (factor.hpp:413-449)
template <int REG>
__device__ __forceinline__ void linearize_point(const Rigid& T, const Corr& c, float genz_alpha, float genz_thr,
                                                PointTerm& out) {
    float tx, ty, tz;
    transform_point(T, c.s.x, c.s.y, c.s.z, tx, ty, tz);
    const float r0 = c.t.x - tx, r1 = c.t.y - ty, r2 = c.t.z - tz;
    float J[3][6];
    se3_jacobian(T, c.s.x, c.s.y, c.s.z, J);
    out.genz_weight = 1.0f;
    if (REG == SP_REG_POINT_TO_POINT) {
        point_to_point(J, r0, r1, r2, out);
    } else if (REG == SP_REG_POINT_TO_PLANE) {
        point_to_plane(J, r0, r1, r2, c.tn, out);
    } else if (REG == SP_REG_GICP) {
        quadratic_form(J, gicp_information(T, c.scov, c.tcov), r0, r1, r2, out);
        out.residual_norm = sqrtf(out.sq);
    } else if (REG == SP_REG_POINT_TO_DISTRIBUTION) {
        quadratic_form(J, inverse(c.tcov), r0, r1, r2, out);  // factor.hpp:311-354
        out.residual_norm = sqrtf(out.sq);
    } else {  // GENZ (factor.hpp:426-443)
        const bool planar = genz_planar(c.tcov, genz_thr);
        const float w = planar ? genz_alpha : (1.0f - genz_alpha);
        if (planar) point_to_plane(J, r0, r1, r2, c.tn, out);
        else point_to_point(J, r0, r1, r2, out);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) out.H[a][cc] *= w;
            out.b[a] *= w;
        }
        out.sq *= w;
        out.genz_weight = w;
    }
}

// calculate_geometry_error<reg> (factor.hpp:459-482): squared error only.
template <int REG>
__device__ __forceinline__ float error_point(const Rigid& T, const Corr& c, float genz_alpha, float genz_thr,
                                             float& genz_weight) {
    float tx, ty, tz;
    transform_point(T, c.s.x, c.s.y, c.s.z, tx, ty, tz);
    const float r0 = c.t.x - tx, r1 = c.t.y - ty, r2 = c.t.z - tz;
    genz_weight = 1.0f;
    auto maha = [&](const Mat3& M) {
        const float v0 = chain3(M.m[0][0], r0, M.m[0][1], r1, M.m[0][2], r2);
        const float v1 = chain3(M.m[1][0], r0, M.m[1][1], r1, M.m[1][2], r2);
        const float v2 = chain3(M.m[2][0], r0, M.m[2][1], r1, M.m[2][2], r2);
        return chain3(r0, v0, r1, v1, r2, v2);
    };
    auto plane = [&]() {
        const float proj = chain3(c.tn.x, r0, c.tn.y, r1, c.tn.z, r2);
        return proj * proj;
    };
    if (REG == SP_REG_POINT_TO_POINT) return chain3(r0, r0, r1, r1, r2, r2);
    if (REG == SP_REG_POINT_TO_PLANE) return plane();
    if (REG == SP_REG_GICP) return maha(gicp_information(T, c.scov, c.tcov));
    if (REG == SP_REG_POINT_TO_DISTRIBUTION) return maha(inverse(c.tcov));
    const bool planar = genz_planar(c.tcov, genz_thr);
    genz_weight = planar ? genz_alpha : (1.0f - genz_alpha);
    return planar ? plane() : chain3(r0, r0, r1, r1, r2, r2);
}

// Rotation constraint: Jensen-Bregman LogDet divergence between R Cs R^T and Ct (rotation_constraint.hpp:15-128).
// D = max(log det((Cs' + Ct)/2) - (log det Cs + log det Ct)/2, 0) with determinants floored at 1e-10;
// gradient g = R^T * (-vex([Cs', M^-1])) in the body frame; the term contributes J J^T to the rotation block of H,
// D J to b[0..2] and D^2/2 as its squared error.
struct RotTerm {
    float D;
    float J[3];
};
template <bool WITH_GRAD>
__device__ __forceinline__ RotTerm rotation_divergence(const Rigid& T, const Mat3& Cs, const Mat3& Ct) {
    Mat3 R;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R.m[i][j] = T.R[i][j];
    const Mat3 Csp = matmul(R, matmul_bt(Cs, R));
    Mat3 M;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M.m[i][j] = (Csp.m[i][j] + Ct.m[i][j]) * 0.5f;
    const float log_det_M = logf(fmaxf(determinant(M), 1e-10f));
    const float log_det_ref = 0.5f * (logf(fmaxf(determinant(Cs), 1e-10f)) + logf(fmaxf(determinant(Ct), 1e-10f)));
    RotTerm out;
    out.D = fmaxf(log_det_M - log_det_ref, 0.0f);
    if (WITH_GRAD) {
        const Mat3 Mi = inverse(M);
        const Mat3 A = matmul(Csp, Mi), B = matmul(Mi, Csp);
        const float g0 = -0.5f * ((A.m[2][1] - B.m[2][1]) - (A.m[1][2] - B.m[1][2]));
        const float g1 = -0.5f * ((A.m[0][2] - B.m[0][2]) - (A.m[2][0] - B.m[2][0]));
        const float g2 = -0.5f * ((A.m[1][0] - B.m[1][0]) - (A.m[0][1] - B.m[0][1]));
#pragma unroll
        for (int i = 0; i < 3; ++i) out.J[i] = chain3(R.m[0][i], g0, R.m[1][i], g1, R.m[2][i], g2);
    }
    return out;
}

struct KParams {
    const float4 *src, *scov, *tgt, *tcov, *tnrm;
    const int32_t* nn_idx;
    const float* nn_d2;
    unsigned n;
    float max_d2, scale, genz_alpha, genz_thr;
    int rot_enable;  // rotation constraint (registration.hpp:559-561): weight and its own robust scale
    float rot_weight, rot_scale;
    Mat4Arg T_val;
    const float* T_dev;
};


template <int REG, int LOSS>
__global__ __launch_bounds__(kBlock) void linearize_kernel(KParams P, float* __restrict__ partials) {
    const Rigid T = load_rigid_colmajor(P.T_dev ? P.T_dev : P.T_val.m);
    float acc[kAcc - 1];
#pragma unroll
    for (int e = 0; e < kAcc - 1; ++e) acc[e] = 0.0f;
    unsigned cnt = 0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < P.n; i += gridDim.x * kBlock) {
        if (P.nn_d2[i] > P.max_d2) continue;
        const Corr c = gather<REG>(i, P.src, P.scov, P.tgt, P.tcov, P.tnrm, P.nn_idx);
        PointTerm pt;
        linearize_point<REG>(T, c, P.genz_alpha, P.genz_thr, pt);
        const float w = robust_weight<LOSS>(pt.residual_norm, P.scale);
        float err = robust_error<LOSS>(pt.residual_norm, P.scale);
        if (REG == SP_REG_GENZ) err = pt.genz_weight * err;
        float rw = 0.0f;  // rotation-constraint term of this correspondence (registration.hpp:630-650)
        RotTerm rot;
        rot.D = rot.J[0] = rot.J[1] = rot.J[2] = 0.0f;
        if (P.rot_enable) {
            const int ti = P.nn_idx[i];
            rot = rotation_divergence<true>(T, load_cov3(P.scov + 4 * (size_t)i), load_cov3(P.tcov + 4 * (size_t)ti));
            const float rn = sqrtf(0.5f * rot.D * rot.D);
            rw = P.rot_weight * robust_weight<LOSS>(rn, P.rot_scale);
            err += P.rot_weight * robust_error<LOSS>(rn, P.rot_scale);
        }
        int e = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int cc = a; cc < 6; ++cc) {
                float v = w * pt.H[a][cc];
                if (a < 3 && cc < 3) v += rw * (rot.J[a] * rot.J[cc]);
                acc[e++] += v;
            }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float v = w * pt.b[a];
            if (a < 3) v += rw * (rot.D * rot.J[a]);
            acc[21 + a] += v;
        }
        acc[27] += err;
        ++cnt;
    }
    block_reduce_store<kAcc - 1>(acc, cnt, partials + (size_t)blockIdx.x * kPartial);
}

template <int REG, int LOSS>
__global__ __launch_bounds__(kBlock) void error_kernel(KParams P, float* __restrict__ partials) {
    const Rigid T = load_rigid_colmajor(P.T_dev ? P.T_dev : P.T_val.m);
    float acc[1] = {0.0f};
    unsigned cnt = 0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < P.n; i += gridDim.x * kBlock) {
        if (P.nn_d2[i] > P.max_d2) continue;
        const Corr c = gather<REG>(i, P.src, P.scov, P.tgt, P.tcov, P.tnrm, P.nn_idx);
        float gw;
        const float sq = error_point<REG>(T, c, P.genz_alpha, P.genz_thr, gw);
        float err = robust_error<LOSS>(sqrtf(sq), P.scale);
        if (REG == SP_REG_GENZ) err = gw * err;
        if (P.rot_enable) {  // registration.hpp:758-766
            const int ti = P.nn_idx[i];
            const RotTerm rot =
                rotation_divergence<false>(T, load_cov3(P.scov + 4 * (size_t)i), load_cov3(P.tcov + 4 * (size_t)ti));
            err += P.rot_weight * robust_error<LOSS>(sqrtf(0.5f * rot.D * rot.D), P.rot_scale);
        }
        acc[0] += err;
        ++cnt;
    }
    block_reduce_store<1>(acc, cnt, partials + (size_t)blockIdx.x * kPartial);
}

template <int REG, int LOSS>
__global__ __launch_bounds__(kBlock) void weights_kernel(KParams P, float* __restrict__ out) {
    const Rigid T = load_rigid_colmajor(P.T_dev ? P.T_dev : P.T_val.m);
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < P.n; i += gridDim.x * kBlock) {
        float w = 0.0f;
        if (P.nn_d2[i] <= P.max_d2) {
            const Corr c = gather<REG>(i, P.src, P.scov, P.tgt, P.tcov, P.tnrm, P.nn_idx);
            float gw;
            const float sq = error_point<REG>(T, c, P.genz_alpha, P.genz_thr, gw);
            w = robust_weight<LOSS>(sqrtf(sq), P.scale);
        }
        out[i] = w;
    }
}

struct GnArgs {  // optional fused Gauss-Newton step (single-GPU loops): T == nullptr -> reduction only
    float* T;
    float lambda, crit_rot, crit_trans;
    float* delta_out8;
};


__global__ __launch_bounds__(kFinalThreads) void final_reduce_kernel(const float* __restrict__ partials,
                                                                     unsigned nblocks, int nv,
                                                                     sp_linearized* __restrict__ out, GnArgs gn) {
    __shared__ float red[kFinalThreads / 32][kPartial];
    __shared__ LdltScratch ldlt_ws;
    reduce_rows_1024(partials, nblocks, nv, red);
    if (threadIdx.x == 0) {
        unpack_totals(red[0], nv, out);
        if (gn.T) gn_update_impl(out, gn.T, gn.lambda, gn.crit_rot, gn.crit_trans, gn.delta_out8, false, ldlt_ws);
    }
}

__global__ void genz_counts_kernel(const float4* __restrict__ tcov, const int32_t* __restrict__ nn_idx,
                                   const float* __restrict__ nn_d2, unsigned n, float max_d2, float thr,
                                   uint32_t* __restrict__ counts) {
    unsigned inl = 0, pl = 0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (nn_d2[i] > max_d2) continue;
        if (genz_planar(load_cov3(tcov + 4 * (size_t)nn_idx[i]), thr)) ++pl;
        ++inl;
    }
    inl = wave_sum_u32(inl);
    pl = wave_sum_u32(pl);
    if ((threadIdx.x & (kWave - 1)) == 0) {  // integer atomics: order-independent, exact
        atomicAdd(&counts[0], inl);
        atomicAdd(&counts[1], pl);
    }
}

__global__ void gn_update_kernel(sp_linearized* lin, float* T, float lambda, float crit_rot, float crit_trans,
                                 float* delta_out8) {
    __shared__ LdltScratch ldlt_ws;
    if (threadIdx.x == 0 && blockIdx.x == 0) gn_update_impl(lin, T, lambda, crit_rot, crit_trans, delta_out8, true, ldlt_ws);
}

// ------------------------------------------------------------------------------------------------------------
// Fused GICP iteration on a prepared target (GridKNN + plane-regularised covariances stored in grid order).
//
// The reference recomputes update_covariance_plane for the source AND the target covariance of every correspondence
// in every iteration (factor.hpp:249-255: two eigen-decompositions per point per iteration). The regularised matrix
// depends only on the covariance itself, so it is computed ONCE per cloud (prepare kernels below) and stored packed
// (6 unique entries, 32-byte rows). The per-iteration kernel then does, per source point, in one pass:
//   q = T p  ->  exact NN on the grid (k = 1)  ->  Ct' of the winner read in grid order (next to the points just
//   scanned)  ->  M = (Ct' + R Cs' R^T)^-1  ->  H (21 unique), b, error accumulated in registers.
// No neighbour arrays are written or re-read unless the caller asks for them. Algorithmic bytes per correspondence:
// 192 B (NN 24 + K11 168); actually moved: 16 + 32 (source) + cell extents + scanned points + 32 (Ct').
// Mathematically identical to K2+K11; numerically the symmetric packing and the upper-triangle H differ from the
// reference's expression order by rounding only (tests: H, b within 2e-5 relative, final pose within 1e-5).

// P2D: the row holds the information matrix of point-to-distribution itself, inverse(Ct) of the RAW covariance
// (compute_target_mahalanobis, factor.hpp:311-317; Zero when |det| < 1e-6 as eigen_utils::inverse returns), symmetrised.
template <bool P2D>
__global__ __launch_bounds__(kBlock) void prepare_cov_kernel(const float4* __restrict__ covs, unsigned n,
                                                             const float4* __restrict__ order_pts,
                                                             const unsigned* __restrict__ order_idx,
                                                             float4* __restrict__ out,
                                                             const float* __restrict__ rho2 = nullptr) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    // row i of the output belongs to the point whose original index is order_pts[i].w / order_idx[i] (or i itself)
    const unsigned src = order_pts ? __float_as_uint(order_pts[i].w) : (order_idx ? order_idx[i] : i);
    const Mat3 C = load_cov3(covs + 4 * (size_t)src);
    const Mat3 P = P2D ? inverse(C) : plane_regularize(C);
    out[2 * (size_t)i] = make_float4(P.m[0][0], (P.m[0][1] + P.m[1][0]) * 0.5f, (P.m[0][2] + P.m[2][0]) * 0.5f, P.m[1][1]);
    // third slot of the second half-row (target side only): the point's squared safe radius (fused_point)
    out[2 * (size_t)i + 1] = make_float4((P.m[1][2] + P.m[2][1]) * 0.5f, P.m[2][2], rho2 ? rho2[src] : 0.0f, 0.0f);
}

// Certificates of the correspondence reuse (fused_point), from one k = 3 self-search on the target grid (row i, original
// order: the point itself at 0, its nearest other point u1, its second-nearest u2; -1 / FLT_MAX when missing):
//   rho2[i]  = (d(t, u1) / 2)^2            first test:  |q - t|^2 < rho2            (stored in t's covariance row)
//   nb[pos]  = (u1.xyz, (d(t, u2) / 2)^2)  second test: |q - t| < |q - u1| and |q - t|^2 < nb.w   (grid order)
// both shrunk by 1e-3 to stay clear of rounding. inv maps an original index to its grid position.
__global__ __launch_bounds__(kBlock) void inverse_order_kernel(const float4* __restrict__ gpts, unsigned n,
                                                               unsigned* __restrict__ inv) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) inv[__float_as_uint(gpts[i].w)] = i;
}
__global__ __launch_bounds__(kBlock) void certificate_kernel(const float4* __restrict__ gpts, unsigned n,
                                                             const int32_t* __restrict__ idx3,
                                                             const float* __restrict__ d23,
                                                             const unsigned* __restrict__ inv,
                                                             float* __restrict__ rho2, float4* __restrict__ nb, float bound2) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;  // grid position
    if (i >= n) return;
    const unsigned o = __float_as_uint(gpts[i].w);
    constexpr float kShrink = 0.25f * (1.0f - 1e-3f);
    // idx3 / d23: the three nearest target points of the point at grid position i (itself first), rows in GRID order.
    // The search was bounded (bound2: a point with nothing nearby does not walk the grid to its end); a neighbour it did not
    // find is farther than the bound, so the bound stands in for its distance: a smaller, still valid, radius.
    rho2[o] = fminf(d23[3 * (size_t)i + 1], bound2) * kShrink;
    const int u1 = idx3[3 * (size_t)i + 1];
    float4 r = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);  // no other point: the distance test always passes
    if (u1 >= 0) {
        const float4 p = gpts[inv[u1]];
        r.x = p.x; r.y = p.y; r.z = p.z;
    }
    r.w = fminf(d23[3 * (size_t)i + 2], bound2) * kShrink;
    nb[i] = r;
}

// Source ordering: key = cell (of the TARGET grid) that T*p falls into, so that consecutive lanes of the fused
// kernel walk consecutive cells of a grid row and their extent / point / covariance loads share cache lines.
__global__ __launch_bounds__(kBlock) void query_cell_kernel(const float4* __restrict__ pts, unsigned n, GridDesc g,
                                                            Mat4Arg T_val, const float* __restrict__ T_dev,
                                                            unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
    const float4 p = pts[i];
    float qx, qy, qz;
    transform_point(T, p.x, p.y, p.z, qx, qy, qz);
    unsigned key = (unsigned)g.nx * g.ny * g.nz;
    if (isfinite(qx) && isfinite(qy) && isfinite(qz)) {
        const int cx = cell_coord(qx, g.ox, g.inv_h, g.nx), cy = cell_coord(qy, g.oy, g.inv_h, g.ny),
                  cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
        key = ((unsigned)cz * g.ny + cy) * g.nx + cx;
    }
    keys[i] = key;
    vals[i] = i;
}
// gather_points + prepare_cov for the source in one pass over the permutation
__global__ __launch_bounds__(kBlock) void prepare_source_kernel(const float4* __restrict__ pts,
                                                                const float4* __restrict__ covs,
                                                                unsigned* __restrict__ order, bool identity, unsigned n,
                                                                unsigned stride, float* __restrict__ out_pts,
                                                                float* __restrict__ out_covp) {
    // Output as planes (x | y | z and xx | xy | xz | yy | yz | zz, `stride` floats apart): the per-iteration kernel is
    // bound by the bytes it streams, and the planes carry 36 bytes per point where float4 rows carry 48.
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned src = identity ? i : order[i];
    if (identity) order[i] = i;  // (the source is used as it lies: the permutation is written here instead of by a launch of its own)
    const float4 p = pts[src];
    out_pts[i] = p.x; out_pts[stride + i] = p.y; out_pts[2 * (size_t)stride + i] = p.z;
    if (!covs) return;  // point-to-distribution: the covariance planes are never read
    const Mat3 P = plane_regularize(load_cov3(covs + 4 * (size_t)src));
    out_covp[i] = P.m[0][0];
    out_covp[stride + i] = (P.m[0][1] + P.m[1][0]) * 0.5f;
    out_covp[2 * (size_t)stride + i] = (P.m[0][2] + P.m[2][0]) * 0.5f;
    out_covp[3 * (size_t)stride + i] = P.m[1][1];
    out_covp[4 * (size_t)stride + i] = (P.m[1][2] + P.m[2][1]) * 0.5f;
    out_covp[5 * (size_t)stride + i] = P.m[2][2];
}


template <int LOSS, bool FAST_NN, bool P2D = false>
__global__ __launch_bounds__(kBlock) void gicp_fused_kernel(FusedParams P, float* __restrict__ partials) {
    const Rigid T = load_rigid_colmajor(P.T_dev ? P.T_dev : P.T_val.m);
    float acc[kAcc - 1];
#pragma unroll
    for (int e = 0; e < kAcc - 1; ++e) acc[e] = 0.0f;
    unsigned cnt = 0, searched = 0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < P.n; i += gridDim.x * kBlock)
        fused_point<LOSS, FAST_NN, P2D, kSeedSearches, kNegCert>(P, T, i, acc, cnt, searched);
    block_reduce_store<kAcc - 1>(acc, cnt, partials + (size_t)blockIdx.x * kPartial, false, searched);
}

// K12 on the prepared path (Registration::compute_error_parallel_reduction, registration.hpp:678-777, as the LM and
// dog-leg trial steps call it, :854, :933): the error at a TRIAL pose with the correspondences FROZEN at those of the last
// linearisation. They are not re-read from neighbour arrays: the correspondence cache (source order, filled or confirmed by
// that linearisation) already holds every point's winner and its prepared covariance row, so a trial step is one coalesced
// stream of 36 + 48 bytes per point — no search, no gather, no eigen-decomposition (the generic sp_gicp_error gathers
// 168 B per point and decomposes two covariances per point). The inlier gate uses the distance at the LINEARISATION pose
// T_lin (nn_d2 in the reference), recomputed here with the search's own arithmetic, i.e. the same bits.
template <int LOSS, bool P2D>
__global__ __launch_bounds__(kBlock) void error_prepared_kernel(FusedParams P, Mat4Arg T_lin_val, const float* T_lin_dev,
                                                                float* __restrict__ partials) {
    const Rigid T = load_rigid_colmajor(P.T_dev ? P.T_dev : P.T_val.m);  // trial pose
    const Rigid TL = load_rigid_colmajor(T_lin_dev ? T_lin_dev : T_lin_val.m);
    float acc[1] = {0.0f};
    unsigned cnt = 0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < P.n; i += gridDim.x * kBlock)
        error_prepared_point<LOSS, P2D>(P, T, TL, i, acc, cnt);
    block_reduce_store<1>(acc, cnt, partials + (size_t)blockIdx.x * kPartial);
}

// ------------------------------------------------------------------------------------------------------------
// Whole alignment on the device (Registration::align's Gauss-Newton loop, registration.hpp:229-276), no host round trip:
// pose, convergence flag and iteration count live in a small state block in HBM that ping-pongs between iterations.
//
//   iteration k = gicp_align_kernel   (-> all-reduce -> align_solve_kernel, sharded runs)
//
// gicp_align_kernel streams the source once: per point the cached correspondence when its certificate holds, else a search
// (fused_point), linearisation, 28 sums in registers, one 32-float partial row per workgroup. On one GPU (ALIGN_PROLOGUE)
// iteration k - 1 is finished at the START of launch k, by every workgroup for itself: sum the previous launch's <= 256
// rows in a fixed order, solve (H + lambda I) delta = -b, update the pose (workgroup 0 also publishes the state).
// Letting the workgroup with the last arrival ticket finish its own launch instead (18 words for the next launch to read
// rather than 32 KB to sum in every workgroup) was built in round 3, measured on the same box and removed: the ticket round
// trip, the sum and the solve then all sit behind the slowest workgroup while 255 CUs idle, and the next launch still starts
// with a dependent read — 30.6 against 28.9 us per steady-state launch (profiles/r03_tail_solve_vs_prologue_same_box.txt).
// Once is_converged() holds, the remaining launches return immediately (the reference breaks out of its loop).
//
// Measured and NOT kept (profiles/r03_search_launch_experiments_not_kept.txt, profiles/README.md): moving the searches of the
// first iterations into a launch of their own (uncertified points compacted per workgroup, two queries per lane in
// lockstep, first block from a 4x duplicated "block row" copy of the target where it is one contiguous run). Standing
// alone, the exact first block of 1M queries takes 71 us (TA 67 % busy: ~44 TA cycles per scattered 16-byte wave load, 39 of
// them per 64 queries) against the 22 us it adds inside this kernel, where it overlaps the linearisation of other waves.

struct AlignState {
    float T[16];          // pose after the iterations finished so far
    float T_lin[16];      // pose of the latest linearisation: the correspondence cache is exact for it
    float delta[8];
    unsigned converged;   // is_converged() held for the step that produced T
    unsigned iterations;  // Gauss-Newton steps applied so far
    unsigned searched;    // source points the last finished iteration had to search for (the others reused their correspondence)
    unsigned pad;
};
constexpr int kStateWords = sizeof(AlignState) / 4;
constexpr int kStateFlagWord = 40;     // word index of AlignState::converged (iterations follows)
static_assert(offsetof(AlignState, converged) == 4 * kStateFlagWord, "AlignState layout");

enum { ALIGN_PROLOGUE = 0, ALIGN_ROWS = 1, ALIGN_FANIN = 2, ALIGN_DIRECT = 3 };

struct AlignArgs {
    // ALIGN_PROLOGUE: the fields below describe the iteration the launch FINISHES first (k = its index, launch index - 1;
    // `first`: launch 0, nothing to finish); every other mode: the iteration the launch (or align_solve_kernel) works on.
    const float* T_init;         // iteration 0
    const AlignState* state_in;  // state after iteration k - 1 (k > 0)
    AlignState* state_out;       // state after iteration k
    int has_prev;
    int first;
    const float* prev_rows;      // ALIGN_PROLOGUE: the partial rows of iteration k (written by the previous launch)
    float lambda, crit_rot, crit_trans;
    sp_linearized* lin_out;      // system of the last finished iteration (may be null)
    // How iteration k is finished: ALIGN_PROLOGUE by the next launch (or align_solve_kernel after the last one) from the
    // partial rows (one GPU); ALIGN_FANIN the launch's last-arriving workgroup sums the rows into ONE 128-byte row, the
    // caller all-reduces it over the ranks and align_solve_kernel
    // finishes; ALIGN_ROWS the caller all-reduces all partial rows (counts travel as floats), then align_solve_kernel;
    // ALIGN_DIRECT that workgroup stores the row into every rank's slot buffer and align_solve_kernel waits for all rows.
    int mode;
    unsigned* searched_log;      // [iteration] -> source points searched for
    int k;                       // index of the iteration that is being FINISHED (prologue / align_solve_kernel)
    int k_launch;                // index of the iteration the streaming launch itself works on (its row, its tag)
    float* fan_row_out;          // this iteration's row (kFanRow floats)
    const float* fan_row_in;     // align_solve_kernel: the same row, all-reduced over the ranks
    unsigned* fan_counter;       // arrival tickets; 0 when a launch starts, reset by the last arriver
    XchgArgs x;                  // ALIGN_DIRECT: where the row goes instead of a collective (sp_xchg.h)
};
// Row of the fan-in: 0..27 the sums, 28 / 29 the inlier count as two floats that stay exact under a float sum over ranks
// (count = hi * 4096 + lo, as sp_linearized carries it), 30 the searched-point count (a float value), 31 unused.
constexpr int kFanRow = 32;

// Pose (-> sT) and flags (-> sflag: converged, iterations) of this iteration. Returns false when an earlier iteration
// converged: workgroup 0 then carries the state forward and the launch has nothing to do.
__device__ __forceinline__ bool align_begin(const float* T_init, const AlignState* state_in, AlignState* state_out,
                                            int has_prev, float* sT, unsigned* sflag, float* zero_row = nullptr) {
    if (has_prev) {
        if (threadIdx.x < 18) {
            const unsigned v = reinterpret_cast<const unsigned*>(state_in)[threadIdx.x < 16 ? threadIdx.x
                                                                                             : kStateFlagWord + threadIdx.x - 16];
            if (threadIdx.x < 16) sT[threadIdx.x] = __uint_as_float(v);
            else sflag[threadIdx.x - 16] = v;
        }
    } else {
        if (threadIdx.x < 16) sT[threadIdx.x] = T_init[threadIdx.x];
        else if (threadIdx.x < 18) sflag[threadIdx.x - 16] = 0u;
    }
    __syncthreads();
    if (sflag[0]) {  // uniform over the grid
        if (blockIdx.x == 0 && state_out) {
            if (threadIdx.x < kStateWords)
                reinterpret_cast<unsigned*>(state_out)[threadIdx.x] = reinterpret_cast<const unsigned*>(state_in)[threadIdx.x];
            if (zero_row && threadIdx.x < kFanRow) zero_row[threadIdx.x] = 0.0f;  // a finished rank adds nothing to the all-reduce
        }
        return false;
    }
    return true;
}

// One thread: totals of iteration k (red0: 28 sums, the uint32 count, the searched count as a float value) at pose sT ->
// solve_linear_system + pose update + is_converged (registration.hpp:791-828, 407-410) -> state after iteration k.
__device__ __forceinline__ void align_finish_iteration(const AlignArgs& A, const float* red0, const float* sT,
                                                       unsigned prev_iterations, unsigned searched, sp_linearized& slin,
                                                       float* sTn, float* sdelta, LdltScratch& ldlt_ws) {
    unpack_totals(red0, kAcc - 1, &slin);
#pragma unroll
    for (int i = 0; i < 16; ++i) sTn[i] = sT[i];
    gn_update_impl(&slin, sTn, A.lambda, A.crit_rot, A.crit_trans, sdelta, false, ldlt_ws);
    AlignState* so = A.state_out;
#pragma unroll
    for (int i = 0; i < 16; ++i) { so->T[i] = sTn[i]; so->T_lin[i] = sT[i]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) so->delta[i] = sdelta[i];
    so->converged = sdelta[6] > 0.5f ? 1u : 0u;
    so->iterations = prev_iterations + 1;
    so->searched = searched;
    so->pad = 0;
    if (A.lin_out) *A.lin_out = slin;
}

// End of a streaming launch in the ALIGN_FANIN / ALIGN_DIRECT modes (MI355X_MICROARCH.md "fanin",
// cdna_hip_programming.md Guideline 16, the counter form with sc1 loads in place of an acquire; every condition of its table
// row holds):
//   * every word of the rows was stored write-through (store_row_word<true>: global_store ... sc1) by lanes of wave 0,
//   * wave 0 drains its stores (s_waitcnt vmcnt(0)) and only then lane 0 takes an agent-scope ticket,
//   * the workgroup whose ticket is the last one learns it from the value its add returned, tells its other waves through
//     LDS + barrier, and reads every row with sc1 loads (never a plain load of another workgroup's bytes),
//   * no fence, no spin: a workgroup either leaves or sums — nothing waits for a workgroup that has not been dispatched.
// The sum is reduce_rows_1024's fixed order over the launch's `grid` rows: bit-reproducible, the same bits as ALIGN_PROLOGUE's.
// The last arriver resets the ticket counter for the next launch.
__device__ __forceinline__ void align_tail(const AlignArgs& A, const float* __restrict__ partials) {
    __shared__ float red[kFinalThreads / 32][kPartial];
    __shared__ unsigned s_last;
    if (threadIdx.x < kWave) {  // the storing lanes (0 .. kAcc) all sit in wave 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(A.fan_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = ticket == gridDim.x - 1 ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_last) return;  // uniform per workgroup
    reduce_rows_1024<true>(partials, gridDim.x, kAcc - 1, red, false);
    const bool log_k = A.searched_log && A.k_launch < kSearchedLog;
    {
        if (threadIdx.x < kFanRow) {
            const unsigned cnt = __float_as_uint(red[0][kAcc - 1]);
            float v = 0.0f;
            if (threadIdx.x < kAcc - 1) v = red[0][threadIdx.x];
            else if (threadIdx.x == kAcc - 1) v = (float)(cnt & 4095u);
            else if (threadIdx.x == kAcc) v = (float)(cnt >> 12);
            else if (threadIdx.x == kAcc + 1) v = red[0][kAcc];
            A.fan_row_out[threadIdx.x] = v;
            if (A.mode == ALIGN_DIRECT) {  // the row, tagged, into slot [epoch & 1][k & 1][rank] of every rank's buffer (sp_xchg.h)
                const unsigned seq = *A.x.epoch * 256u + (unsigned)A.k_launch + 1u;
                const unsigned long long granule = ((unsigned long long)seq << 32) | __float_as_uint(v);
                const size_t slot = ((size_t)xchg_slot(*A.x.epoch, A.k_launch) * A.x.world + A.x.rank) * kFanRow + threadIdx.x;
                for (int r = 0; r < A.x.world; ++r)
                    __hip_atomic_store(A.x.peers[r] + slot, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (threadIdx.x == 0 && log_k) A.searched_log[A.k_launch] = (unsigned)red[0][kAcc];
    }
    if (threadIdx.x == 0) __hip_atomic_store(A.fan_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ALIGN_PROLOGUE, start of launch k + 1: finish iteration k (A describes it) in every workgroup — the previous launch's rows
// summed in reduce_rows_1024's fixed order, solve, pose update — and leave the new pose in sT. Workgroup 0 publishes the
// state. Returns false when the alignment had converged before, or converges with this step: nothing left to do.
// Totals of iteration A.k -> red[0] (28 sums, the uint32 count, the searched count as a float value), by mode: the partial
// rows of one launch (ALIGN_PROLOGUE) or the all-reduced ones (ALIGN_ROWS: counts as float values), the all-reduced fan-in
// row (ALIGN_FANIN), or the rows the ranks stored into this rank's slot buffer (ALIGN_DIRECT: bounded wait, rows added in
// rank order — the same sum on every rank). `after_loads` runs once the loads are issued. Returns false when a peer's row
// did not arrive in time. Ends with a barrier.
template <bool SHARDED, typename Hook>
__device__ __forceinline__ bool align_totals(const AlignArgs& A, const float* __restrict__ rows, unsigned nrows,
                                             float (*red)[kPartial], Hook after_loads, const unsigned* converged = nullptr) {
    if constexpr (!SHARDED) {
        reduce_rows_1024(rows, nrows, kAcc - 1, red, false, after_loads);
        return true;
    } else {
        if (A.mode == ALIGN_ROWS || A.mode == ALIGN_PROLOGUE) {
            reduce_rows_1024(rows, nrows, kAcc - 1, red, A.mode == ALIGN_ROWS, after_loads);
            return true;
        }
        if (A.mode == ALIGN_DIRECT) {
            __shared__ float xrow[kXchgMaxWorld][kFanRow];
            __shared__ unsigned s_late;
            after_loads();
            if (threadIdx.x == 0) s_late = 0u;
            __syncthreads();
            if (converged && *converged) return true;  // (set by `after_loads`: nobody has stored a row, do not wait for one)
            if (threadIdx.x < (unsigned)A.x.world * kFanRow) {
                const unsigned r = threadIdx.x / kFanRow, e = threadIdx.x % kFanRow;
                const unsigned epoch = *A.x.epoch;
                const unsigned long long* const g = A.x.local + ((size_t)xchg_slot(epoch, A.k) * A.x.world + r) * kFanRow + e;
                const unsigned seq = epoch * 256u + (unsigned)A.k + 1u;
                const unsigned long long t0 = wall_clock64();
                unsigned long long v = 0;
                bool ok = false;
                for (;;) {
                    v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((unsigned)(v >> 32) == seq) { ok = true; break; }
                    if (wall_clock64() - t0 > A.x.budget) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                if (!ok) s_late = 1u;
                xrow[r][e] = ok ? __uint_as_float((unsigned)v) : 0.0f;
            }
            __syncthreads();
            if (s_late) return false;
            if (threadIdx.x < kFanRow) {
                float sum = 0.0f;
                for (int r = 0; r < A.x.world; ++r) sum += xrow[r][threadIdx.x];
                red[1][threadIdx.x] = sum;
            }
        } else {  // ALIGN_FANIN
            const float rv = threadIdx.x < kFanRow ? A.fan_row_in[threadIdx.x] : 0.0f;
            after_loads();
            if (threadIdx.x < kFanRow) red[1][threadIdx.x] = rv;
        }
        __syncthreads();
        if (threadIdx.x < kAcc - 1) red[0][threadIdx.x] = red[1][threadIdx.x];
        else if (threadIdx.x == kAcc - 1)  // the count, folded as integers: exact
            red[0][kAcc - 1] = __uint_as_float((unsigned)red[1][kAcc] * 4096u + (unsigned)red[1][kAcc - 1]);
        else if (threadIdx.x == kAcc) red[0][kAcc] = red[1][kAcc + 1];  // searched points, all ranks (a float value)
        __syncthreads();
        return true;
    }
}
// A peer's row did not arrive (ALIGN_DIRECT): the state says converged + error, so the remaining launches return at once and
// sp_gicp_align_status reports it.
__device__ __forceinline__ void align_store_late_state(const AlignArgs& A) {
    if (threadIdx.x < kStateWords) {
        unsigned w = A.has_prev ? reinterpret_cast<const unsigned*>(A.state_in)[threadIdx.x] : 0u;
        if (!A.has_prev && threadIdx.x < 16) w = __float_as_uint(A.T_init[threadIdx.x]);
        if (threadIdx.x == kStateFlagWord) w = 1u;      // converged
        if (threadIdx.x == kStateFlagWord + 3) w = 1u;  // pad = error
        reinterpret_cast<unsigned*>(A.state_out)[threadIdx.x] = w;
    }
}

// (Tried: this step, or the whole prologue, as a noinline function to keep its registers apart from the point loop's. The
// step alone changes nothing; the whole prologue brings the POINT_TO_DISTRIBUTION instantiations from 1-5 spilled registers to
// 1-2 but gives every instantiation a 184-byte stack frame per lane. Inlined, the GICP instantiations have no spill except
// Tukey's one register.)
__device__ __forceinline__ void prologue_step(const float* red0, sp_linearized* slin, float* sT, float lambda,
                                                         float crit_rot, float crit_trans, float* sdelta, LdltScratch* ws) {
    unpack_totals(red0, kAcc - 1, slin);
    gn_update_impl(slin, sT, lambda, crit_rot, crit_trans, sdelta, false, *ws);
}
template <bool SHARDED>
__device__ __forceinline__ bool align_prologue(const AlignArgs& A, float* sT, unsigned* sflag) {
    __shared__ float red[kFinalThreads / 32][kPartial];
    __shared__ sp_linearized slin;
    __shared__ float sdelta[8];
    __shared__ LdltScratch ldlt_ws;
    if (A.first) {
        if (threadIdx.x < 16) sT[threadIdx.x] = A.T_init[threadIdx.x];
        else if (threadIdx.x < 18) sflag[threadIdx.x - 16] = 0u;
        // (a new alignment: the searched-point log starts empty — entry j is written when iteration j is finished; the sharded
        // modes clear it with their tickets and rows before launch 0)
        if (!SHARDED && blockIdx.x == 0 && A.searched_log && threadIdx.x >= 64 && threadIdx.x < 64 + kSearchedLog)
            A.searched_log[threadIdx.x - 64] = 0u;
        __syncthreads();
        return true;
    }
    // the previous state (18 words) is requested before the rows and stored after their loads have been issued: one
    // memory round trip for both
    unsigned sv = 0;
    if (threadIdx.x < 18) {
        if (A.has_prev)
            sv = reinterpret_cast<const unsigned*>(A.state_in)[threadIdx.x < 16 ? threadIdx.x : kStateFlagWord + threadIdx.x - 16];
        else
            sv = threadIdx.x < 16 ? __float_as_uint(A.T_init[threadIdx.x]) : 0u;
    }
    const unsigned nrows = (SHARDED && A.mode == ALIGN_ROWS) ? (unsigned)kAlignMaxBlocks : gridDim.x;
    const bool arrived = align_totals<SHARDED>(A, A.prev_rows, nrows, red, [=] {
        if (threadIdx.x < 16) sT[threadIdx.x] = __uint_as_float(sv);
        else if (threadIdx.x < 18) sflag[threadIdx.x - 16] = sv;
    }, sflag);
    if (!arrived) {  // (uniform per workgroup; every workgroup of every rank runs into the same bound)
        if (blockIdx.x == 0) align_store_late_state(A);
        return false;
    }
    // (a launch that has nothing to do still leaves a defined row for the caller's all-reduce: zeros)
    auto zero_row = [&] {
        if (SHARDED && A.mode == ALIGN_FANIN && blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + kFanRow)
            A.fan_row_out[threadIdx.x - 64] = 0.0f;
    };
    if (sflag[0]) {  // converged earlier (uniform over the grid): carry the state forward
        if (blockIdx.x == 0 && threadIdx.x < kStateWords)
            reinterpret_cast<unsigned*>(A.state_out)[threadIdx.x] = reinterpret_cast<const unsigned*>(A.state_in)[threadIdx.x];
        zero_row();
        return false;
    }
    if (threadIdx.x == 0) {
        // (the step is applied to sT in place: no copy, no second barrier on the path every workgroup waits on)
        AlignState* const so = A.state_out;
        const bool publish = blockIdx.x == 0;
        if (publish) {
#pragma unroll
            for (int i = 0; i < 16; ++i) so->T_lin[i] = sT[i];
        }
        prologue_step(red[0], &slin, sT, A.lambda, A.crit_rot, A.crit_trans, sdelta, &ldlt_ws);
        if (publish) {
            const unsigned searched = (unsigned)red[0][kAcc];
#pragma unroll
            for (int i = 0; i < 16; ++i) so->T[i] = sT[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) so->delta[i] = sdelta[i];
            so->converged = sdelta[6] > 0.5f ? 1u : 0u;
            so->iterations = sflag[1] + 1;
            so->searched = searched;
            so->pad = 0;
            if (A.searched_log && A.k < kSearchedLog) A.searched_log[A.k] = searched;
            if (A.lin_out) *A.lin_out = slin;
        }
    }
    __syncthreads();
    if (sdelta[6] > 0.5f) {  // converged with this step: no further linearisation (registration.hpp:266-268)
        zero_row();
        return false;
    }
    return true;
}

// SHARDED = false: the single-GPU form (ALIGN_PROLOGUE) with nothing of the other modes compiled in — the exchange code
// in the same kernel cost 0.4 us per launch (scalar register pressure at the kernel's start), measured on the same box.
template <int LOSS, bool FAST_NN, bool P2D = false, bool SHARDED = false>
__global__ __launch_bounds__(kAlignBlock) void gicp_align_kernel(FusedParams P, AlignArgs A,
                                                                 float* __restrict__ partials) {
    __shared__ float sT[16];
    __shared__ unsigned sflag[2];
    // (converged: every rank holds the same state and stops at the same launch — nobody waits for a row)
    if (!align_prologue<SHARDED>(A, sT, sflag)) return;
    // the pose is uniform: move it to scalar registers (it would otherwise occupy 12 VGPRs for the whole loop)
    Rigid T = load_rigid_colmajor(sT);
    auto uniform = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) T.R[r][c] = uniform(T.R[r][c]);
        T.t[r] = uniform(T.t[r]);
    }
    float acc[kAcc - 1];
    unsigned cnt = 0, searched = 0;
    const unsigned stride = gridDim.x * kAlignBlock;
#pragma unroll
    for (int e = 0; e < kAcc - 1; ++e) acc[e] = 0.0f;
    // XCD-aware tile order: workgroup b runs on XCD b % 8, so give every XCD one contiguous eighth of the
    // (cell-ordered) source: the target cells it reads then stay in that XCD's own L2 from one grid row / layer to the next
    unsigned tile = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if constexpr (FAST_NN && kWaveTail) {
        if (P.ccache != nullptr) {  // (uniform)
            for (unsigned b = tile * kAlignBlock; b < P.n; b += stride)
                fused_point_wave<LOSS, P2D>(P, T, b + threadIdx.x, b + threadIdx.x < P.n, acc, cnt, searched);
        } else {
            for (unsigned i = tile * kAlignBlock + threadIdx.x; i < P.n; i += stride)
                fused_point<LOSS, FAST_NN, P2D, kSeedSearches, kNegCert>(P, T, i, acc, cnt, searched);
        }
    } else {
        for (unsigned i = tile * kAlignBlock + threadIdx.x; i < P.n; i += stride)
            fused_point<LOSS, FAST_NN, P2D, kSeedSearches, kNegCert>(P, T, i, acc, cnt, searched);
    }
    if constexpr (!SHARDED) {
        block_reduce_store<kAcc - 1, kAlignBlock>(acc, cnt, partials + (size_t)blockIdx.x * kPartial, false, searched);
    } else if (A.mode == ALIGN_ROWS) {
        block_reduce_store<kAcc - 1, kAlignBlock>(acc, cnt, partials + (size_t)blockIdx.x * kPartial, true, searched);
    } else {
        block_reduce_store<kAcc - 1, kAlignBlock, true>(acc, cnt, partials + (size_t)blockIdx.x * kPartial, false, searched);
        align_tail(A, partials);
    }
}

// ------------------------------------------------------------------------------------------------------------
// The whole alignment as ONE launch (single GPU; sp_gicp_align_fused takes it when every workgroup is co-resident: one
// 1024-lane workgroup per CU, at most 256 of them). The iterations are separated by the arrival counter the sharded modes' fan-in
// already uses, not by kernel boundaries:
//   iteration k   per-point loop exactly as gicp_align_kernel -> the workgroup's partial row, stored write-through (sc1) ->
//                 wave 0 drains its stores, lane 0 adds 1 to the counter shard of its XCD (agent scope)
//   then          every workgroup waits until the shards add up to grid * (k + 1) — one lane polls with sc1 loads and s_sleep,
//                 bounded by wall_clock64 (a grid that is not fully resident ends with the error word set, never hangs) — and
//                 runs today's prologue on the rows (sc1 loads, same fixed order: the same bits as the per-launch form)
// (MI355X_MICROARCH.md, hand-off table, first row: one lane signals for all its workgroup's sc1 stores behind the storing
// wave's vmcnt(0) wait; the consumer polls with sc1 loads, the polling wave loads after its poll has matched, the others behind
// a workgroup barrier it joins.) What it removes: the kernel boundary and the launch ramp of every iteration, the launches
// that only find out that an earlier iteration converged (an alignment that converges after 3 of 20 iterations enqueued 17 of
// them, 2.5 us each), and the finish launch (workgroup 0 finishes the last iteration itself).
// Rows ping-pong by the parity of k: a workgroup can only write row k + 1 after every workgroup has stored row k, i.e. after
// every workgroup has finished reading the rows of k - 1 that it overwrites.
struct PersistArgs {
    float* part[2];              // partial rows, ping-pong
    AlignState* state;           // [2]
    unsigned* searched_log;
    unsigned* tickets;           // kTicketShards counters, kTicketStride words apart, zero when the launch starts
    const float* T_init;         // device: initial guess (read before T_out is written)
    float lambda, crit_rot, crit_trans;
    sp_linearized* lin_out;
    float* T_out;                // device: final pose (may be T_init)
    float* delta_out8;
    uint32_t* iterations_out;
    int max_iterations;
    int k_begin;                 // first iteration this launch runs; iterations 0 .. k_begin - 1 were launched one by one, the
                                 // rows of k_begin - 1 are in part[(k_begin - 1) & 1], the state after k_begin - 2 in state[k_begin & 1]
    int cache_valid_later;       // launches after the first may trust the cache (reuse switched on)
    unsigned long long budget;   // wall_clock64 ticks a wait may take
};
template <int LOSS, bool FAST_NN, bool P2D = false>
__global__ __launch_bounds__(kAlignBlock) void gicp_align_persistent_kernel(FusedParams P, PersistArgs A) {
    __shared__ float sT[16];
    __shared__ float red[kFinalThreads / 32][kPartial];
    __shared__ sp_linearized slin;
    __shared__ float sdelta[8];
    __shared__ LdltScratch ldlt_ws;
    __shared__ unsigned s_flag;  // 0 go on, 1 converged / done, 2 a wait ran out
    const unsigned stride = gridDim.x * kAlignBlock;
    unsigned tile = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    __shared__ unsigned s_prev[2];  // converged, iterations of the state this launch starts from
    if (A.k_begin >= 2) {  // the state after iteration k_begin - 2: the pose launch k_begin - 1 ran at, the flags
        if (threadIdx.x < 18) {
            const unsigned v = reinterpret_cast<const unsigned*>(&A.state[A.k_begin & 1])[threadIdx.x < 16 ? threadIdx.x : kStateFlagWord + threadIdx.x - 16];
            if (threadIdx.x < 16) sT[threadIdx.x] = __uint_as_float(v);
            else s_prev[threadIdx.x - 16] = v;
        }
    } else {
        if (threadIdx.x < 16) sT[threadIdx.x] = A.T_init[threadIdx.x];
        else if (threadIdx.x < 18) s_prev[threadIdx.x - 16] = 0u;
    }
    if (A.k_begin == 0 && blockIdx.x == 0 && A.searched_log && threadIdx.x >= 64 && threadIdx.x < 64 + kSearchedLog)
        A.searched_log[threadIdx.x - 64] = 0u;
    __syncthreads();
    if (s_prev[0]) {  // converged before this launch (uniform): workgroup 0 hands out the results of that state
        if (blockIdx.x == 0) {
            const AlignState* const sp_ = &A.state[A.k_begin & 1];
            if (threadIdx.x < kStateWords)
                reinterpret_cast<unsigned*>(&A.state[(A.k_begin + 1) & 1])[threadIdx.x] = reinterpret_cast<const unsigned*>(sp_)[threadIdx.x];
            if (threadIdx.x < 16) A.T_out[threadIdx.x] = sp_->T[threadIdx.x];
            else if (threadIdx.x < 24 && A.delta_out8) A.delta_out8[threadIdx.x - 16] = sp_->delta[threadIdx.x - 16];
            else if (threadIdx.x == 24 && A.iterations_out) *A.iterations_out = sp_->iterations;
        }
        return;
    }
    unsigned iterations = s_prev[1];
    for (int k = A.k_begin;; ++k) {
        if (k > 0) {
            // ---- wait for the rows of iteration k - 1 (a kernel boundary did, for the first one), then finish it: the prologue
            // of gicp_align_kernel
            if (threadIdx.x == 0) {
                const unsigned want = gridDim.x * (unsigned)(k - A.k_begin);
                const unsigned long long t0 = wall_clock64();
                unsigned flag = 0;
                for (;;) {
                    unsigned have = 0;
#pragma unroll
                    for (int sh = 0; sh < kTicketShards; ++sh)
                        have += __hip_atomic_load(A.tickets + sh * kTicketStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (have >= want) break;
                    if (wall_clock64() - t0 > A.budget) { flag = 2; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                s_flag = flag;
            }
            __syncthreads();
            if (s_flag == 2) {  // (uniform per workgroup; every workgroup runs into the same bound)
                if (blockIdx.x == 0 && threadIdx.x < 16) A.T_out[threadIdx.x] = __int_as_float(0x7fc00000);  // loud: NaN pose
                if (blockIdx.x == 0 && threadIdx.x == 16 && A.iterations_out) *A.iterations_out = 0xffffffffu;
                if (blockIdx.x == 0 && threadIdx.x == 17) { A.state[0].pad = 2u; A.state[1].pad = 2u; }
                return;
            }
            reduce_rows_1024<true>(A.part[(k - 1) & 1], gridDim.x, kAcc - 1, red, false, [] {});
            if (threadIdx.x == 0) {
                const bool publish = blockIdx.x == 0;
                AlignState* const so = &A.state[(k - 1) & 1];
                if (publish) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) so->T_lin[i] = sT[i];
                }
                prologue_step(red[0], &slin, sT, A.lambda, A.crit_rot, A.crit_trans, sdelta, &ldlt_ws);
                const bool conv = sdelta[6] > 0.5f;
                const bool last = conv || k == A.max_iterations;
                if (publish) {
                    const unsigned searched = (unsigned)red[0][kAcc];
#pragma unroll
                    for (int i = 0; i < 16; ++i) so->T[i] = sT[i];
#pragma unroll
                    for (int i = 0; i < 8; ++i) so->delta[i] = sdelta[i];
                    so->converged = conv ? 1u : 0u;
                    so->iterations = iterations + 1;
                    so->searched = searched;
                    so->pad = 0;
                    if (A.searched_log && k - 1 < kSearchedLog) A.searched_log[k - 1] = searched;
                    if (A.lin_out) *A.lin_out = slin;
                    if (last) {  // the outputs, and the same state in the other slot (whoever reads state[last & 1] finds it)
                        A.state[k & 1] = *so;
#pragma unroll
                        for (int i = 0; i < 16; ++i) A.T_out[i] = sT[i];
                        if (A.delta_out8)
#pragma unroll
                            for (int i = 0; i < 8; ++i) A.delta_out8[i] = sdelta[i];
                        if (A.iterations_out) *A.iterations_out = iterations + 1;
                    }
                }
                s_flag = last ? 1u : 0u;
            }
            __syncthreads();
            ++iterations;
            if (s_flag) return;  // converged with this step, or the last iteration is finished (uniform over the grid)
        }
        const Rigid T = uniform_pose(sT);
        float acc[kAcc - 1];
        unsigned cnt = 0, searched = 0;
#pragma unroll
        for (int e = 0; e < kAcc - 1; ++e) acc[e] = 0.0f;
        if (k == 1) P.cache_valid = A.cache_valid_later;  // (P is this kernel's own copy of the parameters; a launch that
                                                           //  begins later was handed the right value)
        for (unsigned i = tile * kAlignBlock + threadIdx.x; i < P.n; i += stride)
            fused_point<LOSS, FAST_NN, P2D, kSeedSearches, kNegCert>(P, T, i, acc, cnt, searched);
        block_reduce_store<kAcc - 1, kAlignBlock, true>(acc, cnt, A.part[k & 1] + (size_t)blockIdx.x * kPartial, false, searched);
        if (threadIdx.x < kWave) {  // the storing lanes all sit in wave 0
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0)
                __hip_atomic_fetch_add(A.tickets + (blockIdx.x & (kTicketShards - 1)) * kTicketStride, 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
        // (after the last iteration's rows only workgroup 0 is needed: it finishes the iteration and writes the outputs)
        if (k == A.max_iterations - 1 && blockIdx.x != 0) return;
    }
}

// Sharded runs: finishes iteration k on every rank from the all-reduced row (ALIGN_FANIN) or rows (ALIGN_ROWS) — the same
// sums, the same solve, hence the identical pose on every rank without a broadcast. One workgroup.
struct AlignPublish {  // after the LAST iteration: the results out of the state block (all null: nothing to publish)
    float* T_out;
    float* delta_out8;
    uint32_t* iterations_out;
    unsigned* xchg_epoch;
};
__device__ __forceinline__ void align_publish(const AlignState* state, const AlignPublish& Pb) {
    // (the state was stored by a thread of this workgroup: fence + barrier, then read past the L1)
    __threadfence();
    __syncthreads();
    if (!Pb.T_out) return;
    const unsigned* const w = reinterpret_cast<const unsigned*>(state);
    if (threadIdx.x < 16)
        reinterpret_cast<unsigned*>(Pb.T_out)[threadIdx.x] = __hip_atomic_load(w + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (threadIdx.x < 24 && Pb.delta_out8)
        reinterpret_cast<unsigned*>(Pb.delta_out8)[threadIdx.x - 16] =
            __hip_atomic_load(w + 32 + (threadIdx.x - 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (threadIdx.x == 24 && Pb.iterations_out)
        *Pb.iterations_out = __hip_atomic_load(w + kStateFlagWord + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (threadIdx.x == 32 && Pb.xchg_epoch)
        *Pb.xchg_epoch += 1u;  // direct exchange: the next alignment's tags (sp_xchg.h)
}
static_assert(offsetof(AlignState, delta) == 4 * 32, "AlignState layout");

__global__ __launch_bounds__(kFinalThreads) void align_solve_kernel(AlignArgs A, const float* __restrict__ rows,
                                                                    unsigned nrows, AlignPublish Pb) {
    __shared__ float red[kFinalThreads / 32][kPartial];
    __shared__ float sT[16];
    __shared__ unsigned sflag[2];
    __shared__ sp_linearized slin;
    __shared__ float sTn[16];
    __shared__ float sdelta[8];
    __shared__ LdltScratch ldlt_ws;
    if (!align_begin(A.T_init, A.state_in, A.state_out, A.has_prev, sT, sflag)) {
        align_publish(A.state_out, Pb);
        return;
    }
    if (!align_totals<true>(A, rows, nrows, red, [] {})) {
        align_store_late_state(A);
        align_publish(A.state_out, Pb);
        return;
    }
    if (threadIdx.x == 0) {
        const unsigned searched = (unsigned)red[0][kAcc];
        if ((A.mode == ALIGN_ROWS || A.mode == ALIGN_PROLOGUE) && A.searched_log && A.k < kSearchedLog) A.searched_log[A.k] = searched;
        align_finish_iteration(A, red[0], sT, sflag[1], searched, slin, sTn, sdelta, ldlt_ws);
    }
    align_publish(A.state_out, Pb);
}

unsigned reduce_grid(size_t n) {
    unsigned g = div_up(n, kBlock);
    if (g > kMaxBlocks) g = kMaxBlocks;
    return g ? g : 1;
}

KParams make_params(const float* src, const float* scov, size_t n, const float* tgt, const float* tcov,
                    const float* tnrm, const int32_t* nn_idx, const float* nn_d2, const float* T, int T_dev,
                    const sp_factor_params* fp) {
    KParams P;
    P.src = reinterpret_cast<const float4*>(src);
    P.scov = reinterpret_cast<const float4*>(scov);
    P.tgt = reinterpret_cast<const float4*>(tgt);
    P.tcov = reinterpret_cast<const float4*>(tcov);
    P.tnrm = reinterpret_cast<const float4*>(tnrm);
    P.nn_idx = nn_idx;
    P.nn_d2 = nn_d2;
    P.n = (unsigned)n;
    P.max_d2 = fp->max_correspondence_distance * fp->max_correspondence_distance;  // registration.hpp:554-555
    P.scale = fp->robust_scale;
    P.genz_alpha = fp->genz_alpha;
    P.genz_thr = fp->genz_planarity_threshold;
    P.rot_enable = fp->rotation_constraint_enable;
    P.rot_weight = fp->rotation_constraint_weight;
    P.rot_scale = fp->rotation_robust_scale;
    for (int i = 0; i < 16; ++i) P.T_val.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    if (T && !T_dev)
        for (int i = 0; i < 16; ++i) P.T_val.m[i] = T[i];
    P.T_dev = T_dev ? T : nullptr;
    return P;
}

enum Which { K_LINEARIZE, K_ERROR, K_WEIGHTS };

template <int REG, int LOSS>
void launch_one(Which which, const KParams& P, float* partials, float* weights, unsigned grid, hipStream_t st) {
    if (which == K_LINEARIZE) linearize_kernel<REG, LOSS><<<grid, kBlock, 0, st>>>(P, partials);
    else if (which == K_ERROR) error_kernel<REG, LOSS><<<grid, kBlock, 0, st>>>(P, partials);
    else weights_kernel<REG, LOSS><<<grid, kBlock, 0, st>>>(P, weights);
}
template <int REG>
bool launch_loss(Which which, int loss, const KParams& P, float* partials, float* weights, unsigned grid,
                 hipStream_t st) {
    switch (loss) {  // Registration::dispatch (registration.hpp:372-405)
        case SP_LOSS_NONE: launch_one<REG, LOSS_NONE>(which, P, partials, weights, grid, st); return true;
        case SP_LOSS_HUBER: launch_one<REG, LOSS_HUBER>(which, P, partials, weights, grid, st); return true;
        case SP_LOSS_TUKEY: launch_one<REG, LOSS_TUKEY>(which, P, partials, weights, grid, st); return true;
        case SP_LOSS_CAUCHY: launch_one<REG, LOSS_CAUCHY>(which, P, partials, weights, grid, st); return true;
        case SP_LOSS_GEMAN_MCCLURE: launch_one<REG, LOSS_GEMAN_MCCLURE>(which, P, partials, weights, grid, st); return true;
    }
    return false;
}
bool launch_reg(Which which, int reg, int loss, const KParams& P, float* partials, float* weights, unsigned grid,
                hipStream_t st) {
    switch (reg) {
        case SP_REG_POINT_TO_POINT: return launch_loss<SP_REG_POINT_TO_POINT>(which, loss, P, partials, weights, grid, st);
        case SP_REG_POINT_TO_PLANE: return launch_loss<SP_REG_POINT_TO_PLANE>(which, loss, P, partials, weights, grid, st);
        case SP_REG_POINT_TO_DISTRIBUTION: return launch_loss<SP_REG_POINT_TO_DISTRIBUTION>(which, loss, P, partials, weights, grid, st);
        case SP_REG_GICP: return launch_loss<SP_REG_GICP>(which, loss, P, partials, weights, grid, st);
        case SP_REG_GENZ: return launch_loss<SP_REG_GENZ>(which, loss, P, partials, weights, grid, st);
    }
    return false;
}

// validate_params (registration.hpp:129-193): which attribute arrays a reg_type needs.
int validate(const sp_factor_params* fp, const float* scov, const float* tcov, const float* tnrm) {
    if (!fp) return SP_ERR_INVALID_ARGUMENT;
    if (fp->reg_type == SP_REG_GICP && (!scov || !tcov)) {
        sp_set_error("[Registration::validate_params] Covariance matrices of source and target must be pre-computed "
                     "before performing GICP matching.");
        return SP_ERR_RUNTIME;
    }
    if (fp->reg_type == SP_REG_POINT_TO_DISTRIBUTION && !tcov) {
        sp_set_error("[Registration::validate_params] Covariance matrices of target must be pre-computed before "
                     "performing Point-to-Distribution ICP matching.");
        return SP_ERR_RUNTIME;
    }
    if (fp->reg_type == SP_REG_GENZ && (!tcov || !tnrm)) {
        sp_set_error("[Registration::validate_params] Covariance matrices and normals of target must be pre-computed "
                     "before performing GenZ-ICP matching.");
        return SP_ERR_RUNTIME;
    }
    if (fp->rotation_constraint_enable && !scov) {
        sp_set_error("[Registration::validate_params] Covariance matrices of source are required for performing "
                     "rotation constraint matching.");
        return SP_ERR_RUNTIME;
    }
    if (fp->rotation_constraint_enable && !tcov) {
        sp_set_error("[Registration::validate_params] Covariance matrices of target are required for performing "
                     "rotation constraint matching.");
        return SP_ERR_RUNTIME;
    }
    if (fp->reg_type == SP_REG_POINT_TO_PLANE && !tnrm) {
        sp_set_error("[Registration::validate_params] Normal vector of target must be pre-computed before performing "
                     "Point-to-Plane ICP matching.");
        return SP_ERR_RUNTIME;
    }
    return SP_OK;
}

int run_reduction(Which which, const float* src, const float* scov, size_t n, const float* tgt, const float* tcov,
                  const float* tnrm, const int32_t* nn_idx, const float* nn_d2, const float* T, int T_dev,
                  const sp_factor_params* fp, sp_linearized* out, void* ws, size_t ws_bytes, hipStream_t st) {
    const int v = validate(fp, scov, tcov, tnrm);
    if (v != SP_OK) return v;
    if (n >= (1ull << 32)) { sp_set_error("[Registration] more than 2^32 source points"); return SP_ERR_INVALID_ARGUMENT; }
    if (n == 0) return zero_async(out, sizeof(sp_linearized), st);
    if (!ws || ws_bytes < sp_gicp_workspace_bytes(n)) {
        sp_set_error("[Registration] workspace too small (sp_gicp_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    const KParams P = make_params(src, scov, n, tgt, tcov, tnrm, nn_idx, nn_d2, T, T_dev, fp);
    const unsigned grid = reduce_grid(n);
    float* partials = static_cast<float*>(ws);
    if (!launch_reg(which, fp->reg_type, fp->robust_type, P, partials, nullptr, grid, st)) {
        sp_set_error("[Registration::dispatch] Combination not found in tags!");
        return SP_ERR_RUNTIME;
    }
    final_reduce_kernel<<<1, kFinalThreads, 0, st>>>(partials, grid, which == K_LINEARIZE ? kAcc - 1 : 1, out,
                                                     GnArgs{nullptr, 0.0f, 0.0f, 0.0f, nullptr});
    return launch_status();
}

}  // namespace
}  // namespace sp

extern "C" size_t sp_gicp_workspace_bytes(size_t n) { return (size_t)sp::kMaxBlocks * sp::kPartial * sizeof(float); }

extern "C" int sp_gicp_linearize(const float* src_points, const float* src_covs, size_t n, const float* tgt_points,
                                 const float* tgt_covs, const float* tgt_normals, const int32_t* nn_idx,
                                 const float* nn_d2, const float* transT, int transT_on_device,
                                 const sp_factor_params* params, sp_linearized* out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    return sp::run_reduction(sp::K_LINEARIZE, src_points, src_covs, n, tgt_points, tgt_covs, tgt_normals, nn_idx, nn_d2,
                             transT, transT_on_device, params, out, workspace, workspace_bytes, sp::as_stream(stream));
}
extern "C" int sp_gicp_error(const float* src_points, const float* src_covs, size_t n, const float* tgt_points,
                             const float* tgt_covs, const float* tgt_normals, const int32_t* nn_idx, const float* nn_d2,
                             const float* transT, int transT_on_device, const sp_factor_params* params,
                             sp_linearized* out, void* workspace, size_t workspace_bytes, void* stream) {
    return sp::run_reduction(sp::K_ERROR, src_points, src_covs, n, tgt_points, tgt_covs, tgt_normals, nn_idx, nn_d2,
                             transT, transT_on_device, params, out, workspace, workspace_bytes, sp::as_stream(stream));
}
extern "C" int sp_icp_robust_weights(const float* src_points, const float* src_covs, size_t n, const float* tgt_points,
                                     const float* tgt_covs, const float* tgt_normals, const int32_t* nn_idx,
                                     const float* nn_d2, const float* transT, int transT_on_device,
                                     const sp_factor_params* params, float* weights_out, void* stream) {
    using namespace sp;
    const int v = validate(params, src_covs, tgt_covs, tgt_normals);
    if (v != SP_OK) return v;
    if (n == 0) return SP_OK;
    const KParams P = make_params(src_points, src_covs, n, tgt_points, tgt_covs, tgt_normals, nn_idx, nn_d2, transT,
                                  transT_on_device, params);
    if (!launch_reg(K_WEIGHTS, params->reg_type, params->robust_type, P, nullptr, weights_out, stream_grid(n),
                    as_stream(stream))) {
        sp_set_error("[Registration::dispatch] Combination not found in tags!");
        return SP_ERR_RUNTIME;
    }
    return launch_status();
}
extern "C" int sp_genz_counts(const float* tgt_covs, const int32_t* nn_idx, const float* nn_d2, size_t n, float max_corr,
                              float planarity_threshold, uint32_t* counts_out, void* stream) {
    using namespace sp;
    hipStream_t st = as_stream(stream);
    if (zero_async(counts_out, 8, st) != SP_OK) return SP_ERR_HIP;
    if (n == 0) return SP_OK;
    genz_counts_kernel<<<reduce_grid(n), kBlock, 0, st>>>(reinterpret_cast<const float4*>(tgt_covs), nn_idx, nn_d2,
                                                          (unsigned)n, max_corr * max_corr, planarity_threshold,
                                                          counts_out);
    return launch_status();
}
extern "C" int sp_gn_update(sp_linearized* lin, float* T_dev, float lambda, float crit_rotation, float crit_translation,
                            float* delta_out8, void* stream) {
    sp::gn_update_kernel<<<1, 64, 0, sp::as_stream(stream)>>>(lin, T_dev, lambda, crit_rotation, crit_translation,
                                                              delta_out8);
    return sp::launch_status();
}
extern "C" int sp_gn_update_host(const sp_linearized* lin_host, float* T_host, float lambda, float crit_rotation,
                                 float crit_translation, float* delta_out8_host) {
    sp_linearized tmp = *lin_host;
    sp::LdltScratch w;
    sp::gn_update_impl(&tmp, T_host, lambda, crit_rotation, crit_translation, delta_out8_host, false, w);
    return SP_OK;
}

// ------------------------------------------------------------------ prepared / fused path (C ABI)


extern "C" void sp_gicp_target_destroy(sp_gicp_target* t) {
    if (!t) return;
    // back to the pool tagged with an event per stream the rows were used on: no device-wide wait in a destructor
    sp::pooled_free_after(t->covp, t->streams);
    sp::pooled_free_after(t->rho2, t->streams);
    sp::pooled_free_after(t->nb, t->streams);
    delete t;
}
extern "C" int sp_gicp_target_prepare(sp_gicp_target* t, const float* tgt_covs, int reg_type, void* stream) {
    using namespace sp;
    if (!t || !tgt_covs) {
        sp_set_error(reg_type == SP_REG_POINT_TO_DISTRIBUTION
                         ? "[Registration::validate_params] Covariance matrices of target must be pre-computed before "
                           "performing Point-to-Distribution ICP matching."
                         : "[Registration::validate_params] Covariance matrices of source and target must be pre-computed "
                           "before performing GICP matching.");
        return SP_ERR_RUNTIME;
    }
    if (reg_type != SP_REG_GICP && reg_type != SP_REG_POINT_TO_DISTRIBUTION) {
        sp_set_error("[sp_gicp_target_prepare] only RegType::GICP and RegType::POINT_TO_DISTRIBUTION have a prepared form");
        return SP_ERR_INVALID_ARGUMENT;
    }
    ++t->version;
    t->reg_type = reg_type;
    if (t->n == 0) return SP_OK;
    t->note(as_stream(stream));
    const float4* covs = reinterpret_cast<const float4*>(tgt_covs);
    if (reg_type == SP_REG_GICP)
        prepare_cov_kernel<false><<<div_up(t->n, kBlock), kBlock, 0, as_stream(stream)>>>(covs, (unsigned)t->n, t->grid->d_pts,
                                                                                          nullptr, t->covp, t->rho2);
    else
        prepare_cov_kernel<true><<<div_up(t->n, kBlock), kBlock, 0, as_stream(stream)>>>(covs, (unsigned)t->n, t->grid->d_pts,
                                                                                         nullptr, t->covp, t->rho2);
    return launch_status();
}
extern "C" int sp_gicp_target_update(sp_gicp_target* t, const float* tgt_covs, void* stream) {
    return sp_gicp_target_prepare(t, tgt_covs, t ? t->reg_type : SP_REG_GICP, stream);
}
namespace sp {
namespace {
// Certificates of the correspondence reuse: one k = 3 search of the target's own points on its grid, once per target. The
// queries are the grid's cell-ordered copy of the points, so neighbouring lanes walk neighbouring cells (0.20 ms per 1M
// points; the wave / tile self-kNN kernels are built for long lists and take 0.6 ms here) and the rows come out in grid
// order. Synchronises (temporaries from the library's pool: idle again afterwards).
int build_certificates(sp_gicp_target* t, hipStream_t st) {
    const sp_grid* const grid = t->grid;
    const size_t n = t->n;
    if (n == 0 || t->rho2 != nullptr) return SP_OK;
    ScratchBuf b_idx3, b_d23, b_inv;
    hipError_t e = pooled_alloc(&t->rho2, n * sizeof(float), st);
    if (e == hipSuccess) e = pooled_alloc(&t->nb, n * sizeof(float4), st);
    if (e == hipSuccess) e = b_idx3.get(n * 3 * sizeof(int32_t), st);
    if (e == hipSuccess) e = b_d23.get(n * 3 * sizeof(float), st);
    if (e == hipSuccess) e = b_inv.get(n * sizeof(unsigned), st);
    int32_t* const idx3 = b_idx3.as<int32_t>();
    float* const d23 = b_d23.as<float>();
    unsigned* const inv = b_inv.as<unsigned>();
    int rc = e == hipSuccess ? SP_OK : SP_ERR_HIP;
    // (neighbours beyond three cells are not looked for: isolated points of a scan would walk thousands of empty cells)
    const float bound2 = 9.0f * grid->h * grid->h;
    if (rc == SP_OK) rc = grid_search_own_points(grid, 3, idx3, d23, st, bound2);  // (in cell order already: no sort of the queries)
    if (rc == SP_OK) {
        inverse_order_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(grid->d_pts, (unsigned)n, inv);
        certificate_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(grid->d_pts, (unsigned)n, idx3, d23, inv, t->rho2, t->nb, bound2);
        rc = launch_status();
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == SP_OK) rc = SP_ERR_HIP;
    if (rc != SP_OK) {
        if (e != hipSuccess) sp_set_error(hipGetErrorString(e));
        pooled_free(t->rho2); t->rho2 = nullptr;  // (synchronised: nothing uses them)
        pooled_free(t->nb); t->nb = nullptr;
    }
    return rc;
}
int target_create(const sp_grid* grid, const float* tgt_covs, size_t n, bool certificates, void* stream, sp_gicp_target** out) {
    if (!out || !grid) return SP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (grid->n != n) {
        sp_set_error("[sp_gicp_target_create] the grid was built on a cloud of a different size");
        return SP_ERR_INVALID_ARGUMENT;
    }
    sp_gicp_target* t = new sp_gicp_target();
    t->grid = grid;
    t->n = n;
    t->note(as_stream(stream));
    if (n) {
        const hipError_t e = pooled_alloc(&t->covp, n * 2 * sizeof(float4), as_stream(stream));
        if (e != hipSuccess) {
            sp_set_error(hipGetErrorString(e));
            sp_gicp_target_destroy(t);
            return SP_ERR_HIP;
        }
    }
    if (certificates) {
        const int rc2 = build_certificates(t, as_stream(stream));
        if (rc2 != SP_OK) { sp_gicp_target_destroy(t); return rc2; }
    }
    const int rc = sp_gicp_target_update(t, tgt_covs, stream);
    if (rc != SP_OK) { sp_gicp_target_destroy(t); return rc; }
    *out = t;
    return SP_OK;
}
}  // namespace
}  // namespace sp

extern "C" int sp_gicp_target_create(const sp_grid* grid, const float* tgt_covs, size_t n, void* stream,
                                     sp_gicp_target** out) {
    return sp::target_create(grid, tgt_covs, n, true, stream, out);
}
extern "C" int sp_gicp_target_create_plain(const sp_grid* grid, const float* tgt_covs, size_t n, void* stream,
                                           sp_gicp_target** out) {
    return sp::target_create(grid, tgt_covs, n, false, stream, out);
}
extern "C" int sp_gicp_target_certify(sp_gicp_target* t, const float* tgt_covs, void* stream) {
    if (!t || !tgt_covs) return SP_ERR_INVALID_ARGUMENT;
    if (t->rho2 != nullptr) return SP_OK;
    t->note(sp::as_stream(stream));
    const int rc = sp::build_certificates(t, sp::as_stream(stream));
    if (rc != SP_OK) return rc;
    return sp_gicp_target_update(t, tgt_covs, stream);  // (the rows carry each point's radius: written again, version bumped)
}
extern "C" int sp_gicp_target_has_certificates(const sp_gicp_target* t) { return t && t->rho2 != nullptr ? 1 : 0; }

extern "C" void sp_gicp_source_destroy(sp_gicp_source* s) {
    if (!s) return;
    (void)hipFree(s->pts); (void)hipFree(s->covp); (void)hipFree(s->perm); (void)hipFree(s->ccache); (void)hipFree(s->ccache2); (void)hipFree(s->qcert); (void)hipFree(s->qcert2); (void)hipFree(s->opt_rows);
    (void)hipFree(s->keys_in); (void)hipFree(s->keys_out); (void)hipFree(s->vals_in); (void)hipFree(s->sort_tmp);
    delete s;
}
extern "C" int sp_gicp_source_create(size_t n_max, sp_gicp_source** out) {
    using namespace sp;
    if (!out) return SP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (n_max >= (1ull << 32)) { sp_set_error("[Registration] more than 2^32 source points"); return SP_ERR_INVALID_ARGUMENT; }
    sp_gicp_source* s = new sp_gicp_source();
    s->n_max = n_max;
    const size_t n = n_max ? n_max : 1;
    s->sort_tmp_bytes = radix_sort_u32_workspace_bytes(n);  // radix_sort.hip
    hipError_t e = hipMalloc(&s->pts, (n + 64) * sizeof(float4));  // planes, each padded to a multiple of 64 floats
    if (e == hipSuccess) e = hipMalloc(&s->covp, (n + 64) * 2 * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&s->perm, n * 4);
    if (e == hipSuccess) e = hipMalloc(&s->ccache, n * 3 * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&s->ccache2, std::min<size_t>(n, 2048) * 3 * sizeof(float4));
    s->qcert_points = std::min<size_t>(n, 131072);  // (what the wave-per-point launch serves)
    if (e == hipSuccess) e = hipMalloc(&s->qcert, s->qcert_points * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&s->qcert2, std::min<size_t>(n, 2048) * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&s->opt_rows, sp::kOptRowsBytes);
    if (e == hipSuccess) e = hipMemset(s->opt_rows, 0, sp::kOptRowsBytes);
    if (e == hipSuccess) e = hipMalloc(&s->keys_in, n * 4);
    if (e == hipSuccess) e = hipMalloc(&s->keys_out, n * 4);
    if (e == hipSuccess) e = hipMalloc(&s->vals_in, n * 4);
    if (e == hipSuccess) e = hipMalloc(&s->sort_tmp, s->sort_tmp_bytes ? s->sort_tmp_bytes : 16);
    if (e != hipSuccess) {
        sp_set_error(hipGetErrorString(e));
        sp_gicp_source_destroy(s);
        return SP_ERR_HIP;
    }
    *out = s;
    return SP_OK;
}
extern "C" int sp_gicp_source_prepare(sp_gicp_source* s, const sp_gicp_target* target, const float* src_points,
                                      const float* src_covs, size_t n, const float* transT, int transT_on_device,
                                      int sort_by_cell, void* stream) {
    using namespace sp;
    if (!s || !target) return SP_ERR_INVALID_ARGUMENT;
    if (n > s->n_max) {
        sp_set_error("[sp_gicp_source_prepare] more points than the object was created for");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (!src_covs && n && target->reg_type == SP_REG_GICP) {  // point-to-distribution has no source covariance
        sp_set_error("[Registration::validate_params] Covariance matrices of source and target must be pre-computed "
                     "before performing GICP matching.");
        return SP_ERR_RUNTIME;
    }
    s->n = n;
    s->sorted = sort_by_cell != 0;
    if (n == 0) return SP_OK;
    hipStream_t st = as_stream(stream);
    const float4* pts = reinterpret_cast<const float4*>(src_points);
    const unsigned nb = div_up(n, kBlock);
    if (sort_by_cell == SP_SOURCE_SORT) {
        Mat4Arg tv;
        for (int i = 0; i < 16; ++i) tv.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        if (transT && !transT_on_device)
            for (int i = 0; i < 16; ++i) tv.m[i] = transT[i];
        const GridDesc g = grid_desc(target->grid);
        query_cell_kernel<<<nb, kBlock, 0, st>>>(pts, (unsigned)n, g, tv, transT_on_device ? transT : nullptr, s->keys_in,
                                                 s->vals_in);
        unsigned end_bit = 1;
        while ((1ull << end_bit) <= target->grid->ncells && end_bit < 32) ++end_bit;
        // Only locality matters, not a total order: the top 16 key bits put every run of 32 consecutive cells of an
        // x-row (one 128-byte line of cell extents, ~16 points) together — two radix passes instead of three.
        // The sort is stable, so the order is deterministic.
        const unsigned begin_bit = end_bit > 16 ? end_bit - 16 : 0;
        bool in_b = false;
        if (radix_sort_pairs_u32(s->keys_in, s->keys_out, s->vals_in, s->perm, n, end_bit, s->sort_tmp, s->sort_tmp_bytes, &in_b,
                                 st, begin_bit) != SP_OK) {
            sp_set_error("[Registration] radix sort of the source failed");
            return SP_ERR_HIP;
        }
        if (!in_b) { unsigned* t = s->perm; s->perm = s->vals_in; s->vals_in = t; }  // the sorted permutation is where the last pass wrote
    }
    s->cache_valid = false;  // no previous correspondences
    s->qcert_valid = false;
    s->cache_target = target;
    s->cache_version = target->version;
    prepare_source_kernel<<<nb, kBlock, 0, st>>>(pts, target->reg_type == SP_REG_GICP ? reinterpret_cast<const float4*>(src_covs) : nullptr,
                                                 s->perm, sort_by_cell != SP_SOURCE_SORT, (unsigned)n,
                                                 (unsigned)((n + 63) / 64 * 64), reinterpret_cast<float*>(s->pts),
                                                 reinterpret_cast<float*>(s->covp));
    return launch_status();
}



extern "C" int sp_gicp_iteration_fused(const sp_gicp_target* target, const sp_gicp_source* source, float* transT,
                                       int transT_on_device, const sp_factor_params* params, const sp_gn_params* gn,
                                       int32_t* nn_idx_out, float* nn_d2_out, sp_linearized* out, float* delta_out8,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sp;
    hipStream_t st = as_stream(stream);
    if (!target || !source || !params || !out) return SP_ERR_INVALID_ARGUMENT;
    if (const int rc = check_prepared_reg("iteration_fused", target, params); rc != SP_OK) return rc;
    if (gn && !transT_on_device) {
        sp_set_error("[sp_gicp_iteration_fused] the fused Gauss-Newton update needs the pose on the device");
        return SP_ERR_INVALID_ARGUMENT;
    }
    const size_t n = source->n;
    if (n == 0) return zero_async(out, sizeof(sp_linearized), st);
    if (!workspace || workspace_bytes < sp_gicp_workspace_bytes(n)) {
        sp_set_error("[Registration] workspace too small (sp_gicp_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    target->note(st);
    const FusedParams P = make_fused_params(target, source, params, transT, transT_on_device, nn_idx_out, nn_d2_out);
    const bool fills_cache = P.ccache != nullptr && (source->opt_stage_mask & 1);
    const unsigned grid = reduce_grid(n);
    float* partials = static_cast<float*>(workspace);
    // Unsorted lanes touch unrelated cells: the ring walk (fewest cache lines per query) wins. Cell-sorted lanes share
    // their lines: the branch-light 2x2x2 walk wins (profiles/README.md, r01_c).
    const bool fast = source->opt_fast_nn < 0 ? source->sorted : (source->opt_fast_nn != 0);
    const bool p2d = params->reg_type == SP_REG_POINT_TO_DISTRIBUTION;
#define SP_LAUNCH_FUSED(L)                                                              \
    if (!(source->opt_stage_mask & 1)) {}                                                   \
    else if (fast && p2d) gicp_fused_kernel<L, true, true><<<grid, kBlock, 0, st>>>(P, partials);   \
    else if (fast) gicp_fused_kernel<L, true><<<grid, kBlock, 0, st>>>(P, partials);   \
    else if (p2d) gicp_fused_kernel<L, false, true><<<grid, kBlock, 0, st>>>(P, partials);   \
    else gicp_fused_kernel<L, false><<<grid, kBlock, 0, st>>>(P, partials)
    switch (params->robust_type) {
        case SP_LOSS_NONE: SP_LAUNCH_FUSED(LOSS_NONE); break;
        case SP_LOSS_HUBER: SP_LAUNCH_FUSED(LOSS_HUBER); break;
        case SP_LOSS_TUKEY: SP_LAUNCH_FUSED(LOSS_TUKEY); break;
        case SP_LOSS_CAUCHY: SP_LAUNCH_FUSED(LOSS_CAUCHY); break;
        case SP_LOSS_GEMAN_MCCLURE: SP_LAUNCH_FUSED(LOSS_GEMAN_MCCLURE); break;
        default: sp_set_error("[Registration::dispatch] Combination not found in tags!"); return SP_ERR_RUNTIME;
    }
#undef SP_LAUNCH_FUSED
    if (fills_cache) { source->cache_valid = true; source->qcert_valid = false; }  // (these kernels refresh rows, not margin certificates)
    GnArgs ga{nullptr, 0.0f, 0.0f, 0.0f, nullptr};
    if (gn) ga = GnArgs{transT, gn->lambda, gn->crit_rotation, gn->crit_translation, delta_out8};
    if (source->opt_stage_mask & 2) final_reduce_kernel<<<1, kFinalThreads, 0, st>>>(partials, grid, kAcc - 1, out, ga);
    return launch_status();
}

namespace sp {
namespace {
struct AlignWs {  // workspace: partial rows A | partial rows B | state A | state B | searched log | fan-in rows | tickets
    float* part[2];
    AlignState* state;       // [j & 1]: the state after iteration j
    unsigned* searched_log;  // kSearchedLog entries
    float* fan_row[2];       // fan-in rows (kFanRow floats each), ping-pong like the partial rows
    unsigned* fan_counter;   // arrival tickets of the last-arriver logic
};
AlignWs align_ws(void* workspace) {
    AlignWs w;
    w.part[0] = static_cast<float*>(workspace);
    w.part[1] = w.part[0] + (size_t)kAlignMaxBlocks * kPartial;
    w.state = reinterpret_cast<AlignState*>(w.part[1] + (size_t)kAlignMaxBlocks * kPartial);
    w.searched_log = reinterpret_cast<unsigned*>(w.state + 2);
    w.fan_row[0] = reinterpret_cast<float*>(w.searched_log + kSearchedLog);
    w.fan_row[1] = w.fan_row[0] + kFanRow;
    w.fan_counter = reinterpret_cast<unsigned*>(w.fan_row[1] + kFanRow);
    return w;
}
constexpr size_t kAlignResetBytes = (kSearchedLog + 2 * kFanRow + 4) * sizeof(float);  // log | rows | tickets: zero at k = 0
int align_check(const char* who, const sp_gicp_target* target, const sp_gicp_source* source, const sp_factor_params* params,
                const sp_gn_params* gn, const float* T, void* workspace, size_t workspace_bytes) {
    if (!target || !source || !params || !gn || !T) return SP_ERR_INVALID_ARGUMENT;
    if (const int rc = check_prepared_reg(who, target, params); rc != SP_OK) return rc;
    if (!workspace || workspace_bytes < sp_gicp_workspace_bytes(source->n)) {
        sp_set_error("[Registration] workspace too small (sp_gicp_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    (void)who;
    return SP_OK;
}
AlignArgs align_args(const AlignWs& w, float* transT_device, const sp_gn_params* gn, int j, int mode, sp_linearized* lin_out) {
    AlignArgs A;
    A.T_init = transT_device;
    A.state_in = &w.state[(j + 1) & 1];
    A.state_out = &w.state[j & 1];
    A.has_prev = j > 0;
    A.first = 0;
    A.prev_rows = nullptr;
    A.lambda = gn->lambda;
    A.crit_rot = gn->crit_rotation;
    A.crit_trans = gn->crit_translation;
    A.lin_out = lin_out;
    A.mode = mode;
    A.searched_log = w.searched_log;
    A.k = j;
    A.k_launch = j;
    A.fan_row_out = w.fan_row[j & 1];
    A.fan_row_in = w.fan_row[j & 1];
    A.fan_counter = w.fan_counter;
    A.x = XchgArgs{nullptr, nullptr, 0, 1, nullptr, 0ull};
    return A;
}
// Sharded modes: finish iteration j from the all-reduced row(s) (enqueued behind the caller's collective).
XchgArgs xchg_args(const sp_xchg* x, int j) {
    if (!x) return XchgArgs{nullptr, nullptr, 0, 1, nullptr, 0ull};
    (void)j;
    return XchgArgs{x->peers_dev, x->local, x->rank, x->world, x->epoch_dev,
                    (unsigned long long)x->timeout_ms * 100000ull};  // wall_clock64: 100 MHz
}
// (ALIGN_PROLOGUE: the rows of the last launch, `rows` = its grid; ALIGN_ROWS: all kAlignMaxBlocks all-reduced rows)
void launch_solve(const AlignWs& w, float* transT_device, const sp_gn_params* gn, int j, int mode, sp_linearized* lin_out,
                  hipStream_t st, const sp_xchg* x = nullptr, unsigned rows = kAlignMaxBlocks,
                  AlignPublish Pb = AlignPublish{nullptr, nullptr, nullptr, nullptr}) {
    AlignArgs A = align_args(w, transT_device, gn, j, mode, lin_out);
    A.x = xchg_args(x, j);
    align_solve_kernel<<<1, kFinalThreads, 0, st>>>(A, w.part[j & 1], rows, Pb);
}
}  // namespace
}  // namespace sp

extern "C" int sp_gicp_error_prepared(const sp_gicp_target* target, const sp_gicp_source* source, const float* transT_lin_host,
                                      const float* transT_trial, int trial_on_device, const sp_factor_params* params,
                                      sp_linearized* out, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sp;
    hipStream_t st = as_stream(stream);
    if (!target || !source || !params || !out || !transT_lin_host) return SP_ERR_INVALID_ARGUMENT;
    if (const int rc = check_prepared_reg("error_prepared", target, params); rc != SP_OK) return rc;
    const size_t n = source->n;
    if (n == 0) return zero_async(out, sizeof(sp_linearized), st);
    if (!workspace || workspace_bytes < sp_gicp_workspace_bytes(n)) {
        sp_set_error("[Registration] workspace too small (sp_gicp_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (!source->cache_valid || source->cache_target != target || source->cache_version != target->version) {
        sp_set_error("[sp_gicp_error_prepared] no frozen correspondences: linearise first (sp_gicp_iteration_fused / "
                     "sp_gicp_align_*) with this target");  // compute_error_frozen needs the neighbours of a linearisation
        return SP_ERR_RUNTIME;
    }
    target->note(st);
    const FusedParams P = make_fused_params(target, source, params, transT_trial, trial_on_device, nullptr, nullptr);
    Mat4Arg TL;
    for (int i = 0; i < 16; ++i) TL.m[i] = transT_lin_host[i];
    const unsigned grid = reduce_grid(n);
    float* partials = static_cast<float*>(workspace);
    const bool p2d = params->reg_type == SP_REG_POINT_TO_DISTRIBUTION;
#define SP_LAUNCH_ERR(L)                                                                       \
    if (p2d) error_prepared_kernel<L, true><<<grid, kBlock, 0, st>>>(P, TL, nullptr, partials);        \
    else error_prepared_kernel<L, false><<<grid, kBlock, 0, st>>>(P, TL, nullptr, partials)
    switch (params->robust_type) {
        case SP_LOSS_NONE: SP_LAUNCH_ERR(LOSS_NONE); break;
        case SP_LOSS_HUBER: SP_LAUNCH_ERR(LOSS_HUBER); break;
        case SP_LOSS_TUKEY: SP_LAUNCH_ERR(LOSS_TUKEY); break;
        case SP_LOSS_CAUCHY: SP_LAUNCH_ERR(LOSS_CAUCHY); break;
        case SP_LOSS_GEMAN_MCCLURE: SP_LAUNCH_ERR(LOSS_GEMAN_MCCLURE); break;
        default: sp_set_error("[Registration::dispatch] Combination not found in tags!"); return SP_ERR_RUNTIME;
    }
#undef SP_LAUNCH_ERR
    final_reduce_kernel<<<1, kFinalThreads, 0, st>>>(partials, grid, 1, out, GnArgs{nullptr, 0.0f, 0.0f, 0.0f, nullptr});
    return launch_status();
}

namespace sp {
namespace {
int align_step_impl(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                    const sp_factor_params* params, const sp_gn_params* gn, int k, int rows_all_reduced, int32_t* nn_idx_out,
                    float* nn_d2_out, sp_linearized* lin_out, void* workspace, size_t workspace_bytes, void* stream,
                    const sp_xchg* xchg) {
    hipStream_t st = as_stream(stream);
    const int rc = align_check("step", target, source, params, gn, transT_device, workspace, workspace_bytes);
    if (rc != SP_OK) return rc;
    if (k < 0 || rows_all_reduced < 0 || rows_all_reduced > 3 || ((rows_all_reduced == ALIGN_DIRECT) != (xchg != nullptr)))
        return SP_ERR_INVALID_ARGUMENT;
    const size_t n = source->n;
    const AlignWs w = align_ws(workspace);
    const int mode = rows_all_reduced;
    if (k == 0) {
        // searched-point log, fan-in rows, arrival tickets (the caller's workspace comes as it is); ALIGN_ROWS: every rank
        // all-reduces all kAlignMaxBlocks rows whatever its own tile size, rows a rank does not write stay zero
        if (mode != ALIGN_PROLOGUE && zero_async(w.searched_log, kAlignResetBytes, st) != SP_OK) return SP_ERR_HIP;
        if (mode == ALIGN_ROWS &&
            zero_async(w.part[0], 2 * (size_t)kAlignMaxBlocks * kPartial * sizeof(float), st) != SP_OK)
            return SP_ERR_HIP;
    }
    target->note(st);
    const FusedParams P = make_fused_params(target, source, params, transT_device, 1, nn_idx_out, nn_d2_out);
    const bool fills_cache = P.ccache != nullptr && (source->opt_stage_mask & 1);
    const unsigned grid = align_grid(n);
    const bool fast = source->opt_fast_nn < 0 ? source->sorted : (source->opt_fast_nn != 0);
    // the launch first finishes iteration k - 1: from the previous launch's rows (one GPU), from the all-reduced row(s), or
    // from the rows the peers are storing into this rank's slots
    AlignArgs A = align_args(w, transT_device, gn, k > 0 ? k - 1 : 0, mode, lin_out);
    A.first = k == 0;
    A.prev_rows = w.part[(k + 1) & 1];
    A.k_launch = k;
    A.fan_row_out = w.fan_row[k & 1];
    A.x = xchg_args(xchg, k);
    float* out = w.part[k & 1];
    const bool p2d = params->reg_type == SP_REG_POINT_TO_DISTRIBUTION;
#define SP_LAUNCH_ALIGN2(L, S)                                                                          \
    if (fast && p2d) gicp_align_kernel<L, true, true, S><<<grid, kAlignBlock, 0, st>>>(P, A, out);         \
    else if (fast) gicp_align_kernel<L, true, false, S><<<grid, kAlignBlock, 0, st>>>(P, A, out);          \
    else if (p2d) gicp_align_kernel<L, false, true, S><<<grid, kAlignBlock, 0, st>>>(P, A, out);           \
    else gicp_align_kernel<L, false, false, S><<<grid, kAlignBlock, 0, st>>>(P, A, out)
#define SP_LAUNCH_ALIGN(L)                                                                             \
    if (!(source->opt_stage_mask & 1)) {}                                                                  \
    else if (mode == ALIGN_PROLOGUE) { SP_LAUNCH_ALIGN2(L, false); }                                      \
    else { SP_LAUNCH_ALIGN2(L, true); }
#ifdef SP_DEV_MIN  // development builds only (scratch/devbuild.sh): one instantiation of the kernel, a fraction of the compile time
    if (params->robust_type != SP_LOSS_NONE || !fast || p2d || mode != ALIGN_PROLOGUE) {
        sp_set_error("SP_DEV_MIN build: only GICP / NONE / sorted source / one GPU");
        return SP_ERR_RUNTIME;
    }
    if (source->opt_stage_mask & 1) gicp_align_kernel<LOSS_NONE, true, false, false><<<grid, kAlignBlock, 0, st>>>(P, A, out);
#else
    switch (params->robust_type) {
        case SP_LOSS_NONE: SP_LAUNCH_ALIGN(LOSS_NONE); break;
        case SP_LOSS_HUBER: SP_LAUNCH_ALIGN(LOSS_HUBER); break;
        case SP_LOSS_TUKEY: SP_LAUNCH_ALIGN(LOSS_TUKEY); break;
        case SP_LOSS_CAUCHY: SP_LAUNCH_ALIGN(LOSS_CAUCHY); break;
        case SP_LOSS_GEMAN_MCCLURE: SP_LAUNCH_ALIGN(LOSS_GEMAN_MCCLURE); break;
        default: sp_set_error("[Registration::dispatch] Combination not found in tags!"); return SP_ERR_RUNTIME;
    }
#endif
#undef SP_LAUNCH_ALIGN
#undef SP_LAUNCH_ALIGN2
    if (fills_cache) { source->cache_valid = true; source->qcert_valid = false; }  // (these kernels refresh rows, not margin certificates)
    return launch_status();
}
}  // namespace
}  // namespace sp

extern "C" int sp_gicp_align_step(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                                 const sp_factor_params* params, const sp_gn_params* gn, int k, int rows_all_reduced,
                                 int32_t* nn_idx_out, float* nn_d2_out, sp_linearized* lin_out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (rows_all_reduced == sp::ALIGN_DIRECT) return SP_ERR_INVALID_ARGUMENT;  // (sp_gicp_align_direct)
    return sp::align_step_impl(target, source, transT_device, params, gn, k, rows_all_reduced, nn_idx_out, nn_d2_out, lin_out,
                               workspace, workspace_bytes, stream, nullptr);
}

extern "C" float* sp_gicp_align_rows(void* workspace, int k, size_t* n_floats_out) {
    if (n_floats_out) *n_floats_out = (size_t)sp::kAlignMaxBlocks * sp::kPartial;
    if (!workspace || k < 0) return nullptr;
    return sp::align_ws(workspace).part[k & 1];
}
extern "C" float* sp_gicp_align_row(void* workspace, int k, size_t* n_floats_out) {
    if (n_floats_out) *n_floats_out = (size_t)sp::kFanRow;
    if (!workspace || k < 0) return nullptr;
    return sp::align_ws(workspace).fan_row[k & 1];
}

namespace sp {
namespace {
int align_finish_impl(const sp_gicp_source* source, float* transT_device, const sp_gn_params* gn, int last_k,
                      int rows_all_reduced, sp_linearized* lin_out, float* delta_out8, uint32_t* iterations_out,
                      void* workspace, size_t workspace_bytes, void* stream, const sp_xchg* xchg) {
    hipStream_t st = as_stream(stream);
    if (!source || !transT_device || !gn || last_k < 0 || rows_all_reduced < 0 || rows_all_reduced > 3)
        return SP_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < sp_gicp_workspace_bytes(source->n)) {
        sp_set_error("[Registration] workspace too small (sp_gicp_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    const AlignWs w = align_ws(workspace);
    if (source->opt_stage_mask & 2) {
        launch_solve(w, transT_device, gn, last_k, rows_all_reduced, lin_out, st, xchg,
                     rows_all_reduced == ALIGN_PROLOGUE ? align_grid(source->n) : (unsigned)kAlignMaxBlocks,
                     AlignPublish{transT_device, delta_out8, iterations_out, xchg ? xchg->epoch_dev : nullptr});
    }
    return launch_status();
}
}  // namespace
}  // namespace sp

extern "C" int sp_gicp_align_finish(const sp_gicp_source* source, float* transT_device, const sp_gn_params* gn,
                                   int last_k, int rows_all_reduced, sp_linearized* lin_out, float* delta_out8,
                                   uint32_t* iterations_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (rows_all_reduced == sp::ALIGN_DIRECT) return SP_ERR_INVALID_ARGUMENT;
    return sp::align_finish_impl(source, transT_device, gn, last_k, rows_all_reduced, lin_out, delta_out8, iterations_out,
                                 workspace, workspace_bytes, stream, nullptr);
}

// The sharded loop with the rows exchanged DIRECTLY between the ranks' buffers (sp_xchg.h): per iteration ONE launch — its
// last-arriving workgroup stores the rank's row into every rank's slot buffer, the next launch's prologue waits for all rows —
// no collective, no host involvement; the latency-bound 128-byte all-reduce of sp_gicp_align_sharded (a library launch of its
// own between two kernels) is gone.
extern "C" int sp_gicp_align_direct(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                                   const sp_factor_params* params, const sp_gn_params* gn, int max_iterations, sp_xchg* xchg,
                                   int32_t* nn_idx_out, float* nn_d2_out, sp_linearized* lin_out, float* delta_out8,
                                   uint32_t* iterations_out, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sp;
    if (!xchg || !xchg->connected) {
        sp_set_error("[sp_gicp_align_direct] the exchange is not connected (sp_xchg_connect)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (max_iterations <= 0 || max_iterations > 255) {
        sp_set_error("[sp_gicp_align_direct] max_iterations must be in 1..255");
        return SP_ERR_INVALID_ARGUMENT;
    }
    // (the tags count alignments in a device word that the last launch bumps: every rank must enqueue the same alignments)
    // (a rank whose shard is empty still takes part: its launch writes a zero row)
    for (int k = 0; k < max_iterations; ++k) {
        const int rc = align_step_impl(target, source, transT_device, params, gn, k, ALIGN_DIRECT, nn_idx_out, nn_d2_out, lin_out,
                                       workspace, workspace_bytes, stream, xchg);
        if (rc != SP_OK) {
            // launches of this alignment may be in the queue already: still enqueue the finish launch, which bumps the epoch
            // word — the peers bump theirs, and a rank that fell behind would mismatch every tag from then on
            if (k > 0)
                (void)align_finish_impl(source, transT_device, gn, k - 1, ALIGN_DIRECT, lin_out, delta_out8, iterations_out, workspace,
                                        workspace_bytes, stream, xchg);
            return rc;
        }
    }
    return align_finish_impl(source, transT_device, gn, max_iterations - 1, ALIGN_DIRECT, lin_out, delta_out8, iterations_out,
                             workspace, workspace_bytes, stream, xchg);
}

extern "C" int sp_gicp_align_status(const void* workspace, int last_k, void* stream) {
    if (!workspace || last_k < 0) return SP_ERR_INVALID_ARGUMENT;
    const sp::AlignWs w = sp::align_ws(const_cast<void*>(workspace));
    unsigned err = 0;
    hipStream_t st = sp::as_stream(stream);
    if (hipMemcpyAsync(&err, &w.state[last_k & 1].pad, sizeof err, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return SP_ERR_HIP;
    if (err == 2u) {  // gicp_align_persistent_kernel: a workgroup waited longer than its bound for the others of its own launch
        sp_set_error("[sp_gicp_align_fused] the device-side tail of the alignment ran into its time limit (not every workgroup of the "
                     "launch was resident: another process or a CU-masked stream holds compute units); the pose is NaN. Run the "
                     "alignment again with sp_gicp_source_set_persistent(source, 0)");
        return SP_ERR_RUNTIME;
    }
    if (err) {
        sp_set_error("[sp_gicp_align_direct] the row of another rank did not arrive within the time limit: the alignment was stopped");
        return SP_ERR_RUNTIME;
    }
    return SP_OK;
}

extern "C" int sp_gicp_align_linearization_pose(const void* workspace, int last_k, float* transT_lin_out, void* stream) {
    if (!workspace || last_k < 0 || !transT_lin_out) return SP_ERR_INVALID_ARGUMENT;
    const sp::AlignWs w = sp::align_ws(const_cast<void*>(workspace));
    return sp::hip_status(hipMemcpyAsync(transT_lin_out, w.state[last_k & 1].T_lin, 16 * sizeof(float), hipMemcpyDefault,
                                         sp::as_stream(stream)));
}

namespace sp {
namespace {
// ---- the one-launch form of sp_gicp_align_fused (gicp_align_persistent_kernel)
static_assert(kTicketOffsetBytes + kTicketShards * kTicketStride * 4 <= (size_t)kMaxBlocks * kPartial * sizeof(float), "workspace");
// Whether sp_gicp_align_fused may run the tail of this alignment as one launch; returns the device's guard, locked (the launch
// follows inside the same critical section: launch_persistent releases it), or nullptr.
PersistGuard* persistent_ok(const sp_gicp_target* target, const sp_gicp_source* source, const sp_factor_params* params,
                            const sp_gn_params* gn, const float* T, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (source->opt_persistent == 0 || source->opt_stage_mask != 3) return nullptr;
    if (align_check("fused", target, source, params, gn, T, workspace, workspace_bytes) != SP_OK) return nullptr;  // (reported by the step)
    return persist_acquire(st, align_grid(source->n));
}
int launch_persistent(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                      const sp_factor_params* params, const sp_gn_params* gn, int k_begin, int max_iterations, int32_t* nn_idx_out,
                      float* nn_d2_out, sp_linearized* lin_out, float* delta_out8, uint32_t* iterations_out, void* workspace,
                      hipStream_t st, PersistGuard* guard) {
    struct Release {  // every return below leaves the critical section persistent_ok() entered
        PersistGuard* g; hipStream_t st;
        ~Release() { persist_release(g, st); }
    } release{guard, st};
    const AlignWs w = align_ws(workspace);
    unsigned* const tickets = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + kTicketOffsetBytes);
    if (zero_async(tickets, kTicketShards * kTicketStride * sizeof(unsigned), st) != SP_OK) return SP_ERR_HIP;
    target->note(st);
    const FusedParams P = make_fused_params(target, source, params, transT_device, 1, nn_idx_out, nn_d2_out);
    PersistArgs A;
    A.part[0] = w.part[0]; A.part[1] = w.part[1];
    A.state = w.state;
    A.searched_log = w.searched_log;
    A.tickets = tickets;
    A.T_init = transT_device;
    A.lambda = gn->lambda; A.crit_rot = gn->crit_rotation; A.crit_trans = gn->crit_translation;
    A.lin_out = lin_out;
    A.T_out = transT_device;
    A.delta_out8 = delta_out8;
    A.iterations_out = iterations_out;
    A.max_iterations = max_iterations;
    A.k_begin = k_begin;
    A.cache_valid_later = (source->opt_reuse && P.ccache != nullptr) ? 1 : 0;
    A.budget = 50ull * 100000ull;  // 50 ms of wall_clock64 (100 MHz)
    const unsigned grid = align_grid(source->n);
    const bool fast = source->opt_fast_nn < 0 ? source->sorted : (source->opt_fast_nn != 0);
    const bool p2d = params->reg_type == SP_REG_POINT_TO_DISTRIBUTION;
#define SP_LAUNCH_PERSIST(L)                                                                              \
    if (fast && p2d) gicp_align_persistent_kernel<L, true, true><<<grid, kAlignBlock, 0, st>>>(P, A);      \
    else if (fast) gicp_align_persistent_kernel<L, true, false><<<grid, kAlignBlock, 0, st>>>(P, A);       \
    else if (p2d) gicp_align_persistent_kernel<L, false, true><<<grid, kAlignBlock, 0, st>>>(P, A);        \
    else gicp_align_persistent_kernel<L, false, false><<<grid, kAlignBlock, 0, st>>>(P, A)
#ifdef SP_DEV_MIN
    if (params->robust_type != SP_LOSS_NONE || !fast || p2d) { sp_set_error("SP_DEV_MIN build"); return SP_ERR_RUNTIME; }
    gicp_align_persistent_kernel<LOSS_NONE, true, false><<<grid, kAlignBlock, 0, st>>>(P, A);
#else
    switch (params->robust_type) {
        case SP_LOSS_NONE: SP_LAUNCH_PERSIST(LOSS_NONE); break;
        case SP_LOSS_HUBER: SP_LAUNCH_PERSIST(LOSS_HUBER); break;
        case SP_LOSS_TUKEY: SP_LAUNCH_PERSIST(LOSS_TUKEY); break;
        case SP_LOSS_CAUCHY: SP_LAUNCH_PERSIST(LOSS_CAUCHY); break;
        case SP_LOSS_GEMAN_MCCLURE: SP_LAUNCH_PERSIST(LOSS_GEMAN_MCCLURE); break;
        default: sp_set_error("[Registration::dispatch] Combination not found in tags!"); return SP_ERR_RUNTIME;
    }
#endif
#undef SP_LAUNCH_PERSIST
    if (P.ccache != nullptr) { source->cache_valid = true; source->qcert_valid = false; }
    return launch_status();
}
}  // namespace
}  // namespace sp

namespace sp {
namespace {
constexpr int kMaxGuardDevices = 64;
PersistGuard g_persist[kMaxGuardDevices];
}  // namespace
PersistGuard* persist_acquire(hipStream_t st, unsigned grid) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxGuardDevices) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;  // (no event query in a capture)
    PersistGuard& g = g_persist[dev];
    g.m.lock();
    if (g.cus < 0 && hipDeviceGetAttribute(&g.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) g.cus = 0;
    bool ok = (int)grid <= g.cus;  // every workgroup needs a CU of its own
    if (ok && g.used && g.last != st && hipEventQuery(g.done) != hipSuccess) {
        (void)hipGetLastError();
        ok = false;  // another stream's persistent launch may still be running: the caller takes its per-launch form
    }
    if (!ok) { g.m.unlock(); return nullptr; }
    return &g;
}
void persist_release(PersistGuard* g, hipStream_t st) {
    if (!g) return;
    if (!g->done && hipEventCreateWithFlags(&g->done, hipEventDisableTiming) != hipSuccess) g->done = nullptr;
    if (g->done && hipEventRecord(g->done, st) == hipSuccess) { g->last = st; g->used = true; }
    g->m.unlock();
}
}  // namespace sp

extern "C" int sp_gicp_align_fused(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                                  const sp_factor_params* params, const sp_gn_params* gn, int max_iterations,
                                  int32_t* nn_idx_out, float* nn_d2_out, sp_linearized* lin_out, float* delta_out8,
                                  uint32_t* iterations_out, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sp;
    hipStream_t st = as_stream(stream);
    if (!target || !source || !params || !gn || !transT_device) return SP_ERR_INVALID_ARGUMENT;
    if (source->n == 0 || max_iterations <= 0) {
        if (zero_async(lin_out, sizeof(sp_linearized), st) != SP_OK) return SP_ERR_HIP;
        if (zero_async(delta_out8, 8 * sizeof(float), st) != SP_OK) return SP_ERR_HIP;
        if (zero_async(iterations_out, sizeof(uint32_t), st) != SP_OK) return SP_ERR_HIP;
        return SP_OK;
    }
    // The form. A launch per iteration is the fastest way through an iteration (a grid-wide wait inside one launch costs more
    // than a kernel boundary on this part: 40 against 33.5 us per iteration, same box, profiles/r04_c_*), but an alignment with
    // convergence criteria rarely runs all its iterations, and every launch enqueued after the converged one still costs
    // 2.5 us to find that out (17 of 20 in the benchmark's cloud). So: the first iterations one launch each, the REST as one
    // launch that loops on the device (gicp_align_persistent_kernel) — it returns at once when the alignment has converged,
    // and also writes the results, so there is no finish launch. With criteria that cannot be met (<= 0: a fixed number of
    // iterations) every iteration gets its own launch.
    const int k_tail = (gn->crit_rotation > 0.0f && gn->crit_translation > 0.0f) ? std::min(max_iterations, source->opt_persistent_from)
                                                                                : max_iterations;
    PersistGuard* const guard = k_tail < max_iterations
                                    ? persistent_ok(target, source, params, gn, transT_device, workspace, workspace_bytes, st)
                                    : nullptr;
    if (guard) {
        for (int k = 0; k < k_tail; ++k) {
            const int rc = sp_gicp_align_step(target, source, transT_device, params, gn, k, 0, nn_idx_out, nn_d2_out, lin_out,
                                              workspace, workspace_bytes, stream);
            if (rc != SP_OK) { persist_release(guard, st); return rc; }
        }
        return launch_persistent(target, source, transT_device, params, gn, k_tail, max_iterations, nn_idx_out, nn_d2_out, lin_out,
                                 delta_out8, iterations_out, workspace, st, guard);
    }
    for (int k = 0; k < max_iterations; ++k) {
        const int rc = sp_gicp_align_step(target, source, transT_device, params, gn, k, 0, nn_idx_out, nn_d2_out, lin_out,
                                          workspace, workspace_bytes, stream);
        if (rc != SP_OK) return rc;
    }
    return sp_gicp_align_finish(source, transT_device, gn, max_iterations - 1, 0, lin_out, delta_out8, iterations_out,
                                workspace, workspace_bytes, stream);
}

// 0: sp_gicp_align_fused / sp_gicp_align_optimize always use one launch per step (no launch whose workgroups wait for each other).
extern "C" int sp_gicp_source_set_persistent(sp_gicp_source* s, int enable) {
    if (!s) return SP_ERR_INVALID_ARGUMENT;
    s->opt_persistent = enable ? 1 : 0;
    return SP_OK;
}

// sp_gicp_align_optimize's linearisation steps: 0 a lane per source point, 1 (default) a wave per point for sources of up to 2048
// points, 2 a wave per point up to 131072 points (for targets with crowded cells: see registration_opt.hip).
extern "C" int sp_gicp_source_set_wave_per_point(sp_gicp_source* s, int mode) {
    if (!s || mode < 0 || mode > 2) return SP_ERR_INVALID_ARGUMENT;
    s->opt_wave_query = mode;
    return SP_OK;
}

// Measurement (sp_internal.h): entry k = source points iteration k of the last alignment searched for (the rest reused their
// correspondence); 0 for iterations that did not run (converged earlier).
extern "C" const uint32_t* sp_internal_align_searched_log(void* workspace, size_t* n_entries_out) {
    if (n_entries_out) *n_entries_out = sp::kSearchedLog;
    return workspace ? sp::align_ws(workspace).searched_log : nullptr;
}
// Per-handle measurement / tuning switches (sp_internal.h): exported for tests/, bench.py and scratch/ only.
extern "C" int sp_internal_source_option(sp_gicp_source* s, int option, int value) {
    if (!s) return SP_ERR_INVALID_ARGUMENT;
    switch (option) {
        case SP_INTERNAL_FUSED_STAGE_MASK: s->opt_stage_mask = value; return SP_OK;
        case SP_INTERNAL_FUSED_REUSE: s->opt_reuse = value; s->cache_valid = false; return SP_OK;
        case SP_INTERNAL_FUSED_FAST_NN: s->opt_fast_nn = value; return SP_OK;
        case SP_INTERNAL_FUSED_PERSISTENT: s->opt_persistent = value; return SP_OK;
        case SP_INTERNAL_FUSED_PERSISTENT_FROM: s->opt_persistent_from = value < 0 ? 0 : value; return SP_OK;
        case SP_INTERNAL_OPT_WAVE_QUERY: s->opt_wave_query = value; return SP_OK;
        case SP_INTERNAL_OPT_FUSE_TRIALS: s->opt_fuse_trials = value; return SP_OK;
    }
    return SP_ERR_INVALID_ARGUMENT;
}

// Device code shared by the translation units of the registration path (registration.hip: K11 / K12 / K13, the prepared
// iteration, the per-iteration and the persistent Gauss-Newton launches; registration_opt.hip: the device-resident optimiser
// loop for Levenberg-Marquardt / Powell dog-leg / annealed robust scales). Everything here is inline / template code in an
// anonymous namespace, or a plain struct: each translation unit gets its own copy, nothing is exported.
#ifndef SP_REGISTRATION_DEVICE_H
#define SP_REGISTRATION_DEVICE_H

#include "grid_device.h"
#include "radix_sort.h"
#include "sp_xchg.h"

void sp_set_error(const char* msg);

#include "sp_internal.h"

namespace sp {
namespace {

constexpr int kAcc = 29;       // 21 H (upper, row-major) + 6 b + error + count
constexpr int kPartial = 32;   // floats per workgroup partial
constexpr int kMaxBlocks = 1024;

// Sum NV lane values over the workgroup in a fixed order and let lane e < NV of wave 0 write partial[e].
// count_as_float: the count slot holds the VALUE as a float (exact below 2^24) instead of the uint32 bit pattern, so
// that partial rows can be summed across ranks by a float all-reduce.
// `extra` is a second integer (the number of points this launch had to search for, fused kernels only); its sum goes to slot
// NV + 1 as a float VALUE (exact below 2^24), so it survives the float row sums of the next prologue / an all-reduce.
// SC1: the row is stored write-through at agent scope (global_store ... sc1) because another workgroup of the SAME launch
// will read it (the fan-in of the sharded loop, fanin_reduce below); plain stores otherwise.
template <bool SC1>
__device__ __forceinline__ void store_row_word(float* p, float v) {
    if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool SC1>
__device__ __forceinline__ float load_row_word(const float* p) {
    return SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <int NV, int BLOCK = kBlock, bool SC1 = false>
__device__ __forceinline__ void block_reduce_store(float (&acc)[NV], unsigned cnt, float* __restrict__ partial,
                                                   bool count_as_float = false, unsigned extra = 0) {
    __shared__ float red[BLOCK / kWave][kPartial];
    const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[e] = wave_sum_to_lane63(acc[e]);
    cnt = wave_sum_u32_to_lane63(cnt);
    extra = wave_sum_u32_to_lane63(extra);
    if (lane == kWave - 1) {
#pragma unroll
        for (int e = 0; e < NV; ++e) red[wave][e] = acc[e];
        red[wave][NV] = __uint_as_float(cnt);
        red[wave][NV + 1] = __uint_as_float(extra);
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < BLOCK / kWave; ++w) s += red[w][threadIdx.x];
        store_row_word<SC1>(partial + threadIdx.x, s);
    } else if (threadIdx.x == NV) {
        unsigned c = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / kWave; ++w) c += __float_as_uint(red[w][NV]);
        store_row_word<SC1>(partial + NV, count_as_float ? (float)c : __uint_as_float(c));
    } else if (threadIdx.x == NV + 1) {
        unsigned c = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / kWave; ++w) c += __float_as_uint(red[w][NV + 1]);
        store_row_word<SC1>(partial + NV + 1, (float)c);
    }
}

// Second stage: one workgroup of 1024 lanes sums the per-workgroup partials in a fixed order:
// lane = (part, slot); a part covers a contiguous range of workgroups (independent loads, summed in order),
// then lanes 0..31 add the 32 parts in order. nv = number of float slots (28 for K11, 1 for K12); slot nv holds the
// uint32 count.
constexpr int kFinalThreads = 1024;

__host__ __device__ inline void gn_update_impl(sp_linearized* lin, float* T, float lambda, float crit_rot,
                                               float crit_trans, float* delta_out8, bool fold_inlier, LdltScratch& w);

// Sum `rows` partial rows with a 1024-lane workgroup in a fixed order: lane = (part, slot); the 32 parts cover
// contiguous row ranges (independent loads, added in row order), then lanes 0..31 add the 32 parts in order.
// On return (after a barrier) red[0][e] holds the totals; slot nv is the uint32 count.
// `after_loads` runs once the row loads have been issued (and before the first barrier): the per-iteration kernel stores
// the previous state there, which it loaded BEFORE the rows, so that both arrive in the same memory round trip.
template <bool SC1 = false, typename Hook>
__device__ __forceinline__ void reduce_rows_1024(const float* __restrict__ partials, unsigned rows, int nv,
                                                 float (*red)[kPartial], bool count_is_float, Hook after_loads) {
    constexpr unsigned kParts = kFinalThreads / 32;
    const unsigned e = threadIdx.x & 31, part = threadIdx.x >> 5;
    const unsigned per = (rows + kParts - 1) / kParts;
    const unsigned lo = part * per, hi = min(rows, lo + per);
    float s = 0.0f;
    unsigned c = 0;
    const bool is_count = ((int)e == nv);
    if (per <= 8) {  // at most 256 rows (the per-iteration kernel): all loads of a lane in flight at once
        float v[8];
#pragma unroll
        for (unsigned j = 0; j < 8; ++j)
            v[j] = (lo + j < hi) ? load_row_word<SC1>(partials + (size_t)(lo + j) * kPartial + e) : 0.0f;
        after_loads();
#pragma unroll
        for (unsigned j = 0; j < 8; ++j) {
            if (lo + j < hi) {
                if (is_count) c += count_is_float ? (unsigned)v[j] : __float_as_uint(v[j]);
                else s += v[j];
            }
        }
    } else {
#pragma unroll 8
        for (unsigned b = lo; b < hi; ++b) {
            const float v = load_row_word<SC1>(partials + (size_t)b * kPartial + e);
            if (is_count) c += count_is_float ? (unsigned)v : __float_as_uint(v);
            else s += v;
        }
        after_loads();
    }
    red[part][e] = is_count ? __uint_as_float(c) : s;
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.0f;
        unsigned ct = 0;
#pragma unroll
        for (unsigned p = 0; p < kParts; ++p) {
            const float v = red[p][e];
            if (is_count) ct += __float_as_uint(v);
            else t += v;
        }
        red[0][e] = is_count ? __uint_as_float(ct) : t;
    }
    __syncthreads();
}
template <bool SC1 = false>
__device__ __forceinline__ void reduce_rows_1024(const float* __restrict__ partials, unsigned rows, int nv,
                                                 float (*red)[kPartial], bool count_is_float = false) {
    reduce_rows_1024<SC1>(partials, rows, nv, red, count_is_float, [] {});
}

// reduce_rows_1024 for a workgroup of BLOCK lanes (a multiple of 64): lane = (part, slot) with BLOCK / 32 parts, every part sums
// its contiguous range of rows in order, lanes 0..31 add the parts in order. Fixed order: bit-reproducible for a given BLOCK.
template <int BLOCK, bool SC1>
__device__ __forceinline__ void reduce_rows_block(const float* __restrict__ partials, unsigned rows, int nv, float (*red)[kPartial]) {
    constexpr unsigned kParts = BLOCK / 32;
    const unsigned e = threadIdx.x & 31, part = threadIdx.x >> 5;
    const unsigned per = (rows + kParts - 1) / kParts;
    const unsigned lo = part * per, hi = min(rows, lo + per);
    float s = 0.0f;
    unsigned c = 0;
    const bool is_count = ((int)e == nv);
#pragma unroll 4
    for (unsigned b = lo; b < hi; ++b) {
        const float v = load_row_word<SC1>(partials + (size_t)b * kPartial + e);
        if (is_count) c += __float_as_uint(v);
        else s += v;
    }
    red[part][e] = is_count ? __uint_as_float(c) : s;
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.0f;
        unsigned ct = 0;
#pragma unroll
        for (unsigned p = 0; p < kParts; ++p) {
            const float v = red[p][e];
            if (is_count) ct += __float_as_uint(v);
            else t += v;
        }
        red[0][e] = is_count ? __uint_as_float(ct) : t;
    }
    __syncthreads();
}

// totals (21 upper-triangle H, 6 b, error | error only) + count -> sp_linearized
__device__ __forceinline__ void unpack_totals(const float* tot, int nv, sp_linearized* out) {
    if (nv == kAcc - 1) {
        // (unrolled: one lane runs this between two steps of a device-resident loop, and a rolled loop is 21 dependent LDS round
        // trips — read a total, store it twice — where the unrolled form issues the 27 reads back to back)
        float v[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = tot[k];
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int cc = a; cc < 6; ++cc) {
                out->H[a * 6 + cc] = v[k];
                out->H[cc * 6 + a] = v[k];
                ++k;
            }
#pragma unroll
        for (int a = 0; a < 6; ++a) out->b[a] = v[21 + a];
        out->error = tot[27];
    } else {
        out->error = tot[0];
    }
    const unsigned cnt = __float_as_uint(tot[nv]);
    out->inlier = cnt;
    out->inlier_lo = (float)(cnt & 4095u);
    out->inlier_hi = (float)(cnt >> 12);
    out->pad[0] = out->pad[1] = 0.0f;
}

// optimize_gauss_newton (registration.hpp:803-828) + solve_linear_system (:791-801) + is_converged (:407-410)
__host__ __device__ inline void gn_update_impl(sp_linearized* lin, float* T, float lambda, float crit_rot,
                                               float crit_trans, float* delta_out8, bool fold_inlier, LdltScratch& w) {
    if (fold_inlier) lin->inlier = (uint32_t)lin->inlier_hi * 4096u + (uint32_t)lin->inlier_lo;  // integer fold: exact
    float H[36], nb[6], delta[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) H[i] = lin->H[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        H[i * 6 + i] = lin->H[i * 6 + i] + lambda * 1.0f;
        nb[i] = -lin->b[i];
    }
    const bool ok = ldlt6_solve(H, nb, delta, w);
    const float nr = sqrtf(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    const float nt = sqrtf(delta[3] * delta[3] + delta[4] * delta[4] + delta[5] * delta[5]);
    const bool conv = ok && (nr < crit_rot) && (nt < crit_trans);
    const Rigid cur = load_rigid_colmajor(T);
    const Rigid upd = rigid_mul(cur, se3_exp(delta));
    store_rigid_colmajor(upd, T);
    if (delta_out8) {
        for (int i = 0; i < 6; ++i) delta_out8[i] = delta[i];
        delta_out8[6] = conv ? 1.0f : 0.0f;
        delta_out8[7] = ok ? 1.0f : 0.0f;
    }
}

struct Sym3 {
    float xx, xy, xz, yy, yz, zz;
};

__device__ __forceinline__ Sym3 load_sym(const float4* __restrict__ p) {
    const float4 a = p[0], b = p[1];
    return Sym3{a.x, a.y, a.z, a.w, b.x, b.y};
}

struct FusedParams {
    const float* src;      // prepared source: planes x | y | z, sstride floats apart
    const float* scovp;    // prepared source covariances: planes xx | xy | xz | yy | yz | zz
    unsigned sstride;
    const float4* tpts;    // grid-ordered target points
    const unsigned* tstart;
    const float4* tcovp;   // grid-ordered prepared target covariances
    const float4* tnb;     // grid order: second certificate of the reuse test (certificate_kernel), may be null
    GridDesc g;
    unsigned n;
    float max_d2, scale;
    Mat4Arg T_val;
    const float* T_dev;
    const unsigned* perm;  // prepared-source order -> original source index (for the optional neighbour outputs)
    float4* ccache;        // per prepared source point: its previous correspondence (point, covariance row), 3 x float4
    int cache_valid;       // 0: first linearisation after sp_gicp_source_prepare, the cache holds nothing yet
    int32_t* nn_idx;
    float* nn_d2;
};

// Linearisation of one correspondence (source point s with q = T s, winner nn, packed covariances) and accumulation.
// P2D (linearize_point_to_distribution, factor.hpp:311-354): Ct is the target's information matrix M = inverse(Ct_raw)
// (prepared once per target) and there is no source covariance: N = R^T M R, nothing to invert per point.
// ERR_ONLY (calculate_gicp_error / calculate_point_to_distribution_error, factor.hpp:280-306, 356-373): only the robust
// error and the count, into acc[0] — the K12 of a trial pose (sp_gicp_error_prepared).
template <int LOSS, bool P2D = false, bool ERR_ONLY = false, int NACC = kAcc - 1>
__device__ __forceinline__ void fused_math(const FusedParams& P, const Rigid& T, const float4 s, float qx, float qy,
                                           float qz, const Nearest& nn, const Sym3& Cs, const Sym3& Ct,
                                           float (&acc)[NACC], unsigned& cnt) {
    const float r0 = nn.x - qx, r1 = nn.y - qy, r2 = nn.z - qz;
    // Source-frame form of factor.hpp:239-278. With S = skew(p), J = [R S | -R] and M = (Ct' + R Cs' R^T)^-1:
    //   N := R^T M R = (Cs' + R^T Ct' R)^-1,  v := R^T r,  u := N v,  G := S N
    //   H = [[-G S, G], [G^T, N]],   b = [u x p, -u],   e = v . u
    // (same mathematics as J^T M J / J^T M r, about half the multiply-adds).
    const float (&R)[3][3] = T.R;
    float W[3][3];  // W = Ct' R
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        W[0][j] = chain3(Ct.xx, R[0][j], Ct.xy, R[1][j], Ct.xz, R[2][j]);
        W[1][j] = chain3(Ct.xy, R[0][j], Ct.yy, R[1][j], Ct.yz, R[2][j]);
        W[2][j] = chain3(Ct.xz, R[0][j], Ct.yz, R[1][j], Ct.zz, R[2][j]);
    }
    // A = Cs' + R^T Ct' R (symmetric); P2D: R^T M R with no source term, and A is N itself
    const float g00 = chain3(R[0][0], W[0][0], R[1][0], W[1][0], R[2][0], W[2][0]);
    const float g01 = chain3(R[0][0], W[0][1], R[1][0], W[1][1], R[2][0], W[2][1]);
    const float g02 = chain3(R[0][0], W[0][2], R[1][0], W[1][2], R[2][0], W[2][2]);
    const float g11 = chain3(R[0][1], W[0][1], R[1][1], W[1][1], R[2][1], W[2][1]);
    const float g12 = chain3(R[0][1], W[0][2], R[1][1], W[1][2], R[2][1], W[2][2]);
    const float g22 = chain3(R[0][2], W[0][2], R[1][2], W[1][2], R[2][2], W[2][2]);
    float n00, n01, n02, n11, n12, n22;
    if constexpr (P2D) {
        n00 = g00; n01 = g01; n02 = g02; n11 = g11; n12 = g12; n22 = g22;
    } else {
        const float a00 = g00 + Cs.xx, a01 = g01 + Cs.xy, a02 = g02 + Cs.xz, a11 = g11 + Cs.yy, a12 = g12 + Cs.yz,
                    a22 = g22 + Cs.zz;
        // symmetric inverse by the adjugate; Zero when |det| < 1e-6 (eigen_utils::inverse, eigen_utils.hpp:403-423;
        // det is invariant under the rotation)
        const float c00 = fmaf(a11, a22, -a12 * a12);
        const float c01 = fmaf(a02, a12, -a01 * a22);
        const float c02 = fmaf(a01, a12, -a02 * a11);
        const float det = fmaf(a00, c00, fmaf(a01, c01, a02 * c02));
        const float inv_det = fabsf(det) < 1e-6f ? 0.0f : 1.0f / det;
        n00 = c00 * inv_det; n01 = c01 * inv_det; n02 = c02 * inv_det;
        n11 = fmaf(a00, a22, -a02 * a02) * inv_det;
        n12 = fmaf(a01, a02, -a00 * a12) * inv_det;
        n22 = fmaf(a00, a11, -a01 * a01) * inv_det;
    }
    const float v0 = chain3(R[0][0], r0, R[1][0], r1, R[2][0], r2);
    const float v1 = chain3(R[0][1], r0, R[1][1], r1, R[2][1], r2);
    const float v2 = chain3(R[0][2], r0, R[1][2], r1, R[2][2], r2);
    const float u0 = chain3(n00, v0, n01, v1, n02, v2);
    const float u1 = chain3(n01, v0, n11, v1, n12, v2);
    const float u2 = chain3(n02, v0, n12, v1, n22, v2);
    const float sq = chain3(v0, u0, v1, u1, v2, u2);
    const float rn = sqrtf(sq);
    if constexpr (ERR_ONLY) {
        acc[0] += robust_error<LOSS>(rn, P.scale);
        ++cnt;
    } else {
    const float w = robust_weight<LOSS>(rn, P.scale);
    const float px = s.x, py = s.y, pz = s.z;
    // G = S N, rows: p x (columns of N)
    const float g00 = fmaf(py, n02, -pz * n01), g01 = fmaf(py, n12, -pz * n11), g02 = fmaf(py, n22, -pz * n12);
    const float g10 = fmaf(pz, n00, -px * n02), g11 = fmaf(pz, n01, -px * n12), g12 = fmaf(pz, n02, -px * n22);
    const float g20 = fmaf(px, n01, -py * n00), g21 = fmaf(px, n11, -py * n01), g22 = fmaf(px, n12, -py * n02);
    // H_rr = -G S (symmetric)
    const float h00 = fmaf(g02, py, -g01 * pz), h01 = fmaf(g00, pz, -g02 * px), h02 = fmaf(g01, px, -g00 * py);
    const float h11 = fmaf(g10, pz, -g12 * px), h12 = fmaf(g11, px, -g10 * py), h22 = fmaf(g21, px, -g20 * py);
    acc[0] += w * h00; acc[1] += w * h01; acc[2] += w * h02; acc[3] += w * g00; acc[4] += w * g01; acc[5] += w * g02;
    acc[6] += w * h11; acc[7] += w * h12; acc[8] += w * g10; acc[9] += w * g11; acc[10] += w * g12;
    acc[11] += w * h22; acc[12] += w * g20; acc[13] += w * g21; acc[14] += w * g22;
    acc[15] += w * n00; acc[16] += w * n01; acc[17] += w * n02; acc[18] += w * n11; acc[19] += w * n12; acc[20] += w * n22;
    // b = [u x p, -u]
    acc[21] += w * fmaf(u1, pz, -u2 * py); acc[22] += w * fmaf(u2, px, -u0 * pz); acc[23] += w * fmaf(u0, py, -u1 * px);
    acc[24] += w * -u0; acc[25] += w * -u1; acc[26] += w * -u2;
    acc[27] += robust_error<LOSS>(rn, P.scale);
    ++cnt;
    }
}

// The cached correspondence of a source point: 3 x float4 in SOURCE order (sp_gicp_source::ccache)
//   row[0] = (t.x, t.y, t.z, rho_t^2)                  what the reuse certificate needs comes first
//   row[1] = (xx, xy, xz, yy)                          prepared covariance row of t ...
//   row[2] = (yz, zz, index bits, grid position bits)  ... index -1 (and rho^2 = 0): nothing found, searched again next time
// Written by whoever searched for the point (fused_point inline, or gicp_search_kernel); the winner's prepared row is
// gathered here, next to the points the search has just scanned.
// NOTHING FOUND below the search bound (idx = -1): the row records WHERE the search was made, q, and the margin m by which its
// bound exceeded max_correspondence_distance B (search_margin2: squared, shrunk against rounding; 0 when there was none). No
// target lies within B + m of q, so while the point stays within m of q no target lies within B of it: the same "no
// correspondence" a fresh search would return, proven by the first test of the certificate (|q' - q|^2 < m^2) with no search —
// a source point outside the overlap is searched once, not in every iteration, and it is the dearest search there is (the
// whole ball of the bound).
__device__ __forceinline__ void store_correspondence(float4* __restrict__ row, const FusedParams& P, const Nearest& nn,
                                                     Sym3& Ct, float qx = 0.0f, float qy = 0.0f, float qz = 0.0f,
                                                     float margin2 = 0.0f) {
    float4 r0 = make_float4(qx, qy, qz, margin2), r1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f),
           r2 = make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f);
    if (nn.idx >= 0) {
        const float4 c0 = P.tcovp[2 * (size_t)nn.pos], c1 = P.tcovp[2 * (size_t)nn.pos + 1];
        r0 = make_float4(nn.x, nn.y, nn.z, c1.z);
        r1 = c0;
        r2 = make_float4(c1.x, c1.y, __int_as_float(nn.idx), __uint_as_float(nn.pos));
        Ct = Sym3{c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};
    }
    row[0] = r0;
    row[1] = r1;
    row[2] = r2;
}

// Reuse certificate. Correspondences rarely change from one iteration to the next. The previous winner t is PROVABLY still
// the nearest neighbour — same index, same distance as a fresh search — when |q - t| < rho_t, half the distance from t to
// its nearest other target point: any other target u then has |q - u| >= |t - u| - |q - t| > 2 rho_t - rho_t > |q - t|.
// Second chance (t has a close neighbour u1, so rho_t is tiny — such points would be searched in EVERY iteration): t is
// also certified when |q - t| < |q - u1| and |q - t| < d(t, u2) / 2, u2 being t's second-nearest other point — every target
// u other than t and u1 then has |q - u| >= |t - u| - |q - t| >= d(t, u2) - |q - t| > |q - t|. One 16-byte gather.
// (rho^2 carries a 1e-3 margin against rounding, certificate_kernel.)
__device__ __forceinline__ bool certified(const FusedParams& P, float d, float rho2, float qx, float qy, float qz,
                                          unsigned pos) {
    if (d < rho2) return true;
    if (!P.tnb || !(rho2 > 0.0f)) return false;
    const float4 nb = P.tnb[pos];
    return d < nb.w && d < dist2(qx, qy, qz, nb.x, nb.y, nb.z);
}

// The search of one query as the prepared paths run it. Nobody reads the neighbours when there is no nn_idx output: a
// correspondence beyond max_correspondence_distance is rejected whatever it is, so the search need not find it. The next
// float above max_d2 keeps a neighbour at exactly that distance.
__device__ __forceinline__ float search_bound2(const FusedParams& P) {
    return (P.nn_idx == nullptr && P.max_d2 < FLT_MAX) ? __uint_as_float(__float_as_uint(P.max_d2) + 1u) : FLT_MAX;
}
// The same bound widened by a margin of a quarter of max_correspondence_distance, for the searches whose failure is recorded in
// the correspondence cache (store_correspondence): margin2 = what the row's certificate may use (squared, 2 % short of the
// margin: the bound's ball and the point's motion are compared in float arithmetic).
__device__ __forceinline__ float search_bound2_margin(const FusedParams& P, float& margin2) {
    margin2 = 0.0f;
    if (P.nn_idx != nullptr || !(P.max_d2 < FLT_MAX) || P.ccache == nullptr) return search_bound2(P);
    const float B = sqrtf(P.max_d2), m = 0.25f * B;
    const float wide = (B + m) * (B + m);
    if (!(wide < FLT_MAX)) return search_bound2(P);
    margin2 = (0.98f * m) * (0.98f * m);
    return wide;
}

// One source point of the fused iteration: q = T p -> correspondence (cache row by certificate, else exact NN on the target
// grid) -> linearise -> accumulate.
// P.cache_valid: 0 the cache holds nothing (every point is searched), 1 a row is used when its certificate holds: while
// correspondences hold, an iteration is a pure coalesced stream of 84 bytes per point (12 p + 24 Cs' as planes + 48 cache
// row) with no search and no gather, and a wave whose lanes all pass never enters the search code.
// SEED: a point whose certificate fails starts its search from the previous winner (grid_nn1_fast's `seed`).
// NEG: a search that finds nothing is made with the widened bound and recorded as a negative certificate (store_correspondence).
template <int LOSS, bool FAST_NN, bool P2D = false, bool SEED = true, bool NEG = true>
__device__ __forceinline__ void fused_point(const FusedParams& P, const Rigid& T, unsigned i, float (&acc)[kAcc - 1],
                                            unsigned& cnt, unsigned& searched) {
    const float4 s = make_float4(P.src[i], P.src[P.sstride + i], P.src[2 * (size_t)P.sstride + i], 1.0f);
    float qx, qy, qz;
    transform_point(T, s.x, s.y, s.z, qx, qy, qz);
    Nearest nn;
    Sym3 Ct{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool hit = false;
    bool seeded = false;
    Nearest seed;
    seed.d2 = FLT_MAX; seed.idx = -1; seed.pos = 0; seed.x = seed.y = seed.z = 0.0f;
    float4* const row = P.ccache ? P.ccache + 3 * (size_t)i : nullptr;
    if (row && P.cache_valid) {
        const float4 r0 = row[0], r1 = row[1], r2 = row[2];
        const float d = dist2(qx, qy, qz, r0.x, r0.y, r0.z);
        const unsigned pos = __float_as_uint(r2.w);
        if (certified(P, d, r0.w, qx, qy, qz, pos)) {
            hit = true;
            nn.idx = __float_as_int(r2.z); nn.pos = pos; nn.x = r0.x; nn.y = r0.y; nn.z = r0.z;
            nn.d2 = nn.idx >= 0 ? d : FLT_MAX;
            Ct = Sym3{r1.x, r1.y, r1.z, r1.w, r2.x, r2.y};
        } else if (SEED && __float_as_int(r2.z) >= 0) {  // the previous winner: a real point, an upper bound
            seed.d2 = d; seed.idx = __float_as_int(r2.z); seed.pos = pos; seed.x = r0.x; seed.y = r0.y; seed.z = r0.z;
            seeded = true;
        }
    }
    if (!hit) {
        ++searched;
        float margin2 = 0.0f;
        const float bound2 = NEG ? search_bound2_margin(P, margin2) : search_bound2(P);
        seeded = seeded && seed.d2 < bound2;
        if (!seeded) { seed.d2 = bound2; seed.idx = -1; seed.pos = 0; seed.x = seed.y = seed.z = 0.0f; }
        if (FAST_NN) {
            nn = grid_nn1_auto(P.tpts, P.tstart, P.g, qx, qy, qz, bound2, &seed);
        } else {
            nn = grid_nn1(P.tpts, P.tstart, P.g, qx, qy, qz, &seed, 0);
        }
        if (row) {
            if (NEG) store_correspondence(row, P, nn, Ct, qx, qy, qz, margin2);
            else store_correspondence(row, P, nn, Ct);
        } else if (nn.idx >= 0) Ct = load_sym(P.tcovp + 2 * (size_t)nn.pos);
    }
    if (P.nn_idx) {
        const unsigned o = P.perm[i];
        P.nn_idx[o] = nn.idx;
        P.nn_d2[o] = nn.d2;
    }
    if (nn.idx < 0 || nn.d2 > P.max_d2) return;
    const float* const cp = P.scovp + i;
    const size_t st = P.sstride;
    const Sym3 Cs = P2D ? Sym3{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}
                        : Sym3{cp[0], cp[st], cp[2 * st], cp[3 * st], cp[4 * st], cp[5 * st]};
    fused_math<LOSS, P2D>(P, T, s, qx, qy, qz, nn, Cs, Ct, acc, cnt);
}

// fused_point with the wave working together where a lane alone would hold it up: EVERY lane of the wave calls this
// (`valid`: the lane has a point), the first two search stages run per lane as in fused_point, and the queries still open
// after them (nothing proven inside the 4x4x4 block: points outside the overlap, holes, the rim) are finished one at a time
// by all 64 lanes (grid_nn1_ball_wave). Same correspondences, same sums per point as fused_point.
#ifdef SP_OPT_TIMING
__device__ unsigned long long g_sp_dbg[24 * 16];
__device__ unsigned g_sp_step;
__device__ unsigned long long g_sp_wg[24 * 8];
#define SP_PSTAMP(k) if (blockIdx.x == 0 && threadIdx.x == 0 && g_sp_step < 24) g_sp_dbg[g_sp_step * 16 + (k)] = wall_clock64()
#else
#define SP_PSTAMP(k)
#endif
template <int LOSS, bool P2D = false>
__device__ __forceinline__ void fused_point_wave(const FusedParams& P, const Rigid& T, unsigned i, bool valid,
                                                 float (&acc)[kAcc - 1], unsigned& cnt, unsigned& searched) {
    const unsigned ii = valid ? i : 0u;
    SP_PSTAMP(0);
    const float4 s = make_float4(P.src[ii], P.src[P.sstride + ii], P.src[2 * (size_t)P.sstride + ii], 1.0f);
    float qx, qy, qz;
    transform_point(T, s.x, s.y, s.z, qx, qy, qz);
    Nearest nn;
    nn.d2 = FLT_MAX; nn.idx = -1; nn.pos = 0; nn.x = nn.y = nn.z = 0.0f;
    Sym3 Ct{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool hit = false;
    float4* const row = P.ccache + 3 * (size_t)ii;
    Nearest seed;
    seed.d2 = FLT_MAX; seed.idx = -1; seed.pos = 0; seed.x = seed.y = seed.z = 0.0f;
    if (valid && P.cache_valid) {
        const float4 r0 = row[0], r1 = row[1], r2 = row[2];
        const float d = dist2(qx, qy, qz, r0.x, r0.y, r0.z);
        const unsigned pos = __float_as_uint(r2.w);
        if (certified(P, d, r0.w, qx, qy, qz, pos)) {
            hit = true;
            nn.idx = __float_as_int(r2.z); nn.pos = pos; nn.x = r0.x; nn.y = r0.y; nn.z = r0.z;
            nn.d2 = nn.idx >= 0 ? d : FLT_MAX;
            Ct = Sym3{r1.x, r1.y, r1.z, r1.w, r2.x, r2.y};
        } else if (__float_as_int(r2.z) >= 0) {  // the previous winner: a real point, an upper bound (grid_nn1_fast's seed)
            seed.d2 = d; seed.idx = __float_as_int(r2.z); seed.pos = pos; seed.x = r0.x; seed.y = r0.y; seed.z = r0.z;
        }
    }
    const bool search = valid && !hit;
    SP_PSTAMP(1);
    float margin2 = 0.0f;
    const float bound2 = search_bound2_margin(P, margin2);
    bool open = false;
    bool fast_ok = true;
    // (a point with a previous winner starts from it, grid_nn1_fast's seed. Sending it straight to the ball scan of that
    // distance instead of through the block stages was measured on the reference's example: slower, 462 against 397 us per
    // alignment — the ball scan's batches of four dependent loads cost more than the first block's batches of eight.)
    const bool seeded = search && seed.d2 < bound2;
    if (search) {
        ++searched;
        if (!seeded) { seed.d2 = bound2; seed.idx = -1; seed.pos = 0; seed.x = seed.y = seed.z = 0.0f; }
        fast_ok = grid_nn1_fast(P.tpts, P.tstart, P.g, qx, qy, qz, nn, bound2, &seed);
    }
    SP_PSTAMP(2);
    if (search && !fast_ok) open = !grid_nn1_block4(P.tpts, P.tstart, P.g, qx, qy, qz, nn);
    SP_PSTAMP(3);
#ifdef SP_OPT_TIMING
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const unsigned long long a = __ballot(search), b = __ballot(search && !fast_ok), c = __ballot(open);
        const unsigned long long sd = __ballot(seeded);
        if (threadIdx.x == 0 && g_sp_step < 24) {
            unsigned long long* const d = g_sp_dbg + g_sp_step * 16;
            d[8] = __popcll(a); d[9] = __popcll(b); d[10] = __popcll(c); d[11] = __popcll(sd);
        }
    }
#endif
    if (__ballot(open)) {  // (uniform)
        if (open && !(nn.d2 < FLT_MAX)) {  // no bound at all (the caller wants the neighbour whatever its distance): the lane's own walk
            const Nearest seed = nn;
            nn = grid_nn1(P.tpts, P.tstart, P.g, qx, qy, qz, &seed, 2);
            open = false;
        }
        grid_nn1_ball_wave(P.tpts, P.tstart, P.g, open, qx, qy, qz, nn);
    }
    SP_PSTAMP(4);
    if (search) store_correspondence(row, P, nn, Ct, qx, qy, qz, margin2);
    SP_PSTAMP(5);
    if (valid && P.nn_idx) {
        const unsigned o = P.perm[ii];
        P.nn_idx[o] = nn.idx;
        P.nn_d2[o] = nn.d2;
    }
    if (!valid || nn.idx < 0 || nn.d2 > P.max_d2) return;
    const float* const cp = P.scovp + ii;
    const size_t st = P.sstride;
    const Sym3 Cs = P2D ? Sym3{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}
                        : Sym3{cp[0], cp[st], cp[2 * st], cp[3 * st], cp[4 * st], cp[5 * st]};
    fused_math<LOSS, P2D>(P, T, s, qx, qy, qz, nn, Cs, Ct, acc, cnt);
    SP_PSTAMP(6);
}

// One source point, the WHOLE WAVE (every lane calls it with the same i; what it returns in acc / cnt / searched is the same on
// every lane: the caller keeps one lane's). For sources of a few thousand points — the reference pipeline aligns a random sample
// of 1000 — there are more SIMDs than queries, and what a linearisation costs is the longest chain of dependent loads any one
// lane runs into. Here a query is: cache row + certificates, else ONE ball scan by all 64 lanes (grid_nn1_query_wave) of
//   the previous winner's distance (a seed: a real point, so the scan is exact and its ball a fraction of a cell), or
//   half a cell when there is no previous winner (anything found inside is the nearest neighbour), then the search bound's ball,
// the winner's prepared row, the arithmetic of fused_point (same correspondences, same per-point terms).
//
// MARGIN CERTIFICATE (qcert, optional: a float4 per source point beside its cache row). The rho certificate asks how near the
// query is to its winner against the spacing of the TARGET; on a real scan — returns millimetres apart, the two clouds
// centimetres — it never holds, and every iteration searched every point although the pose had stopped moving. The scan knows
// more: d1, the winner's distance, and a lower bound d2 of every other target's (the runner-up it met, or its own radius).
// With the query at q_s then and at q' now: |q' - u| >= d2 - |q' - q_s| for every other target u and |q' - t| <= d1 + |q' - q_s|,
// so while the point has moved less than (d2 - d1) / 2 since it was searched, t is still the strict nearest neighbour — the
// winner a fresh search would return, at the distance computed here. qcert = (q_s, margin^2), the margin shrunk against
// rounding; a seeded scan reaches a little past its seed (twice the point's last move) so that there is a runner-up to measure.
//
// rows_out / qc_out (optional): the rows this linearisation LEAVES — another set than the one it reads when the step is
// speculative (registration_opt.hip: an LM / dog-leg trial and the linearisation at its pose in one step; the rows read stay
// what the trial's frozen correspondences are). Every point's rows are then written: found anew, or copied when a certificate held.
template <int LOSS, bool P2D = false>
__device__ __forceinline__ void fused_query_wave(const FusedParams& P, const Rigid& T, unsigned i, float (&acc)[kAcc - 1],
                                                 unsigned& cnt, unsigned& searched, float4* rows_out = nullptr,
                                                 const float4* qc_in = nullptr, float4* qc_out = nullptr) {
    SP_PSTAMP(0);
    const float4 s = make_float4(P.src[i], P.src[P.sstride + i], P.src[2 * (size_t)P.sstride + i], 1.0f);
    float qx, qy, qz;
    transform_point(T, s.x, s.y, s.z, qx, qy, qz);
    Nearest nn;
    nn.d2 = FLT_MAX; nn.idx = -1; nn.pos = 0; nn.x = nn.y = nn.z = 0.0f;
    Sym3 Ct{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    bool hit = false, seeded = false;
    float moved2 = 0.0f;  // how far the point is from where it was last searched (squared)
    float4* const row = P.ccache + 3 * (size_t)i;
    float4* const row_out = rows_out ? rows_out + 3 * (size_t)i : row;
    if (qc_in != nullptr && qc_out == nullptr) qc_out = const_cast<float4*>(qc_in);
    if (P.cache_valid) {
        const float4 r0 = row[0], r1 = row[1], r2 = row[2];
        const float4 qc = qc_in ? qc_in[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float d = dist2(qx, qy, qz, r0.x, r0.y, r0.z);
        const unsigned pos = __float_as_uint(r2.w);
        const bool real = __float_as_int(r2.z) >= 0;
        if (qc_in) moved2 = dist2(qx, qy, qz, qc.x, qc.y, qc.z);
        if (certified(P, d, r0.w, qx, qy, qz, pos) || (qc_in && real && moved2 < qc.w)) {
            hit = true;
            nn.idx = __float_as_int(r2.z); nn.pos = pos; nn.x = r0.x; nn.y = r0.y; nn.z = r0.z;
            nn.d2 = nn.idx >= 0 ? d : FLT_MAX;
            Ct = Sym3{r1.x, r1.y, r1.z, r1.w, r2.x, r2.y};
            if (row_out != row) { row_out[0] = r0; row_out[1] = r1; row_out[2] = r2; }
            if (qc_out && qc_out != qc_in) qc_out[i] = qc;
        } else if (real) {  // the previous winner: a real point, an upper bound
            nn.d2 = d; nn.idx = __float_as_int(r2.z); nn.pos = pos; nn.x = r0.x; nn.y = r0.y; nn.z = r0.z;
            seeded = true;
        }
    }
    SP_PSTAMP(1);
    if (!hit) {  // (uniform)
        ++searched;
        float margin2 = 0.0f;
        const float bound2 = search_bound2_margin(P, margin2);
        float second2 = 0.0f, d1 = 0.0f;
        if (seeded && nn.d2 < bound2) {
            // (a little past the seed — twice the point's last move, a quarter of a cell at most: room for a runner-up that tells
            // how far the point may move before its winner can change)
            const float pad = qc_in ? fminf(2.0f * sqrtf(moved2), 0.25f * P.g.h) : 0.0f;
            const float reach = sqrtf(nn.d2) + pad;
            grid_nn1_query_wave(P.tpts, P.tstart, P.g, qx, qy, qz, fminf(reach * reach, bound2), nn, second2);
        } else {
            const float half = 0.5f * P.g.h;
            const float first2 = fminf(half * half, bound2);
            nn.d2 = first2; nn.idx = -1; nn.pos = 0; nn.x = nn.y = nn.z = 0.0f;
            grid_nn1_query_wave(P.tpts, P.tstart, P.g, qx, qy, qz, first2, nn, second2);
            if (nn.idx < 0 && first2 < bound2) {
                nn.d2 = bound2;
                grid_nn1_query_wave(P.tpts, P.tstart, P.g, qx, qy, qz, bound2, nn, second2);
            }
        }
        d1 = nn.d2;
        if (nn.idx < 0) nn.d2 = FLT_MAX;
        SP_PSTAMP(2); SP_PSTAMP(3); SP_PSTAMP(4);
        store_correspondence(row_out, P, nn, Ct, qx, qy, qz, margin2);  // (64 lanes, one address each, the same bits)
        if (qc_out) {
            // margin = (d2 - d1) / 2, 5 % short and less the rounding of the three distances involved (a difference of coordinates
            // of size s is off by an ulp of s, 6e-8 s: 2e-6 s covers them several times); never more than a cell (a second
            // target may simply not exist)
            float m = 0.0f;
            if (nn.idx >= 0) {
                const float scale = fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fmaxf(fabsf(qz), 1.0f));
                m = 0.475f * (sqrtf(second2) - sqrtf(d1)) - 2.0e-6f * scale;
                m = fminf(fmaxf(m, 0.0f), P.g.h);
            }
            qc_out[i] = make_float4(qx, qy, qz, m * m);
        }
        SP_PSTAMP(5);
    }
    if (nn.idx < 0 || nn.d2 > P.max_d2) return;
    const float* const cp = P.scovp + i;
    const size_t st = P.sstride;
    const Sym3 Cs = P2D ? Sym3{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}
                        : Sym3{cp[0], cp[st], cp[2 * st], cp[3 * st], cp[4 * st], cp[5 * st]};
    fused_math<LOSS, P2D>(P, T, s, qx, qy, qz, nn, Cs, Ct, acc, cnt);
    SP_PSTAMP(6);
}

// One source point of the K12 on the prepared path (Registration::compute_error_parallel_reduction, registration.hpp:678-777, as
// the LM and dog-leg trial steps call it, :854, :933): the error at the TRIAL pose T with the correspondence FROZEN at that
// of the last linearisation (pose TL) — read from the point's cache row, not from neighbour arrays. The inlier gate uses the
// distance at TL (nn_d2 in the reference), recomputed with the search's own arithmetic, i.e. the same bits.
template <int LOSS, bool P2D>
__device__ __forceinline__ void error_prepared_point(const FusedParams& P, const Rigid& T, const Rigid& TL, unsigned i,
                                                     float (&acc)[1], unsigned& cnt) {
    const float4 s = make_float4(P.src[i], P.src[P.sstride + i], P.src[2 * (size_t)P.sstride + i], 1.0f);
    const float4* const row = P.ccache + 3 * (size_t)i;
    const float4 tp = row[0], c0 = row[1], c1 = row[2];
    if (__float_as_int(c1.z) < 0) return;  // no neighbour found
    float lx, ly, lz;
    transform_point(TL, s.x, s.y, s.z, lx, ly, lz);
    if (dist2(lx, ly, lz, tp.x, tp.y, tp.z) > P.max_d2) return;  // registration.hpp:716-718
    float qx, qy, qz;
    transform_point(T, s.x, s.y, s.z, qx, qy, qz);
    Nearest nn;
    nn.x = tp.x; nn.y = tp.y; nn.z = tp.z; nn.idx = __float_as_int(c1.z); nn.pos = 0; nn.d2 = 0.0f;
    const float* const cp = P.scovp + i;
    const size_t st = P.sstride;
    Sym3 Cs{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (!P2D) Cs = Sym3{cp[0], cp[st], cp[2 * st], cp[3 * st], cp[4 * st], cp[5 * st]};
    const Sym3 Ct{c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};
    fused_math<LOSS, P2D, true, 1>(P, T, s, qx, qy, qz, nn, Cs, Ct, acc, cnt);
}

// block_reduce_store for a launch of ONE workgroup: the totals go straight to tot[0 .. NV + 1] in LDS (NV sums, the uint32
// count, the second integer as a float value) instead of a partial row in memory that the same workgroup would read back.
// Ends with a barrier.
template <int NV, int BLOCK>
__device__ __forceinline__ void block_reduce_lds(float (&acc)[NV], unsigned cnt, unsigned extra, float* __restrict__ tot) {
    __shared__ float red[BLOCK / kWave][kPartial];
    const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[e] = wave_sum_to_lane63(acc[e]);
    cnt = wave_sum_u32_to_lane63(cnt);
    extra = wave_sum_u32_to_lane63(extra);
    if (lane == kWave - 1) {
#pragma unroll
        for (int e = 0; e < NV; ++e) red[wave][e] = acc[e];
        red[wave][NV] = __uint_as_float(cnt);
        red[wave][NV + 1] = __uint_as_float(extra);
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < BLOCK / kWave; ++w) s += red[w][threadIdx.x];
        tot[threadIdx.x] = s;
    } else if (threadIdx.x == NV || threadIdx.x == NV + 1) {
        unsigned c = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / kWave; ++w) c += __float_as_uint(red[w][threadIdx.x]);
        tot[threadIdx.x] = threadIdx.x == NV ? __uint_as_float(c) : (float)c;
    }
    __syncthreads();
}

constexpr int kAlignBlock = 1024;      // 16 waves: one workgroup per CU at 4 waves/SIMD -> 256 partial rows
constexpr int kAlignMaxBlocks = 256;
constexpr int kSearchedLog = 64;       // launches of an alignment whose searched-point counts are kept
// Whether a point whose certificate failed starts its search from the previous winner (fused_point's SEED) in the per-iteration
// kernels of registration.hip. Off: at their 128-register budget it costs the benchmarked instantiation one spilled register.
#ifndef SP_SEED_SEARCHES
#define SP_SEED_SEARCHES 1
#endif
constexpr bool kSeedSearches = SP_SEED_SEARCHES != 0;
#ifndef SP_NEG_CERT
#define SP_NEG_CERT 0
#endif
constexpr bool kNegCert = SP_NEG_CERT != 0;  // negative certificates in the per-iteration kernels of registration.hip (fused_point's NEG)
// Whether the per-iteration kernels finish their open queries (nothing proven inside the 4x4x4 block) with the whole wave
// (fused_point_wave) instead of each lane for itself. A / B on the same box: profiles/r05_c_*.
#ifndef SP_WAVE_TAIL
#define SP_WAVE_TAIL 0
#endif
constexpr bool kWaveTail = SP_WAVE_TAIL != 0;

// The pose of a launch into scalar registers (it is uniform; it would otherwise occupy 12 VGPRs for the whole loop).
__device__ __forceinline__ Rigid uniform_pose(const float* sT) {
    Rigid T = load_rigid_colmajor(sT);
    auto uniform = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) T.R[r][c] = uniform(T.R[r][c]);
        T.t[r] = uniform(T.t[r]);
    }
    return T;
}
constexpr int kTicketShards = 8;
constexpr int kTicketStride = 32;  // uint32 words between two shards (128 bytes: a line each)

}  // namespace
}  // namespace sp

struct sp_gicp_target {
    const sp_grid* grid = nullptr;  // borrowed: must outlive this object
    float4* covp = nullptr;         // 2 x float4 per target point, grid order: (xx,xy,xz,yy | yz,zz,rho^2,0)
    float* rho2 = nullptr;          // per target point (original order): squared safe radius of the reuse test
    float4* nb = nullptr;           // per target point (grid order): second certificate (nearest neighbour, second radius)
    unsigned long long version = 0; // bumped by every sp_gicp_target_update (cached copies of rows become stale)
    int reg_type = SP_REG_GICP;     // what the rows hold: plane(Ct) for GICP, inverse(Ct) for POINT_TO_DISTRIBUTION
    size_t n = 0;
    mutable sp::StreamSet streams;  // streams the rows (and the borrowed grid) have been used on
    void note(hipStream_t st) const { streams.note(st); if (grid) sp::grid_use(grid, st); }
};
namespace sp {
// Tagged partial rows of the wave-per-point optimiser launch: 2 (step parity) x kAlignMaxBlocks rows x 32 granules of
// {float value, uint32 tag} (registration_opt.hip, tagged_rows_exchange).
constexpr size_t kOptRowsBytes = 2 * (size_t)256 * 32 * sizeof(unsigned long long);
}  // namespace sp
struct sp_gicp_source {
    size_t n_max = 0, n = 0;
    float4* pts = nullptr;    // n points in prepared order
    float4* covp = nullptr;   // 2 x float4 per point, prepared order
    unsigned* perm = nullptr; // prepared position -> original index
    float4* ccache = nullptr;      // 3 x float4 per prepared point: its previous correspondence (see fused_point)
    float4* qcert = nullptr;       // margin certificates of the wave-per-point optimiser launch (fused_query_wave), qcert_points of them
    float4* qcert2 = nullptr;      // ... and of the other row set of fused steps (min(n_max, 2048) points)
    size_t qcert_points = 0;
    mutable bool qcert_valid = false;  // the margin certificates belong to the cache rows as they are (only a wave-per-point launch keeps them so)
    float4* ccache2 = nullptr;     // a second set of cache rows for sp_gicp_align_optimize's fused trial + linearisation steps (min(n_max, 2048) points)
    unsigned long long* opt_rows = nullptr;  // sp_gicp_align_optimize, wave-per-point launches: tagged partial rows (kOptRowsBytes)
    mutable unsigned opt_epoch = 0;          // ... and the epoch of the latest such launch (12 bits of every row tag)
    mutable bool cache_valid = false;  // set by the first linearisation after prepare
    const sp_gicp_target* cache_target = nullptr;  // the copies are of this target ...
    unsigned long long cache_version = 0;           // ... at this covariance version
    bool sorted = false;
    // measurement / tuning switches (sp_internal.h; not part of the C ABI), per prepared source:
    int opt_stage_mask = 3;  // bit 0 = per-iteration kernel, bit 1 = final reduce (+ solve) / finish kernel
    int opt_reuse = 2;       // 0 always search, 1 reuse on the first certificate, 2 also the second
    int opt_fast_nn = -1;    // -1: automatic (2x2x2 fast path when the source is cell-sorted), 0 / 1: forced
    int opt_persistent = 1;  // sp_gicp_align_fused: 1 the tail of an alignment as one launch when the grid is resident, 0 a launch per iteration
    int opt_persistent_from = 4;  // first iteration of that tail
    int opt_fuse_trials = 1; // sp_gicp_align_optimize, wave-per-point launches: LM / dog-leg trial steps also linearise at the trial pose (sp_internal.h)
    int opt_wave_query = 1;  // sp_gicp_align_optimize, small sources: one wave per point in the linearisation steps
    unsigned *keys_in = nullptr, *keys_out = nullptr, *vals_in = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
};

namespace sp {
namespace {
FusedParams make_fused_params(const sp_gicp_target* target, const sp_gicp_source* source, const sp_factor_params* params,
                              const float* transT, int transT_on_device, int32_t* nn_idx_out, float* nn_d2_out) {
    const size_t n = source->n;
    FusedParams P;
    P.src = reinterpret_cast<const float*>(source->pts);
    P.scovp = reinterpret_cast<const float*>(source->covp);
    P.sstride = (unsigned)((n + 63) / 64 * 64);
    P.tpts = target->grid->d_pts;
    P.tstart = target->grid->d_start;
    P.tcovp = target->covp;
    P.tnb = source->opt_reuse > 1 ? target->nb : nullptr;  // (null too when the target has no certificates)
    P.g = grid_desc(target->grid);
    P.n = (unsigned)n;
    P.max_d2 = params->max_correspondence_distance * params->max_correspondence_distance;
    P.scale = params->robust_scale;
    for (int i = 0; i < 16; ++i) P.T_val.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    if (transT && !transT_on_device)
        for (int i = 0; i < 16; ++i) P.T_val.m[i] = transT[i];
    P.T_dev = transT_on_device ? transT : nullptr;
    P.perm = source->perm;
    // the cache rows are always written (sp_gicp_error_prepared reads the frozen correspondences from them); the reuse
    // switch only decides whether a later linearisation may trust them instead of searching
    P.ccache = source->ccache;  // (a target without certificates still fills the rows: every linearisation then searches, seeded)
    P.cache_valid = (source->opt_reuse && source->cache_valid && source->cache_target == target &&
                     source->cache_version == target->version) ? 1 : 0;
    P.nn_idx = (nn_idx_out && nn_d2_out) ? nn_idx_out : nullptr;
    P.nn_d2 = nn_d2_out;
    return P;
}

// The prepared forms exist for RegType::GICP and POINT_TO_DISTRIBUTION; the target's rows must be of the factor asked for.
int check_prepared_reg(const char* who, const sp_gicp_target* target, const sp_factor_params* params) {
    if (params->reg_type != SP_REG_GICP && params->reg_type != SP_REG_POINT_TO_DISTRIBUTION) {
        sp_set_error("[sp_gicp_*] only RegType::GICP and RegType::POINT_TO_DISTRIBUTION have a prepared/fused form");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (params->reg_type != target->reg_type) {
        sp_set_error("[sp_gicp_*] the prepared target holds the rows of another RegType: call sp_gicp_target_prepare");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (params->rotation_constraint_enable) {
        sp_set_error("[sp_gicp_*] the rotation constraint needs the raw covariances: use sp_gicp_linearize");
        return SP_ERR_INVALID_ARGUMENT;
    }
    (void)who;
    return SP_OK;
}

unsigned align_grid(size_t n) {
    unsigned grid = div_up(n, kAlignBlock);
    return grid > (unsigned)kAlignMaxBlocks ? (unsigned)kAlignMaxBlocks : (grid ? grid : 1u);
}

}  // namespace
}  // namespace sp


#include <mutex>
namespace sp {
// One persistent launch (a kernel whose workgroups wait for each other: gicp_align_persistent_kernel, gicp_optimize_kernel) at a
// time per DEVICE of this process: two in flight on different streams would each hold CUs the other one is waiting for, and
// both would run into their time limit. persist_acquire() returns the device's guard LOCKED when such a launch may go onto
// `st` now — its grid fits the device's CUs, `st` is not capturing, and no other stream's persistent launch can still be
// running — or nullptr. The caller launches and calls persist_release(), which records the completion event and unlocks:
// check and launch are one critical section. (Other processes on the device and CU-masked streams are not seen by this guard:
// the kernels' waits are bounded and a wait that runs out is reported — sp_align_result::status, sp_gicp_align_status.)
struct PersistGuard {
    std::mutex m;
    hipStream_t last = nullptr;
    hipEvent_t done = nullptr;
    bool used = false;
    int cus = -1;
};
PersistGuard* persist_acquire(hipStream_t st, unsigned grid);
void persist_release(PersistGuard* g, hipStream_t st);
constexpr size_t kTicketOffsetBytes = 96 * 1024;  // in the workspace, behind everything align_ws() lays out (< 66 KB)
}  // namespace sp

#endif  // SP_REGISTRATION_DEVICE_H

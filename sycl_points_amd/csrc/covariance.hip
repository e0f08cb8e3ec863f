// K5/K6/K7 — neighbourhood covariance and normals for gfx950
// (replaces algorithms/feature/covariance.hpp:16-74, 260-311, 417-503).
//
// One lane per point. The k neighbour indices of a point are contiguous (k*4 bytes), the k gathered points are
// random 16-byte reads served by L2 / Infinity Cache; the output covariance is one 64-byte row written as four
// 16-byte stores. Sums follow the reference order exactly (sequential over j, multiply then add, no fusion), so
// K5 is bit-identical to an IEEE evaluation of the reference; K6/K7 go through acosf/cosf and agree to a few ulp.
// Algorithmic bytes per point (k = 20): 80 idx + 320 gathered + 64 written = 464 B.
#include "sp_common.h"
#include "sp_math.h"

void sp_set_error(const char* msg);

namespace sp {
namespace {

// covariance::kernel::estimate (covariance.hpp:16-47). Returns false when fewer than 4 neighbours: identity.
__device__ __forceinline__ bool estimate_cov(const float4* __restrict__ pts, const int32_t* __restrict__ nbr, int k,
                                             Mat3& C) {
    // outer(p,p) is bitwise symmetric (p_i*p_j == p_j*p_i), so 6 running sums carry all 9 entries.
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    float oxx = 0.0f, oxy = 0.0f, oxz = 0.0f, oyy = 0.0f, oyz = 0.0f, ozz = 0.0f;
    unsigned cnt = 0;
    for (int j = 0; j < k; ++j) {
        const int idx = nbr[j];
        if (idx < 0) continue;
        const float4 p = pts[idx];
        sx += p.x; sy += p.y; sz += p.z;
        oxx += p.x * p.x; oxy += p.x * p.y; oxz += p.x * p.z;
        oyy += p.y * p.y; oyz += p.y * p.z; ozz += p.z * p.z;
        ++cnt;
    }
    if (cnt < 4) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) C.m[i][j] = (i == j) ? 1.0f : 0.0f;
        return false;
    }
    const float inv = 1.0f / (float)cnt;  // 1.0f / correspondences (size_t -> float), covariance.hpp:44
    const float mx = sx * inv, my = sy * inv, mz = sz * inv;
    // (sum_outer * inv) - outer(mean, mean), then ensure_symmetric (eigen_utils.hpp:208-219): (a + a) * 0.5
    const float cxx = oxx * inv - mx * mx, cxy = oxy * inv - mx * my, cxz = oxz * inv - mx * mz;
    const float cyy = oyy * inv - my * my, cyz = oyz * inv - my * mz, czz = ozz * inv - mz * mz;
    const float sxy = (cxy + cxy) * 0.5f, sxz = (cxz + cxz) * 0.5f, syz = (cyz + cyz) * 0.5f;
    C.m[0][0] = cxx; C.m[0][1] = sxy; C.m[0][2] = sxz;
    C.m[1][0] = sxy; C.m[1][1] = cyy; C.m[1][2] = syz;
    C.m[2][0] = sxz; C.m[2][1] = syz; C.m[2][2] = czz;
    return true;
}

__device__ __forceinline__ void store_cov(float4* __restrict__ out, const Mat3& C) {
    out[0] = make_float4(C.m[0][0], C.m[1][0], C.m[2][0], 0.0f);  // column 0
    out[1] = make_float4(C.m[0][1], C.m[1][1], C.m[2][1], 0.0f);
    out[2] = make_float4(C.m[0][2], C.m[1][2], C.m[2][2], 0.0f);
    out[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
__device__ __forceinline__ Mat3 load_cov(const float4* __restrict__ in) {
    const float4 c0 = in[0], c1 = in[1], c2 = in[2];
    Mat3 C;
    C.m[0][0] = c0.x; C.m[1][0] = c0.y; C.m[2][0] = c0.z;
    C.m[0][1] = c1.x; C.m[1][1] = c1.y; C.m[2][1] = c1.z;
    C.m[0][2] = c2.x; C.m[1][2] = c2.y; C.m[2][2] = c2.z;
    return C;
}

// covariance::kernel::extract_normal (covariance.hpp:49-65): smallest-eigenvalue eigenvector, flipped when n.p > 1.
__device__ __forceinline__ float4 normal_of(const Mat3& C, const float4 p) {
    float ev[3];
    Mat3 V;
    symmetric_eigen3(C, ev, V);
    const float nx = V.m[0][0], ny = V.m[1][0], nz = V.m[2][0];
    const float d = chain3(nx, p.x, ny, p.y, nz, p.z);
    if (d <= 1.0f) return make_float4(nx, ny, nz, 0.0f);
    return make_float4(-nx, -ny, -nz, 0.0f);
}

__global__ __launch_bounds__(kBlock) void cov_direct_kernel(const float4* __restrict__ pts, unsigned n,
                                                            const int32_t* __restrict__ knn, int k,
                                                            float4* __restrict__ covs) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Mat3 C;
    estimate_cov(pts, knn + (size_t)i * k, k, C);
    store_cov(covs + 4 * (size_t)i, C);
}
// K5 with the two streamed arrays moved in whole lines: a workgroup's 256 index lists are one contiguous k KB block, copied
// into LDS with 16-byte loads (a lane reading its own list from global memory strides k*4 bytes: every one of its k loads
// touches 40 lines per wave); the 256 covariances (64 B each) go back through the same LDS region and leave as 16 KB of
// consecutive 16-byte stores. What remains scattered is what the algorithm scatters: the k gathered points per lane.
// Same arithmetic, same order as cov_direct_kernel. `knn` 16-byte aligned, LDS = max(k, 16) KB.
__global__ __launch_bounds__(kBlock) void cov_kernel(const float4* __restrict__ pts, unsigned n,
                                                     const int32_t* __restrict__ knn, int k,
                                                     float4* __restrict__ covs) {
    extern __shared__ int4 s_stage[];
    int32_t* s_idx = reinterpret_cast<int32_t*>(s_stage);
    const unsigned t = threadIdx.x;
    const unsigned base = blockIdx.x * kBlock;
    const unsigned cnt = min((unsigned)kBlock, n - base);
    const unsigned total = cnt * (unsigned)k;
    const int32_t* g = knn + (size_t)base * k;  // (base * k * 4 bytes: a multiple of 1 KB)
    const int4* g4 = reinterpret_cast<const int4*>(g);
    for (unsigned e = t; e < total / 4; e += kBlock) s_stage[e] = g4[e];
    for (unsigned e = (total & ~3u) + t; e < total; e += kBlock) s_idx[e] = g[e];
    __syncthreads();
    Mat3 C;
    if (t < cnt) estimate_cov(pts, s_idx + t * k, k, C);
    __syncthreads();  // every list has been read: the region now carries the covariances
    float4* s_cov = reinterpret_cast<float4*>(s_stage);
    if (t < cnt) store_cov(s_cov + 4 * t, C);
    __syncthreads();
    float4* out = covs + 4 * (size_t)base;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned e = t + r * kBlock;
        if (e < 4 * cnt) out[e] = s_cov[e];
    }
}
__global__ __launch_bounds__(kBlock) void normal_knn_kernel(const float4* __restrict__ pts, unsigned n,
                                                            const int32_t* __restrict__ knn, int k,
                                                            float4* __restrict__ normals) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Mat3 C;
    estimate_cov(pts, knn + (size_t)i * k, k, C);
    normals[i] = normal_of(C, pts[i]);
}
__global__ __launch_bounds__(kBlock) void normal_cov_kernel(const float4* __restrict__ pts,
                                                            const float4* __restrict__ covs, unsigned n,
                                                            float4* __restrict__ normals) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    normals[i] = normal_of(load_cov(covs + 4 * (size_t)i), pts[i]);
}
__global__ __launch_bounds__(kBlock) void cov_plane_kernel(const float4* __restrict__ covs, unsigned n,
                                                           float4* __restrict__ out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Mat3 P = plane_regularize(load_cov(covs + 4 * (size_t)i));
    store_cov(out + 4 * (size_t)i, P);
}

// ---- K8: M-estimated covariance (covariance.hpp:97-250, 323-381) and normalize_covariance (:76-95)
// One lane per point, a literal restatement: weighted sums in neighbour order, Mahalanobis distances to the current
// estimate, their median by insertion sort, IRLS weights, repeat. The two per-lane work arrays (weights / squared
// distances, up to 64 entries each, indexed at run time) live in LDS as [entry][lane] columns — conflict-free, and a
// private array indexed at run time would be placed in scratch memory.
constexpr int kRobustBlock = 128;
constexpr int kRobustMaxK = 64;

__device__ __forceinline__ bool estimate_cov_weighted(const float4* __restrict__ pts, const int32_t* __restrict__ nbr,
                                                      int k, const float* w /* LDS column, stride kRobustBlock */,
                                                      Mat3& C, float& mx, float& my, float& mz) {
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    float oxx = 0.0f, oxy = 0.0f, oxz = 0.0f, oyy = 0.0f, oyz = 0.0f, ozz = 0.0f;
    float tw = 0.0f;
    unsigned cnt = 0;
    for (int j = 0; j < k; ++j) {
        const int idx = nbr[j];
        if (idx < 0) continue;
        const float4 p = pts[idx];
        const float wj = w[j * kRobustBlock];
        sx += p.x * wj; sy += p.y * wj; sz += p.z * wj;
        oxx += (p.x * p.x) * wj; oxy += (p.x * p.y) * wj; oxz += (p.x * p.z) * wj;
        oyy += (p.y * p.y) * wj; oyz += (p.y * p.z) * wj; ozz += (p.z * p.z) * wj;
        ++cnt;
        tw += wj;
    }
    if (cnt < 4 || tw < FLT_EPSILON) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) C.m[i][j] = (i == j) ? 1.0f : 0.0f;
        return false;
    }
    const float inv = 1.0f / tw;
    mx = sx * inv; my = sy * inv; mz = sz * inv;
    const float cxx = oxx * inv - mx * mx, cxy = oxy * inv - mx * my, cxz = oxz * inv - mx * mz;
    const float cyy = oyy * inv - my * my, cyz = oyz * inv - my * mz, czz = ozz * inv - mz * mz;
    const float sxy = (cxy + cxy) * 0.5f, sxz = (cxz + cxz) * 0.5f, syz = (cyz + cyz) * 0.5f;
    C.m[0][0] = cxx; C.m[0][1] = sxy; C.m[0][2] = sxz;
    C.m[1][0] = sxy; C.m[1][1] = cyy; C.m[1][2] = syz;
    C.m[2][0] = sxz; C.m[2][1] = syz; C.m[2][2] = czz;
    return true;
}

template <int LOSS>
__global__ __launch_bounds__(kRobustBlock) void cov_robust_kernel(const float4* __restrict__ pts, unsigned n,
                                                                  const int32_t* __restrict__ knn, int k,
                                                                  float mad_scale, float min_scale, unsigned max_iter,
                                                                  float4* __restrict__ covs) {
    __shared__ float s_w[kRobustMaxK * kRobustBlock];
    __shared__ float s_d[kRobustMaxK * kRobustBlock];
    const unsigned i = blockIdx.x * kRobustBlock + threadIdx.x;
    if (i >= n) return;
    float* const w = s_w + threadIdx.x;
    float* const d = s_d + threadIdx.x;
    const int32_t* nbr = knn + (size_t)i * k;
    for (int j = 0; j < k; ++j) { w[j * kRobustBlock] = 1.0f; d[j * kRobustBlock] = 0.0f; }
    Mat3 C;
    float mx = 0.0f, my = 0.0f, mz = 0.0f;
    bool ok = estimate_cov_weighted(pts, nbr, k, w, C, mx, my, mz);
    for (unsigned it = 0; ok && it < max_iter; ++it) {
        const Mat3 Ci = inverse(C);
        for (int j = 0; j < k; ++j) {
            const int idx = nbr[j];
            if (idx < 0) continue;
            const float4 p = pts[idx];
            const float d0 = p.x - mx, d1 = p.y - my, d2 = p.z - mz;
            // dot<4>(diff, multiply<4,4>(cov_inv, diff)): fma chains from 0, the fourth terms are exact zeros
            const float v0 = fmaf(Ci.m[0][2], d2, fmaf(Ci.m[0][1], d1, fmaf(Ci.m[0][0], d0, 0.0f)));
            const float v1 = fmaf(Ci.m[1][2], d2, fmaf(Ci.m[1][1], d1, fmaf(Ci.m[1][0], d0, 0.0f)));
            const float v2 = fmaf(Ci.m[2][2], d2, fmaf(Ci.m[2][1], d1, fmaf(Ci.m[2][0], d0, 0.0f)));
            d[j * kRobustBlock] = fmaf(d2, v2, fmaf(d1, v1, fmaf(d0, v0, 0.0f)));
        }
        // compute_median (covariance.hpp:143-173): insertion sort of a copy (the weights array is the buffer)
        for (int j = 0; j < k; ++j) w[j * kRobustBlock] = d[j * kRobustBlock];
        for (int a = 1; a < k; ++a) {
            const float key = w[a * kRobustBlock];
            int j = a;
            while (j > 0 && w[(j - 1) * kRobustBlock] > key) {
                w[j * kRobustBlock] = w[(j - 1) * kRobustBlock];
                --j;
            }
            w[j * kRobustBlock] = key;
        }
        const int mid = k / 2;
        const float median = (k % 2 == 0) ? (w[(mid - 1) * kRobustBlock] + w[mid * kRobustBlock]) * 0.5f
                                          : w[mid * kRobustBlock];
        float scale = mad_scale * median;
        if (scale < min_scale) scale = min_scale;
        for (int j = 0; j < k; ++j) w[j * kRobustBlock] = robust_weight<LOSS>(d[j * kRobustBlock], scale);
        ok = estimate_cov_weighted(pts, nbr, k, w, C, mx, my, mz);
    }
    store_cov(covs + 4 * (size_t)i, C);
}

__global__ __launch_bounds__(kBlock) void cov_normalize_kernel(const float4* __restrict__ covs, unsigned n,
                                                               float4* __restrict__ out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    store_cov(out + 4 * (size_t)i, normalize_cov(load_cov(covs + 4 * (size_t)i)));
}

}  // namespace
}  // namespace sp

extern "C" int sp_cov_estimate(const float* points, size_t n, const int32_t* knn_idx, size_t k, float* covs_out,
                               void* stream) {
    using namespace sp;
    if (n == 0) return SP_OK;
    const float4* p = reinterpret_cast<const float4*>(points);
    float4* c = reinterpret_cast<float4*>(covs_out);
    if (k >= 1 && k <= 64 && (reinterpret_cast<uintptr_t>(knn_idx) & 15) == 0) {
        const size_t lds = (size_t)kBlock * 4 * (k < 16 ? 16 : k);
        cov_kernel<<<div_up(n, kBlock), kBlock, lds, as_stream(stream)>>>(p, (unsigned)n, knn_idx, (int)k, c);
    } else {
        cov_direct_kernel<<<div_up(n, kBlock), kBlock, 0, as_stream(stream)>>>(p, (unsigned)n, knn_idx, (int)k, c);
    }
    return launch_status();
}
extern "C" int sp_normals_from_knn(const float* points, size_t n, const int32_t* knn_idx, size_t k, float* normals_out,
                                   void* stream) {
    using namespace sp;
    if (n == 0) return SP_OK;
    normal_knn_kernel<<<div_up(n, kBlock), kBlock, 0, as_stream(stream)>>>(
        reinterpret_cast<const float4*>(points), (unsigned)n, knn_idx, (int)k, reinterpret_cast<float4*>(normals_out));
    return launch_status();
}
extern "C" int sp_normals_from_cov(const float* points, const float* covs, size_t n, float* normals_out, void* stream) {
    using namespace sp;
    if (n == 0) return SP_OK;
    if (!covs) {
        sp_set_error("[covariance::extract_normals_async] covariances not computed");
        return SP_ERR_RUNTIME;
    }
    normal_cov_kernel<<<div_up(n, kBlock), kBlock, 0, as_stream(stream)>>>(
        reinterpret_cast<const float4*>(points), reinterpret_cast<const float4*>(covs), (unsigned)n,
        reinterpret_cast<float4*>(normals_out));
    return launch_status();
}
extern "C" int sp_cov_update_plane(const float* covs, size_t n, float* covs_out, void* stream) {
    using namespace sp;
    if (n == 0) return SP_OK;
    cov_plane_kernel<<<div_up(n, kBlock), kBlock, 0, as_stream(stream)>>>(reinterpret_cast<const float4*>(covs),
                                                                          (unsigned)n, reinterpret_cast<float4*>(covs_out));
    return launch_status();
}

extern "C" int sp_cov_estimate_robust(const float* points, size_t n, const int32_t* knn_idx, size_t k, int robust_type,
                                      float mad_scale, float min_robust_scale, size_t robust_max_iterations,
                                      float* covs_out, void* stream) {
    using namespace sp;
    if (k > (size_t)kRobustMaxK) {
        sp_set_error("[covariance::estimate_robust_async] neighbor K is too large. MAX_K is 64");
        return SP_ERR_RUNTIME;
    }
    if (n == 0) return SP_OK;
    if (k == 0) return SP_ERR_INVALID_ARGUMENT;
    if (robust_type == SP_LOSS_NONE) return sp_cov_estimate(points, n, knn_idx, k, covs_out, stream);
    const unsigned grid = div_up(n, kRobustBlock);
    hipStream_t st = as_stream(stream);
    const float4* p = reinterpret_cast<const float4*>(points);
    float4* c = reinterpret_cast<float4*>(covs_out);
    const unsigned it = (unsigned)robust_max_iterations;
#define SP_ROBUST(L) cov_robust_kernel<L><<<grid, kRobustBlock, 0, st>>>(p, (unsigned)n, knn_idx, (int)k, mad_scale, min_robust_scale, it, c)
    switch (robust_type) {
        case SP_LOSS_HUBER: SP_ROBUST(LOSS_HUBER); break;
        case SP_LOSS_TUKEY: SP_ROBUST(LOSS_TUKEY); break;
        case SP_LOSS_CAUCHY: SP_ROBUST(LOSS_CAUCHY); break;
        case SP_LOSS_GEMAN_MCCLURE: SP_ROBUST(LOSS_GEMAN_MCCLURE); break;
        default: sp_set_error("[covariance::estimate_robust_async] unknown robust loss type"); return SP_ERR_INVALID_ARGUMENT;
    }
#undef SP_ROBUST
    return launch_status();
}

extern "C" int sp_cov_normalize(const float* covs, size_t n, float* covs_out, void* stream) {
    using namespace sp;
    if (n == 0) return SP_OK;
    cov_normalize_kernel<<<div_up(n, kBlock), kBlock, 0, as_stream(stream)>>>(reinterpret_cast<const float4*>(covs),
                                                                              (unsigned)n,
                                                                              reinterpret_cast<float4*>(covs_out));
    return launch_status();
}

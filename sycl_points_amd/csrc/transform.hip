// K14 — rigid/affine transform of points, covariances and normals for gfx950
// (replaces algorithms/common/transform.hpp:14-37, 45-94).
// The general 4x4 forms of the reference are kept (full homogeneous product, T*C*T^T as two 4x4x4 fma products,
// normals multiplied by T and — as in the reference, whose normalize() result is discarded — not re-normalised).
// HBM-bound: 32 B/pt (points), 128 B/pt (covariances), 32 B/pt (normals).
#include "sp_common.h"
#include "sp_math.h"

namespace sp {
namespace {

__device__ __forceinline__ float4 mat4_vec(const float* T, const float4 v) {  // eigen_utils.hpp:113-127
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        r[i] = fmaf(T[12 + i], v.w, fmaf(T[8 + i], v.z, fmaf(T[4 + i], v.y, fmaf(T[i], v.x, 0.0f))));
    return make_float4(r[0], r[1], r[2], r[3]);
}

__global__ __launch_bounds__(kBlock) void transform_vec_kernel(const float4* __restrict__ in, unsigned n, Mat4Arg T,
                                                               float4* __restrict__ out) {
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = mat4_vec(T.m, in[i]);
}

// T * (C * T^T), both products accumulated k-ascending with fma (eigen_utils.hpp:88-105), column-major storage.
__global__ __launch_bounds__(kBlock) void transform_cov_kernel(const float4* __restrict__ in, unsigned n, Mat4Arg T,
                                                               float4* __restrict__ out) {
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        float C[16], Y[16], R[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 col = in[4 * (size_t)i + c];
            C[c * 4 + 0] = col.x; C[c * 4 + 1] = col.y; C[c * 4 + 2] = col.z; C[c * 4 + 3] = col.w;
        }
        // Y = C * T^T : Y(r,c) = sum_k C(r,k) * T(c,k)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) s = fmaf(C[k * 4 + r], T.m[k * 4 + c], s);
                Y[c * 4 + r] = s;
            }
        // R = T * Y
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) s = fmaf(T.m[k * 4 + r], Y[c * 4 + k], s);
                R[c * 4 + r] = s;
            }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            out[4 * (size_t)i + c] = make_float4(R[c * 4 + 0], R[c * 4 + 1], R[c * 4 + 2], R[c * 4 + 3]);
    }
}

}  // namespace
}  // namespace sp

extern "C" int sp_transform(const float* points, const float* covs, const float* normals, size_t n,
                            const float* transT_host, float* points_out, float* covs_out, float* normals_out,
                            void* stream) {
    using namespace sp;
    if (n == 0) return SP_OK;
    Mat4Arg T;
    for (int i = 0; i < 16; ++i) T.m[i] = transT_host[i];
    hipStream_t st = as_stream(stream);
    if (covs && covs_out)
        transform_cov_kernel<<<stream_grid(n), kBlock, 0, st>>>(reinterpret_cast<const float4*>(covs), (unsigned)n, T,
                                                                reinterpret_cast<float4*>(covs_out));
    if (normals && normals_out)
        transform_vec_kernel<<<stream_grid(n), kBlock, 0, st>>>(reinterpret_cast<const float4*>(normals), (unsigned)n,
                                                                T, reinterpret_cast<float4*>(normals_out));
    if (points && points_out)
        transform_vec_kernel<<<stream_grid(n), kBlock, 0, st>>>(reinterpret_cast<const float4*>(points), (unsigned)n, T,
                                                                reinterpret_cast<float4*>(points_out));
    return launch_status();
}

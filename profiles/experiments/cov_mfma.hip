// EXPERIMENT (not part of the shipped library): K5's sum of [x y z 1]^T [x y z 1] over a point's k neighbours on the
// matrix cores — v_mfma_f32_4x4x1_16b_f32, one 4x4 outer product per 4-lane block, 16 queries per instruction.
// Reference site: algorithms/feature/covariance.hpp:16-47. Built and timed by profiles/experiments/k5_experiment.py;
// the outcome is in profiles/r03_k5_lds_and_mfma.txt. The shipped kernel is csrc/covariance.hip (VALU, bit-exact).
//
// Lane l = 4*q + c (q: query of the wave's group of 16, c: component). Per neighbour j the lane loads ONE float, component
// c of point nbr[q][j] (the 4 lanes of a block read the 16 bytes of one point), and feeds it as both A and B operand:
// D_q += p p^T with p = (x, y, z, 1). After k steps lane c holds column c of sum p p^T in D[0..2] and sum p_c in D[3];
// lane 3's D[3] is the neighbour count. A wave covers 64 queries in 4 such groups (4 accumulators in flight).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void cov_mfma_kernel(const float* __restrict__ pts, unsigned n,
                                                       const int32_t* __restrict__ knn, int k, float4* __restrict__ covs) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const unsigned c = lane & 3;
    floatx4 acc[4];
    long q[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        acc[g] = floatx4{0.f, 0.f, 0.f, 0.f};
        const unsigned qi = wave * 64 + g * 16 + (lane >> 2);
        q[g] = qi < n ? (long)qi : -1;
    }
    for (int j = 0; j < k; ++j) {
        float v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int idx = q[g] >= 0 ? knn[q[g] * k + j] : -1;
            v[g] = idx >= 0 ? pts[4 * (size_t)idx + c] : 0.0f;  // (w = 1 in the cloud: c = 3 counts the neighbour)
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(v[g], v[g], acc[g], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // block-wide values: count = lane 3's D[3]; mean_i = lane i's D[3] / count
        const float cnt = __shfl(acc[g][3], (lane & ~3u) + 3);
        const float inv = 1.0f / cnt;
        const float m0 = __shfl(acc[g][3], (lane & ~3u) + 0) * inv;
        const float m1 = __shfl(acc[g][3], (lane & ~3u) + 1) * inv;
        const float m2 = __shfl(acc[g][3], (lane & ~3u) + 2) * inv;
        const float mc = c == 0 ? m0 : (c == 1 ? m1 : m2);
        float4 col;
        if (cnt < 4.0f) col = make_float4(c == 0 ? 1.f : 0.f, c == 1 ? 1.f : 0.f, c == 2 ? 1.f : 0.f, 0.f);
        else col = make_float4(acc[g][0] * inv - m0 * mc, acc[g][1] * inv - m1 * mc, acc[g][2] * inv - m2 * mc, 0.f);
        if (c == 3) col = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q[g] >= 0) covs[4 * q[g] + c] = col;  // 64 lanes: 1 KB of consecutive 16-byte stores
    }
}

extern "C" int exp_cov_mfma(const float* points, size_t n, const int32_t* knn, size_t k, float* covs, void* stream) {
    if (n == 0) return 0;
    cov_mfma_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(points, (unsigned)n, knn, (int)k,
                                                                                  reinterpret_cast<float4*>(covs));
    return (int)hipGetLastError();
}

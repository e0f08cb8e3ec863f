// Shared launch helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/sycl_points_amd.h"

namespace sp {

constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kNumCU = 256;          // MI355X
constexpr int kBlock = 256;          // default workgroup: 4 waves, one per SIMD

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

inline int hip_status(hipError_t e) { return e == hipSuccess ? SP_OK : SP_ERR_HIP; }

// Launch-time check: converts a failed launch into SP_ERR_HIP (no sync; graph-capture safe).
inline int launch_status() { return hip_status(hipGetLastError()); }

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// Grid for a memory-bound grid-stride kernel: enough workgroups to fill 256 CUs x 8, no more.
inline unsigned stream_grid(size_t n, int block = kBlock, int per_thread = 1) {
    const size_t want = (n + (size_t)block * per_thread - 1) / ((size_t)block * per_thread);
    const size_t cap = (size_t)kNumCU * 8;
    return (unsigned)(want < 1 ? 1 : (want > cap ? cap : want));
}

// wave64 butterfly sum; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
    return v;
}

struct Mat4Arg {  // a 4x4 passed by value in the kernarg segment (column-major)
    float m[16];
};

}  // namespace sp

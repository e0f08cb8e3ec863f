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

// Launch-time check: converts a failed launch into SP_ERR_HIP (no sync; graph-capture safe). It also reports, once, a
// failure a KERNEL of an earlier call raised through device_error_word() — by then the call that hit it has returned
// SP_OK, so the message names the kernel and says so (capi_common.hip).
int launch_status();
// One word of pinned host memory per process that kernels can write (system scope) when a "cannot happen" guard trips — a
// look-back poll that ran out (sp_lookback.h) — so that it is not lost: nullptr when the allocation failed (no device).
unsigned* device_error_word();
constexpr unsigned kDevErrLookback = 1u;
constexpr unsigned kDevErrBounds = 2u;  // sp_grid_create_bounded: a finite point outside the bounds the caller vouched for

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// Grid for a memory-bound grid-stride kernel: enough workgroups to fill 256 CUs x 8, no more.
inline unsigned stream_grid(size_t n, int block = kBlock, int per_thread = 1) {
    const size_t want = (n + (size_t)block * per_thread - 1) / ((size_t)block * per_thread);
    const size_t cap = (size_t)kNumCU * 8;
    return (unsigned)(want < 1 ? 1 : (want > cap ? cap : want));
}

// Zero `bytes` bytes (a multiple of 4) at p with a KERNEL. Every enqueue-only entry point uses this instead of
// hipMemsetAsync: on ROCm 7.2 a memset NODE of a captured hipGraph was observed (tests/test_gpu_multirank.py, the fan-in
// ticket counter; also the 64 KB row reset of the rows_all_reduced = 1 loop) not to be ordered before the kernel node that
// follows it in the captured stream when the graph is replayed — eager launches and the first replay hide it.
int zero_async(void* p, size_t bytes, hipStream_t st);  // capi_common.hip; returns SP_OK / SP_ERR_HIP

// Device scratch for the structure builds (sp_grid_create and friends): the temporaries of a build (sort keys, rocPRIM
// workspaces, a few counters) are idle again when the build returns (it synchronises its stream), so they are kept and
// handed to the next build instead of going back to hipFree / hipMalloc, which cost ~0.1 ms apiece on this runtime
// (eleven of them made a 1M-point grid build 1.9 ms for 0.2 ms of kernels). Bounded: at most kKeep idle buffers are kept,
// the smallest is dropped first. The arrays of objects that outlive the call (a grid, a prepared target) come from the same
// pool through pooled_alloc / pooled_free: their destroy waits for the device to go idle (what hipFree does implicitly)
// before the buffers are offered to the next build.
hipError_t scratch_acquire(void** ptr, size_t bytes, hipStream_t st = nullptr);  // capi_common.hip; st: the stream the buffer will be used on (a buffer released behind the same stream's work is taken without waiting)
void* pinned_mailbox();  // capi_common.hip: 256 bytes of pinned host memory per host thread (nullptr: none), for small read-backs
void scratch_release(void* ptr);
template <class T> hipError_t pooled_alloc(T** ptr, size_t bytes, hipStream_t st = nullptr) { return scratch_acquire(reinterpret_cast<void**>(ptr), bytes, st); }
inline void pooled_free(void* ptr) { if (ptr) scratch_release(ptr); }  // caller: nothing on the device still uses ptr

// The streams an object's arrays have been used on (every entry point notes its stream): when the object is destroyed its
// buffers go back to the pool TAGGED with one event per stream (recorded then), and the pool hands a buffer out again only
// once its events have completed — no hipDeviceSynchronize in a destructor, so destroying an object neither stalls other
// streams nor breaks a stream capture that is running beside it. More than kMax distinct streams: the destroy path falls
// back to the device-wide wait.
struct StreamSet {
    static constexpr int kMax = 8;
    hipStream_t s[kMax];
    int n = 0;
    bool overflow = false;
    void note(hipStream_t st) {
        for (int i = 0; i < n; ++i)
            if (s[i] == st) return;
        if (n < kMax) s[n++] = st; else overflow = true;
    }
    void merge(const StreamSet& o) {
        for (int i = 0; i < o.n; ++i) note(o.s[i]);
        overflow = overflow || o.overflow;
    }
};
void scratch_release_after(void* ptr, const StreamSet& streams);  // capi_common.hip
inline void pooled_free_after(void* ptr, const StreamSet& streams) { if (ptr) scratch_release_after(ptr, streams); }
struct ScratchBuf {  // RAII handle
    void* p = nullptr;
    hipError_t get(size_t bytes, hipStream_t st = nullptr) { return scratch_acquire(&p, bytes, st); }
    template <class T> T* as() const { return static_cast<T*>(p); }
    ~ScratchBuf() { if (p) scratch_release(p); }
    /// hand the buffer back while work on `streams` may still use it (the pool waits for their events before reuse)
    void release_after(const StreamSet& streams) { if (p) { scratch_release_after(p, streams); p = nullptr; } }
    ScratchBuf() = default;
    ScratchBuf(const ScratchBuf&) = delete;
    ScratchBuf& operator=(const ScratchBuf&) = delete;
};

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

// wave64 sum delivered to lane 63 only, on the VALU's DPP path (no LDS crossbar): an inclusive scan inside each row of 16
// lanes (row_shr 1, 2, 4, 8; lanes shifted in from outside the row read 0), then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3. Six adds per value; `__shfl_xor` costs a ds_bpermute per step (measured: the 28-value
// workgroup reduction at the end of the GICP kernels took 6 us of a 27 us launch with it). Fixed tree: reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_zero(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += __int_as_float(dpp_or_zero<0x111, 0xf>(__float_as_int(v)));  // row_shr:1
    v += __int_as_float(dpp_or_zero<0x112, 0xf>(__float_as_int(v)));  // row_shr:2
    v += __int_as_float(dpp_or_zero<0x114, 0xf>(__float_as_int(v)));  // row_shr:4
    v += __int_as_float(dpp_or_zero<0x118, 0xf>(__float_as_int(v)));  // row_shr:8   -> lane 15 of every row: the row total
    v += __int_as_float(dpp_or_zero<0x142, 0xa>(__float_as_int(v)));  // row_bcast:15 -> rows 1 and 3 add the row before
    v += __int_as_float(dpp_or_zero<0x143, 0xc>(__float_as_int(v)));  // row_bcast:31 -> rows 2 and 3 add lane 31
    return v;                                                          // lane 63: the wave total
}
__device__ __forceinline__ unsigned wave_sum_u32_to_lane63(unsigned v) {
    v += (unsigned)dpp_or_zero<0x111, 0xf>((int)v);
    v += (unsigned)dpp_or_zero<0x112, 0xf>((int)v);
    v += (unsigned)dpp_or_zero<0x114, 0xf>((int)v);
    v += (unsigned)dpp_or_zero<0x118, 0xf>((int)v);
    v += (unsigned)dpp_or_zero<0x142, 0xa>((int)v);
    v += (unsigned)dpp_or_zero<0x143, 0xc>((int)v);
    return v;
}

struct Mat4Arg {  // a 4x4 passed by value in the kernarg segment (column-major)
    float m[16];
};

}  // namespace sp

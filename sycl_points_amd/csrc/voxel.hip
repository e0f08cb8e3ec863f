// K9 + voxel aggregation, K10 box filter and flag compaction for gfx950
// (replaces algorithms/common/voxel_constants.hpp:36-62, filter/voxel_downsampling.hpp:50-79,146-288,
//  filter/preprocess_operator/{common.hpp:15-25, box_filter_operator.hpp:36-44}, common/filter_by_flags.hpp:30-99).
//
// The reference computes keys on the device, then copies them to the host, std::sort's an index array and walks
// the runs sequentially. Here everything stays in HBM:
//   keys (24 B/pt) -> device LSD radix sort of (key, index u32) -> one pass that finds run heads and lets the
//   head lane sum its run in ascending point-index order (the sort is stable) -> exclusive scan of the workgroups' kept
//   counts -> scatter in ascending key order.
// Sort and scan are this library's own (radix_sort.hip: stable LSD radix sort of 32- or 64-bit keys, look-back scan); the
// key, aggregation and scatter kernels are written here. Headline algorithmic bytes: 40 B/pt (key kernel + one
// aggregation pass).
#include <cstring>

#include "radix_sort.h"
#include "sp_common.h"
#include "sp_math.h"
#include "sp_wave_select.h"

void sp_set_error(const char* msg);

namespace sp {
namespace {

constexpr uint64_t kInvalidKey = ~0ull;  // VoxelConstants::invalid_coord

// filter::kernel::compute_voxel_bit (voxel_constants.hpp:36-62)
__device__ __forceinline__ uint64_t voxel_key(const float4 p, float inv) {
    constexpr int64_t mask = (1 << 21) - 1;
    constexpr int64_t offset = 1 << 20;
    if (!isfinite(p.x) || !isfinite(p.y) || !isfinite(p.z)) return kInvalidKey;
    const int64_t c0 = (int64_t)floorf(p.x * inv) + offset;
    const int64_t c1 = (int64_t)floorf(p.y * inv) + offset;
    const int64_t c2 = (int64_t)floorf(p.z * inv) + offset;
    if (c0 < 0 || mask < c0 || c1 < 0 || mask < c1 || c2 < 0 || mask < c2) return kInvalidKey;
    return ((uint64_t)(c0 & mask)) | ((uint64_t)(c1 & mask) << 21) | ((uint64_t)(c2 & mask) << 42);
}

__global__ __launch_bounds__(kBlock) void key_kernel(const float4* __restrict__ pts, unsigned n, float inv,
                                                     uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        keys[i] = voxel_key(pts[i], inv);
        if (vals) vals[i] = i;
    }
}

// Voxel coordinates (the three 21-bit fields of the key) of a point, false when the key would be invalid.
__device__ __forceinline__ bool voxel_coords(const float4 p, float inv, int& c0, int& c1, int& c2) {
    constexpr int64_t mask = (1 << 21) - 1;
    constexpr int64_t offset = 1 << 20;
    if (!isfinite(p.x) || !isfinite(p.y) || !isfinite(p.z)) return false;
    const int64_t a = (int64_t)floorf(p.x * inv) + offset, b = (int64_t)floorf(p.y * inv) + offset,
                  c = (int64_t)floorf(p.z * inv) + offset;
    if (a < 0 || mask < a || b < 0 || mask < b || c < 0 || mask < c) return false;
    c0 = (int)a; c1 = (int)b; c2 = (int)c;
    return true;
}

// Bounding box of the voxel coordinates (integer atomics: order-independent). box = {min x,y,z, max x,y,z}.
__global__ void box_init_kernel(int32_t* box, bool eight = false) {  // eight: a whole record of voxel_report (no point outside)
    if (threadIdx.x < 3) box[threadIdx.x] = INT32_MAX;
    else if (threadIdx.x < 6) box[threadIdx.x] = INT32_MIN;
    else if (eight && threadIdx.x < 8) box[threadIdx.x] = 0;
}
__global__ __launch_bounds__(kBlock) void key_box_kernel(const float4* __restrict__ pts, unsigned n, float inv,
                                                         int32_t* __restrict__ box) {
    int lo0 = INT32_MAX, lo1 = INT32_MAX, lo2 = INT32_MAX, hi0 = INT32_MIN, hi1 = INT32_MIN, hi2 = INT32_MIN;
    // four independent loads per trip (the loop is a latency chain otherwise: 13.8 us per 1M points with one)
    const unsigned stride = gridDim.x * kBlock;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += 4 * stride) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = pts[min(i + u * stride, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int c0, c1, c2;
            if (i + u * stride < n && voxel_coords(p[u], inv, c0, c1, c2)) {
                lo0 = min(lo0, c0); lo1 = min(lo1, c1); lo2 = min(lo2, c2);
                hi0 = max(hi0, c0); hi1 = max(hi1, c1); hi2 = max(hi2, c2);
            }
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        lo0 = min(lo0, __shfl_xor(lo0, off)); lo1 = min(lo1, __shfl_xor(lo1, off)); lo2 = min(lo2, __shfl_xor(lo2, off));
        hi0 = max(hi0, __shfl_xor(hi0, off)); hi1 = max(hi1, __shfl_xor(hi1, off)); hi2 = max(hi2, __shfl_xor(hi2, off));
    }
    // one set of atomics per workgroup (a few hundred per launch): wave results meet in LDS first
    __shared__ int red[kBlock / kWave][6];
    const unsigned wave = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        red[wave][0] = lo0; red[wave][1] = lo1; red[wave][2] = lo2;
        red[wave][3] = hi0; red[wave][4] = hi1; red[wave][5] = hi2;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int v = red[0][threadIdx.x];
        for (unsigned w = 1; w < kBlock / kWave; ++w) v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&box[threadIdx.x], v);
        else atomicMax(&box[threadIdx.x], v);
    }
}

// Compressed key: the position of the voxel in its bounding box, z-major like the 63-bit key, so the order is the same;
// `invalid` (= number of cells of the box) sorts after every valid key. A point outside the box (the box did not come
// from this cloud) is counted in *status and dropped.
struct KeyBox {
    int x0, y0, z0;
    unsigned nx, ny, nz;
    unsigned invalid;
};
constexpr int kBoxShards = SP_VOXEL_BOX_SHARDS, kBoxStride = SP_VOXEL_BOX_SHARD_STRIDE;
__global__ __launch_bounds__(kBlock) void voxel_init_kernel(uint32_t* status, int32_t* box_shards) {  // one launch for both
    if (threadIdx.x == 0 && status) *status = 0u;
    if (box_shards && threadIdx.x < kBoxShards * 6)
        box_shards[(threadIdx.x / 6) * kBoxStride + threadIdx.x % 6] = (threadIdx.x % 6) < 3 ? INT32_MAX : INT32_MIN;
}
// box_shards (optional, initialised by voxel_init_kernel): the bounding box of THIS cloud's voxel coordinates, found on the
// way (the next call's guess; the exact box for a redo when the cloud left the one it was given) — a separate pass over the
// points cost 13.8 us per 1M. Sharded, every shard on a 128-byte line of its own: atomics on one LINE queue up at the L2
// (two thousand workgroups on sixteen shards packed into six lines took 12 us); the caller folds the shards.
__global__ __launch_bounds__(kBlock) void key32_kernel(const float4* __restrict__ pts, unsigned n, float inv, KeyBox b,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                       uint32_t* __restrict__ status, int32_t* __restrict__ box_shards,
                                                       int32_t* __restrict__ wg_records = nullptr) {
    // wg_records (sp_voxel_downsample_report): instead of atomics on the status word and the sharded box — which need a launch to
    // initialise them — every workgroup stores ONE record {box lo xyz, hi xyz, points outside the given box, 0}; the call's last
    // kernel folds the records (voxel_report).
    int lo0 = INT32_MAX, lo1 = INT32_MAX, lo2 = INT32_MAX, hi0 = INT32_MIN, hi1 = INT32_MIN, hi2 = INT32_MIN;
    int outside = 0;
    // four independent loads per trip (a latency chain otherwise)
    const unsigned stride = gridDim.x * kBlock;
    for (unsigned i0 = blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += 4 * stride) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = pts[min(i0 + u * stride, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = i0 + u * stride;
            if (i >= n) continue;
            int c0, c1, c2;
            uint32_t key = b.invalid;
            if (voxel_coords(p[u], inv, c0, c1, c2)) {
                lo0 = min(lo0, c0); lo1 = min(lo1, c1); lo2 = min(lo2, c2);
                hi0 = max(hi0, c0); hi1 = max(hi1, c1); hi2 = max(hi2, c2);
                const unsigned x = (unsigned)(c0 - b.x0), y = (unsigned)(c1 - b.y0), z = (unsigned)(c2 - b.z0);
                if (x < b.nx && y < b.ny && z < b.nz) key = (z * b.ny + y) * b.nx + x;
                else if (wg_records) ++outside;
                else if (status) atomicAdd(status, 1u);
            }
            keys[i] = key;
            vals[i] = i;
        }
    }
    if (!box_shards && !wg_records) return;  // uniform
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        lo0 = min(lo0, __shfl_xor(lo0, off)); lo1 = min(lo1, __shfl_xor(lo1, off)); lo2 = min(lo2, __shfl_xor(lo2, off));
        hi0 = max(hi0, __shfl_xor(hi0, off)); hi1 = max(hi1, __shfl_xor(hi1, off)); hi2 = max(hi2, __shfl_xor(hi2, off));
        outside += __shfl_xor(outside, off);
    }
    __shared__ int red[kBlock / kWave][7];
    const unsigned wave = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        red[wave][0] = lo0; red[wave][1] = lo1; red[wave][2] = lo2;
        red[wave][3] = hi0; red[wave][4] = hi1; red[wave][5] = hi2;
        red[wave][6] = outside;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        int v = red[0][threadIdx.x];
        for (unsigned w = 1; w < kBlock / kWave; ++w)
            v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : (threadIdx.x < 6 ? max(v, red[w][threadIdx.x]) : v + red[w][threadIdx.x]);
        if (wg_records) {
            wg_records[blockIdx.x * 8 + threadIdx.x] = v;
        } else if (threadIdx.x < 6) {
            int32_t* const dst = box_shards + kBoxStride * (blockIdx.x & (kBoxShards - 1)) + threadIdx.x;
            if (threadIdx.x < 3) { if (v != INT32_MAX) atomicMin(dst, v); }
            else if (v != INT32_MIN) atomicMax(dst, v);
        }
    }
}
// sp_voxel_downsample_report: the records of the key kernel's workgroups folded into report8 = {voxels, points outside the given
// box, this cloud's key box lo xyz, hi xyz}, by one workgroup of kBlock lanes (all of them call). Word 0 goes last, behind a
// system-scope fence: report8 may be host-mapped memory the caller spins on.
__device__ __forceinline__ void voxel_report(const int32_t* __restrict__ records, unsigned n_records, unsigned voxels,
                                             uint32_t* __restrict__ report8) {
    __shared__ int rred[kBlock / kWave][7];
    int v[7] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN, 0};
    for (unsigned r = threadIdx.x; r < n_records; r += kBlock) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int x = records[r * 8 + j];
            v[j] = j < 3 ? min(v[j], x) : (j < 6 ? max(v[j], x) : v[j] + x);
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int x = __shfl_xor(v[j], off);
            v[j] = j < 3 ? min(v[j], x) : (j < 6 ? max(v[j], x) : v[j] + x);
        }
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
#pragma unroll
        for (int j = 0; j < 7; ++j) rred[threadIdx.x / kWave][j] = v[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            int x = rred[0][j];
            for (unsigned w = 1; w < kBlock / kWave; ++w) x = j < 3 ? min(x, rred[w][j]) : (j < 6 ? max(x, rred[w][j]) : x + rred[w][j]);
            report8[j < 6 ? 2 + j : 1] = (uint32_t)x;
        }
        __threadfence_system();
        __hip_atomic_store(report8, voxels, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(kBlock) void voxel_report_kernel(const int32_t* __restrict__ records, unsigned n_records,
                                                              const uint32_t* __restrict__ voxels_dev, uint32_t* __restrict__ report8) {
    voxel_report(records, n_records, voxels_dev ? *voxels_dev : 0u, report8);
}
// key32_kernel for sp_voxel_downsample_report, tile by tile of the sort that follows (radix_sort.h, RadixFirstPass): the keys are
// in registers here, so the sort's first count — a launch and a read of every key — is this kernel's by-product:
// tile_hist[digit * tiles + tile] for every digit of the first pass. One record per workgroup as in key32_kernel.
template <int BINS>
__global__ __launch_bounds__(kBlock) void key32_tiles_kernel(const float4* __restrict__ pts, unsigned n, float inv, KeyBox b,
                                                             uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                             int32_t* __restrict__ wg_records, unsigned tiles, unsigned tile_keys,
                                                             unsigned mask, unsigned* __restrict__ tile_hist) {
    constexpr int kWaves = kBlock / kWave;
    __shared__ unsigned h[kWaves][BINS];
    int lo0 = INT32_MAX, lo1 = INT32_MAX, lo2 = INT32_MAX, hi0 = INT32_MIN, hi1 = INT32_MIN, hi2 = INT32_MIN;
    int outside = 0;
    const unsigned wave = threadIdx.x / kWave;
    constexpr int kPer = 8;  // keys a lane and tile (tile_keys == kPer * kBlock, checked by the caller)
    for (unsigned tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        for (unsigned d = threadIdx.x; d < kWaves * BINS; d += kBlock) (&h[0][0])[d] = 0u;
        __syncthreads();
        const unsigned base = tile * tile_keys;
        float4 p[kPer];
#pragma unroll
        for (int c = 0; c < kPer; ++c) p[c] = pts[min(base + c * kBlock + threadIdx.x, n - 1)];
#pragma unroll
        for (int c = 0; c < kPer; ++c) {
            const unsigned i = base + c * kBlock + threadIdx.x;
            if (i >= n) continue;
            int c0, c1, c2;
            uint32_t key = b.invalid;
            if (voxel_coords(p[c], inv, c0, c1, c2)) {
                lo0 = min(lo0, c0); lo1 = min(lo1, c1); lo2 = min(lo2, c2);
                hi0 = max(hi0, c0); hi1 = max(hi1, c1); hi2 = max(hi2, c2);
                const unsigned x = (unsigned)(c0 - b.x0), y = (unsigned)(c1 - b.y0), z = (unsigned)(c2 - b.z0);
                if (x < b.nx && y < b.ny && z < b.nz) key = (z * b.ny + y) * b.nx + x;
                else ++outside;
            }
            keys[i] = key;
            vals[i] = i;
            atomicAdd(&h[wave][key & mask], 1u);
        }
        __syncthreads();
        for (unsigned d = threadIdx.x; d < (unsigned)BINS; d += kBlock) {
            unsigned v = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) v += h[w][d];
            tile_hist[(size_t)d * tiles + tile] = v;
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        lo0 = min(lo0, __shfl_xor(lo0, off)); lo1 = min(lo1, __shfl_xor(lo1, off)); lo2 = min(lo2, __shfl_xor(lo2, off));
        hi0 = max(hi0, __shfl_xor(hi0, off)); hi1 = max(hi1, __shfl_xor(hi1, off)); hi2 = max(hi2, __shfl_xor(hi2, off));
        outside += __shfl_xor(outside, off);
    }
    __shared__ int red[kWaves][7];
    if ((threadIdx.x & (kWave - 1)) == 0) {
        red[wave][0] = lo0; red[wave][1] = lo1; red[wave][2] = lo2;
        red[wave][3] = hi0; red[wave][4] = hi1; red[wave][5] = hi2;
        red[wave][6] = outside;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        int v = red[0][threadIdx.x];
        for (unsigned w = 1; w < (unsigned)kWaves; ++w)
            v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : (threadIdx.x < 6 ? max(v, red[w][threadIdx.x]) : v + red[w][threadIdx.x]);
        wg_records[blockIdx.x * 8 + threadIdx.x] = v;
    }
}
__device__ __forceinline__ uint64_t expand_key(uint32_t k, const KeyBox& b) {
    if (k >= b.invalid) return kInvalidKey;
    const unsigned x = k % b.nx, yz = k / b.nx, y = yz % b.ny, z = yz / b.ny;
    return (uint64_t)(x + (unsigned)b.x0) | ((uint64_t)(y + (unsigned)b.y0) << 21) | ((uint64_t)(z + (unsigned)b.z0) << 42);
}

// VoxelGrid::compute_median (voxel_downsampling.hpp:82-98) without a scratch array: the element of rank r in a run
// is the one with exactly r elements ordered before it ((value, position) lexicographic). O(L^2) per run.
__device__ float run_select(const float* __restrict__ inten, const uint32_t* __restrict__ sv, unsigned b, unsigned e,
                            unsigned rank) {
    for (unsigned a = b; a < e; ++a) {
        const float va = inten[sv[a]];
        unsigned before = 0;
        for (unsigned c = b; c < e; ++c) {
            const float vc = inten[sv[c]];
            before += (vc < va || (vc == va && c < a)) ? 1u : 0u;
        }
        if (before == rank) return va;
    }
    return 0.0f;
}

struct AggPtrs {
    const float4* rgb;
    const float* inten;
    const float* ts;
    float4* t_rgb;
    float* t_inten;
    float* t_ts;
};

// (Round 3, measured and not kept: aggregation, offsets and scatter as ONE launch with the workgroups' output offsets by
// decoupled look-back over one word per workgroup (sp_lookback.h). With 3907 tiles of 256 positions, two thousand of them
// resident at once and all publishing their counts at the same moment, the window walks are long: 79 us against the
// 22.7 + 5.1 + 8.1 us of the three launches below, gpurun_out -> profiles/r03_voxel_trace_fused_aggregate_not_kept.txt.)
// One lane per sorted position; the head lane of a run owns the whole run (voxel_downsampling.hpp:194-208,243-268).
template <typename KEY>
__global__ __launch_bounds__(kBlock) void aggregate_kernel(const KEY* __restrict__ sk, KEY invalid,
                                                           const uint32_t* __restrict__ sv, unsigned n,
                                                           const float4* __restrict__ pts, float min_count, AggPtrs a,
                                                           float4* __restrict__ t_pts, uint32_t* __restrict__ flag,
                                                           uint32_t* __restrict__ block_keep) {
    // Every lane fetches its own sorted key, index and point at once (independent loads, one gather per lane) and parks them
    // in LDS; a run's head then walks its run out of LDS (global memory only past the end of the workgroup's 256 positions)
    // instead of chasing key -> index -> point for every member in turn.
    __shared__ KEY lkey[kBlock];
    __shared__ uint32_t lsrc[kBlock];
    __shared__ float4 lpt[kBlock];
    const unsigned base = blockIdx.x * kBlock;
    const unsigned i = base + threadIdx.x;
    const KEY key = i < n ? sk[i] : invalid;
    const KEY prev = (i > 0 && i < n) ? sk[i - 1] : invalid;
    const uint32_t my_src = i < n ? sv[i] : 0u;
    lkey[threadIdx.x] = key;
    lsrc[threadIdx.x] = my_src;
    lpt[threadIdx.x] = (i < n && key != invalid) ? pts[my_src] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
    const bool head = key != invalid && (i == 0 || prev != key);
    uint32_t keep = 0;
    float px = 0.0f, py = 0.0f, pz = 0.0f, pw = 0.0f;
    float cx = 0.0f, cy = 0.0f, cz = 0.0f, cw = 0.0f, tsum = 0.0f;
    unsigned e = i;
    if (head) {  // the part of the run inside the workgroup's 256 positions: out of LDS
        // eight members per step: their LDS reads are independent (one latency per step, not per member); the sums stay in
        // index order, and a member past the end of the run ends the walk
        const unsigned lim = min(n - base, (unsigned)kBlock);  // positions of this workgroup that exist
        bool go = true;
        while (go && e - base < lim) {
            const unsigned l0 = e - base;
            KEY kk[8];
            float4 pp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned l = min(l0 + j, (unsigned)kBlock - 1u);
                kk[j] = lkey[l];
                pp[j] = lpt[l];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (!go) break;
                if (l0 + j >= lim || kk[j] != key) { go = false; break; }
                px += pp[j].x; py += pp[j].y; pz += pp[j].z; pw += pp[j].w;
                if (a.rgb || a.ts) {
                    const uint32_t src = lsrc[l0 + j];
                    if (a.rgb) { const float4 c = a.rgb[src]; cx += c.x; cy += c.y; cz += c.z; cw += c.w; }
                    if (a.ts) tsum += a.ts[src];
                }
                ++e;
            }
        }
    }
    // A run that goes on past the workgroup's positions (a voxel of hundreds of points: the near range of a LiDAR scan at
    // 0.25 m) used to be chased by its head alone — key -> index -> point, three dependent loads per member, 55 us of a
    // 70 k-point call. The sums must stay in index order (bit-identical results), but the LOADS need not wait for each other:
    // the head's wave fetches the next 64 members at once, parks them in its quarter of the LDS tile, and the head adds them
    // in order. One head at a time per wave (there is at most one such run per workgroup boundary, rarely two).
    __syncthreads();  // every head has finished reading the tile
    {
        const unsigned lane = threadIdx.x & 63u, w0 = threadIdx.x & ~63u;
        bool more = head && e < n && e - base >= (unsigned)kBlock;
        unsigned long long todo = __ballot(more);
        while (todo) {
            const int owner = __ffsll((long long)todo) - 1;
            const unsigned e0 = (unsigned)__builtin_amdgcn_readlane((int)e, owner);
            KEY k0;
            if constexpr (sizeof(KEY) == 8) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, owner);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)key >> 32), owner);
                k0 = (KEY)(((unsigned long long)hi << 32) | lo);
            } else {
                k0 = (KEY)__builtin_amdgcn_readlane((int)key, owner);
            }
            // four batches of 64 members, all their loads in flight together (a voxel of a thousand points is four such rounds)
            constexpr int kU = 4;
            bool match[kU];
            uint32_t src[kU];
            float4 pt[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const unsigned idx = e0 + 64u * u + lane;
                const bool in = idx < n;
                const KEY kk = in ? sk[idx] : invalid;
                src[u] = in ? sv[idx] : 0u;  // (not behind the key's compare: key and index in ONE memory round trip, the points in the next)
                match[u] = in && kk == k0;
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) pt[u] = match[u] ? pts[src[u]] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            unsigned cnt = 64u;
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (cnt < 64u) break;  // (uniform: the run ended inside the previous batch)
                const unsigned long long mm = __ballot(match[u]);
                cnt = mm == ~0ull ? 64u : (unsigned)__builtin_ctzll(~mm);  // members of the run among these 64
                if (cnt == 0u) break;
                // The point sums of these cnt members, sequentially, WITHOUT the head adding them one by one out of LDS (40 ns a
                // member): lane 0 starts from the head's sums, then lane j takes lane j - 1's sums (one DPP move, wave_shr:1) plus its
                // own point, cnt - 1 times — lane j is final after j steps and stays so; eight steps a trip (a step too many changes
                // nothing, a taken branch costs as much as a step). The same adds in the same order: the same bits.
                float sx = __fadd_rn(bcast_f(px, owner), pt[u].x), sy = __fadd_rn(bcast_f(py, owner), pt[u].y),
                      sz = __fadd_rn(bcast_f(pz, owner), pt[u].z), sw = __fadd_rn(bcast_f(pw, owner), pt[u].w);
                const bool moving = lane > 0u && lane < cnt;
                for (unsigned it = 0; it + 1u < cnt; it += 8) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float tx = __fadd_rn(shift_up1_f(sx), pt[u].x), ty = __fadd_rn(shift_up1_f(sy), pt[u].y),
                                    tz = __fadd_rn(shift_up1_f(sz), pt[u].z), tw = __fadd_rn(shift_up1_f(sw), pt[u].w);
                        sx = moving ? tx : sx; sy = moving ? ty : sy; sz = moving ? tz : sz; sw = moving ? tw : sw;
                    }
                }
                const float lx = bcast_f(sx, (int)cnt - 1), ly = bcast_f(sy, (int)cnt - 1), lz = bcast_f(sz, (int)cnt - 1),
                            lw = bcast_f(sw, (int)cnt - 1);
                if (a.rgb || a.ts) {  // (uniform) the other attributes: by the head, member by member, through the staged indices
                    lsrc[w0 + lane] = src[u];
                    __builtin_amdgcn_wave_barrier();
                    if ((int)lane == owner) {
                        for (unsigned j = 0; j < cnt; ++j) {
                            if (a.rgb) { const float4 c = a.rgb[lsrc[w0 + j]]; cx += c.x; cy += c.y; cz += c.z; cw += c.w; }
                            if (a.ts) tsum += a.ts[lsrc[w0 + j]];
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if ((int)lane == owner) {
                    px = lx; py = ly; pz = lz; pw = lw;
                    e += cnt;
                }
            }
            if (cnt < 64u) todo &= todo - 1ull;  // the run has ended: the next head of this wave
        }
    }
    if (head) {
        if (pw >= min_count) {
            keep = 1;
            t_pts[i] = make_float4(px / pw, py / pw, pz / pw, pw / pw);
            if (a.rgb) a.t_rgb[i] = make_float4(cx / pw, cy / pw, cz / pw, cw / pw);
            if (a.ts) a.t_ts[i] = tsum / pw;
            if (a.inten) {
                const unsigned L = e - i, mid = L / 2;
                const float upper = run_select(a.inten, sv, i, e, mid);
                a.t_inten[i] = (L & 1u) ? upper : 0.5f * (run_select(a.inten, sv, i, e, mid - 1) + upper);
            }
        }
    }
    if (i < n) flag[i] = keep;
    // voxels kept by this workgroup: the compaction needs no scan over the n flags, only over the workgroups' counts
    const int kept = __syncthreads_count((int)keep);
    if (threadIdx.x == 0) block_keep[blockIdx.x] = (uint32_t)kept;
}

// Exclusive scan of the workgroups' kept-voxel counts, in place (one workgroup; a few thousand values), and the total.
__global__ __launch_bounds__(1024) void block_offsets_kernel(uint32_t* __restrict__ block_keep, unsigned nblocks,
                                                             uint32_t* __restrict__ n_out) {
    __shared__ unsigned wave_tot[16];
    __shared__ unsigned s_carry;
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (unsigned b0 = 0; b0 < nblocks; b0 += 1024) {
        const unsigned b = b0 + threadIdx.x;
        const unsigned v = b < nblocks ? block_keep[b] : 0u;
        unsigned inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(inc, off, 64);
            if ((int)lane >= off) inc += o;
        }
        if (lane == 63u) wave_tot[w] = inc;
        __syncthreads();
        unsigned before = s_carry, all = 0;
        for (unsigned i = 0; i < 16; ++i) { const unsigned t = wave_tot[i]; before += i < w ? t : 0u; all += t; }
        if (b < nblocks) block_keep[b] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = s_carry;
}


// FOLD (up to kFoldBlocks workgroups: 1 M positions): there is no block_offsets launch — block_off holds the RAW kept counts and
// every workgroup sums the counts of the workgroups before its own (at most 16 KB, L2-resident; the last one also writes the
// total): at these sizes a launch costs more than reading the counts again, as in the sort's scatter (radix_sort.hip).
constexpr unsigned kFoldBlocks = 4096;
template <typename KEY, bool FOLD>
__global__ __launch_bounds__(kBlock) void scatter_kernel(const uint32_t* __restrict__ flag,
                                                         const uint32_t* __restrict__ block_off, unsigned n,
                                                         const KEY* __restrict__ sk, KeyBox box,
                                                         const float4* __restrict__ t_pts, AggPtrs a,
                                                         float4* __restrict__ o_pts, float4* __restrict__ o_rgb,
                                                         float* __restrict__ o_inten, float* __restrict__ o_ts,
                                                         uint64_t* __restrict__ o_keys, uint32_t* __restrict__ n_out,
                                                         const int32_t* __restrict__ records = nullptr, unsigned n_records = 0,
                                                         uint32_t* __restrict__ report8 = nullptr) {
    __shared__ unsigned wave_kept[kBlock / 64];
    __shared__ unsigned s_all;
    __shared__ unsigned wave_before[kBlock / 64];
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t f = i < n ? flag[i] : 0u;
    unsigned before = 0;
    if constexpr (FOLD) {
        for (unsigned b = threadIdx.x; b < blockIdx.x; b += kBlock) before += block_off[b];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
        if (lane == 0u) wave_before[w] = before;
    }
    // position = kept voxels of the earlier workgroups + of the earlier waves of this one + of the earlier lanes of this wave
    const unsigned long long m = __ballot(f != 0u);
    if (lane == 0u) wave_kept[w] = (unsigned)__builtin_popcountll(m);
    __syncthreads();
    if constexpr (FOLD) {
        before = 0;
#pragma unroll
        for (unsigned j = 0; j < kBlock / 64; ++j) before += wave_before[j];
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
            unsigned all = before;
#pragma unroll
            for (unsigned j = 0; j < kBlock / 64; ++j) all += wave_kept[j];
            if (n_out) *n_out = all;
            s_all = all;
        }
    } else {
        before = block_off[blockIdx.x];
    }
    unsigned p = before + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    for (unsigned j = 0; j < w; ++j) p += wave_kept[j];
    if (f) {
        o_pts[p] = t_pts[i];
        if (a.rgb) o_rgb[p] = a.t_rgb[i];
        if (a.inten) o_inten[p] = a.t_inten[i];
        if (a.ts) o_ts[p] = a.t_ts[i];
        if (o_keys) o_keys[p] = sizeof(KEY) == 8 ? (uint64_t)sk[i] : expand_key((uint32_t)sk[i], box);
    }
    if constexpr (FOLD) {
        if (report8 && blockIdx.x == gridDim.x - 1) {  // (uniform per workgroup)
            __syncthreads();
            voxel_report(records, n_records, s_all, report8);
        }
    }
}

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
constexpr unsigned kKeyGridMax = 1024;  // workgroups of key32_kernel at most

struct VoxelWs {
    size_t keys_in, keys_out, vals_in, vals_out, flag, pos, t_pts, t_rgb, t_inten, t_ts, prim, prim_bytes, records, total;
};

VoxelWs voxel_ws(size_t n) {
    VoxelWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes); return at; };
    w.keys_in = take(n * 8); w.keys_out = take(n * 8);
    w.vals_in = take(n * 4); w.vals_out = take(n * 4);
    w.flag = take(n * 4); w.pos = take(n * 4);
    w.t_pts = take(n * 16); w.t_rgb = take(n * 16); w.t_inten = take(n * 4); w.t_ts = take(n * 4);
    w.prim_bytes = radix_sort_u64_workspace_bytes(n);
    if (radix_sort_u32_workspace_bytes(n) > w.prim_bytes) w.prim_bytes = radix_sort_u32_workspace_bytes(n);
    if (exclusive_scan_u32_workspace_bytes(n) > w.prim_bytes) w.prim_bytes = exclusive_scan_u32_workspace_bytes(n);
    w.prim = take(w.prim_bytes);
    w.records = take(kKeyGridMax * 8 * sizeof(int32_t));  // one record per workgroup of the key kernel (sp_voxel_downsample_report)
    w.total = o;
    return w;
}

// ---- box filter (K10) and stable compaction
__global__ __launch_bounds__(kBlock) void box_filter_kernel(const float4* __restrict__ pts, unsigned n, float mn,
                                                            float mx, uint8_t* __restrict__ flags) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    uint8_t f = 1;
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && isfinite(p.w))) {
        f = 0;
    } else {
        const float linf = sycl_max(fabsf(p.x), sycl_max(fabsf(p.y), fabsf(p.z)));
        if (linf < mn || linf > mx) f = 0;
    }
    flags[i] = f;
}
}  // namespace
}  // namespace sp

extern "C" int sp_voxel_keys(const float* points, size_t n, float inv_voxel_size, uint64_t* keys_out, void* stream) {
    using namespace sp;
    if (n == 0) return SP_OK;
    key_kernel<<<stream_grid(n), kBlock, 0, as_stream(stream)>>>(reinterpret_cast<const float4*>(points), (unsigned)n,
                                                                 inv_voxel_size, keys_out, nullptr);
    return launch_status();
}

extern "C" size_t sp_voxel_downsample_workspace_bytes(size_t n) { return n ? sp::voxel_ws(n).total : 0; }

namespace sp {
namespace {
int voxel_downsample_impl(const float* points, size_t n, float inv_voxel_size, size_t min_voxel_count, const float* rgb,
                          const float* intensities, const float* timestamps, float* points_out, float* rgb_out,
                          float* intensities_out, float* timestamps_out, uint64_t* keys_out_opt, uint32_t* n_out_dev,
                          const int32_t* box6_host, uint32_t* status_dev, int32_t* box_shards_dev, void* workspace,
                          size_t workspace_bytes, hipStream_t st, uint32_t* report8 = nullptr) {
    if (!(inv_voxel_size > 0.0f)) {
        sp_set_error("voxel_size must be positive");  // voxel_downsampling.hpp:23-25
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (status_dev || box_shards_dev) voxel_init_kernel<<<1, kBlock, 0, st>>>(status_dev, box_shards_dev);
    if (n == 0) {
        if (report8) voxel_report_kernel<<<1, kBlock, 0, st>>>(nullptr, 0u, nullptr, report8);  // {0, 0, empty box}
        return n_out_dev ? zero_async(n_out_dev, 4, st) : launch_status();
    }
    if (n >= (1ull << 32)) {
        sp_set_error("[VoxelGrid::downsampling] more than 2^32 points");
        return SP_ERR_INVALID_ARGUMENT;
    }
    const VoxelWs w = voxel_ws(n);
    if (!workspace || workspace_bytes < w.total) {
        sp_set_error("[VoxelGrid::downsampling] workspace too small (sp_voxel_downsample_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    char* base = static_cast<char*>(workspace);
    uint32_t* vals_in = (uint32_t*)(base + w.vals_in);
    uint32_t* vals_sorted = (uint32_t*)(base + w.vals_out);
    uint32_t* flag = (uint32_t*)(base + w.flag);
    uint32_t* pos = (uint32_t*)(base + w.pos);
    float4* t_pts = (float4*)(base + w.t_pts);
    const float4* pts = reinterpret_cast<const float4*>(points);
    AggPtrs a;
    a.rgb = reinterpret_cast<const float4*>(rgb);
    a.inten = intensities;
    a.ts = timestamps;
    a.t_rgb = (float4*)(base + w.t_rgb);
    a.t_inten = (float*)(base + w.t_inten);
    a.t_ts = (float*)(base + w.t_ts);
    int32_t* const records = report8 ? (int32_t*)(base + w.records) : nullptr;
    const unsigned agg_blocks = div_up(n, kBlock);

    // compressed 32-bit keys when the caller knows the bounding box of the voxel coordinates and it has < 2^32 cells
    KeyBox kb{0, 0, 0, 0, 0, 0, 0};
    bool boxed = false;
    if (box6_host && box6_host[0] <= box6_host[3] && box6_host[1] <= box6_host[4] && box6_host[2] <= box6_host[5]) {
        const uint64_t nx = (uint64_t)((int64_t)box6_host[3] - box6_host[0] + 1), ny = (uint64_t)((int64_t)box6_host[4] - box6_host[1] + 1),
                       nz = (uint64_t)((int64_t)box6_host[5] - box6_host[2] + 1);
        if (nx < (1ull << 21) + 1 && ny < (1ull << 21) + 1 && nz < (1ull << 21) + 1 && nx * ny < (1ull << 32) &&
            nx * ny * nz < 0xFFFFFFFFull) {
            kb = KeyBox{box6_host[0], box6_host[1], box6_host[2], (unsigned)nx, (unsigned)ny, (unsigned)nz,
                        (unsigned)(nx * ny * nz)};
            boxed = true;
        }
    }
    if (boxed) {
        uint32_t* k_in = (uint32_t*)(base + w.keys_in);
        uint32_t* k_sorted = (uint32_t*)(base + w.keys_out);
        unsigned end_bit = 1;
        while ((1ull << end_bit) <= (uint64_t)kb.invalid && end_bit < 32) ++end_bit;  // `invalid` itself must be representable
        unsigned key_grid = std::min(stream_grid(n, kBlock, 4), kKeyGridMax);
        const RadixFirstPass fp = radix_first_pass(n, end_bit);
        const bool counted = records != nullptr && fp.tile_keys == 8u * kBlock;  // (the report call: the key kernel also counts)
        if (counted) {
            key_grid = std::min(fp.tiles, kKeyGridMax);
            unsigned* const hist = reinterpret_cast<unsigned*>(base + w.prim);
            if (fp.digit_bits == 9u)
                key32_tiles_kernel<512><<<key_grid, kBlock, 0, st>>>(pts, (unsigned)n, inv_voxel_size, kb, k_in, vals_in, records, fp.tiles,
                                                                     fp.tile_keys, fp.mask, hist);
            else
                key32_tiles_kernel<256><<<key_grid, kBlock, 0, st>>>(pts, (unsigned)n, inv_voxel_size, kb, k_in, vals_in, records, fp.tiles,
                                                                     fp.tile_keys, fp.mask, hist);
        } else {
            key32_kernel<<<key_grid, kBlock, 0, st>>>(pts, (unsigned)n, inv_voxel_size, kb, k_in, vals_in, status_dev, box_shards_dev,
                                                      records);
        }
        bool in_b = false;  // the hand-written sort (radix_sort.hip) ping-pongs between the two buffer pairs
        if (radix_sort_pairs_u32(k_in, k_sorted, vals_in, vals_sorted, n, end_bit, base + w.prim, w.prim_bytes, &in_b, st, 0, counted) != SP_OK) {
            sp_set_error("[VoxelGrid::downsampling] radix sort failed");
            return SP_ERR_HIP;
        }
        if (!in_b) { uint32_t* t = k_in; k_in = k_sorted; k_sorted = t; t = vals_in; vals_in = vals_sorted; vals_sorted = t; }
        aggregate_kernel<uint32_t><<<div_up(n, kBlock), kBlock, 0, st>>>(k_sorted, kb.invalid, vals_sorted, (unsigned)n, pts,
                                                                         (float)min_voxel_count, a, t_pts, flag, pos);
        if (agg_blocks <= kFoldBlocks) {
            scatter_kernel<uint32_t, true><<<agg_blocks, kBlock, 0, st>>>(
                flag, pos, (unsigned)n, k_sorted, kb, t_pts, a, reinterpret_cast<float4*>(points_out),
                reinterpret_cast<float4*>(rgb_out), intensities_out, timestamps_out, keys_out_opt, n_out_dev, records, key_grid, report8);
            return launch_status();
        }
        uint32_t* const total = n_out_dev ? n_out_dev : reinterpret_cast<uint32_t*>(base + w.prim);  // (the sort is over: its scratch is free)
        block_offsets_kernel<<<1, 1024, 0, st>>>(pos, agg_blocks, total);
        scatter_kernel<uint32_t, false><<<agg_blocks, kBlock, 0, st>>>(
            flag, pos, (unsigned)n, k_sorted, kb, t_pts, a, reinterpret_cast<float4*>(points_out),
            reinterpret_cast<float4*>(rgb_out), intensities_out, timestamps_out, keys_out_opt, n_out_dev);
        if (report8) voxel_report_kernel<<<1, kBlock, 0, st>>>(records, key_grid, total, report8);
        return launch_status();
    }
    uint64_t* keys_in = (uint64_t*)(base + w.keys_in);
    uint64_t* keys_sorted = (uint64_t*)(base + w.keys_out);
    // no usable box (none given, or one of >= 2^32 cells): the 63-bit keys themselves, eight passes of the same sort
    key_kernel<<<stream_grid(n), kBlock, 0, st>>>(pts, (unsigned)n, inv_voxel_size, keys_in, vals_in);
    if (records) {  // this cloud's box for the report: one record, by integer atomics (no point is "outside": there is no box)
        unsigned grid = div_up(n, kBlock * 16);
        if (grid > 256u) grid = 256u;
        box_init_kernel<<<1, 64, 0, st>>>(records, true);
        key_box_kernel<<<grid ? grid : 1u, kBlock, 0, st>>>(pts, (unsigned)n, inv_voxel_size, records);
    }
    if (box_shards_dev) {  // this cloud's box for the caller, as on the boxed path
        unsigned grid = div_up(n, kBlock * 16);
        if (grid > 256u) grid = 256u;
        key_box_kernel<<<grid ? grid : 1u, kBlock, 0, st>>>(pts, (unsigned)n, inv_voxel_size, box_shards_dev);
    }
    bool in_b64 = false;
    if (radix_sort_pairs_u64(keys_in, keys_sorted, vals_in, vals_sorted, n, 64, base + w.prim, w.prim_bytes, &in_b64, st) != SP_OK) {
        sp_set_error("[VoxelGrid::downsampling] radix sort failed");
        return SP_ERR_HIP;
    }
    if (!in_b64) { uint64_t* t = keys_in; keys_in = keys_sorted; keys_sorted = t; uint32_t* tv = vals_in; vals_in = vals_sorted; vals_sorted = tv; }
    aggregate_kernel<uint64_t><<<div_up(n, kBlock), kBlock, 0, st>>>(keys_sorted, kInvalidKey, vals_sorted, (unsigned)n, pts,
                                                                     (float)min_voxel_count, a, t_pts, flag, pos);
    if (agg_blocks <= kFoldBlocks) {
        scatter_kernel<uint64_t, true><<<agg_blocks, kBlock, 0, st>>>(
            flag, pos, (unsigned)n, keys_sorted, kb, t_pts, a, reinterpret_cast<float4*>(points_out),
            reinterpret_cast<float4*>(rgb_out), intensities_out, timestamps_out, keys_out_opt, n_out_dev, records, 1u, report8);
        return launch_status();
    }
    uint32_t* const total = n_out_dev ? n_out_dev : reinterpret_cast<uint32_t*>(base + w.prim);
    block_offsets_kernel<<<1, 1024, 0, st>>>(pos, agg_blocks, total);
    scatter_kernel<uint64_t, false><<<agg_blocks, kBlock, 0, st>>>(
        flag, pos, (unsigned)n, keys_sorted, kb, t_pts, a, reinterpret_cast<float4*>(points_out),
        reinterpret_cast<float4*>(rgb_out), intensities_out, timestamps_out, keys_out_opt, n_out_dev);
    if (report8) voxel_report_kernel<<<1, kBlock, 0, st>>>(records, 1u, total, report8);
    return launch_status();
}
}  // namespace
}  // namespace sp

extern "C" int sp_voxel_downsample(const float* points, size_t n, float inv_voxel_size, size_t min_voxel_count,
                                   const float* rgb, const float* intensities, const float* timestamps,
                                   float* points_out, float* rgb_out, float* intensities_out, float* timestamps_out,
                                   uint64_t* keys_out_opt, uint32_t* n_out_dev, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    return sp::voxel_downsample_impl(points, n, inv_voxel_size, min_voxel_count, rgb, intensities, timestamps, points_out,
                                     rgb_out, intensities_out, timestamps_out, keys_out_opt, n_out_dev, nullptr, nullptr, nullptr,
                                     workspace, workspace_bytes, sp::as_stream(stream));
}

extern "C" int sp_voxel_key_box(const float* points, size_t n, float inv_voxel_size, int32_t* box6_dev, void* stream) {
    using namespace sp;
    if (!box6_dev) return SP_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    box_init_kernel<<<1, 64, 0, st>>>(box6_dev);
    if (n) {
        unsigned grid = div_up(n, kBlock * 16);  // few workgroups: their six atomics each meet on the same six words
        if (grid > 256u) grid = 256u;            // (977 workgroups: 25.8 us per 1M points; 245 with one load per trip: 13.8)
        key_box_kernel<<<grid ? grid : 1u, kBlock, 0, st>>>(reinterpret_cast<const float4*>(points), (unsigned)n,
                                                            inv_voxel_size, box6_dev);
    }
    return launch_status();
}

extern "C" int sp_voxel_downsample_boxed(const float* points, size_t n, float inv_voxel_size, size_t min_voxel_count,
                                         const float* rgb, const float* intensities, const float* timestamps,
                                         float* points_out, float* rgb_out, float* intensities_out,
                                         float* timestamps_out, uint64_t* keys_out_opt, uint32_t* n_out_dev,
                                         const int32_t* box6_host, uint32_t* status_dev_opt, int32_t* box_shards_dev_opt,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    return sp::voxel_downsample_impl(points, n, inv_voxel_size, min_voxel_count, rgb, intensities, timestamps, points_out,
                                     rgb_out, intensities_out, timestamps_out, keys_out_opt, n_out_dev, box6_host,
                                     status_dev_opt, box_shards_dev_opt, workspace, workspace_bytes, sp::as_stream(stream));
}

extern "C" int sp_voxel_downsample_report(const float* points, size_t n, float inv_voxel_size, size_t min_voxel_count,
                                          const float* rgb, const float* intensities, const float* timestamps,
                                          float* points_out, float* rgb_out, float* intensities_out,
                                          float* timestamps_out, uint64_t* keys_out_opt, uint32_t* n_out_dev_opt,
                                          const int32_t* box6_host, uint32_t* report8, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    if (!report8) return SP_ERR_INVALID_ARGUMENT;
    return sp::voxel_downsample_impl(points, n, inv_voxel_size, min_voxel_count, rgb, intensities, timestamps, points_out,
                                     rgb_out, intensities_out, timestamps_out, keys_out_opt, n_out_dev_opt, box6_host, nullptr,
                                     nullptr, workspace, workspace_bytes, sp::as_stream(stream), report8);
}

extern "C" int sp_box_filter_flags(const float* points, size_t n, float min_distance, float max_distance,
                                   uint8_t* flags_out, void* stream) {
    using namespace sp;
    if (n == 0) return SP_OK;
    box_filter_kernel<<<div_up(n, kBlock), kBlock, 0, as_stream(stream)>>>(reinterpret_cast<const float4*>(points),
                                                                           (unsigned)n, min_distance, max_distance,
                                                                           flags_out);
    return launch_status();
}

namespace sp {
namespace {
struct GatherArrays {
    const uint32_t* src[16];
    uint32_t* dst[16];
    unsigned words[16];   // 32-bit words per row
    unsigned first[17];   // prefix sums of words: one output element = first[n_arrays] words
    int n_arrays;
};
// out row j of every array = row indices[j] of its source: one lane per 32-bit word of the output
__global__ __launch_bounds__(kBlock) void gather_rows_multi_kernel(GatherArrays A, const uint32_t* __restrict__ indices, unsigned m) {
    const unsigned per = A.first[A.n_arrays];
    const size_t total = (size_t)m * per;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        const unsigned j = (unsigned)(t / per), w = (unsigned)(t % per);
        int a = 0;
        while (w >= A.first[a + 1]) ++a;
        const unsigned k = w - A.first[a];
        A.dst[a][(size_t)j * A.words[a] + k] = A.src[a][(size_t)indices[j] * A.words[a] + k];
    }
}
}  // namespace
}  // namespace sp

extern "C" int sp_gather_rows_multi(const void* const* rows, const size_t* row_bytes, void* const* rows_out, int n_arrays,
                                    const uint32_t* indices, size_t m, void* stream) {
    using namespace sp;
    if (n_arrays < 1 || n_arrays > 16 || !rows || !row_bytes || !rows_out || (!indices && m)) return SP_ERR_INVALID_ARGUMENT;
    if (m == 0) return SP_OK;
    GatherArrays A;
    A.n_arrays = n_arrays;
    A.first[0] = 0;
    for (int a = 0; a < n_arrays; ++a) {
        if (row_bytes[a] % 4 != 0 || row_bytes[a] == 0 || m >= (1ull << 30)) {
            sp_set_error("[gather_rows] row_bytes must be a positive multiple of 4 and m < 2^30");
            return SP_ERR_INVALID_ARGUMENT;
        }
        A.src[a] = static_cast<const uint32_t*>(rows[a]);
        A.dst[a] = static_cast<uint32_t*>(rows_out[a]);
        A.words[a] = (unsigned)(row_bytes[a] / 4);
        A.first[a + 1] = A.first[a] + A.words[a];
    }
    const size_t total = m * A.first[n_arrays];
    gather_rows_multi_kernel<<<(unsigned)std::min<size_t>(div_up(total, kBlock), 4096), kBlock, 0, as_stream(stream)>>>(A, indices, (unsigned)m);
    return launch_status();
}

extern "C" size_t sp_compact_workspace_bytes(size_t n) {
    if (n == 0) return 0;
    return sp::align_up(n * 4) * 2 + sp::align_up(sp::exclusive_scan_u32_workspace_bytes(n));
}

extern "C" int sp_compact_by_flags(const void* rows, size_t n, size_t row_bytes, const uint8_t* flags, void* rows_out,
                                   int32_t* new_indices_out_opt, uint32_t* n_out_dev, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    return sp_compact_by_flags_multi(&rows, &row_bytes, &rows_out, 1, n, flags, new_indices_out_opt, n_out_dev, workspace,
                                     workspace_bytes, stream);
}

namespace sp {
namespace {
int check_compact_arrays(const void* const* rows, const size_t* row_bytes, void* const* rows_out, int n_arrays, size_t n,
                         CompactArrays* out) {
    if (n_arrays < 0 || n_arrays > 16 || (n_arrays && (!rows || !row_bytes || !rows_out))) return SP_ERR_INVALID_ARGUMENT;
    bool ok_rows = n < (1ull << 30);
    for (int a = 0; a < n_arrays; ++a) ok_rows = ok_rows && row_bytes[a] % 4 == 0 && row_bytes[a] != 0;
    if (!ok_rows) {
        sp_set_error("[FilterByFlags] row_bytes must be a positive multiple of 4 and n < 2^30");
        return SP_ERR_INVALID_ARGUMENT;
    }
    int k = 0;
    for (int a = 0; a < n_arrays; ++a) {
        if (rows_out[a] == nullptr) continue;  // (an attribute that is only counted)
        out->src[k] = static_cast<const uint32_t*>(rows[a]);
        out->dst[k] = static_cast<uint32_t*>(rows_out[a]);
        out->words[k] = (unsigned)(row_bytes[a] / 4);
        ++k;
    }
    out->n_arrays = k;
    return SP_OK;
}
}  // namespace
}  // namespace sp

extern "C" int sp_compact_by_flags_multi(const void* const* rows, const size_t* row_bytes, void* const* rows_out, int n_arrays,
                                         size_t n, const uint8_t* flags, int32_t* new_indices_out_opt, uint32_t* n_out_dev,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sp;
    hipStream_t st = as_stream(stream);
    if (n_arrays < 1) return SP_ERR_INVALID_ARGUMENT;
    if (n == 0) return zero_async(n_out_dev, 4, st);
    CompactArrays A;
    if (const int rc = check_compact_arrays(rows, row_bytes, rows_out, n_arrays, n, &A); rc != SP_OK) return rc;
    if (!workspace || workspace_bytes < sp_compact_workspace_bytes(n)) {
        sp_set_error("[FilterByFlags] workspace too small (sp_compact_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    // one launch: the flags' scan by look-back and every array through it (radix_sort.hip, compact_fused_kernel)
    return compact_rows_fused(A, n, flags, nullptr, 0.0f, 0.0f, nullptr, new_indices_out_opt, n_out_dev, workspace, workspace_bytes, st);
}

extern "C" int sp_box_filter_compact_multi(const float* points, size_t n, float min_distance, float max_distance,
                                           const void* const* rows, const size_t* row_bytes, void* const* rows_out, int n_arrays,
                                           uint8_t* flags_out_opt, int32_t* new_indices_out_opt, uint32_t* n_out_dev, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    using namespace sp;
    hipStream_t st = as_stream(stream);
    if (n == 0) return zero_async(n_out_dev, 4, st);
    if (!points) return SP_ERR_INVALID_ARGUMENT;
    CompactArrays A;
    if (const int rc = check_compact_arrays(rows, row_bytes, rows_out, n_arrays, n, &A); rc != SP_OK) return rc;
    if (!workspace || workspace_bytes < sp_compact_workspace_bytes(n)) {
        sp_set_error("[BoxFilter] workspace too small (sp_compact_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    return compact_rows_fused(A, n, nullptr, reinterpret_cast<const float4*>(points), min_distance, max_distance, flags_out_opt,
                              new_indices_out_opt, n_out_dev, workspace, workspace_bytes, st);
}

// GridKNN — exact kNN on a uniform cell grid, built and searched on the device (gfx950).
//
// Why it exists: the reference's NN structure is a KD-tree whose per-query cost is a chain of ~40 dependent node
// loads (kdtree.hpp:463-553); on a GPU that chain, not bandwidth, sets the time (profiles/r01_a: 263 us per 1M
// queries). A uniform grid turns the search into a handful of INDEPENDENT loads per query: the extents of the 9
// x-rows of the 3x3x3 cell block (one round trip), then the few points of the rows that can still hold a closer
// point (one or two more). It plugs into the same seam (`KNNBase::knn_search_async`, knn/knn.hpp:14-61).
//
// Semantics: exact k nearest neighbours of transT*q, squared distances computed with the reference's fma chain
// (sp::dist2), rows ascending by (distance, target index): ties go to the LOWEST index, i.e. the result is
// bit-identical to knn_search_bruteforce (bruteforce.hpp:24-96) for every input. (The KD-tree breaks exact ties by
// visiting order instead; on tie-free data all three agree bit for bit.)
//
// Build (all on the device, deterministic): bounding box -> cell id per point -> radix sort (cell, index; radix_sort.hip) ->
// gather points into cell order as float4 {x,y,z,index-bits} -> cell_start[] by binary search.
// Search: one query per lane, rings of cells around the query's cell; a row / cell is skipped when its box lies
// farther than the current k-th distance; the search stops when the k-th distance is inside the scanned block.
// Box tests are made conservative by `eps` so that float rounding in the cell assignment can never prune a true
// neighbour.
#include <cstring>


#include <mutex>

#include "grid_device.h"
#include "radix_sort.h"

void sp_set_error(const char* msg);

#include "sp_internal.h"
#include "sp_wave_select.h"

// The (cell id, index) pairs are sorted by radix_sort.hip (3 passes for a 22-bit cell id: 63 us per 1M pairs; the library's
// Onesweep took 110-125 us at this size, its merge sort 155 us).

namespace sp {
namespace {

__device__ __forceinline__ unsigned enc(float f) {  // order-preserving float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float dec(unsigned u) {
    const unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &v, 4);
    return f;
}

// bbox[0..2] = min xyz, bbox[3..5] = max xyz (encoded), over finite points.
// Grid-stride over <= 256 workgroups, wave butterfly, LDS across the 4 waves, then 6 integer atomics per workgroup
// (exact, order independent).
__global__ __launch_bounds__(kBlock) void bbox_kernel(const float4* __restrict__ pts, unsigned n, unsigned* bbox) {
    __shared__ unsigned red[kBlock / kWave][6];
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    // (four loads in flight per lane: one at a time, 16 dependent trips made this the longest kernel of the build, 18 us)
    const unsigned stride = gridDim.x * kBlock;
    for (unsigned i0 = blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += 4 * stride) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = pts[min(i0 + u * stride, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * stride < n && isfinite(p[u].x) && isfinite(p[u].y) && isfinite(p[u].z)) {
                const unsigned e[3] = {enc(p[u].x), enc(p[u].y), enc(p[u].z)};
#pragma unroll
                for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], e[a]); mx[a] = max(mx[a], e[a]); }
            }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = min(mn[a], (unsigned)__shfl_xor((int)mn[a], o, 64));
            mx[a] = max(mx[a], (unsigned)__shfl_xor((int)mx[a], o, 64));
        }
    }
    const unsigned wave = threadIdx.x / kWave;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[wave][a] = mn[a]; red[wave][3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned v = red[0][threadIdx.x];
        for (int w = 1; w < kBlock / kWave; ++w) v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&bbox[threadIdx.x], v);
        else atomicMax(&bbox[threadIdx.x], v);
    }
}

// bounds_error (sp_grid_create_bounded): the box came from the caller — a finite point outside it would sit in a border cell whose
// geometry does not contain it, and the searches' pruning would be wrong without anybody noticing: it raises the device error word.
__global__ __launch_bounds__(kBlock) void cell_id_kernel(const float4* __restrict__ pts, GridDesc g,
                                                         unsigned* __restrict__ keys, unsigned* __restrict__ vals,
                                                         unsigned* __restrict__ bounds_error = nullptr) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= g.n) return;
    const float4 p = pts[i];
    unsigned key = (unsigned)g.nx * g.ny * g.nz;  // non-finite points: a trash cell past the grid, never searched
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        if (bounds_error) {
            const float hx = g.ox + g.nx * g.h, hy = g.oy + g.ny * g.h, hz = g.oz + g.nz * g.h;
            if (p.x < g.ox || p.y < g.oy || p.z < g.oz || p.x > hx || p.y > hy || p.z > hz)
                __hip_atomic_store(bounds_error, kDevErrBounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const int cx = cell_coord(p.x, g.ox, g.inv_h, g.nx), cy = cell_coord(p.y, g.oy, g.inv_h, g.ny),
                  cz = cell_coord(p.z, g.oz, g.inv_h, g.nz);
        key = ((unsigned)cz * g.ny + cy) * g.nx + cx;
    }
    keys[i] = key;
    vals[i] = i;
}
__global__ __launch_bounds__(kBlock) void gather_sorted_kernel(const float4* __restrict__ pts,
                                                               const unsigned* __restrict__ order, unsigned n,
                                                               float4* __restrict__ out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned src = order[i];
    float4 p = pts[src];
    p.w = __uint_as_float(src);
    out[i] = p;
}
// start[c] = first sorted position whose key >= c  (c in [0, ncells]); keys ascending.
// One lane per sorted position i in [0, n]: the cells in (key[i-1], key[i]] all start at i (position n closes the table up to
// ncells). Gaps of a few cells — the usual case — are written by the lane itself; a long gap (empty space between surfaces)
// is written by its whole wave, 64 cells per step. Coalesced key reads, near-coalesced table writes: 8 us per 1M points at
// 0.5 points per cell, against 28 us for a binary search per cell.
__global__ __launch_bounds__(kBlock) void cell_start_kernel(const unsigned* __restrict__ keys, unsigned n,
                                                            unsigned ncells, unsigned* __restrict__ start) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    const unsigned lane = threadIdx.x & (kWave - 1);
    unsigned lo = 0, hi = 0;  // cells [lo, hi) start at position i
    if (i <= n) {
        lo = i == 0 ? 0u : min(keys[i - 1], ncells) + 1u;
        hi = (i == n ? ncells : min(keys[i], ncells)) + 1u;
        if (lo > hi) lo = hi;
    }
    const unsigned len = hi - lo;
    if (len <= 8u)
        for (unsigned c = lo; c < hi; ++c) start[c] = i;
    unsigned long long m = __ballot(len > 8u);
    while (m) {
        const int L = __builtin_ctzll(m);
        m &= m - 1;
        const unsigned blo = (unsigned)__builtin_amdgcn_readlane((int)lo, L), bhi = (unsigned)__builtin_amdgcn_readlane((int)hi, L);
        const unsigned bi = (unsigned)__builtin_amdgcn_readlane((int)i, L);
        for (unsigned c = blo + lane; c < bhi; c += kWave) start[c] = bi;
    }
}

// The whole build of a SMALL grid (up to kSmallBuildPoints points, fewer than kSmallBuildCells cells) by ONE workgroup: cell ids, the
// stable LSD radix sort of (cell id, index) with the tile in LDS (the ranking of rs_scatter_kernel, radix_sort.hip: wave
// match-any by eight ballots, per-wave digit counters, wave order = position order; the pairs stay in registers between passes),
// the gather into cell order and the cell table (a suffix minimum over the runs' first positions). It replaces seven launches
// (cell ids | 2 x (count, scatter) | gather | cell table): at the size of a voxel-downsampled scan those are launch latency
// (45 us on the example's 6 k-point target against the 15 us of this kernel). Same structure bit for bit.
constexpr unsigned kSmallBuildPoints = 8192, kSmallBuildCells = 32768;  // (cells: 0 .. ncells, ncells < kSmallBuildCells)
bool g_small_build = true;  // (sp_internal_grid_small_build: tests and measurements compare the two builds)
constexpr int kSbThreads = 1024, kSbWaves = kSbThreads / 64, kSbItems = kSmallBuildPoints / kSbThreads;
__global__ __launch_bounds__(kSbThreads) void grid_build_small_kernel(const float4* __restrict__ pts, GridDesc g, unsigned ncells,
                                                                       unsigned passes, float4* __restrict__ out_pts,
                                                                       unsigned* __restrict__ start,
                                                                       unsigned* __restrict__ bounds_error) {
    __shared__ unsigned cnt[kSbWaves][256];
    __shared__ unsigned tstart[256];
    __shared__ unsigned wave_tot[kSbWaves];
    __shared__ unsigned lk[kSmallBuildPoints];
    __shared__ unsigned short lv[kSmallBuildPoints];
    const unsigned tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const unsigned n = g.n;
    // wave w owns the positions [w * 512, (w + 1) * 512), 64 at a time: position order = (wave, chunk, lane)
    unsigned k[kSbItems], v[kSbItems], rank[kSbItems];
    bool out_of_box = false;
#pragma unroll
    for (int c = 0; c < kSbItems; ++c) {
        const unsigned e = w * (kSmallBuildPoints / kSbWaves) + c * 64 + lane;
        k[c] = ncells;  // non-finite points: the trash cell past the grid (cell_id_kernel)
        v[c] = e;
        if (e < n) {
            const float4 p = pts[e];
            if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
                if (bounds_error) {
                    const float hx = g.ox + g.nx * g.h, hy = g.oy + g.ny * g.h, hz = g.oz + g.nz * g.h;
                    out_of_box |= p.x < g.ox || p.y < g.oy || p.z < g.oz || p.x > hx || p.y > hy || p.z > hz;
                }
                const int cx = cell_coord(p.x, g.ox, g.inv_h, g.nx), cy = cell_coord(p.y, g.oy, g.inv_h, g.ny),
                          cz = cell_coord(p.z, g.oz, g.inv_h, g.nz);
                k[c] = ((unsigned)cz * g.ny + cy) * g.nx + cx;
            }
        }
    }
    if (out_of_box) __hip_atomic_store(bounds_error, kDevErrBounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (unsigned pass = 0; pass < passes; ++pass) {
        const unsigned shift = 8u * pass;
        for (unsigned i = tid; i < kSbWaves * 256; i += kSbThreads) (&cnt[0][0])[i] = 0u;
        __syncthreads();
        volatile unsigned* const my_cnt = cnt[w];
#pragma unroll
        for (int c = 0; c < kSbItems; ++c) {
            const unsigned e = w * (kSmallBuildPoints / kSbWaves) + c * 64 + lane;
            const bool valid = e < n;
            const unsigned d = (k[c] >> shift) & 255u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long mm = __ballot(bit);
                peers &= bit ? mm : ~mm;
            }
            const unsigned old = my_cnt[d];  // read by every lane BEFORE the digit's first lane bumps it (LDS ops of a wave are in order)
            const unsigned lower = (unsigned)__builtin_popcountll(peers & lt);
            rank[c] = old + lower;
            if (valid && lower == 0u) my_cnt[d] = old + (unsigned)__builtin_popcountll(peers);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        {   // digit tid: the waves' totals -> each wave's offset inside the digit; the digit's first position
            unsigned tot = 0;
            if (tid < 256u) {
#pragma unroll
                for (int i = 0; i < kSbWaves; ++i) { const unsigned c = cnt[i][tid]; cnt[i][tid] = tot; tot += c; }
            }
            unsigned inc = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(inc, off, 64);
                if ((int)lane >= off) inc += o;
            }
            if (lane == 63u) wave_tot[w] = inc;
            __syncthreads();
            unsigned before = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) before += (unsigned)i < w ? wave_tot[i] : 0u;  // (digits live in waves 0..3)
            if (tid < 256u) tstart[tid] = before + inc - tot;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < kSbItems; ++c) {
            const unsigned e = w * (kSmallBuildPoints / kSbWaves) + c * 64 + lane;
            if (e < n) {
                const unsigned d = (k[c] >> shift) & 255u;
                const unsigned pos = tstart[d] + cnt[w][d] + rank[c];
                lk[pos] = k[c];
                lv[pos] = (unsigned short)v[c];
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < kSbItems; ++c) {
            const unsigned e = w * (kSmallBuildPoints / kSbWaves) + c * 64 + lane;
            if (e < n) { k[c] = lk[e]; v[c] = lv[e]; }
        }
        __syncthreads();
    }
    // points into cell order, {x, y, z, index bits} (gather_sorted_kernel)
#pragma unroll
    for (int c = 0; c < kSbItems; ++c) {
        const unsigned e = w * (kSmallBuildPoints / kSbWaves) + c * 64 + lane;
        if (e < n) {
            float4 p = pts[v[c]];
            p.w = __uint_as_float(v[c]);
            out_pts[e] = p;
        }
    }
    // start[c] = first sorted position whose key >= c, c in [0, ncells]: every run's first position is marked at its cell in an
    // LDS table, and the table's suffix minimum is the answer (32 consecutive cells a lane; the scan across lanes by shuffles,
    // across waves through LDS). cell_start_kernel's walk over the empty cells between two runs is serial per gap: on a cloud
    // of surfaces (most cells empty) it was 15 of this kernel's 25 us.
    __shared__ unsigned short table[kSmallBuildCells];
    __shared__ unsigned wave_min[kSbWaves];
    constexpr unsigned kPerLane = kSmallBuildCells / kSbThreads;  // 32
    static_assert(kPerLane % 8 == 0, "whole 16-byte groups");
    {
        uint4* const t4 = reinterpret_cast<uint4*>(table);
        for (unsigned i = tid; i < kSmallBuildCells / 8; i += kSbThreads) t4[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kSbItems; ++c) {
        const unsigned e = w * (kSmallBuildPoints / kSbWaves) + c * 64 + lane;
        if (e < n) {
            const unsigned key = min(k[c], ncells);
            if (e == 0u || min(lk[e - 1], ncells) != key) table[key] = (unsigned short)e;
        }
    }
    __syncthreads();
    unsigned sfx[kPerLane];  // suffix minima inside the lane's 32 cells
    {
        const uint4* const t4 = reinterpret_cast<const uint4*>(table) + tid * (kPerLane / 8);
        unsigned run = 0xffffu;
#pragma unroll
        for (int q = kPerLane / 8 - 1; q >= 0; --q) {
            const uint4 u = t4[q];
            const unsigned wds[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                run = min(run, wds[j] >> 16);
                sfx[q * 8 + j * 2 + 1] = run;
                run = min(run, wds[j] & 0xffffu);
                sfx[q * 8 + j * 2] = run;
            }
        }
        // exclusive suffix minimum over the lanes to the right: inside the wave, then over the waves
        unsigned inc = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_down(inc, off, 64);
            if ((int)lane + off < 64) inc = min(inc, o);
        }
        if (lane == 0u) wave_min[w] = inc;
        __syncthreads();
        unsigned carry = __shfl_down(inc, 1, 64);
        if (lane == 63u) carry = 0xffffu;
#pragma unroll
        for (int i = 0; i < kSbWaves; ++i) carry = (unsigned)i > w ? min(carry, wave_min[i]) : carry;
        carry = min(carry, n);  // (no run at or after the cell: the table closes at n)
        const unsigned c0 = tid * kPerLane;
        if (c0 + kPerLane <= ncells + 1u) {  // (all 32 cells exist: whole 16-byte stores)
            uint4* const dst = reinterpret_cast<uint4*>(start + c0);
#pragma unroll
            for (int q = 0; q < (int)kPerLane / 4; ++q)
                dst[q] = make_uint4(min(sfx[q * 4], carry), min(sfx[q * 4 + 1], carry), min(sfx[q * 4 + 2], carry), min(sfx[q * 4 + 3], carry));
        } else {
#pragma unroll
            for (int j = 0; j < (int)kPerLane; ++j)
                if (c0 + j <= ncells) start[c0 + j] = min(sfx[j], carry);
        }
    }
}

template <int KCAP>
__global__ __launch_bounds__(kBlock) void grid_search_kernel(const float4* __restrict__ pts,
                                                             const unsigned* __restrict__ start, GridDesc g,
                                                             const float4* __restrict__ queries, unsigned nq, int k,
                                                             Mat4Arg T_val, const float* __restrict__ T_dev,
                                                             int32_t* __restrict__ idx_out,
                                                             float* __restrict__ d2_out,
                                                             const unsigned* __restrict__ todo = nullptr,
                                                             const unsigned* __restrict__ todo_count = nullptr,
                                                             float bound2 = FLT_MAX) {
    // (todo: only the listed queries — what grid_search_select_kernel could not prove)
    // (bound2: only neighbours closer than this squared distance are wanted — rows stay padded beyond it and the walk ends
    // at the ring that reaches it: a query with nothing nearby does not walk the grid to its end)
    const unsigned t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= (todo ? *todo_count : nq)) return;
    const unsigned qi = todo ? todo[t] : t;
    const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
    const float4 q4 = queries[qi];
    float qx, qy, qz;
    transform_point(T, q4.x, q4.y, q4.z, qx, qy, qz);

    float bd[KCAP];
    int bi[KCAP];
#pragma unroll
    for (int i = 0; i < KCAP; ++i) { bd[i] = FLT_MAX; bi[i] = -1; }
    float kth = bound2;
    int kth_idx = -1;

    const bool finite_q = isfinite(qx) && isfinite(qy) && isfinite(qz);
    if (finite_q && g.n > 0) {
        const int cx = cell_coord(qx, g.ox, g.inv_h, g.nx), cy = cell_coord(qy, g.oy, g.inv_h, g.ny),
                  cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
        const int rmax = max(max(g.nx, g.ny), g.nz);
        for (int r = 0; r <= rmax; ++r) {
            const int z0 = max(cz - r, 0), z1 = min(cz + r, g.nz - 1);
            const int y0 = max(cy - r, 0), y1 = min(cy + r, g.ny - 1);
            const int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
            for (int z = z0; z <= z1; ++z) {
                const float dz2 = gap2(qz, g.oz + z * g.h, g.oz + (z + 1) * g.h, g.eps);
                if (dz2 > kth) continue;
                for (int y = y0; y <= y1; ++y) {
                    const float dyz2 = dz2 + gap2(qy, g.oy + y * g.h, g.oy + (y + 1) * g.h, g.eps);
                    if (dyz2 > kth) continue;
                    const bool shell_row = (r == 0) || (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
                    const unsigned row = ((unsigned)z * g.ny + y) * g.nx;
                    // a shell row is scanned over its whole x-range; an interior row only at its two end cells
                    const int nseg = shell_row ? 1 : 2;
                    for (int sgi = 0; sgi < nseg; ++sgi) {
                        int xa, xb;
                        if (shell_row) { xa = x0; xb = x1; }
                        else if (sgi == 0) { xa = cx - r; xb = cx - r; if (xa < 0) continue; }
                        else { xa = cx + r; xb = cx + r; if (xb > g.nx - 1) continue; }
                        const float d2box = dyz2 + gap2(qx, g.ox + xa * g.h, g.ox + (xb + 1) * g.h, g.eps);
                        if (d2box > kth) continue;
                        const unsigned s = start[row + xa], e = start[row + xb + 1];
                        for (unsigned i = s; i < e; i += 4) {
                            // up to four independent 16-byte loads in flight
                            const float4 p0 = pts[i];
                            const float4 p1 = pts[min(i + 1, e - 1)];
                            const float4 p2 = pts[min(i + 2, e - 1)];
                            const float4 p3 = pts[min(i + 3, e - 1)];
                            const float d0 = dist2(qx, qy, qz, p0.x, p0.y, p0.z);
                            const float d1 = dist2(qx, qy, qz, p1.x, p1.y, p1.z);
                            const float d2 = dist2(qx, qy, qz, p2.x, p2.y, p2.z);
                            const float d3 = dist2(qx, qy, qz, p3.x, p3.y, p3.z);
                            const int i0 = __float_as_int(p0.w), i1 = __float_as_int(p1.w), i2 = __float_as_int(p2.w),
                                      i3 = __float_as_int(p3.w);
                            if (d0 < kth || (d0 == kth && i0 < kth_idx)) lex_insert<KCAP>(bd, bi, k, d0, i0, kth, kth_idx);
                            if (i + 1 < e && (d1 < kth || (d1 == kth && i1 < kth_idx)))
                                lex_insert<KCAP>(bd, bi, k, d1, i1, kth, kth_idx);
                            if (i + 2 < e && (d2 < kth || (d2 == kth && i2 < kth_idx)))
                                lex_insert<KCAP>(bd, bi, k, d2, i2, kth, kth_idx);
                            if (i + 3 < e && (d3 < kth || (d3 == kth && i3 < kth_idx)))
                                lex_insert<KCAP>(bd, bi, k, d3, i3, kth, kth_idx);
                            if (bound2 < kth) { kth = bound2; kth_idx = -1; }  // (a list that is not full yet says FLT_MAX)
                        }
                    }
                }
            }
            // distance from the query to the faces of the scanned block; faces on the grid boundary do not count
            float cov = FLT_MAX;
            if (cx - r > 0) cov = fminf(cov, qx - (g.ox + (cx - r) * g.h));
            if (cx + r < g.nx - 1) cov = fminf(cov, (g.ox + (cx + r + 1) * g.h) - qx);
            if (cy - r > 0) cov = fminf(cov, qy - (g.oy + (cy - r) * g.h));
            if (cy + r < g.ny - 1) cov = fminf(cov, (g.oy + (cy + r + 1) * g.h) - qy);
            if (cz - r > 0) cov = fminf(cov, qz - (g.oz + (cz - r) * g.h));
            if (cz + r < g.nz - 1) cov = fminf(cov, (g.oz + (cz + r + 1) * g.h) - qz);
            if (cov == FLT_MAX) break;  // the block is the whole grid
            cov = fmaxf(cov - g.eps, 0.0f);
            if (kth < cov * cov) break;  // strict: an unseen point at exactly the k-th distance could win a tie
        }
    }
    const size_t o = (size_t)qi * (size_t)k;
#pragma unroll
    for (int i = 0; i < KCAP; ++i)
        if (i < k) { d2_out[o + i] = bd[i]; idx_out[o + i] = bi[i]; }
}

// ---------------------------------------------------------------------------------------------------------------
// Self-kNN on the grid (every point of the cloud queries its own cloud: the covariance / normal preprocessing of
// the pipeline, pipeline/pointcloud_processing.hpp:144-156 and examples/example_registration.cpp:92-109).
// Work unit = up to 64 consecutive points of one x-row of cells, one wave per unit. The unit's candidate set is the
// 3 x 3 block of rows around it over the unit's x-span +- 1 cell: nine contiguous segments of the cell-ordered point
// array. Candidates are streamed 64 at a time: one coalesced 16-byte load per lane, parked in LDS, then every lane
// tests all of them with wave-uniform (broadcast) LDS reads — a brute force over a small local tile, with no
// per-lane pointer chasing and no divergence except in the sorted insertion. A query whose k-th distance reaches
// outside the scanned block (sparse regions) is appended to a to-do list and finished by the ring walk.
// The work units of every x-row (ceil(points in the row / 64); entry `rows` = 0, so the scan that follows needs no memset).
// (A hand-written single-workgroup scan of these ~16 k values took 16-45 us in three forms against 7.6 us for the library's.)
// The fullest cell (a statistic of the build: callers choose between the grid and the hierarchy by it, sp_grid_max_cell_points).
__global__ __launch_bounds__(kBlock) void cell_max_kernel(const unsigned* __restrict__ start, unsigned ncells,
                                                          unsigned* __restrict__ out) {
    unsigned m = 0;
    for (unsigned c = blockIdx.x * kBlock + threadIdx.x; c < ncells; c += gridDim.x * kBlock) m = max(m, start[c + 1] - start[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// Occupied cells and the fullest cell in one pass (out[0] += occupied, out[1] = max): what the density-adaptive build steers by.
__global__ __launch_bounds__(kBlock) void cell_stats_kernel(const unsigned* __restrict__ start, unsigned ncells,
                                                            unsigned* __restrict__ out) {
    unsigned m = 0, occ = 0;
    for (unsigned c = blockIdx.x * kBlock + threadIdx.x; c < ncells; c += gridDim.x * kBlock) {
        const unsigned k = start[c + 1] - start[c];
        m = max(m, k);
        occ += k ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
        occ += (unsigned)__shfl_xor((int)occ, o, 64);
    }
    if ((threadIdx.x & 63) == 0 && occ) {
        atomicAdd(out, occ);
        atomicMax(out + 1, m);
    }
}
__global__ void row_units_kernel(const unsigned* __restrict__ start, unsigned nx, unsigned rows,
                                 unsigned* __restrict__ units) {
    const unsigned r = blockIdx.x * kBlock + threadIdx.x;
    if (r <= rows) units[r] = r < rows ? (start[(size_t)(r + 1) * nx] - start[(size_t)r * nx] + 63u) / 64u : 0u;
}
struct TileOut {
    int32_t* knn_idx;  // [n][k] original order (may be null)
    float* knn_d2;
    float4* covs;      // [n][4] (may be null)
    float4* normals;   // [n] (may be null)
    unsigned* todo;    // queries (positions in grid order) the ring walk must finish
    unsigned* todo_count;
    unsigned pos_lo, pos_hi;  // only the queries at grid positions [pos_lo, pos_hi) are searched (sp_grid_self_knn_range)
};

// covariance::kernel::estimate (feature/covariance.hpp:16-47) over the neighbour list in ascending order, reading
// the neighbours from the cell-ordered copy (same values as points[idx], nearby in memory).
__device__ __forceinline__ void cov_from_list(const float4* __restrict__ pts, const int* bp, int k, float c[6],
                                              bool& identity) {
    float sx = 0.0f, sy = 0.0f, sz = 0.0f, oxx = 0.0f, oxy = 0.0f, oxz = 0.0f, oyy = 0.0f, oyz = 0.0f, ozz = 0.0f;
    unsigned cnt = 0;
    for (int j = 0; j < k; ++j) {
        const int pos = bp[j];
        if (pos < 0) continue;
        const float4 p = pts[pos];
        sx += p.x; sy += p.y; sz += p.z;
        oxx += p.x * p.x; oxy += p.x * p.y; oxz += p.x * p.z;
        oyy += p.y * p.y; oyz += p.y * p.z; ozz += p.z * p.z;
        ++cnt;
    }
    identity = cnt < 4;
    if (identity) return;
    const float inv = 1.0f / (float)cnt;
    const float mx = sx * inv, my = sy * inv, mz = sz * inv;
    const float cxy = oxy * inv - mx * my, cxz = oxz * inv - mx * mz, cyz = oyz * inv - my * mz;
    c[0] = oxx * inv - mx * mx; c[1] = (cxy + cxy) * 0.5f; c[2] = (cxz + cxz) * 0.5f;
    c[3] = oyy * inv - my * my; c[4] = (cyz + cyz) * 0.5f; c[5] = ozz * inv - mz * mz;
}

// What a lane that holds the sorted neighbour list of point q (distances, original indices, grid positions) writes:
// the list at q's ORIGINAL index, the covariance of the neighbours (covariance::kernel::estimate order) and / or the normal.
template <int KCAP>
__device__ __forceinline__ void self_knn_outputs(const float4* __restrict__ pts, const float4& q, const float (&bd)[KCAP],
                                                 const int (&bi)[KCAP], const int (&bp)[KCAP], int k, const TileOut& out) {
    const unsigned orig = __float_as_uint(q.w);
    if (out.knn_idx) {
        const size_t o = (size_t)orig * (size_t)k;
#pragma unroll
        for (int i = 0; i < KCAP; ++i)
            if (i < k) { out.knn_idx[o + i] = bi[i]; out.knn_d2[o + i] = bd[i]; }
    }
    if (out.covs || out.normals) {
        float c[6];
        bool identity;
        cov_from_list(pts, bp, k, c, identity);
        Mat3 C;
        if (identity) {
            C.m[0][0] = C.m[1][1] = C.m[2][2] = 1.0f;
            C.m[0][1] = C.m[0][2] = C.m[1][0] = C.m[1][2] = C.m[2][0] = C.m[2][1] = 0.0f;
        } else {
            C.m[0][0] = c[0]; C.m[0][1] = c[1]; C.m[0][2] = c[2];
            C.m[1][0] = c[1]; C.m[1][1] = c[3]; C.m[1][2] = c[4];
            C.m[2][0] = c[2]; C.m[2][1] = c[4]; C.m[2][2] = c[5];
        }
        if (out.covs) {
            float4* o4 = out.covs + 4 * (size_t)orig;
            o4[0] = make_float4(C.m[0][0], C.m[1][0], C.m[2][0], 0.0f);
            o4[1] = make_float4(C.m[0][1], C.m[1][1], C.m[2][1], 0.0f);
            o4[2] = make_float4(C.m[0][2], C.m[1][2], C.m[2][2], 0.0f);
            o4[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        if (out.normals) {  // covariance::kernel::extract_normal (covariance.hpp:49-65)
            float ev[3];
            Mat3 V;
            symmetric_eigen3(C, ev, V);
            const float nx_ = V.m[0][0], ny_ = V.m[1][0], nz_ = V.m[2][0];
            const float dd = chain3(nx_, q.x, ny_, q.y, nz_, q.z);
            out.normals[orig] = (dd <= 1.0f) ? make_float4(nx_, ny_, nz_, 0.0f) : make_float4(-nx_, -ny_, -nz_, 0.0f);
        }
    }
}

template <int KCAP>
__global__ __launch_bounds__(kWave) void grid_self_knn_tile_kernel(const float4* __restrict__ pts,
                                                                   const unsigned* __restrict__ start,
                                                                   const unsigned* __restrict__ unit_off, GridDesc g,
                                                                   int k, TileOut out) {
    __shared__ float4 tile[kWave];
    const unsigned unit = blockIdx.x;
    const unsigned rows = (unsigned)g.ny * g.nz;
    // row of this unit: last r with unit_off[r] <= unit (wave-uniform binary search)
    unsigned lo = 0, hi = rows;
    while (hi - lo > 1) {
        const unsigned mid = (lo + hi) >> 1;
        if (unit_off[mid] <= unit) lo = mid;
        else hi = mid;
    }
    const unsigned row = lo;
    const int ry = (int)(row % g.ny), rz = (int)(row / g.ny);
    const unsigned row_s = start[(size_t)row * g.nx], row_e = start[(size_t)(row + 1) * g.nx];
    const unsigned qs = row_s + (unit - unit_off[row]) * 64u;
    const unsigned qe = min(qs + 64u, row_e);
    if (qe <= out.pos_lo || qs >= out.pos_hi) return;  // wave-uniform: the unit lies outside the requested range
    const unsigned lane = threadIdx.x;
    const bool active = qs + lane < qe && qs + lane >= out.pos_lo && qs + lane < out.pos_hi;
    const float4 q = pts[min(qs + lane, qe - 1)];
    // x-span of the unit's queries (cells are ascending along the row)
    const float4 qf = pts[qs], ql = pts[qe - 1];
    const int cxa = cell_coord(qf.x, g.ox, g.inv_h, g.nx), cxb = cell_coord(ql.x, g.ox, g.inv_h, g.nx);
    const int xa = max(cxa - 1, 0), xb = min(cxb + 1, g.nx - 1);
    const int ya = max(ry - 1, 0), yb = min(ry + 1, g.ny - 1), za = max(rz - 1, 0), zb = min(rz + 1, g.nz - 1);

    float bd[KCAP];
    int bi[KCAP], bp[KCAP];
#pragma unroll
    for (int i = 0; i < KCAP; ++i) { bd[i] = FLT_MAX; bi[i] = -1; bp[i] = -1; }
    float kth = FLT_MAX;
    int kth_idx = -1;

    // the queries' own row first, then the four rows sharing a face with it, then the corner rows: the k-th best is tight
    // before the far rows arrive and they rarely take the insertion branch (which costs the whole wave when any lane takes it)
    for (int ring = 0; ring < 3; ++ring)
      for (int z = za; z <= zb; ++z)
        for (int y = ya; y <= yb; ++y) {
            if (abs(z - rz) + abs(y - ry) != ring) continue;
            const unsigned rr = ((unsigned)z * g.ny + y) * g.nx;
            const unsigned s = start[rr + xa], e = start[rr + xb + 1];
            for (unsigned base = s; base < e; base += kWave) {
                const unsigned cnt = min((unsigned)kWave, e - base);
                __syncthreads();
                if (lane < cnt) tile[lane] = pts[base + lane];
                __syncthreads();
                for (unsigned c = 0; c < cnt; ++c) {
                    const float4 p = tile[c];
                    const float d = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
                    const int pi = __float_as_int(p.w);
                    if (d < kth || (d == kth && pi < kth_idx)) {
                        // sorted insertion by (distance, original index); positions ride along
                        float cd = d;
                        int ci = pi, cp = (int)(base + c);
                        bool shifting = false;
#pragma unroll
                        for (int i = 0; i < KCAP; ++i) {
                            if (i < k) {
                                const bool sw = shifting || cd < bd[i] || (cd == bd[i] && ci < bi[i]);
                                const float td = bd[i];
                                const int ti = bi[i], tp = bp[i];
                                const float nd = sw ? cd : td;
                                const int ni = sw ? ci : ti;
                                bd[i] = nd; bi[i] = ni; bp[i] = sw ? cp : tp;
                                cd = sw ? td : cd; ci = sw ? ti : ci; cp = sw ? tp : cp;
                                shifting = sw;
                                kth = nd; kth_idx = ni;
                            }
                        }
                    }
                }
            }
        }
    if (!active) return;
    // is the k-th neighbour provably inside the scanned block?
    float cov = FLT_MAX;
    if (xa > 0) cov = fminf(cov, q.x - (g.ox + xa * g.h));
    if (xb < g.nx - 1) cov = fminf(cov, (g.ox + (xb + 1) * g.h) - q.x);
    if (ya > 0) cov = fminf(cov, q.y - (g.oy + ya * g.h));
    if (yb < g.ny - 1) cov = fminf(cov, (g.oy + (yb + 1) * g.h) - q.y);
    if (za > 0) cov = fminf(cov, q.z - (g.oz + za * g.h));
    if (zb < g.nz - 1) cov = fminf(cov, (g.oz + (zb + 1) * g.h) - q.z);
    bool exact = true;
    if (cov != FLT_MAX) {
        cov = fmaxf(cov - g.eps, 0.0f);
        exact = kth < cov * cov;
    }
    if (!exact) {
        const unsigned slot = atomicAdd(out.todo_count, 1u);
        out.todo[slot] = qs + lane;
        return;
    }
    self_knn_outputs<KCAP>(pts, q, bd, bi, bp, k, out);
}

// ---------------------------------------------------------------------------------------------------------------
// Lane per query for LONG lists (7 <= k <= 24): bound, collect, sort — no insertion anywhere.
// The wave-cooperative kernel below spends ~630 vector instructions on ONE query (a 64-lane sorting network and ballot-ranked
// insertions); a lane-per-query kernel does 64 queries per instruction but could not keep a sorted 20-entry list cheaply
// (every insertion is a 20-step select chain that the whole wave walks). Here nothing is kept sorted while scanning:
//   1. the lane's candidates — its 3x3x3 cells = nine x-segments of the cell-ordered array, up to kSelCand points — are
//      fetched once; their squared distances stay in the lane's LDS column as 7-bit keys (192 bytes per lane; a register
//      array, fully unrolled, made the compiler spill 700-900 registers; 16-bit keys allowed one wave per SIMD only and the
//      kernel was latency-bound at 1.45 ms per 1 M points);
//   2. a threshold t with k <= #{key <= t} <= 32 is found on those keys by the t^1.5 law of points in a ball, two or three
//      counting passes (the key scale comes from the lane's own candidate count);
//   3. the <= 32 candidates within t go to a per-lane list in LDS, come back as exact 64-bit (distance, index) keys and each
//      is RANKED among the lane's keys (a branch-free count): lexicographic order = brute force's tie rule; ranks < k are
//      the list, written straight to their places;
//   4. exactness as in the tile kernel: the k-th distance must be inside the block's coverage, otherwise (or when a segment
//      is longer than its slots, or no threshold is found) the query goes to the to-do list of the ring-walk kernel.
// Same lists, distances and covariances as the other kernels, bit for bit.
constexpr int kSelCand = 192;  // candidates of one query (27 cells: 162 at the default density, sigma 13)
constexpr int kSelList = 32;   // entries that may pass the threshold

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}

// Phases 1 - 3 and the exactness test of the lane-per-query selection (see above) for the query q in cell (cx, ry, rz):
// its k nearest of the 27 cells go to list_idx / list_d2[0 .. k) (when not null), their grid positions in ascending order to
// the lane's column of l_pos. Returns true when the lists are proven exact, false when the query has to be searched again by
// a kernel without the 27-cell limit. `qpos`: any valid position (idle slots read it).
__device__ __forceinline__ bool self_knn_select_core(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                                     const GridDesc& g, int k, const float4 q, int cx, int ry, int rz,
                                                     bool active, unsigned qpos, unsigned long long (*l_key8)[kWave],
                                                     int (*l_pos)[kWave], int32_t* list_idx, float* list_d2) {
    unsigned char* const l_key = reinterpret_cast<unsigned char*>(&l_key8[0][0]);
    const int xa = max(cx - 1, 0), xb = min(cx + 1, g.nx - 1);
    const int ya = max(ry - 1, 0), yb = min(ry + 1, g.ny - 1), za = max(rz - 1, 0), zb = min(rz + 1, g.nz - 1);
    const unsigned lane = threadIdx.x;
    auto key_slot = [&](unsigned j) -> unsigned char& { return l_key[((j >> 3) * kWave + lane) * 8 + (j & 7u)]; };

    // 1. the nine segments; the distance of every point in them as a 7-BIT key in the lane's LDS column:
    //        key = floor(d * 64 / t0), saturating at 127,   t0 = the squared radius of the ball that holds 26 points at the
    //        density of the lane's own 27 cells (so the threshold sought below sits near key 64, one key step = 1.6 %).
    // A monotone key is all the threshold needs: if a point outside the collected set {key <= t} were among the k nearest,
    // the >= k collected points, whose keys are smaller, would all be strictly nearer.
    unsigned sbeg[9], slen[9], cbase[9];
    unsigned total = 0, maxlen = 0;
#pragma unroll
    for (int s9 = 0; s9 < 9; ++s9) {
        const int y = ry + (s9 % 3) - 1, z = rz + (s9 / 3) - 1;
        unsigned b = 0, e = 0;
        if (y >= 0 && y < g.ny && z >= 0 && z < g.nz) {  // (wave-uniform for the self-kNN, per lane for external queries)
            const unsigned rr = ((unsigned)z * g.ny + y) * g.nx;
            b = start[rr + xa];
            e = start[rr + xb + 1];
        }
        sbeg[s9] = b;
        slen[s9] = e - b;
        total += e - b;
    }
    bool fallback = !active || total > (unsigned)kSelCand;
    if (fallback) total = 0;
    {
        unsigned acc = 0;
#pragma unroll
        for (int s9 = 0; s9 < 9; ++s9) {
            if (fallback) slen[s9] = 0u;
            cbase[s9] = acc;
            acc += slen[s9];
            maxlen = max(maxlen, slen[s9]);
        }
    }
    const float cells = (float)((xb - xa + 1) * (yb - ya + 1) * (zb - za + 1));
    const float t0 = g.h * g.h * __powf(26.0f * cells / (4.18879f * fmaxf((float)total, 1.0f)), 0.6666667f);
    const float key_scale = 64.0f / fmaxf(t0, FLT_MIN);
    // All nine segments advance together, four points each per trip: 36 independent loads in flight and ~8 trips, where a
    // loop per segment made 27 dependent round trips.
    const unsigned trips = wave_max_u32(maxlen);
    for (unsigned o = 0; o < trips; o += 4) {
        float4 p[4][9];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int s9 = 0; s9 < 9; ++s9) p[u][s9] = pts[(o + u < slen[s9]) ? sbeg[s9] + o + u : qpos];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int s9 = 0; s9 < 9; ++s9)
                if (o + u < slen[s9]) {
                    const float kd = dist2(q.x, q.y, q.z, p[u][s9].x, p[u][s9].y, p[u][s9].z) * key_scale;
                    key_slot(cbase[s9] + o + u) = (unsigned char)(kd < 127.0f ? (unsigned)kd : 127u);  // (NaN -> 127)
                }
    }
    const unsigned groups = (wave_max_u32(total) + 7u) >> 3;  // the counting loops run to the longest list of the wave
    for (unsigned j = total; j < 8u * groups; ++j) key_slot(j) = 0xffu;  // (high bit set: never counted)
    // 2. a threshold key t < 127 with k .. kSelList candidates at or below it. The count grows like t^1.5 (points in a ball),
    // so a miss is corrected by that law — two or three counting passes for the whole wave — inside a bracket that falls back
    // to bisection. Four keys are counted at once (the classic "bytes less than n" mask + a population count).
    unsigned t = 64u;
    unsigned t_lo = 0u, t_hi = 127u;  // keys below t_lo are known to count < k, keys from t_hi on > kSelList (127 = saturated)
    bool found = false;
    for (int it = 0; it < 10; ++it) {
        const unsigned cmpc = 0x01010101u * (127u + t + 1u);  // (t + 1 <= 127)
        unsigned cnt = 0;
#pragma unroll 4
        for (unsigned gq = 0; gq < groups; ++gq) {
            const unsigned long long w = l_key8[gq][lane];
            const unsigned w0 = (unsigned)w, w1 = (unsigned)(w >> 32);
            cnt += __builtin_popcount((cmpc - (w0 & 0x7f7f7f7fu)) & ~w0 & 0x80808080u);
            cnt += __builtin_popcount((cmpc - (w1 & 0x7f7f7f7fu)) & ~w1 & 0x80808080u);
        }
        if (!found && !fallback) {
            if (cnt >= (unsigned)k && cnt <= (unsigned)kSelList) {
                found = true;
            } else {
                if (cnt < (unsigned)k) t_lo = t + 1u;
                else t_hi = t;
                if (t_lo >= t_hi) {
                    fallback = true;  // the keys cannot separate k .. kSelList points (ties, or sparser than the estimate)
                } else {
                    const float aim = 0.5f * (float)(k + kSelList);
                    const float r = __powf(aim / fmaxf((float)cnt, 0.5f), 0.6666667f);
                    unsigned tn = (unsigned)(((float)t + 0.5f) * fminf(fmaxf(r, 0.3f), 3.0f));
                    if (tn < t_lo || tn >= t_hi) tn = (t_lo + t_hi) >> 1;  // outside the bracket: bisect
                    t = tn;
                }
            }
        }
        if (__ballot(!found && !fallback) == 0ull) break;
    }
    fallback = fallback || !found;
    // 3. the candidates at or below the threshold: their numbers in the lane's column, to the lane's list ...
    unsigned c = 0;
#pragma unroll 2
    for (unsigned gq = 0; gq < groups; ++gq) {
        const unsigned long long w = l_key8[gq][lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned kk = (unsigned)(w >> (8 * u)) & 0xffu;
            if (kk <= t && !fallback) {
                l_pos[c][lane] = (int)(8u * gq + u);
                ++c;
            }
        }
    }
    // ... each number back to its position (which segment it falls in), exact distance and original index as a 64-bit key,
    // on registers
    const unsigned cmax = wave_max_u32(c);
    unsigned long long key[kSelList];
    int pos[kSelList];
#pragma unroll
    for (int i = 0; i < kSelList; ++i) {
        key[i] = kNoCand;
        pos[i] = -1;
        if ((unsigned)i < cmax) {  // (wave-uniform)
            const bool have = (unsigned)i < c;
            const unsigned j = have ? (unsigned)l_pos[i][lane] : 0u;
            unsigned off = sbeg[0];  // position = number + (start of its segment - numbers before the segment)
#pragma unroll
            for (int s9 = 1; s9 < 9; ++s9) off = j >= cbase[s9] ? sbeg[s9] - cbase[s9] : off;
            const int pp = have ? (int)(j + off) : (int)qpos;
            const float4 p = pts[pp];
            if (have) {
                key[i] = cand_key(dist2(q.x, q.y, q.z, p.x, p.y, p.z), __float_as_int(p.w));
                pos[i] = pp;
            }
        }
    }
    // rank of every entry among the lane's entries (keys are distinct: the index is part of them) = its place in the list,
    // counted on registers. (A 32-input sorting network on registers was the first form: 240 compare-exchanges of three
    // registers each, which the register allocator answered with 560 spills; counting only reads the keys.)
    int (*l_spos)[kWave] = l_pos;  // the list region, rewritten in sorted order (every position is on registers by now)
    const bool write_lists = list_idx != nullptr && active && !fallback;
    float kth = FLT_MAX;
#pragma unroll
    for (int i = 0; i < kSelList; ++i) {
        unsigned rank = 0;
#pragma unroll
        for (int j = 0; j < kSelList; ++j) rank += key[j] < key[i] ? 1u : 0u;
        if ((unsigned)i < c) {
            l_spos[rank][lane] = pos[i];
            if (rank == (unsigned)k - 1u) kth = key_d2(key[i]);
            // (written before exactness is known: a query that turns out unproven is rewritten by the to-do kernel, which
            // runs after this one)
            if (write_lists && rank < (unsigned)k) {
                list_idx[rank] = key_idx(key[i]);
                list_d2[rank] = key_d2(key[i]);
            }
        }
    }
    if (!active) return true;
    // 4. is the k-th neighbour provably inside the scanned block?
    if (!fallback) {
        float cov = FLT_MAX;
        if (xa > 0) cov = fminf(cov, q.x - (g.ox + xa * g.h));
        if (xb < g.nx - 1) cov = fminf(cov, (g.ox + (xb + 1) * g.h) - q.x);
        if (ya > 0) cov = fminf(cov, q.y - (g.oy + ya * g.h));
        if (yb < g.ny - 1) cov = fminf(cov, (g.oy + (yb + 1) * g.h) - q.y);
        if (za > 0) cov = fminf(cov, q.z - (g.oz + za * g.h));
        if (zb < g.nz - 1) cov = fminf(cov, (g.oz + (zb + 1) * g.h) - q.z);
        if (cov != FLT_MAX) {
            cov = fmaxf(cov - g.eps, 0.0f);
            fallback = !(kth < cov * cov);
        }
    }
    return !fallback;
}

__global__ __launch_bounds__(kWave) void grid_self_knn_select_kernel(const float4* __restrict__ pts,
                                                                     const unsigned* __restrict__ start,
                                                                     const unsigned* __restrict__ unit_off, GridDesc g,
                                                                     int k, TileOut out) {
    // the lane's keys, one byte each, in groups of eight (one ds_read_b64 per eight candidates), [group][lane]: conflict-free
    __shared__ unsigned long long l_key8[kSelCand / 8][kWave];  // 12 KB
    __shared__ int l_pos[kSelList][kWave];                      //  8 KB  (20 KB per wave: eight waves per CU)
    const unsigned unit = blockIdx.x;
    const unsigned rows = (unsigned)g.ny * g.nz;
    unsigned lo = 0, hi = rows;
    while (hi - lo > 1) {  // row of this unit: last r with unit_off[r] <= unit (wave-uniform)
        const unsigned mid = (lo + hi) >> 1;
        if (unit_off[mid] <= unit) lo = mid;
        else hi = mid;
    }
    const unsigned row = lo;
    const int ry = (int)(row % g.ny), rz = (int)(row / g.ny);
    const unsigned row_s = start[(size_t)row * g.nx], row_e = start[(size_t)(row + 1) * g.nx];
    const unsigned qs = row_s + (unit - unit_off[row]) * 64u;
    const unsigned qe = min(qs + 64u, row_e);
    if (qe <= out.pos_lo || qs >= out.pos_hi) return;  // wave-uniform: the unit lies outside the requested range
    const unsigned lane = threadIdx.x;
    const unsigned qpos = min(qs + lane, qe - 1);
    const bool active = qs + lane < qe && qs + lane >= out.pos_lo && qs + lane < out.pos_hi;
    const float4 q = pts[qpos];
    const int cx = cell_coord(q.x, g.ox, g.inv_h, g.nx);
    const unsigned orig = __float_as_uint(q.w);
    // (the lists are written before exactness is known: a query that turns out unproven is rewritten by the list kernel,
    // which runs after this one)
    const bool proven = self_knn_select_core(pts, start, g, k, q, cx, ry, rz, active, qpos, l_key8, l_pos,
                                             out.knn_idx ? out.knn_idx + (size_t)orig * (size_t)k : nullptr,
                                             out.knn_idx ? out.knn_d2 + (size_t)orig * (size_t)k : nullptr);
    if (!active) return;
    int (*l_spos)[kWave] = l_pos;
    const bool fallback = !proven;
    if (fallback) {
        const unsigned slot = atomicAdd(out.todo_count, 1u);
        out.todo[slot] = qs + lane;
        return;
    }
    if (out.covs || out.normals) {
        // covariance::kernel::estimate over the list in ascending order (covariance.hpp:16-47), k >= 7 valid neighbours here
        float sx = 0.0f, sy = 0.0f, sz = 0.0f, oxx = 0.0f, oxy = 0.0f, oxz = 0.0f, oyy = 0.0f, oyz = 0.0f, ozz = 0.0f;
#pragma unroll 12
        for (int j = 0; j < k; ++j) {
            const float4 p = pts[l_spos[j][lane]];
            sx += p.x; sy += p.y; sz += p.z;
            oxx += p.x * p.x; oxy += p.x * p.y; oxz += p.x * p.z;
            oyy += p.y * p.y; oyz += p.y * p.z; ozz += p.z * p.z;
        }
        const float inv = 1.0f / (float)k;
        const float mx = sx * inv, my = sy * inv, mz = sz * inv;
        const float cxy = oxy * inv - mx * my, cxz = oxz * inv - mx * mz, cyz = oyz * inv - my * mz;
        Mat3 C;
        C.m[0][0] = oxx * inv - mx * mx; C.m[0][1] = (cxy + cxy) * 0.5f; C.m[0][2] = (cxz + cxz) * 0.5f;
        C.m[1][1] = oyy * inv - my * my; C.m[1][2] = (cyz + cyz) * 0.5f; C.m[2][2] = ozz * inv - mz * mz;
        C.m[1][0] = C.m[0][1]; C.m[2][0] = C.m[0][2]; C.m[2][1] = C.m[1][2];
        if (out.covs) {
            float4* o4 = out.covs + 4 * (size_t)orig;
            o4[0] = make_float4(C.m[0][0], C.m[1][0], C.m[2][0], 0.0f);
            o4[1] = make_float4(C.m[0][1], C.m[1][1], C.m[2][1], 0.0f);
            o4[2] = make_float4(C.m[0][2], C.m[1][2], C.m[2][2], 0.0f);
            o4[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        if (out.normals) {  // covariance::kernel::extract_normal (covariance.hpp:49-65)
            float ev[3];
            Mat3 V;
            symmetric_eigen3(C, ev, V);
            const float nx_ = V.m[0][0], ny_ = V.m[1][0], nz_ = V.m[2][0];
            const float dd = chain3(nx_, q.x, ny_, q.y, nz_, q.z);
            out.normals[orig] = (dd <= 1.0f) ? make_float4(nx_, ny_, nz_, 0.0f) : make_float4(-nx_, -ny_, -nz_, 0.0f);
        }
    }
}

// The same selection for EXTERNAL queries (KNNBase::knn_search_async on a GridKNN, 8 <= k <= 24): a lane per query in the
// caller's order, its cell from its (transformed) coordinates; what cannot be proven inside the 27 cells (and non-finite
// queries) goes to the ring-walk kernel through a list. 1 M queries against 1 M points, k = 20: 7.4 ms with the ring walk
// alone (a 20-step insertion per candidate in every lane).
__global__ __launch_bounds__(kWave) void grid_search_select_kernel(const float4* __restrict__ pts,
                                                                   const unsigned* __restrict__ start, GridDesc g,
                                                                   const float4* __restrict__ queries, unsigned nq, int k,
                                                                   Mat4Arg T_val, const float* __restrict__ T_dev,
                                                                   int32_t* __restrict__ idx_out, float* __restrict__ d2_out,
                                                                   unsigned* __restrict__ todo, unsigned* __restrict__ todo_count,
                                                                   const unsigned* __restrict__ order = nullptr) {
    __shared__ unsigned long long l_key8[kSelCand / 8][kWave];
    __shared__ int l_pos[kSelList][kWave];
    const unsigned gi = blockIdx.x * kWave + threadIdx.x;
    // (order: the queries sorted by cell, grid_query_cell_kernel — an idle lane of the last wave is number nq, as without it)
    const unsigned qi = (order && gi < nq) ? order[gi] : gi;
    const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
    const float4 q4 = queries[min(qi, nq - 1u)];
    float4 q;
    transform_point(T, q4.x, q4.y, q4.z, q.x, q.y, q.z);
    q.w = 0.0f;
    const bool finite_q = isfinite(q.x) && isfinite(q.y) && isfinite(q.z);
    const bool active = qi < nq && finite_q && g.n > 0;
    const int cx = cell_coord(q.x, g.ox, g.inv_h, g.nx), cy = cell_coord(q.y, g.oy, g.inv_h, g.ny),
              cz = cell_coord(q.z, g.oz, g.inv_h, g.nz);
    const bool proven = self_knn_select_core(pts, start, g, k, q, cx, cy, cz, active, 0u, l_key8, l_pos,
                                             idx_out + (size_t)min(qi, nq - 1u) * (size_t)k,
                                             d2_out + (size_t)min(qi, nq - 1u) * (size_t)k);
    if (qi < nq && !active) {  // a non-finite query (or an empty grid): the empty list (knn/result.hpp:21-27)
        for (int i = 0; i < k; ++i) { idx_out[(size_t)qi * k + i] = -1; d2_out[(size_t)qi * k + i] = FLT_MAX; }
    } else if (qi < nq && !proven) {
        todo[atomicAdd(todo_count, 1u)] = qi;
    }
}

// Lane per point, for short lists (k <= 10): the ring walk of grid_search_kernel over the grid's own cell-ordered points
// (neighbouring lanes walk neighbouring cells), positions carried along for the fused covariance / normal. Exact by
// construction (the walk ends when the k-th neighbour is proven), so there is no to-do list. On 1M points: k = 10 in
// 0.6 ms against 1.1 ms for the wave-cooperative kernel, k = 6 in 0.35 ms against 0.9 ms for the tile kernel.
template <int KCAP>
__global__ __launch_bounds__(kBlock) void grid_self_knn_lane_kernel(const float4* __restrict__ pts,
                                                                    const unsigned* __restrict__ start, GridDesc g, int k,
                                                                    TileOut out) {
    const unsigned qi = out.pos_lo + blockIdx.x * kBlock + threadIdx.x;
    if (qi >= out.pos_hi) return;
    const float4 q = pts[qi];
    const float qx = q.x, qy = q.y, qz = q.z;
    float bd[KCAP];
    int bi[KCAP], bp[KCAP];
#pragma unroll
    for (int i = 0; i < KCAP; ++i) { bd[i] = FLT_MAX; bi[i] = -1; bp[i] = -1; }
    float kth = FLT_MAX;
    int kth_idx = -1;
    auto consider = [&](float d, int pi, int pos) {
        if (!(d < kth || (d == kth && pi < kth_idx))) return;
        float cd = d;
        int ci = pi, cp = pos;
        bool shifting = false;
#pragma unroll
        for (int i = 0; i < KCAP; ++i) {
            if (i < k) {
                const bool sw = shifting || cd < bd[i] || (cd == bd[i] && ci < bi[i]);
                const float td = bd[i];
                const int ti = bi[i], tp = bp[i];
                bd[i] = sw ? cd : td; bi[i] = sw ? ci : ti; bp[i] = sw ? cp : tp;
                cd = sw ? td : cd; ci = sw ? ti : ci; cp = sw ? tp : cp;
                shifting = sw;
                kth = bd[i]; kth_idx = bi[i];
            }
        }
    };
    if (isfinite(qx) && isfinite(qy) && isfinite(qz)) {
        const int cx = cell_coord(qx, g.ox, g.inv_h, g.nx), cy = cell_coord(qy, g.oy, g.inv_h, g.ny),
                  cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
        const int rmax = max(max(g.nx, g.ny), g.nz);
        for (int r = 0; r <= rmax; ++r) {
            const int z0 = max(cz - r, 0), z1 = min(cz + r, g.nz - 1);
            const int y0 = max(cy - r, 0), y1 = min(cy + r, g.ny - 1);
            const int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
            for (int z = z0; z <= z1; ++z) {
                const float dz2 = gap2(qz, g.oz + z * g.h, g.oz + (z + 1) * g.h, g.eps);
                if (dz2 > kth) continue;
                for (int y = y0; y <= y1; ++y) {
                    const float dyz2 = dz2 + gap2(qy, g.oy + y * g.h, g.oy + (y + 1) * g.h, g.eps);
                    if (dyz2 > kth) continue;
                    const bool shell_row = (r == 0) || (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
                    const unsigned row = ((unsigned)z * g.ny + y) * g.nx;
                    const int nseg = shell_row ? 1 : 2;  // a shell row over its whole x-range, an interior row at its two end cells
                    for (int sgi = 0; sgi < nseg; ++sgi) {
                        int xa, xb;
                        if (shell_row) { xa = x0; xb = x1; }
                        else if (sgi == 0) { xa = cx - r; xb = cx - r; if (xa < 0) continue; }
                        else { xa = cx + r; xb = cx + r; if (xb > g.nx - 1) continue; }
                        if (dyz2 + gap2(qx, g.ox + xa * g.h, g.ox + (xb + 1) * g.h, g.eps) > kth) continue;
                        const unsigned s = start[row + xa], e = start[row + xb + 1];
                        for (unsigned i = s; i < e; i += 4) {  // up to four independent 16-byte loads in flight
                            const float4 p0 = pts[i], p1 = pts[min(i + 1, e - 1)], p2 = pts[min(i + 2, e - 1)],
                                         p3 = pts[min(i + 3, e - 1)];
                            const float d0 = dist2(qx, qy, qz, p0.x, p0.y, p0.z), d1 = dist2(qx, qy, qz, p1.x, p1.y, p1.z),
                                        d2 = dist2(qx, qy, qz, p2.x, p2.y, p2.z), d3 = dist2(qx, qy, qz, p3.x, p3.y, p3.z);
                            consider(d0, __float_as_int(p0.w), (int)i);
                            if (i + 1 < e) consider(d1, __float_as_int(p1.w), (int)i + 1);
                            if (i + 2 < e) consider(d2, __float_as_int(p2.w), (int)i + 2);
                            if (i + 3 < e) consider(d3, __float_as_int(p3.w), (int)i + 3);
                        }
                    }
                }
            }
            float cov = FLT_MAX;  // distance to the faces of the scanned block; faces on the grid boundary do not count
            if (cx - r > 0) cov = fminf(cov, qx - (g.ox + (cx - r) * g.h));
            if (cx + r < g.nx - 1) cov = fminf(cov, (g.ox + (cx + r + 1) * g.h) - qx);
            if (cy - r > 0) cov = fminf(cov, qy - (g.oy + (cy - r) * g.h));
            if (cy + r < g.ny - 1) cov = fminf(cov, (g.oy + (cy + r + 1) * g.h) - qy);
            if (cz - r > 0) cov = fminf(cov, qz - (g.oz + (cz - r) * g.h));
            if (cz + r < g.nz - 1) cov = fminf(cov, (g.oz + (cz + r + 1) * g.h) - qz);
            if (cov == FLT_MAX) break;
            cov = fmaxf(cov - g.eps, 0.0f);
            if (kth < cov * cov) break;  // strict: an unseen point at exactly the k-th distance could win a tie
        }
    }
    self_knn_outputs<KCAP>(pts, q, bd, bi, bp, k, out);
}


// ---------------------------------------------------------------------------------------------------------------
// Wave-cooperative self-kNN (k <= 32): the 64 lanes of a wave work on ONE query at a time.
//   * candidates = the 3x3x3 block of cells around the query's cell: nine contiguous segments of the cell-ordered
//     point array, flattened into one list and consumed 64 at a time — lane l evaluates candidate (base + l): coalesced
//     16-byte loads, no LDS, no per-lane lists;
//   * the sorted top-k lives ONE ENTRY PER LANE (lane i holds the i-th best (d, idx, pos)); the first 64 candidates are
//     sorted with a wave bitonic network, later candidates that beat the k-th entry (found by ballot) are inserted with
//     a rank computed by ballot + a one-lane shift: O(1) wave instructions per insertion, whatever k is;
//   * queries are walked in cell order (work unit = 64 consecutive points of one x-row), so consecutive queries reuse
//     the same nine segments out of L1/L2;
//   * results are ordered by (distance, original index): bit-identical to brute force; the covariance (optional) is
//     accumulated over the list in ascending order exactly as covariance::kernel::estimate does.
// A candidate is ordered by (squared distance, index): for non-negative floats the bit pattern orders like the value, so
// the pair packs into one 64-bit key and "nearer, ties to the lower index" is a single unsigned compare.

// nq <= 64 queries of x-row (ry, rz): lane j holds query j in `myq` (coordinates + the bits of its output row); `qs`: the grid
// position of query 0 (self-kNN: tested against the requested range).
__device__ __forceinline__ void self_knn_wave_queries(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                                      const GridDesc& g, int k, const TileOut& out, float (*sm_terms)[33],
                                                      unsigned qs, unsigned nq, int ry, int rz, const float4 myq) {
    const unsigned lane = threadIdx.x;
    const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);

    int cur_cx = -1, cur_variant = -1;
    unsigned seg_s[9], seg_c[10];  // start of each segment, running candidate counts (seg_c[9] = total)
    int xa = 0, xb = 0;
    for (unsigned j = 0; j < nq; ++j) {
        if (qs + j < out.pos_lo || qs + j >= out.pos_hi) continue;  // wave-uniform
        // the query, made wave-uniform
        const float qx = bcast_f(myq.x, (int)j), qy = bcast_f(myq.y, (int)j), qz = bcast_f(myq.z, (int)j);
        const unsigned qorig = (unsigned)bcast_i(__float_as_int(myq.w), (int)j);
        const int cx = cell_coord(qx, g.ox, g.inv_h, g.nx);
        // which half of its cell the query is in, per axis: the rows on that side come first
        const int sy = ((qy - g.oy) * g.inv_h - (float)ry < 0.5f) ? -1 : 1, sz = ((qz - g.oz) * g.inv_h - (float)rz < 0.5f) ? -1 : 1;
        const int variant = (sy > 0 ? 1 : 0) | (sz > 0 ? 2 : 0);
        if (cx != cur_cx || variant != cur_variant) {  // wave-uniform: new cell or new side, new segments
            cur_cx = cx;
            cur_variant = variant;
            xa = max(cx - 1, 0);
            xb = min(cx + 1, g.nx - 1);
            unsigned c = 0;
            int r = 0;
            // centre row, the two face rows on the query's side, the corner row between them, the two far face rows, the
            // other corners: the nearest candidates arrive early, the k-th best tightens fast, later chunks insert little
            constexpr int kDy[9] = {0, 1, 0, 1, -1, 0, 1, -1, -1}, kDz[9] = {0, 0, 1, 1, 0, -1, -1, 1, -1};
#pragma unroll
            for (int o = 0; o < 9; ++o) {
                const int y = ry + kDy[o] * sy, z = rz + kDz[o] * sz;
                unsigned s0 = 0, e0 = 0;
                if (y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
                    const unsigned rr = ((unsigned)z * g.ny + y) * g.nx;
                    s0 = start[rr + xa];
                    e0 = start[rr + xb + 1];
                }
                seg_s[r] = s0;
                seg_c[r] = c;
                c += e0 - s0;
                ++r;
            }
            for (; r < 9; ++r) { seg_s[r] = 0; seg_c[r] = c; }
            seg_c[9] = c;
        }
        const unsigned total = seg_c[9];
        Cand best;  // lane i holds the i-th best
        best.key = kNoCand; best.pos = -1;
        unsigned long long kth = kNoCand;  // key of the current k-th best (wave-uniform)
        auto fetch = [&](unsigned base, float4& p, unsigned& pos, bool& valid) {
            const unsigned f = base + lane;
            valid = f < total;
            const unsigned ff = valid ? f : total - 1;
            unsigned off = seg_s[0];  // position of candidate ff = ff + (start of its segment - candidates before the segment)
#pragma unroll
            for (int r = 1; r < 9; ++r) off = (ff >= seg_c[r]) ? seg_s[r] - seg_c[r] : off;  // (uniform difference: scalar)
            pos = ff + off;
            p = pts[pos];
        };
        // Up to 192 candidates (the usual 27 cells at 6 points per cell hold 162): all three chunks are evaluated first and
        // the network sorts each lane's NEAREST candidate of its three. The k-th of those 64 minima is already within a few
        // places of the final k-th neighbour, so the two passes over the lanes' other candidates insert ~4 of them instead of
        // the ~14 that beat the k-th of an arbitrary first chunk: a third fewer wave instructions per query, same lists.
        bool selected = false;
        if (total <= 192u && total > 0u) {
            Cand c3[3];
            {
                float4 p3[3];
                unsigned pos3[3];
                bool valid3[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if ((unsigned)r * 64u < total) fetch(r * 64u, p3[r], pos3[r], valid3[r]);
                    else { valid3[r] = false; pos3[r] = 0; p3[r] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    c3[r].key = valid3[r] ? cand_key(dist2(qx, qy, qz, p3[r].x, p3[r].y, p3[r].z), __float_as_int(p3[r].w)) : kNoCand;
                    c3[r].pos = (int)pos3[r];
                }
            }
            // lane minimum into c3[0] (two compare-exchanges by key)
            if (c3[1].key < c3[0].key) { const Cand t = c3[0]; c3[0] = c3[1]; c3[1] = t; }
            if (c3[2].key < c3[0].key) { const Cand t = c3[0]; c3[0] = c3[2]; c3[2] = t; }
            best = bitonic_sort64(c3[0], lane);
            kth = bcast_k(best.key, k - 1);
            insert_candidates(c3[1], best, kth, k, kmask, lane);
            insert_candidates(c3[2], best, kth, k, kmask, lane);
            selected = true;
        }
        float4 p_next;
        unsigned pos_next;
        bool valid_next;
        // (total == 0 cannot happen for the cloud's own points — a query is in its own cell — but an external query can sit in an
        // empty neighbourhood: nothing to fetch, the rings below find its neighbours)
        p_next = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        pos_next = 0u;
        valid_next = false;
        if (!selected && total > 0u) fetch(0, p_next, pos_next, valid_next);
        for (unsigned base = 0; !selected && base < total; base += 64) {
            const float4 p = p_next;
            const unsigned pos = pos_next;
            const bool valid = valid_next;
            if (base + 64 < total) fetch(base + 64, p_next, pos_next, valid_next);  // in flight while this chunk is merged
            Cand c;
            c.key = valid ? cand_key(dist2(qx, qy, qz, p.x, p.y, p.z), __float_as_int(p.w)) : kNoCand;
            c.pos = (int)pos;
            if (base == 0) {
                best = bitonic_sort64(c, lane);
                kth = bcast_k(best.key, k - 1);
            } else {
                insert_candidates(c, best, kth, k, kmask, lane);
            }
        }
        // Exactness: is the k-th neighbour inside the scanned block? If not (sparse neighbourhoods), keep adding rings
        // of cells — still wave-cooperatively, one shell segment at a time — until it is.
        const int cy = ry, cz = rz;
        const int rmax = max(max(g.nx, g.ny), g.nz);
        for (int R = 1; R <= rmax; ++R) {
            float cov = FLT_MAX;
            if (cx - R > 0) cov = fminf(cov, qx - (g.ox + (cx - R) * g.h));
            if (cx + R < g.nx - 1) cov = fminf(cov, (g.ox + (cx + R + 1) * g.h) - qx);
            if (cy - R > 0) cov = fminf(cov, qy - (g.oy + (cy - R) * g.h));
            if (cy + R < g.ny - 1) cov = fminf(cov, (g.oy + (cy + R + 1) * g.h) - qy);
            if (cz - R > 0) cov = fminf(cov, qz - (g.oz + (cz - R) * g.h));
            if (cz + R < g.nz - 1) cov = fminf(cov, (g.oz + (cz + R + 1) * g.h) - qz);
            if (cov == FLT_MAX) break;  // the block is the whole grid
            cov = fmaxf(cov - g.eps, 0.0f);
            if (key_d2(kth) < cov * cov) break;  // proven exact (strict: an unseen point at exactly kth could win a tie)
            const int Rn = R + 1;        // add the shell at Chebyshev distance Rn
            const int z0 = max(cz - Rn, 0), z1 = min(cz + Rn, g.nz - 1), y0 = max(cy - Rn, 0), y1 = min(cy + Rn, g.ny - 1);
            const int x0 = max(cx - Rn, 0), x1 = min(cx + Rn, g.nx - 1);
            for (int z = z0; z <= z1; ++z)
                for (int y = y0; y <= y1; ++y) {
                    // only the part of the shell the ball of the current k-th distance reaches (the k-th best moves as
                    // candidates are inserted; the test uses its current value)
                    const float dyz2 = gap2(qz, g.oz + z * g.h, g.oz + (z + 1) * g.h, g.eps) +
                                       gap2(qy, g.oy + y * g.h, g.oy + (y + 1) * g.h, g.eps);
                    if (dyz2 > key_d2(kth)) continue;
                    const bool shell_row = (z == cz - Rn) || (z == cz + Rn) || (y == cy - Rn) || (y == cy + Rn);
                    const unsigned rr = ((unsigned)z * g.ny + y) * g.nx;
                    for (int sgi = 0; sgi < (shell_row ? 1 : 2); ++sgi) {
                        int sxa, sxb;
                        if (shell_row) { sxa = x0; sxb = x1; }
                        else if (sgi == 0) { sxa = sxb = cx - Rn; if (sxa < 0) continue; }
                        else { sxa = sxb = cx + Rn; if (sxb > g.nx - 1) continue; }
                        if (dyz2 + gap2(qx, g.ox + sxa * g.h, g.ox + (sxb + 1) * g.h, g.eps) > key_d2(kth)) continue;
                        const unsigned s0 = start[rr + sxa], e0 = start[rr + sxb + 1];
                        for (unsigned base = s0; base < e0; base += 64) {
                            const unsigned pos = base + lane;
                            const bool valid = pos < e0;
                            const float4 p = pts[valid ? pos : e0 - 1];
                            Cand c;
                            c.key = valid ? cand_key(dist2(qx, qy, qz, p.x, p.y, p.z), __float_as_int(p.w)) : kNoCand;
                            c.pos = (int)pos;
                            insert_candidates(c, best, kth, k, kmask, lane);
                        }
                    }
                }
        }
        const bool have = (int)lane < k && key_d2(best.key) != FLT_MAX;
        if (out.knn_idx && (int)lane < k) {
            const size_t o = (size_t)qorig * (size_t)k + lane;
            out.knn_idx[o] = have ? key_idx(best.key) : -1;
            out.knn_d2[o] = key_d2(best.key);
        }
        if (out.covs || out.normals) {
            const float4 np = pts[have ? best.pos : 0];
            const unsigned cnt = (unsigned)__builtin_popcountll(__ballot(have));
            // The nine sums of covariance::kernel::estimate, each in ascending neighbour order (its order): lane t < cnt
            // forms the nine terms of its own neighbour once and parks them in LDS, then lane r < 9 adds up quantity r
            // over t = 0 .. cnt-1 — 20 additions on nine lanes in parallel instead of 9 x 20 on every lane.
            __syncthreads();  // (one wave per workgroup) the previous query's sums have been read
            if (have) {
                sm_terms[0][lane] = np.x; sm_terms[1][lane] = np.y; sm_terms[2][lane] = np.z;
                sm_terms[3][lane] = np.x * np.x; sm_terms[4][lane] = np.x * np.y; sm_terms[5][lane] = np.x * np.z;
                sm_terms[6][lane] = np.y * np.y; sm_terms[7][lane] = np.y * np.z; sm_terms[8][lane] = np.z * np.z;
            }
            __syncthreads();
            float acc9 = 0.0f;
            {
                const float* const col = sm_terms[lane < 9 ? lane : 0];
#pragma unroll
                for (int t = 0; t < 20; ++t)
                    if ((unsigned)t < cnt) acc9 += col[t];  // cnt is wave-uniform: a scalar branch
            }
            const float sx = bcast_f(acc9, 0), sy = bcast_f(acc9, 1), sz = bcast_f(acc9, 2), oxx = bcast_f(acc9, 3),
                        oxy = bcast_f(acc9, 4), oxz = bcast_f(acc9, 5), oyy = bcast_f(acc9, 6), oyz = bcast_f(acc9, 7),
                        ozz = bcast_f(acc9, 8);
            Mat3 C;
            if (cnt < 4) {
                C.m[0][0] = C.m[1][1] = C.m[2][2] = 1.0f;
                C.m[0][1] = C.m[0][2] = C.m[1][0] = C.m[1][2] = C.m[2][0] = C.m[2][1] = 0.0f;
            } else {
                const float inv = 1.0f / (float)cnt;
                const float mx = sx * inv, my = sy * inv, mz = sz * inv;
                const float cxy = oxy * inv - mx * my, cxz = oxz * inv - mx * mz, cyz = oyz * inv - my * mz;
                C.m[0][0] = oxx * inv - mx * mx; C.m[1][1] = oyy * inv - my * my; C.m[2][2] = ozz * inv - mz * mz;
                C.m[0][1] = C.m[1][0] = (cxy + cxy) * 0.5f;
                C.m[0][2] = C.m[2][0] = (cxz + cxz) * 0.5f;
                C.m[1][2] = C.m[2][1] = (cyz + cyz) * 0.5f;
            }
            if (out.covs && lane < 4) {
                const float4 col = lane == 0 ? make_float4(C.m[0][0], C.m[1][0], C.m[2][0], 0.0f)
                                 : lane == 1 ? make_float4(C.m[0][1], C.m[1][1], C.m[2][1], 0.0f)
                                 : lane == 2 ? make_float4(C.m[0][2], C.m[1][2], C.m[2][2], 0.0f)
                                             : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                out.covs[4 * (size_t)qorig + lane] = col;
            }
            if (out.normals) {
                float ev[3];
                Mat3 V;
                symmetric_eigen3(C, ev, V);
                const float nx_ = V.m[0][0], ny_ = V.m[1][0], nz_ = V.m[2][0];
                const float dd = chain3(nx_, qx, ny_, qy, nz_, qz);
                if (lane == 0)
                    out.normals[qorig] = (dd <= 1.0f) ? make_float4(nx_, ny_, nz_, 0.0f) : make_float4(-nx_, -ny_, -nz_, 0.0f);
            }
        }
    }
}

__global__ __launch_bounds__(kWave) void grid_self_knn_wave_kernel(const float4* __restrict__ pts,
                                                                   const unsigned* __restrict__ start,
                                                                   const unsigned* __restrict__ unit_off, GridDesc g,
                                                                   int k, TileOut out) {
    __shared__ float sm_terms[9][33];  // covariance terms of the k <= 20 neighbours (row stride 33: conflict-free columns)
    const unsigned unit = blockIdx.x;
    const unsigned rows = (unsigned)g.ny * g.nz;
    unsigned lo = 0, hi = rows;
    while (hi - lo > 1) {
        const unsigned mid = (lo + hi) >> 1;
        if (unit_off[mid] <= unit) lo = mid;
        else hi = mid;
    }
    const unsigned row = lo;
    const int ry = (int)(row % g.ny), rz = (int)(row / g.ny);
    const unsigned row_s = start[(size_t)row * g.nx], row_e = start[(size_t)(row + 1) * g.nx];
    const unsigned qs = row_s + (unit - unit_off[row]) * 64u;
    const unsigned nq = min(64u, row_e - qs);
    if (qs + nq <= out.pos_lo || qs >= out.pos_hi) return;  // the unit lies outside the requested range
    self_knn_wave_queries(pts, start, g, k, out, sm_terms, qs, nq, ry, rz, pts[min(qs + threadIdx.x, row_e - 1u)]);
}
// The same search for a LIST of queries (the to-do list of the lane-per-query kernels): a fixed grid of waves takes them one
// at a time. The ring walk that used to finish them kept 27 k of 1 M queries busy for 1.5 ms (a lane each, a 20-step
// insertion per candidate); a wave each, they take a few tens of microseconds.
__global__ __launch_bounds__(kWave) void grid_self_knn_wave_list_kernel(const float4* __restrict__ pts,
                                                                        const unsigned* __restrict__ start, GridDesc g, int k,
                                                                        TileOut out) {
    __shared__ float sm_terms[9][33];
    const unsigned n_items = *out.todo_count;
    for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x) {  // (wave-uniform)
        const unsigned pos = out.todo[item];
        const float4 q = pts[pos];
        const int ry = cell_coord(q.y, g.oy, g.inv_h, g.ny), rz = cell_coord(q.z, g.oz, g.inv_h, g.nz);
        self_knn_wave_queries(pts, start, g, k, out, sm_terms, pos, 1u, ry, rz, q);
    }
}
// ... and for a list of EXTERNAL queries (what grid_search_select_kernel could not prove): rows by query number.
__global__ __launch_bounds__(kWave) void grid_search_wave_list_kernel(const float4* __restrict__ pts,
                                                                      const unsigned* __restrict__ start, GridDesc g,
                                                                      const float4* __restrict__ queries, int k, Mat4Arg T_val,
                                                                      const float* __restrict__ T_dev, TileOut out) {
    __shared__ float sm_terms[9][33];
    const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
    const unsigned n_items = *out.todo_count;
    for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x) {  // (wave-uniform)
        const unsigned qi = out.todo[item];
        const float4 q4 = queries[qi];
        float4 q;
        transform_point(T, q4.x, q4.y, q4.z, q.x, q.y, q.z);
        q.w = __uint_as_float(qi);
        const int ry = cell_coord(q.y, g.oy, g.inv_h, g.ny), rz = cell_coord(q.z, g.oz, g.inv_h, g.nz);
        self_knn_wave_queries(pts, start, g, k, out, sm_terms, 0u, 1u, ry, rz, q);
    }
}

// Ring walk for the queries the tile kernel could not prove exact (their positions are listed in `todo`).
template <int KCAP>
__global__ __launch_bounds__(kBlock) void grid_self_knn_todo_kernel(const float4* __restrict__ pts,
                                                                    const unsigned* __restrict__ start, GridDesc g,
                                                                    int k, TileOut out) {
    const unsigned t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= *out.todo_count) return;
    const float4 q = pts[out.todo[t]];
    float bd[KCAP];
    int bi[KCAP], bp[KCAP];
#pragma unroll
    for (int i = 0; i < KCAP; ++i) { bd[i] = FLT_MAX; bi[i] = -1; bp[i] = -1; }
    float kth = FLT_MAX;
    int kth_idx = -1;
    const int cx = cell_coord(q.x, g.ox, g.inv_h, g.nx), cy = cell_coord(q.y, g.oy, g.inv_h, g.ny),
              cz = cell_coord(q.z, g.oz, g.inv_h, g.nz);
    const int rmax = max(max(g.nx, g.ny), g.nz);
    for (int r = 0; r <= rmax; ++r) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, g.nz - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, g.ny - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
        for (int z = z0; z <= z1; ++z) {
            const float dz2 = gap2(q.z, g.oz + z * g.h, g.oz + (z + 1) * g.h, g.eps);
            if (dz2 > kth) continue;
            for (int y = y0; y <= y1; ++y) {
                const float dyz2 = dz2 + gap2(q.y, g.oy + y * g.h, g.oy + (y + 1) * g.h, g.eps);
                if (dyz2 > kth) continue;
                const bool shell_row = (r == 0) || (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
                const unsigned row = ((unsigned)z * g.ny + y) * g.nx;
                const int nseg = shell_row ? 1 : 2;
                for (int sgi = 0; sgi < nseg; ++sgi) {
                    int xa, xb;
                    if (shell_row) { xa = x0; xb = x1; }
                    else if (sgi == 0) { xa = cx - r; xb = cx - r; if (xa < 0) continue; }
                    else { xa = cx + r; xb = cx + r; if (xb > g.nx - 1) continue; }
                    if (dyz2 + gap2(q.x, g.ox + xa * g.h, g.ox + (xb + 1) * g.h, g.eps) > kth) continue;
                    const unsigned s = start[row + xa], e = start[row + xb + 1];
                    for (unsigned i = s; i < e; ++i) {
                        const float4 p = pts[i];
                        const float d = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
                        const int pi = __float_as_int(p.w);
                        if (d < kth || (d == kth && pi < kth_idx)) {
                            float cd = d;
                            int ci = pi, cp = (int)i;
                            bool shifting = false;
#pragma unroll
                            for (int j = 0; j < KCAP; ++j) {
                                if (j < k) {
                                    const bool sw = shifting || cd < bd[j] || (cd == bd[j] && ci < bi[j]);
                                    const float td = bd[j];
                                    const int ti = bi[j], tp = bp[j];
                                    const float nd = sw ? cd : td;
                                    const int ni = sw ? ci : ti;
                                    bd[j] = nd; bi[j] = ni; bp[j] = sw ? cp : tp;
                                    cd = sw ? td : cd; ci = sw ? ti : ci; cp = sw ? tp : cp;
                                    shifting = sw;
                                    kth = nd; kth_idx = ni;
                                }
                            }
                        }
                    }
                }
            }
        }
        float cov = FLT_MAX;
        if (cx - r > 0) cov = fminf(cov, q.x - (g.ox + (cx - r) * g.h));
        if (cx + r < g.nx - 1) cov = fminf(cov, (g.ox + (cx + r + 1) * g.h) - q.x);
        if (cy - r > 0) cov = fminf(cov, q.y - (g.oy + (cy - r) * g.h));
        if (cy + r < g.ny - 1) cov = fminf(cov, (g.oy + (cy + r + 1) * g.h) - q.y);
        if (cz - r > 0) cov = fminf(cov, q.z - (g.oz + (cz - r) * g.h));
        if (cz + r < g.nz - 1) cov = fminf(cov, (g.oz + (cz + r + 1) * g.h) - q.z);
        if (cov == FLT_MAX) break;
        cov = fmaxf(cov - g.eps, 0.0f);
        if (kth < cov * cov) break;
    }
    const unsigned orig = __float_as_uint(q.w);
    if (out.knn_idx) {
        const size_t o = (size_t)orig * (size_t)k;
#pragma unroll
        for (int i = 0; i < KCAP; ++i)
            if (i < k) { out.knn_idx[o + i] = bi[i]; out.knn_d2[o + i] = bd[i]; }
    }
    if (out.covs || out.normals) {
        float c[6];
        bool identity;
        cov_from_list(pts, bp, k, c, identity);
        Mat3 C;
        if (identity) {
            C.m[0][0] = C.m[1][1] = C.m[2][2] = 1.0f;
            C.m[0][1] = C.m[0][2] = C.m[1][0] = C.m[1][2] = C.m[2][0] = C.m[2][1] = 0.0f;
        } else {
            C.m[0][0] = c[0]; C.m[0][1] = c[1]; C.m[0][2] = c[2];
            C.m[1][0] = c[1]; C.m[1][1] = c[3]; C.m[1][2] = c[4];
            C.m[2][0] = c[2]; C.m[2][1] = c[4]; C.m[2][2] = c[5];
        }
        if (out.covs) {
            float4* o4 = out.covs + 4 * (size_t)orig;
            o4[0] = make_float4(C.m[0][0], C.m[1][0], C.m[2][0], 0.0f);
            o4[1] = make_float4(C.m[0][1], C.m[1][1], C.m[2][1], 0.0f);
            o4[2] = make_float4(C.m[0][2], C.m[1][2], C.m[2][2], 0.0f);
            o4[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        if (out.normals) {
            float ev[3];
            Mat3 V;
            symmetric_eigen3(C, ev, V);
            const float nx_ = V.m[0][0], ny_ = V.m[1][0], nz_ = V.m[2][0];
            const float dd = chain3(nx_, q.x, ny_, q.y, nz_, q.z);
            out.normals[orig] = (dd <= 1.0f) ? make_float4(nx_, ny_, nz_, 0.0f) : make_float4(-nx_, -ny_, -nz_, 0.0f);
        }
    }
}

__global__ void fill_todo_kernel(unsigned* todo, unsigned* count, unsigned n) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) todo[i] = i;
    if (i == 0) *count = n;
}

// The work units of the self-kNN tiling (ceil(points in row / 64) per x-row, prefix-summed), made on first use: allocates, and
// reads their number back (one synchronisation per grid that is ever asked for a self-kNN).
int ensure_units(const sp_grid* gr, hipStream_t st) {
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (gr->units_ready || gr->n == 0) return SP_OK;
    const unsigned rows = (unsigned)gr->dims[1] * (unsigned)gr->dims[2];
    const size_t stmp_bytes = exclusive_scan_u32_workspace_bytes(rows + 1);
    ScratchBuf b_units, b_stmp;
    hipError_t e = b_units.get((rows + 1) * 4, st);
    if (e == hipSuccess) e = b_stmp.get(std::max<size_t>(stmp_bytes, 16), st);
    if (e == hipSuccess && gr->d_unit_off == nullptr) e = pooled_alloc(&gr->d_unit_off, (rows + 1) * 4, st);
    if (e != hipSuccess) { sp_set_error(hipGetErrorString(e)); return SP_ERR_HIP; }
    row_units_kernel<<<div_up(rows + 1, kBlock), kBlock, 0, st>>>(gr->d_start, (unsigned)gr->dims[0], rows, b_units.as<unsigned>());
    if (exclusive_scan_u32(b_units.as<unsigned>(), gr->d_unit_off, rows + 1, nullptr, b_stmp.p, stmp_bytes, st) != SP_OK) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipMemcpyAsync(&gr->n_units, gr->d_unit_off + rows, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // (the temporaries are idle from here on)
    if (e != hipSuccess) { sp_set_error(hipGetErrorString(e)); return SP_ERR_HIP; }
    gr->units_ready = true;
    return SP_OK;
}

template <int KCAP>
int launch_self(const sp_grid* gr, int k, const TileOut& out, hipStream_t st) {
    if (const int rc = ensure_units(gr, st); rc != SP_OK) return rc;
    const GridDesc g = grid_desc(gr);
    if (zero_async(out.todo_count, 4, st) != SP_OK) return SP_ERR_HIP;
    if (KCAP <= 10 && gr->self_knn_mode == 0 && k <= 7) {  // short lists: lane per point, exact without a to-do pass
        grid_self_knn_lane_kernel<KCAP><<<div_up(out.pos_hi - out.pos_lo, kBlock), kBlock, 0, st>>>(gr->d_pts, gr->d_start, g, k, out);
        return launch_status();
    }
    if (gr->n_units) {
        // lane-per-query tile kernel for short lists (k = 3 on 1M points: 0.70 ms against 1.94 ms; it loses from k = 7 up: scratch/selfknn_modes.py)
        if (KCAP <= 10 && (gr->self_knn_mode == 1 || (gr->self_knn_mode == 0 && k <= 6)))
            grid_self_knn_tile_kernel<KCAP><<<gr->n_units, kWave, 0, st>>>(gr->d_pts, gr->d_start, gr->d_unit_off, g, k, out);
        else if (k >= 7 && k <= 24 && (gr->self_knn_mode == 0 || gr->self_knn_mode == 3)) {
            // (1 M points: k = 8 / 10 / 12 / 16 / 20 in 0.48 / 0.49 / 0.49 / 0.52 / 0.60 ms against 0.54 / 0.68 for the lane kernel
            // and 0.84 ... 0.95 for the wave-cooperative one; with covariances 0.47 ... 0.57 against 1.0 ... 1.15)
            grid_self_knn_select_kernel<<<gr->n_units, kWave, 0, st>>>(gr->d_pts, gr->d_start, gr->d_unit_off, g, k, out);
            // its unproven queries (a few per cent: the k-th neighbour not provably inside the 27 cells), a wave each
            grid_self_knn_wave_list_kernel<<<kNumCU * 32, kWave, 0, st>>>(gr->d_pts, gr->d_start, g, k, out);
            return launch_status();
        } else
            grid_self_knn_wave_kernel<<<gr->n_units, kWave, 0, st>>>(gr->d_pts, gr->d_start, gr->d_unit_off, g, k, out);
    }
    grid_self_knn_todo_kernel<KCAP><<<div_up(gr->n, kBlock), kBlock, 0, st>>>(gr->d_pts, gr->d_start, g, k, out);
    return launch_status();
}

// Radius search = the k nearest within the radius (knn/kdtree.hpp:574-719: a candidate enters when d <= radius^2 and
// beats the k-th best): rows are ascending, so the entries beyond the radius are a suffix, turned into padding here.
__global__ __launch_bounds__(kBlock) void radius_filter_kernel(int32_t* __restrict__ idx, float* __restrict__ d2, size_t total,
                                                               float radius_sq) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    if (d2[i] > radius_sq) { idx[i] = -1; d2[i] = FLT_MAX; }
}

__global__ __launch_bounds__(kBlock) void grid_search_k1_kernel(const float4* __restrict__ pts,
                                                                const unsigned* __restrict__ start, GridDesc g,
                                                                const float4* __restrict__ queries, unsigned nq,
                                                                Mat4Arg T_val, const float* __restrict__ T_dev,
                                                                int32_t* __restrict__ idx_out,
                                                                float* __restrict__ d2_out,
                                                                const unsigned* __restrict__ order = nullptr) {
    const unsigned gi = blockIdx.x * kBlock + threadIdx.x;
    if (gi >= nq) return;
    const unsigned qi = order ? order[gi] : gi;
    const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
    const float4 q4 = queries[qi];
    float qx, qy, qz;
    transform_point(T, q4.x, q4.y, q4.z, qx, qy, qz);
    const Nearest nn = grid_nn1_auto(pts, start, g, qx, qy, qz);
    idx_out[qi] = nn.idx;
    d2_out[qi] = nn.d2;
}

// External queries in cell order (1 M queries in arbitrary order, k = 20: 2.1 ms against 0.68 ms for the same queries in cell
// order): key = the cell the transformed query falls into (non-finite queries last), sorted by the library's radix sort.
__global__ __launch_bounds__(kBlock) void grid_query_cell_kernel(const float4* __restrict__ queries, unsigned nq, GridDesc g,
                                                                 Mat4Arg T_val, const float* __restrict__ T_dev,
                                                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                                 uint32_t* __restrict__ count) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) *count = nq;
    if (i >= nq) return;
    const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
    const float4 q4 = queries[i];
    float qx, qy, qz;
    transform_point(T, q4.x, q4.y, q4.z, qx, qy, qz);
    unsigned key = (unsigned)g.nx * g.ny * g.nz;
    if (isfinite(qx) && isfinite(qy) && isfinite(qz))
        key = ((unsigned)cell_coord(qz, g.oz, g.inv_h, g.nz) * g.ny + cell_coord(qy, g.oy, g.inv_h, g.ny)) * g.nx +
              cell_coord(qx, g.ox, g.inv_h, g.nx);
    keys[i] = key;
    vals[i] = i;
}

template <int KCAP>
int launch(const sp_grid* gr, const float* q, size_t nq, size_t k, const float* T, int T_dev, int32_t* idx, float* d2,
           hipStream_t st, bool queries_in_cell_order = false, float bound2 = FLT_MAX) {
    Mat4Arg tv;
    for (int i = 0; i < 16; ++i) tv.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    if (T && !T_dev)
        for (int i = 0; i < 16; ++i) tv.m[i] = T[i];
    const GridDesc g = grid_desc(gr);
    // many queries: in cell order (0.1 ms per million for the sort; a loss below a few hundred thousand)
    uint32_t* sortbuf = nullptr;
    const unsigned* order = nullptr;
    unsigned* order_count = nullptr;
    struct Release {
        uint32_t*& p; hipStream_t st;
        ~Release() { if (p) { StreamSet used; used.note(st); pooled_free_after(p, used); } }
    } release{sortbuf, st};
    if (nq >= 400000 && gr->sort_queries && !queries_in_cell_order && gr->n != 0 && gr->ncells < 0xffffffffull) {
        const size_t wsb = radix_sort_u32_workspace_bytes(nq);
        if (pooled_alloc(&sortbuf, (4 * nq + 4) * sizeof(uint32_t) + wsb, st) != hipSuccess) return SP_ERR_HIP;
        uint32_t *ka = sortbuf, *kb = ka + nq, *va = kb + nq, *vb = va + nq;
        order_count = vb + nq;
        grid_query_cell_kernel<<<div_up(nq, kBlock), kBlock, 0, st>>>(reinterpret_cast<const float4*>(q), (unsigned)nq, g, tv,
                                                                     T_dev ? T : nullptr, ka, va, order_count);
        unsigned bits = 1;
        while ((1ull << bits) <= (unsigned long long)gr->ncells && bits < 32) ++bits;  // (the key `ncells` itself must fit)
        bool in_b = false;
        if (radix_sort_pairs_u32(ka, kb, va, vb, nq, bits, order_count + 4, wsb, &in_b, st) != SP_OK) return SP_ERR_HIP;
        order = in_b ? vb : va;
    }
    if (KCAP == 1) {
        grid_search_k1_kernel<<<div_up(nq, kBlock), kBlock, 0, st>>>(gr->d_pts, gr->d_start, g,
                                                                      reinterpret_cast<const float4*>(q), (unsigned)nq,
                                                                      tv, T_dev ? T : nullptr, idx, d2, order);
        return launch_status();
    }
    if (k > 10 && k <= 24 && gr->n != 0 && (gr->self_knn_mode == 0 || gr->self_knn_mode == 3)) {
        // lane-per-query selection inside the 27 cells; the queries it cannot prove (a few per cent at the densities the grid
        // is built for) are listed and finished a wave each. (k <= 10 stays on the ring walk: its row pruning scans fewer
        // candidates than all 27 cells, which decides for queries in random order — 0.8 against 2.2 ms at k = 10.) The list lives in the library's buffer pool for the
        // duration of the call (handed back tagged with an event on this stream).
        unsigned* todo = nullptr;
        if (pooled_alloc(&todo, (nq + 1) * sizeof(unsigned), st) != hipSuccess) return SP_ERR_HIP;
        unsigned* const todo_count = todo + nq;
        int rc = zero_async(todo_count, 4, st);
        if (rc == SP_OK) {
            grid_search_select_kernel<<<div_up(nq, kWave), kWave, 0, st>>>(gr->d_pts, gr->d_start, g,
                                                                        reinterpret_cast<const float4*>(q), (unsigned)nq, (int)k, tv,
                                                                        T_dev ? T : nullptr, idx, d2, todo, todo_count, order);
            TileOut lo;
            lo.knn_idx = idx; lo.knn_d2 = d2; lo.covs = nullptr; lo.normals = nullptr;
            lo.todo = todo; lo.todo_count = todo_count; lo.pos_lo = 0u; lo.pos_hi = 0xffffffffu;
            grid_search_wave_list_kernel<<<kNumCU * 32, kWave, 0, st>>>(gr->d_pts, gr->d_start, g, reinterpret_cast<const float4*>(q),
                                                                      (int)k, tv, T_dev ? T : nullptr, lo);
            rc = launch_status();
        }
        StreamSet used;
        used.note(st);
        pooled_free_after(todo, used);
        return rc;
    }
    grid_search_kernel<KCAP><<<div_up(nq, kBlock), kBlock, 0, st>>>(gr->d_pts, gr->d_start, g,
                                                                  reinterpret_cast<const float4*>(q), (unsigned)nq,
                                                                  (int)k, tv, T_dev ? T : nullptr, idx, d2, order, order_count, bound2);
    return launch_status();
}

}  // namespace
}  // namespace sp

extern "C" void sp_grid_destroy(sp_grid* g) {
    if (!g) return;
    // the arrays go back to the pool tagged with an event per stream they were used on (no device-wide wait)
    sp::pooled_free_after(g->d_pts, g->streams);
    sp::pooled_free_after(g->d_start, g->streams);
    sp::pooled_free_after(g->d_unit_off, g->streams);
    if (g->built_ev) (void)hipEventDestroy(g->built_ev);  // (an event still pending is released when it completes)
    delete g;
}

namespace sp {
namespace {
int grid_create_impl(const float* points, size_t n, float cell_size, float points_per_cell, bool adaptive, void* stream,
                     sp_grid** out, const float* bounds6 = nullptr) {
    if (!out) return SP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (n >= (1ull << 31)) {
        sp_set_error("[GridKNN::build] more than 2^31 points: indices are int32");
        return SP_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = as_stream(stream);
    sp_grid* g = new sp_grid();
    g->n = n;
    g->streams.note(st);
    auto fail = [&](hipError_t e) {
        sp_set_error(hipGetErrorString(e));
        (void)hipStreamSynchronize(st);  // nothing of this build may still be running when its scratch goes back to the pool
        sp_grid_destroy(g);
        return SP_ERR_HIP;
    };
    hipError_t e;
    if (n == 0) {
        if ((e = pooled_alloc(&g->d_start, 4 * sizeof(uint32_t), st)) != hipSuccess) return fail(e);
        if ((e = hipMemsetAsync(g->d_start, 0, 2 * sizeof(uint32_t), st)) != hipSuccess) return fail(e);
        *out = g;
        return SP_OK;
    }
    const float4* pts = reinterpret_cast<const float4*>(points);
    // 1. bounding box of the finite points
    ScratchBuf bbox_buf;
    if ((e = bbox_buf.get(6 * sizeof(unsigned), st)) != hipSuccess) return fail(e);
    unsigned* const d_bbox = bbox_buf.as<unsigned>();
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    unsigned h_bbox_own[6];
    unsigned* const h_bbox = pinned_mailbox() ? static_cast<unsigned*>(pinned_mailbox()) : h_bbox_own;
    float mn[3], mx[3];
    if (bounds6) {  // the caller knows a box that holds every finite point (sp_grid_create_bounded): no kernel, no read-back, no wait
        for (int a = 0; a < 3; ++a) { mn[a] = bounds6[a]; mx[a] = bounds6[3 + a]; }
        e = hipSuccess;
    } else {
        e = hipMemcpyAsync(d_bbox, init, sizeof init, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            bbox_kernel<<<std::min(stream_grid(n, kBlock, 4), 256u), kBlock, 0, st>>>(pts, (unsigned)n, d_bbox);  // (1024 workgroups: 26 us — their 6 atomics each share one cache line)
            e = hipMemcpyAsync(h_bbox, d_bbox, 6 * sizeof(unsigned), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return fail(e);
        bool any = h_bbox[0] != 0xffffffffu;
        for (int a = 0; a < 3; ++a) {
            mn[a] = any ? dec(h_bbox[a]) : 0.0f;
            mx[a] = any ? dec(h_bbox[3 + a]) : 0.0f;
        }
    }
    // 2. cell size: given, or chosen for `points_per_cell` points per cell on average over the bounding box
    float ext[3], ext_max = 0.0f, scale = 0.0f;
    for (int a = 0; a < 3; ++a) {
        ext[a] = mx[a] - mn[a];
        ext_max = std::max(ext_max, ext[a]);
        scale = std::max(scale, std::max(std::fabs(mn[a]), std::fabs(mx[a])));
    }
    float h = cell_size;
    if (!(h > 0.0f)) {
        const float ppc = points_per_cell > 0.0f ? points_per_cell : 2.0f;
        // treat degenerate (flat) extents as one cell thick
        double vol = 1.0;
        int live = 0;
        for (int a = 0; a < 3; ++a)
            if (ext[a] > 1e-6f * std::max(ext_max, 1e-30f)) { vol *= ext[a]; ++live; }
        h = live ? (float)std::pow(vol * ppc / (double)n, 1.0 / live) : 1.0f;
    }
    if (!(h > 0.0f) || !std::isfinite(h)) h = 1.0f;
    const size_t max_cells = 1ull << 25;  // 32M cells = 128 MB of cell_start at most
    // 3. sort points by cell id, gather, cell_start (temporaries from the scratch pool: idle again when this returns)
    unsigned *keys_in = nullptr, *keys_out = nullptr, *vals_in = nullptr, *vals_out = nullptr;
    ScratchBuf b_kin, b_kout, b_vin, b_vout, b_tmp, b_stats;
    size_t tmp_bytes = radix_sort_u32_workspace_bytes(n);  // radix_sort.hip
    e = b_kin.get(n * 4, st);
    if (e == hipSuccess) e = b_kout.get(n * 4, st);
    if (e == hipSuccess) e = b_vin.get(n * 4, st);
    if (e == hipSuccess) e = b_vout.get(n * 4, st);
    if (e == hipSuccess) e = b_tmp.get(std::max<size_t>(tmp_bytes, 16), st);
    if (e == hipSuccess) e = b_stats.get(16, st);
    if (e == hipSuccess) e = pooled_alloc(&g->d_pts, n * sizeof(float4), st);
    if (e != hipSuccess) return fail(e);
    void* const tmp = b_tmp.p;
    // Density-adaptive build (sp_grid_create_adaptive): the cell size of the volume rule assumes the points fill their bounding
    // box. A cloud of SURFACES (a voxel-downsampled LiDAR scan) leaves most cells empty and packs the occupied ones — 40 points
    // per occupied cell on the reference's bundled scan at the 0.5 target, and a query scans hundreds of candidates per block of
    // cells. So the build measures what it made (occupied cells, in the same synchronisation it ends with anyway) and, while the
    // occupied cells hold well more than a uniform cloud's would, shrinks h by the dimension the two measurements imply (2 — a
    // surface — for the first correction) and builds again: at most three more sorts, none for a uniform cloud.
    const double ppc_target = (cell_size > 0.0f) ? 0.0 : (points_per_cell > 0.0f ? points_per_cell : 2.0f);
    const double occ_target = ppc_target > 0.0 ? ppc_target / (1.0 - std::exp(-ppc_target)) : 0.0;  // Poisson: mean of the non-empty cells
    double prev_h = 0.0, prev_occ_cells = 0.0;
    for (int attempt = 0;; ++attempt) {
        for (;;) {
            size_t nc = 1;
            for (int a = 0; a < 3; ++a) {
                const double d = std::floor((double)ext[a] / h) + 1.0;
                g->dims[a] = (int)std::min(d, 2.0e6);
                nc *= (size_t)g->dims[a];
            }
            if (nc <= max_cells) { g->ncells = nc; break; }
            h *= 1.26f;
            adaptive = false;  // the table is as large as it may get
        }
        g->h = h;
        g->inv_h = 1.0f / h;
        g->eps = 4.0e-6f * (scale + ext_max + h);  // bounds the float rounding of (p - org) * inv_h cell assignment
        for (int a = 0; a < 3; ++a) g->org[a] = mn[a];
        unsigned end_bit = 1;
        while ((1ull << end_bit) <= g->ncells && end_bit < 32) ++end_bit;
        keys_in = b_kin.as<unsigned>(); keys_out = b_kout.as<unsigned>(); vals_in = b_vin.as<unsigned>(); vals_out = b_vout.as<unsigned>();
        const unsigned rows = (unsigned)g->dims[1] * (unsigned)g->dims[2];
        ScratchBuf b_units, b_stmp;
        const size_t stmp_bytes = exclusive_scan_u32_workspace_bytes(rows + 1);
        e = pooled_alloc(&g->d_start, (g->ncells + 3) * sizeof(uint32_t), st);  // (+ 2: fast_extents reads three words at the last cell)
        if (adaptive) {  // (this form of the build ends every attempt with a read-back anyway: the work units ride along)
            if (e == hipSuccess) e = b_units.get((rows + 1) * 4, st);
            if (e == hipSuccess) e = pooled_alloc(&g->d_unit_off, (rows + 1) * 4, st);
            if (e == hipSuccess) e = b_stmp.get(std::max<size_t>(stmp_bytes, 16), st);
        }
        unsigned* const units = b_units.as<unsigned>();
        void* const stmp = b_stmp.p;
        bool small = false;
        if (e == hipSuccess) {
            GridDesc gd{g->inv_h, g->h, g->eps, g->org[0], g->org[1], g->org[2], g->dims[0], g->dims[1], g->dims[2],
                        (unsigned)n};
            small = !adaptive && n <= kSmallBuildPoints && g->ncells < kSmallBuildCells && g_small_build;
            if (small) {  // cell ids, sort, gather and cell table in one launch of one workgroup
                grid_build_small_kernel<<<1, kSbThreads, 0, st>>>(pts, gd, (unsigned)g->ncells, (end_bit + 7u) / 8u, g->d_pts, g->d_start,
                                                                  bounds6 ? device_error_word() : nullptr);
            } else {
                cell_id_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(pts, gd, keys_in, vals_in, bounds6 ? device_error_word() : nullptr);
                bool in_b = false;
                if (const int rc = radix_sort_pairs_u32(keys_in, keys_out, vals_in, vals_out, n, end_bit, tmp, tmp_bytes, &in_b, st); rc != SP_OK) {
                    // (with the message the sort's own status check left: a launch error — or the device error word, raised by then
                    // by this build's cell_id_kernel for a point outside the caller's box)
                    (void)hipStreamSynchronize(st);
                    sp_grid_destroy(g);
                    return rc;
                }
                if (!in_b) { keys_out = keys_in; vals_out = vals_in; }  // the passes ping-pong: the sorted pairs are where the last one wrote
            }
        }
        unsigned h_stats[2] = {0u, 0u};
        if (e == hipSuccess) {
            if (!small) {
                gather_sorted_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(pts, vals_out, (unsigned)n, g->d_pts);
                cell_start_kernel<<<div_up(n + 1, kBlock), kBlock, 0, st>>>(keys_out, (unsigned)n, (unsigned)g->ncells,
                                                                                  g->d_start);
            }
            if (!adaptive) {
                // Nothing is read back and nothing waited for: the temporaries go to the pool behind the stream's work, an event
                // marks the end of the build for other streams (grid_use), and the work units of the self-kNN tiling are made
                // when a self-kNN first asks for them (ensure_units).
                if (hipEventCreateWithFlags(&g->built_ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(g->built_ev, st) != hipSuccess)
                    return fail(hipErrorUnknown);
                if (const int ls = launch_status(); ls != SP_OK) {  // (with its own message: a launch error, or the device error word)
                    (void)hipStreamSynchronize(st);
                    sp_grid_destroy(g);
                    return ls;
                }
                g->build_stream = st;
                for (ScratchBuf* b : {&bbox_buf, &b_kin, &b_kout, &b_vin, &b_vout, &b_tmp, &b_stats}) b->release_after(g->streams);
                break;
            }
            // work units of the self-kNN tiling: ceil(points in row / 64) per x-row, prefix-summed (same stream: no sync between)
            row_units_kernel<<<div_up(rows + 1, kBlock), kBlock, 0, st>>>(g->d_start, (unsigned)g->dims[0], rows, units);
            if (exclusive_scan_u32(units, g->d_unit_off, rows + 1, nullptr, stmp, stmp_bytes, st) != SP_OK) e = hipErrorUnknown;
            g->units_ready = true;
            if (adaptive && e == hipSuccess) {
                e = zero_async(b_stats.p, 8, st) == SP_OK ? hipSuccess : hipErrorUnknown;
                cell_stats_kernel<<<std::min(div_up(g->ncells, kBlock), 1024u), kBlock, 0, st>>>(g->d_start, (unsigned)g->ncells,
                                                                                                 b_stats.as<unsigned>());
                if (e == hipSuccess) e = hipMemcpyAsync(h_stats, b_stats.p, 8, hipMemcpyDeviceToHost, st);
            }
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&g->n_units, g->d_unit_off + rows, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // the temporaries are idle from here on
        if (e != hipSuccess) return fail(e);
        if (!adaptive || attempt >= 3 || h_stats[0] == 0) break;
        g->max_cell = h_stats[1];
        g->max_cell_known = true;
        const double occ_cells = (double)h_stats[0];
        const double mean_occ = (double)n / occ_cells;
        // (act when an occupied cell holds more than twice what it would in a uniform cloud at four times the target — 4 points at
        // the in-loop search's 0.5 — and aim at 3: the block stages of the search cover 0.5 h and 1.5 h around the query, so a
        // cell should span about two point spacings, not one)
        const double occ_aim = std::max(occ_target, 3.0), occ_act = std::max(1.6 * occ_target, 4.0);
        if (mean_occ <= occ_act) break;
        // dimension of the point set between the two latest cell sizes (occupied cells ~ h^-D), 2 before there are two
        double D = 2.0;
        if (prev_h > 0.0 && prev_occ_cells > 0.0 && occ_cells > prev_occ_cells)
            D = std::min(3.0, std::max(1.0, std::log(occ_cells / prev_occ_cells) / std::log(prev_h / (double)h)));
        prev_h = h;
        prev_occ_cells = occ_cells;
        const float h_new = (float)((double)h * std::pow(occ_aim / mean_occ, 1.0 / D));
        if (!(h_new > 0.0f) || !(h_new < 0.97f * h)) break;
        pooled_free(g->d_start); g->d_start = nullptr;       // (synchronised above: nothing on the device uses them)
        pooled_free(g->d_unit_off); g->d_unit_off = nullptr;
        g->max_cell_known = false;
        h = h_new;
    }
    *out = g;
    return SP_OK;
}
}  // namespace
}  // namespace sp

extern "C" int sp_internal_grid_small_build(int enable) {
    const int was = sp::g_small_build ? 1 : 0;
    if (enable >= 0) sp::g_small_build = enable != 0;
    return was;
}
extern "C" int sp_grid_create(const float* points, size_t n, float cell_size, float points_per_cell, void* stream,
                              sp_grid** out) {
    return sp::grid_create_impl(points, n, cell_size, points_per_cell, false, stream, out);
}
extern "C" int sp_grid_create_adaptive(const float* points, size_t n, float points_per_cell, void* stream, sp_grid** out) {
    return sp::grid_create_impl(points, n, 0.0f, points_per_cell, true, stream, out);
}
extern "C" int sp_grid_create_bounded(const float* points, size_t n, const float* bounds_min_max6, float cell_size,
                                      float points_per_cell, void* stream, sp_grid** out) {
    if (!bounds_min_max6) return SP_ERR_INVALID_ARGUMENT;
    for (int a = 0; a < 3; ++a)
        if (!(bounds_min_max6[a] <= bounds_min_max6[3 + a]) || !std::isfinite(bounds_min_max6[a]) || !std::isfinite(bounds_min_max6[3 + a])) {
            sp_set_error("[GridKNN::build] bounds must be finite with min <= max");
            return SP_ERR_INVALID_ARGUMENT;
        }
    return sp::grid_create_impl(points, n, cell_size, points_per_cell, false, stream, out, bounds_min_max6);
}

// ---- lazy delete (the grid's counterpart of KDTree::remove_nodes_by_flags, kdtree.hpp:282-284, 721-765) -------------
// The cell order survives the removal of points, so nothing is re-sorted: kept points are compacted in place of order,
// re-labelled with their new indices, and every cell's start becomes the number of kept points before its old start.
namespace sp {
namespace {
__global__ __launch_bounds__(kBlock) void grid_keep_kernel(const float4* __restrict__ pts, unsigned n,
                                                           const uint8_t* __restrict__ flags, unsigned n_flags,
                                                           unsigned* __restrict__ keep) {
    const unsigned pos = blockIdx.x * kBlock + threadIdx.x;
    if (pos > n) return;
    if (pos == n) { keep[pos] = 0; return; }  // the scan's last entry: the number of kept points
    const int p = __float_as_int(pts[pos].w);
    keep[pos] = (p >= 0 && ((unsigned)p >= n_flags || flags[p])) ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void grid_compact_kernel(const float4* __restrict__ pts, unsigned n,
                                                              const uint8_t* __restrict__ flags,
                                                              const int32_t* __restrict__ new_idx, unsigned n_flags,
                                                              const unsigned* __restrict__ scan, float4* __restrict__ out) {
    const unsigned pos = blockIdx.x * kBlock + threadIdx.x;
    if (pos >= n) return;
    float4 s = pts[pos];
    const int p = __float_as_int(s.w);
    if (p < 0) return;
    if ((unsigned)p < n_flags) {
        if (!flags[p]) return;
        s.w = __int_as_float(new_idx[p]);
    }
    out[scan[pos]] = s;
}
__global__ __launch_bounds__(kBlock) void grid_restart_kernel(const unsigned* __restrict__ old_start, size_t ncells,
                                                              const unsigned* __restrict__ scan,
                                                              unsigned* __restrict__ new_start) {
    const size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (c <= ncells) new_start[c] = scan[old_start[c]];
}
}  // namespace
}  // namespace sp

extern "C" int sp_grid_remove_by_flags(sp_grid* g, const uint8_t* flags, const int32_t* new_indices, size_t n_flags,
                                       void* stream) {
    using namespace sp;
    if (!g || (n_flags && (!flags || !new_indices))) return SP_ERR_INVALID_ARGUMENT;
    const size_t n = g->n;
    if (n == 0 || n_flags == 0) return SP_OK;
    hipStream_t st = as_stream(stream);
    grid_use(g, st);
    ScratchBuf b_keep, b_scan, b_tmp, b_units, b_start;
    float4* new_pts = nullptr;
    size_t tmp_bytes = 0, tmp2_bytes = 0;
    const unsigned rows = (unsigned)g->dims[1] * (unsigned)g->dims[2];
    tmp_bytes = exclusive_scan_u32_workspace_bytes(n + 1);
    tmp2_bytes = exclusive_scan_u32_workspace_bytes(rows + 1);
    hipError_t e = b_keep.get((n + 1) * 4, st);
    if (e == hipSuccess) e = b_scan.get((n + 1) * 4, st);
    if (e == hipSuccess) e = b_tmp.get(std::max<size_t>(std::max(tmp_bytes, tmp2_bytes), 16), st);
    if (e == hipSuccess) e = b_units.get((rows + 1) * 4, st);
    if (e == hipSuccess) e = b_start.get((g->ncells + 1) * 4, st);
    if (e == hipSuccess) e = pooled_alloc(&new_pts, n * sizeof(float4), st);
    auto fail = [&](hipError_t err) {
        sp_set_error(hipGetErrorString(err));
        (void)hipStreamSynchronize(st);
        pooled_free(new_pts);
        return SP_ERR_HIP;
    };
    if (e != hipSuccess) return fail(e);
    unsigned* const keep = b_keep.as<unsigned>();
    unsigned* const scan = b_scan.as<unsigned>();
    unsigned* const units = b_units.as<unsigned>();
    unsigned* const new_start = b_start.as<unsigned>();
    grid_keep_kernel<<<div_up(n + 1, kBlock), kBlock, 0, st>>>(g->d_pts, (unsigned)n, flags, (unsigned)n_flags, keep);
    if (exclusive_scan_u32(keep, scan, n + 1, nullptr, b_tmp.p, tmp_bytes, st) != SP_OK) return fail(hipErrorUnknown);
    grid_compact_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(g->d_pts, (unsigned)n, flags, new_indices, (unsigned)n_flags,
                                                              scan, new_pts);
    grid_restart_kernel<<<div_up(g->ncells + 1, kBlock), kBlock, 0, st>>>(g->d_start, g->ncells, scan, new_start);
    unsigned kept = 0;
    e = hipMemcpyAsync(&kept, scan + n, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(g->d_start, new_start, (g->ncells + 1) * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(e);
    if (g->d_unit_off == nullptr && (e = pooled_alloc(&g->d_unit_off, (rows + 1) * 4, st)) != hipSuccess) return fail(e);
    row_units_kernel<<<div_up(rows + 1, kBlock), kBlock, 0, st>>>(g->d_start, (unsigned)g->dims[0], rows, units);
    if (exclusive_scan_u32(units, g->d_unit_off, rows + 1, nullptr, b_tmp.p, tmp2_bytes, st) != SP_OK) e = hipErrorUnknown;
    g->units_ready = true;
    if (e == hipSuccess) e = hipMemcpyAsync(&g->n_units, g->d_unit_off + rows, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(e);
    pooled_free_after(g->d_pts, g->streams);  // readers of the old array on other streams: the pool waits for them
    g->d_pts = new_pts;
    g->n = kept;
    return SP_OK;
}

extern "C" size_t sp_grid_size(const sp_grid* g) { return g ? g->n : 0; }
extern "C" float sp_grid_cell_size(const sp_grid* g) { return g ? g->h : 0.0f; }

namespace sp {
namespace {
__global__ __launch_bounds__(kBlock) void grid_order_kernel(const float4* __restrict__ pts, unsigned n,
                                                            uint32_t* __restrict__ out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = __float_as_uint(pts[i].w);
}
}  // namespace
}  // namespace sp

extern "C" int sp_grid_order(const sp_grid* grid, uint32_t* idx_out, void* stream) {
    if (!grid || (!idx_out && grid->n)) return SP_ERR_INVALID_ARGUMENT;
    if (grid->n == 0) return SP_OK;
    sp::grid_use(grid, sp::as_stream(stream));
    sp::grid_order_kernel<<<sp::div_up(grid->n, sp::kBlock), sp::kBlock, 0, sp::as_stream(stream)>>>(
        grid->d_pts, (unsigned)grid->n, idx_out);
    return sp::launch_status();
}

extern "C" int sp_grid_search(const sp_grid* grid, const float* queries, size_t nq, size_t k, const float* transT,
                              int transT_on_device, int32_t* idx_out, float* d2_out, void* stream) {
    using namespace sp;
    if (!grid || k == 0) {
        sp_set_error("[GridKNN::knn_search_async] null grid or k == 0");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (k > 20) {
        sp_set_error("[GridKNN::knn_search_async] `k` is too large (max 20).");
        return SP_ERR_RUNTIME;
    }
    if (nq == 0) return SP_OK;
    hipStream_t st = as_stream(stream);
    grid_use(grid, st);
    if (k == 1) return launch<1>(grid, queries, nq, k, transT, transT_on_device, idx_out, d2_out, st);
    if (k <= 10) return launch<10>(grid, queries, nq, k, transT, transT_on_device, idx_out, d2_out, st);
    return launch<20>(grid, queries, nq, k, transT, transT_on_device, idx_out, d2_out, st);
}

// The grid's own cell-ordered points as queries (row i = neighbours of the point at grid position i): the caller knows the
// order, no sort (internal: the certificates of sp_gicp_target_create).
namespace sp {
int grid_search_own_points(const sp_grid* grid, size_t k, int32_t* idx_out, float* d2_out, hipStream_t st, float bound2) {
    if (!grid || k == 0 || k > 10) return SP_ERR_INVALID_ARGUMENT;
    if (grid->n == 0) return SP_OK;
    grid_use(grid, st);
    const float* q = reinterpret_cast<const float*>(grid->d_pts);
    if (k == 1) return launch<1>(grid, q, grid->n, k, nullptr, 0, idx_out, d2_out, st, true);
    return launch<10>(grid, q, grid->n, k, nullptr, 0, idx_out, d2_out, st, true, bound2);
}
}  // namespace sp

// Measured the first time it is asked for (one small kernel + a blocking read-back on the default stream): the grids of the
// registration path never ask, and their build stays at its 0.18 ms.
extern "C" uint32_t sp_grid_max_cell_points(const sp_grid* grid) {
    using namespace sp;
    if (!grid || grid->n == 0 || grid->ncells == 0) return 0u;
    if (!grid->max_cell_known) {
        ScratchBuf word;
        unsigned h = 0;
        if (word.get(4) != hipSuccess) return 0u;
        if (hipMemsetAsync(word.p, 0, 4, nullptr) != hipSuccess) return 0u;
        cell_max_kernel<<<std::min(div_up(grid->ncells, kBlock), 256u), kBlock, 0, nullptr>>>(grid->d_start, (unsigned)grid->ncells,
                                                                                                word.as<unsigned>());
        if (hipMemcpy(&h, word.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return 0u;  // (synchronises)
        grid->max_cell = h;
        grid->max_cell_known = true;
    }
    return grid->max_cell;
}

extern "C" size_t sp_grid_self_workspace_bytes(const sp_grid* grid) { return grid ? (grid->n + 2) * 4 : 0; }

extern "C" int sp_grid_self_knn(const sp_grid* grid, size_t k, int32_t* idx_out, float* d2_out, float* covs_out,
                                float* normals_out, void* workspace, size_t workspace_bytes, void* stream) {
    return sp_grid_self_knn_range(grid, k, 0, grid ? grid->n : 0, idx_out, d2_out, covs_out, normals_out, workspace,
                                  workspace_bytes, stream);
}

extern "C" int sp_grid_self_knn_range(const sp_grid* grid, size_t k, size_t pos_first, size_t pos_count, int32_t* idx_out,
                                      float* d2_out, float* covs_out, float* normals_out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    using namespace sp;
    if (!grid || k == 0) {
        sp_set_error("[GridKNN::self_knn] null grid or k == 0");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (k > 20) {
        sp_set_error("[GridKNN::knn_search_async] `k` is too large (max 20).");
        return SP_ERR_RUNTIME;
    }
    if (pos_first > grid->n || pos_count > grid->n - pos_first) {
        sp_set_error("[GridKNN::self_knn] position range outside the grid");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (grid->n == 0 || pos_count == 0) return SP_OK;
    if (!workspace || workspace_bytes < sp_grid_self_workspace_bytes(grid)) {
        sp_set_error("[GridKNN::self_knn] workspace too small (sp_grid_self_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    TileOut out;
    out.knn_idx = (idx_out && d2_out) ? idx_out : nullptr;
    out.knn_d2 = d2_out;
    out.covs = reinterpret_cast<float4*>(covs_out);
    out.normals = reinterpret_cast<float4*>(normals_out);
    out.todo_count = static_cast<unsigned*>(workspace);
    out.todo = out.todo_count + 2;
    out.pos_lo = (unsigned)pos_first;
    out.pos_hi = (unsigned)(pos_first + pos_count);
    hipStream_t st = as_stream(stream);
    grid_use(grid, st);
    if (k <= 10) return launch_self<10>(grid, (int)k, out, st);
    return launch_self<20>(grid, (int)k, out, st);
}

// Tuning hook, not part of the stable surface.
extern "C" int sp_grid_radius_search(const sp_grid* grid, const float* queries, size_t nq, size_t max_k, float radius,
                                     const float* transT, int transT_on_device, int32_t* idx_out, float* d2_out,
                                     void* stream) {
    using namespace sp;
    if (max_k == 0 || nq == 0) return SP_OK;
    if (max_k > 20) {
        sp_set_error("[GridKNN::radius_search_async] `max_k` is too large (max 20).");
        return SP_ERR_RUNTIME;
    }
    const int rc = sp_grid_search(grid, queries, nq, max_k, transT, transT_on_device, idx_out, d2_out, stream);
    if (rc != SP_OK) return rc;
    const size_t total = nq * max_k;
    radius_filter_kernel<<<div_up(total, kBlock), kBlock, 0, as_stream(stream)>>>(idx_out, d2_out, total, radius * radius);
    return launch_status();
}

// Per-handle tuning switch (sp_internal.h): exported for tests/ and scratch/ only.
extern "C" int sp_internal_grid_option(sp_grid* grid, int option, int value) {
    if (!grid) return SP_ERR_INVALID_ARGUMENT;
    if (option == SP_INTERNAL_SELF_KNN_MODE) grid->self_knn_mode = value;
    else if (option == SP_INTERNAL_GRID_SORT_QUERIES) grid->sort_queries = value;
    else return SP_ERR_INVALID_ARGUMENT;
    return SP_OK;
}

// Rows between the caller's (original) order and the grid's position order, for clouds whose per-point results are computed
// by ranges of grid positions on different ranks (sp_grid_self_knn_range) and exchanged with one all-gather.
namespace sp {
namespace {
template <bool TO_POSITIONS>
__global__ __launch_bounds__(kBlock) void grid_rows_kernel(const float4* __restrict__ pts, unsigned pos_lo, unsigned count,
                                                           unsigned quads, const float4* __restrict__ in,
                                                           float4* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (size_t)count * quads) return;
    const unsigned j = (unsigned)(t / quads), q = (unsigned)(t % quads);
    const unsigned orig = __float_as_uint(pts[pos_lo + j].w);
    if (TO_POSITIONS) out[(size_t)j * quads + q] = in[(size_t)orig * quads + q];
    else out[(size_t)orig * quads + q] = in[(size_t)j * quads + q];
}
int grid_rows(const sp_grid* grid, const void* in, size_t row_bytes, size_t pos_first, size_t pos_count, void* out,
              bool to_positions, hipStream_t st) {
    if (!grid || (pos_count && (!in || !out))) return SP_ERR_INVALID_ARGUMENT;
    if (row_bytes == 0 || row_bytes % 16 != 0 || pos_first > grid->n || pos_count > grid->n - pos_first) {
        sp_set_error("[GridKNN] rows must be a multiple of 16 bytes and the position range inside the grid");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (pos_count == 0) return SP_OK;
    grid_use(grid, st);
    const unsigned quads = (unsigned)(row_bytes / 16);
    const size_t total = pos_count * quads;
    if (to_positions)
        grid_rows_kernel<true><<<div_up(total, kBlock), kBlock, 0, st>>>(grid->d_pts, (unsigned)pos_first, (unsigned)pos_count, quads,
                                                                        static_cast<const float4*>(in), static_cast<float4*>(out));
    else
        grid_rows_kernel<false><<<div_up(total, kBlock), kBlock, 0, st>>>(grid->d_pts, (unsigned)pos_first, (unsigned)pos_count, quads,
                                                                         static_cast<const float4*>(in), static_cast<float4*>(out));
    return launch_status();
}
}  // namespace
}  // namespace sp
extern "C" int sp_grid_gather_rows(const sp_grid* grid, const void* rows, size_t row_bytes, size_t pos_first, size_t pos_count,
                                   void* out_by_position, void* stream) {
    return sp::grid_rows(grid, rows, row_bytes, pos_first, pos_count, out_by_position, true, sp::as_stream(stream));
}
extern "C" int sp_grid_scatter_rows(const sp_grid* grid, const void* in_by_position, size_t row_bytes, size_t pos_first,
                                    size_t pos_count, void* rows, void* stream) {
    return sp::grid_rows(grid, in_by_position, row_bytes, pos_first, pos_count, rows, false, sp::as_stream(stream));
}

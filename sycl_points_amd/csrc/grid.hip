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
// Build (all on the device, deterministic): bounding box -> cell id per point -> radix sort (cell, index) ->
// gather points into cell order as float4 {x,y,z,index-bits} -> cell_start[] by binary search.
// Search: one query per lane, rings of cells around the query's cell; a row / cell is skipped when its box lies
// farther than the current k-th distance; the search stops when the k-th distance is inside the scanned block.
// Box tests are made conservative by `eps` so that float rounding in the cell assignment can never prune a true
// neighbour.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "grid_device.h"

void sp_set_error(const char* msg);

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

// bbox[0..2] = min xyz, bbox[3..5] = max xyz (encoded), over finite points
__global__ __launch_bounds__(kBlock) void bbox_kernel(const float4* __restrict__ pts, unsigned n, unsigned* bbox) {
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 p = pts[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            const unsigned e[3] = {enc(p.x), enc(p.y), enc(p.z)};
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
    if ((threadIdx.x & 63) == 0) {  // integer atomics: exact and order independent
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(&bbox[a], mn[a]); atomicMax(&bbox[3 + a], mx[a]); }
    }
}

__global__ __launch_bounds__(kBlock) void cell_id_kernel(const float4* __restrict__ pts, GridDesc g,
                                                         unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= g.n) return;
    const float4 p = pts[i];
    unsigned key = (unsigned)g.nx * g.ny * g.nz;  // non-finite points: a trash cell past the grid, never searched
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
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
// start[c] = first sorted position whose key >= c  (c in [0, ncells]); keys ascending
__global__ __launch_bounds__(kBlock) void cell_start_kernel(const unsigned* __restrict__ keys, unsigned n,
                                                            unsigned ncells, unsigned* __restrict__ start) {
    const unsigned c = blockIdx.x * kBlock + threadIdx.x;
    if (c > ncells) return;
    unsigned lo = 0, hi = n;
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (keys[mid] < c) lo = mid + 1;
        else hi = mid;
    }
    start[c] = lo;
}

template <int KCAP>
__global__ __launch_bounds__(kBlock) void grid_search_kernel(const float4* __restrict__ pts,
                                                             const unsigned* __restrict__ start, GridDesc g,
                                                             const float4* __restrict__ queries, unsigned nq, int k,
                                                             Mat4Arg T_val, const float* __restrict__ T_dev,
                                                             int32_t* __restrict__ idx_out,
                                                             float* __restrict__ d2_out) {
    const unsigned qi = blockIdx.x * kBlock + threadIdx.x;
    if (qi >= nq) return;
    const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
    const float4 q4 = queries[qi];
    float qx, qy, qz;
    transform_point(T, q4.x, q4.y, q4.z, qx, qy, qz);

    float bd[KCAP];
    int bi[KCAP];
#pragma unroll
    for (int i = 0; i < KCAP; ++i) { bd[i] = FLT_MAX; bi[i] = -1; }
    float kth = FLT_MAX;
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

__global__ __launch_bounds__(kBlock) void grid_search_k1_kernel(const float4* __restrict__ pts,
                                                                const unsigned* __restrict__ start, GridDesc g,
                                                                const float4* __restrict__ queries, unsigned nq,
                                                                Mat4Arg T_val, const float* __restrict__ T_dev,
                                                                int32_t* __restrict__ idx_out,
                                                                float* __restrict__ d2_out) {
    const unsigned qi = blockIdx.x * kBlock + threadIdx.x;
    if (qi >= nq) return;
    const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
    const float4 q4 = queries[qi];
    float qx, qy, qz;
    transform_point(T, q4.x, q4.y, q4.z, qx, qy, qz);
    const Nearest nn = grid_nn1_auto(pts, start, g, qx, qy, qz);
    idx_out[qi] = nn.idx;
    d2_out[qi] = nn.d2;
}

template <int KCAP>
int launch(const sp_grid* gr, const float* q, size_t nq, size_t k, const float* T, int T_dev, int32_t* idx, float* d2,
           hipStream_t st) {
    Mat4Arg tv;
    for (int i = 0; i < 16; ++i) tv.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    if (T && !T_dev)
        for (int i = 0; i < 16; ++i) tv.m[i] = T[i];
    const GridDesc g = grid_desc(gr);
    if (KCAP == 1) {
        grid_search_k1_kernel<<<div_up(nq, kBlock), kBlock, 0, st>>>(gr->d_pts, gr->d_start, g,
                                                                      reinterpret_cast<const float4*>(q), (unsigned)nq,
                                                                      tv, T_dev ? T : nullptr, idx, d2);
        return launch_status();
    }
    grid_search_kernel<KCAP><<<div_up(nq, kBlock), kBlock, 0, st>>>(gr->d_pts, gr->d_start, g,
                                                                  reinterpret_cast<const float4*>(q), (unsigned)nq,
                                                                  (int)k, tv, T_dev ? T : nullptr, idx, d2);
    return launch_status();
}

}  // namespace
}  // namespace sp

extern "C" void sp_grid_destroy(sp_grid* g) {
    if (!g) return;
    if (g->d_pts) (void)hipFree(g->d_pts);
    if (g->d_start) (void)hipFree(g->d_start);
    delete g;
}

extern "C" int sp_grid_create(const float* points, size_t n, float cell_size, float points_per_cell, void* stream,
                              sp_grid** out) {
    using namespace sp;
    if (!out) return SP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (n >= (1ull << 31)) {
        sp_set_error("[GridKNN::build] more than 2^31 points: indices are int32");
        return SP_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = as_stream(stream);
    sp_grid* g = new sp_grid();
    g->n = n;
    auto fail = [&](hipError_t e) {
        sp_set_error(hipGetErrorString(e));
        sp_grid_destroy(g);
        return SP_ERR_HIP;
    };
    hipError_t e;
    if (n == 0) {
        if ((e = hipMalloc(&g->d_start, 2 * sizeof(uint32_t))) != hipSuccess) return fail(e);
        if ((e = hipMemsetAsync(g->d_start, 0, 2 * sizeof(uint32_t), st)) != hipSuccess) return fail(e);
        *out = g;
        return SP_OK;
    }
    const float4* pts = reinterpret_cast<const float4*>(points);
    // 1. bounding box of the finite points
    unsigned* d_bbox = nullptr;
    if ((e = hipMalloc(&d_bbox, 6 * sizeof(unsigned))) != hipSuccess) return fail(e);
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    unsigned h_bbox[6];
    e = hipMemcpyAsync(d_bbox, init, sizeof init, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        bbox_kernel<<<stream_grid(n), kBlock, 0, st>>>(pts, (unsigned)n, d_bbox);
        e = hipMemcpyAsync(h_bbox, d_bbox, sizeof h_bbox, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_bbox);
    if (e != hipSuccess) return fail(e);
    float mn[3], mx[3];
    bool any = h_bbox[0] != 0xffffffffu;
    for (int a = 0; a < 3; ++a) {
        mn[a] = any ? dec(h_bbox[a]) : 0.0f;
        mx[a] = any ? dec(h_bbox[3 + a]) : 0.0f;
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
    for (;;) {
        size_t nc = 1;
        for (int a = 0; a < 3; ++a) {
            const double d = std::floor((double)ext[a] / h) + 1.0;
            g->dims[a] = (int)std::min(d, 2.0e6);
            nc *= (size_t)g->dims[a];
        }
        if (nc <= max_cells) { g->ncells = nc; break; }
        h *= 1.26f;
    }
    g->h = h;
    g->inv_h = 1.0f / h;
    g->eps = 4.0e-6f * (scale + ext_max + h);  // bounds the float rounding of (p - org) * inv_h cell assignment
    for (int a = 0; a < 3; ++a) g->org[a] = mn[a];

    // 3. sort points by cell id, gather, cell_start
    unsigned *keys_in = nullptr, *keys_out = nullptr, *vals_in = nullptr, *vals_out = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    unsigned end_bit = 1;
    while ((1ull << end_bit) <= g->ncells && end_bit < 32) ++end_bit;
    (void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, st);
    e = hipMalloc(&keys_in, n * 4);
    if (e == hipSuccess) e = hipMalloc(&keys_out, n * 4);
    if (e == hipSuccess) e = hipMalloc(&vals_in, n * 4);
    if (e == hipSuccess) e = hipMalloc(&vals_out, n * 4);
    if (e == hipSuccess) e = hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16));
    if (e == hipSuccess) e = hipMalloc(&g->d_pts, n * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&g->d_start, (g->ncells + 1) * sizeof(uint32_t));
    if (e == hipSuccess) {
        GridDesc gd{g->inv_h, g->h, g->eps, g->org[0], g->org[1], g->org[2], g->dims[0], g->dims[1], g->dims[2],
                    (unsigned)n};
        cell_id_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(pts, gd, keys_in, vals_in);
        e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, st);
    }
    if (e == hipSuccess) {
        gather_sorted_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(pts, vals_out, (unsigned)n, g->d_pts);
        cell_start_kernel<<<div_up(g->ncells + 1, kBlock), kBlock, 0, st>>>(keys_out, (unsigned)n, (unsigned)g->ncells,
                                                                          g->d_start);
        e = hipStreamSynchronize(st);
    }
    (void)hipFree(keys_in); (void)hipFree(keys_out); (void)hipFree(vals_in); (void)hipFree(vals_out); (void)hipFree(tmp);
    if (e != hipSuccess) return fail(e);
    *out = g;
    return SP_OK;
}

extern "C" size_t sp_grid_size(const sp_grid* g) { return g ? g->n : 0; }
extern "C" float sp_grid_cell_size(const sp_grid* g) { return g ? g->h : 0.0f; }

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
    if (k == 1) return launch<1>(grid, queries, nq, k, transT, transT_on_device, idx_out, d2_out, st);
    if (k <= 10) return launch<10>(grid, queries, nq, k, transT, transT_on_device, idx_out, d2_out, st);
    return launch<20>(grid, queries, nq, k, transT, transT_on_device, idx_out, d2_out, st);
}

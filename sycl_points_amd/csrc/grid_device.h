// Device-side pieces of the uniform-grid search shared by grid.hip (GridKNN) and gicp_fused.hip.
#pragma once
#include "sp_common.h"
#include "sp_math.h"

struct sp_grid {
    size_t n = 0;
    float h = 1.0f, inv_h = 1.0f, eps = 0.0f;
    float org[3] = {0, 0, 0};
    int dims[3] = {1, 1, 1};
    size_t ncells = 1;
    float4* d_pts = nullptr;      // n points in cell order, w = original index bits
    uint32_t* d_start = nullptr;  // ncells + 1 (+ 2 spare words: fast_extents loads three at a time)
    // rows + 1: first 64-query work unit of every x-row (self-kNN tiling) and their number. Made when a self-kNN first asks
    // (ensure_units, grid.hip): a grid built for Registration::align's searches never does — three launches and a read-back less
    // per build
    mutable uint32_t* d_unit_off = nullptr;
    mutable uint32_t n_units = 0;
    mutable bool units_ready = false;
    // sp_grid_create returns without waiting for the device (round 5): the build's stream, and an event behind its last kernel
    // that every OTHER stream waits for before it touches the arrays (grid_use)
    hipStream_t build_stream = nullptr;
    hipEvent_t built_ev = nullptr;
    mutable uint32_t max_cell = 0;    // points in the fullest cell, measured on first request (sp_grid_max_cell_points)
    mutable bool max_cell_known = false;
    // tuning switch (sp_internal.h): self-kNN kernel — 0 chosen by k (lane per point for k <= 10, wave-cooperative above),
    // 1 LDS-tile kernel (k <= 10), 2 wave-cooperative kernel. Results are identical.
    int self_knn_mode = 0;
    int sort_queries = 1;  // external queries (400 k or more) are searched in cell order (sp_internal.h; 0: as given)
    mutable sp::StreamSet streams;  // every stream the arrays have been handed to (sp_grid_destroy tags the pool entries)
};

namespace sp {

// Every entry point that hands the grid's arrays to a stream: note the stream (sp_grid_destroy tags the pool entries with it)
// and, when it is not the stream the grid was built on, make it wait for the build.
inline void grid_use(const sp_grid* g, hipStream_t st) {
    if (g->built_ev != nullptr && st != g->build_stream) (void)hipStreamWaitEvent(st, g->built_ev, 0);
    g->streams.note(st);
}

// grid.hip: kNN of the grid's own cell-ordered points (k <= 10), rows by grid position; enqueues only. bound2 (k > 1): only
// neighbours closer than this squared distance are looked for, the rest of a row stays padded (-1 / FLT_MAX)
int grid_search_own_points(const sp_grid* grid, size_t k, int32_t* idx_out, float* d2_out, hipStream_t st, float bound2 = FLT_MAX);

struct GridDesc {
    float inv_h, h, eps;
    float ox, oy, oz;
    int nx, ny, nz;
    unsigned n;
};

inline GridDesc grid_desc(const sp_grid* gr) {
    return GridDesc{gr->inv_h, gr->h, gr->eps, gr->org[0], gr->org[1], gr->org[2], gr->dims[0], gr->dims[1],
                    gr->dims[2], (unsigned)gr->n};
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_h, int dim) {
    const float t = floorf((v - o) * inv_h);
    return (int)fminf(fmaxf(t, 0.0f), (float)(dim - 1));
}

// (distance, index) lexicographic sorted insertion into the first k slots.
template <int KCAP>
__device__ __forceinline__ void lex_insert(float (&bd)[KCAP], int (&bi)[KCAP], int k, float d, int idx, float& kth,
                                           int& kth_idx) {
    if (KCAP == 1) {
        const bool better = d < bd[0] || (d == bd[0] && idx < bi[0]);
        bi[0] = better ? idx : bi[0];
        bd[0] = better ? d : bd[0];
        kth = bd[0];
        kth_idx = bi[0];
        return;
    }
    float cd = d;
    int ci = idx;
    bool shifting = false;
#pragma unroll
    for (int i = 0; i < KCAP; ++i) {
        if (i < k) {
            const bool sw = shifting || cd < bd[i] || (cd == bd[i] && ci < bi[i]);
            const float td = bd[i];
            const int ti = bi[i];
            const float nd = sw ? cd : td;
            const int ni = sw ? ci : ti;
            bd[i] = nd;
            bi[i] = ni;
            cd = sw ? td : cd;
            ci = sw ? ti : ci;
            shifting = sw;
            kth = nd;
            kth_idx = ni;
        }
    }
}

// squared distance from coordinate v to the interval [lo, hi], made conservative by eps
__device__ __forceinline__ float gap2(float v, float lo, float hi, float eps) {
    const float g = fmaxf(fmaxf(lo - v, v - hi) - eps, 0.0f);
    return g * g;
}

// Exact nearest neighbour (k = 1) of (qx,qy,qz) on the grid, ties to the lowest original index — the same walk as
// grid_search_kernel<1>, additionally reporting the winner's position in the cell-ordered point array and its
// coordinates (so a caller can read per-point data stored in grid order without another gather of the point).
struct Nearest {
    float d2;
    int idx;        // original index of the target point (-1: none)
    unsigned pos;   // position in grid order
    float x, y, z;  // its coordinates
};

// `seed` (optional) is an upper bound found earlier — every cell of the rings r < r_start must already have been
// searched (or be provably unable to beat the seed).
__device__ __forceinline__ Nearest grid_nn1(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                            const GridDesc& g, float qx, float qy, float qz,
                                            const Nearest* seed = nullptr, int r_start = 0) {
    Nearest best;
    best.d2 = FLT_MAX; best.idx = -1; best.pos = 0; best.x = best.y = best.z = 0.0f;
    if (!(isfinite(qx) && isfinite(qy) && isfinite(qz)) || g.n == 0) return best;
    if (seed) best = *seed;
    const int cx = cell_coord(qx, g.ox, g.inv_h, g.nx), cy = cell_coord(qy, g.oy, g.inv_h, g.ny),
              cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
    const int rmax = max(max(g.nx, g.ny), g.nz);
    auto consider = [&](const float4 p, unsigned pos) {
        const float d = dist2(qx, qy, qz, p.x, p.y, p.z);
        const int pi = __float_as_int(p.w);
        if (d < best.d2 || (d == best.d2 && pi < best.idx)) {
            best.d2 = d; best.idx = pi; best.pos = pos; best.x = p.x; best.y = p.y; best.z = p.z;
        }
    };
    for (int r = r_start; r <= rmax; ++r) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, g.nz - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, g.ny - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
        for (int z = z0; z <= z1; ++z) {
            const float dz2 = gap2(qz, g.oz + z * g.h, g.oz + (z + 1) * g.h, g.eps);
            if (dz2 > best.d2) continue;
            for (int y = y0; y <= y1; ++y) {
                const float dyz2 = dz2 + gap2(qy, g.oy + y * g.h, g.oy + (y + 1) * g.h, g.eps);
                if (dyz2 > best.d2) continue;
                const bool shell_row = (r == 0) || (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
                const unsigned row = ((unsigned)z * g.ny + y) * g.nx;
                const int nseg = shell_row ? 1 : 2;
                for (int sgi = 0; sgi < nseg; ++sgi) {
                    int xa, xb;
                    if (shell_row) { xa = x0; xb = x1; }
                    else if (sgi == 0) { xa = cx - r; xb = cx - r; if (xa < 0) continue; }
                    else { xa = cx + r; xb = cx + r; if (xb > g.nx - 1) continue; }
                    const float d2box = dyz2 + gap2(qx, g.ox + xa * g.h, g.ox + (xb + 1) * g.h, g.eps);
                    if (d2box > best.d2) continue;
                    const unsigned s = start[row + xa], e = start[row + xb + 1];
                    for (unsigned i = s; i < e; i += 4) {
                        const float4 p0 = pts[i];
                        const float4 p1 = pts[min(i + 1, e - 1)];
                        const float4 p2 = pts[min(i + 2, e - 1)];
                        const float4 p3 = pts[min(i + 3, e - 1)];
                        consider(p0, i);
                        if (i + 1 < e) consider(p1, i + 1);
                        if (i + 2 < e) consider(p2, i + 2);
                        if (i + 3 < e) consider(p3, i + 3);
                    }
                }
            }
        }
        float cov = FLT_MAX;
        if (cx - r > 0) cov = fminf(cov, qx - (g.ox + (cx - r) * g.h));
        if (cx + r < g.nx - 1) cov = fminf(cov, (g.ox + (cx + r + 1) * g.h) - qx);
        if (cy - r > 0) cov = fminf(cov, qy - (g.oy + (cy - r) * g.h));
        if (cy + r < g.ny - 1) cov = fminf(cov, (g.oy + (cy + r + 1) * g.h) - qy);
        if (cz - r > 0) cov = fminf(cov, qz - (g.oz + (cz - r) * g.h));
        if (cz + r < g.nz - 1) cov = fminf(cov, (g.oz + (cz + r + 1) * g.h) - qz);
        if (cov == FLT_MAX) break;
        cov = fmaxf(cov - g.eps, 0.0f);
        if (best.d2 < cov * cov) break;
    }
    return best;
}

// Fast path for k = 1: scan only the 2x2x2 block of cells nearest to the query (own cell plus the neighbour on the
// nearer side of each axis): 4 x-rows of at most 2 cells, 8 independent extent loads up front, no per-row pruning.
// Every point within `cov` (>= h/2 away from the grid boundary cases) of the query lies in that block, so the result
// is exact whenever the winner is closer than cov; otherwise the caller falls back to the later stages.
// The walk is split into stages (extents -> flatten -> candidate batches -> exactness) so that a caller can interleave
// the stages of several queries and keep all their loads in flight together.
struct FastFlat {           // the four x-row ranges as one candidate list
    unsigned o0, o1, o2, o3;  // position of candidate jj in range r = o_r + jj (o_r = first position - first number)
    unsigned c1, c2, c3, total;
};

// Stage 1: the block and its 8 extent loads. Along each axis, with t = fractional position of the query in its (clamped)
// cell: the block is {cell-1, cell} when t < 0.5, else {cell, cell+1} (clipped to the grid), and its nearest face that
// is not the grid boundary is max(t, 1-t) >= 0.5 cells away; cov_cells = the minimum over the axes (conservative when
// clipping removed the face). A non-finite query reads cell 0 and gets an empty list from fast_flatten.
__device__ __forceinline__ void fast_extents(const unsigned* __restrict__ start, const GridDesc& g, float qx, float qy,
                                             float qz, unsigned (&s)[4], unsigned (&e)[4], float& cov_cells) {
    const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
    const float cxf = fminf(fmaxf(floorf(fx), 0.0f), (float)(g.nx - 1));
    const float cyf = fminf(fmaxf(floorf(fy), 0.0f), (float)(g.ny - 1));
    const float czf = fminf(fmaxf(floorf(fz), 0.0f), (float)(g.nz - 1));
    const float tx = fx - cxf, ty = fy - cyf, tz = fz - czf;
    const bool lx = tx < 0.5f, ly = ty < 0.5f, lz = tz < 0.5f;
    const int cx = (int)cxf, cy = (int)cyf, cz = (int)czf;
    const int xa = max(cx - (lx ? 1 : 0), 0), xb = min(cx + (lx ? 0 : 1), g.nx - 1);
    const int ya = max(cy - (ly ? 1 : 0), 0), yb = min(cy + (ly ? 0 : 1), g.ny - 1);
    const int za = max(cz - (lz ? 1 : 0), 0), zb = min(cz + (lz ? 0 : 1), g.nz - 1);
    cov_cells = fminf(fminf(lx ? 1.0f - tx : tx, ly ? 1.0f - ty : ty), lz ? 1.0f - tz : tz);
    // extents of the (up to) four rows; a row that does not exist repeats an existing one with an empty range
    const unsigned r00 = ((unsigned)za * g.ny + ya) * g.nx + xa;
    const unsigned dy = yb != ya ? (unsigned)g.nx : 0u, dz = zb != za ? (unsigned)g.ny * g.nx : 0u;
    const unsigned w = (unsigned)(xb - xa) + 1u;
    const unsigned r01 = r00 + dy, r10 = r00 + dz, r11 = r10 + dy;
    // one 12-byte load per row (start[r], start[r + 1], start[r + 2]: the array carries two spare words) instead of two 4-byte
    // ones: the searching launches are bound by the NUMBER of scattered requests (profiles/r04_i), not by their bytes
    struct Three { unsigned a, b, c; };
    const Three v0 = *reinterpret_cast<const Three*>(start + r00), v1 = *reinterpret_cast<const Three*>(start + r01),
                v2 = *reinterpret_cast<const Three*>(start + r10), v3 = *reinterpret_cast<const Three*>(start + r11);
    const bool two = w == 2u;
    s[0] = v0.a; e[0] = two ? v0.c : v0.b;
    s[1] = v1.a; e[1] = two ? v1.c : v1.b;
    s[2] = v2.a; e[2] = two ? v2.c : v2.b;
    s[3] = v3.a; e[3] = two ? v3.c : v3.b;
    if (dy == 0u) { e[1] = s[1]; e[3] = s[3]; }
    if (dz == 0u) { e[2] = s[2]; e[3] = s[3]; }
}

// Stage 2: flatten the four ranges into one list.
__device__ __forceinline__ FastFlat fast_flatten(const unsigned (&s)[4], const unsigned (&e)[4], bool usable) {
    FastFlat f;
    const unsigned n0 = e[0] - s[0], n1 = e[1] - s[1], n2 = e[2] - s[2], n3 = e[3] - s[3];
    f.c1 = n0; f.c2 = n0 + n1; f.c3 = f.c2 + n2;
    f.o0 = s[0]; f.o1 = s[1] - f.c1; f.o2 = s[2] - f.c2; f.o3 = s[3] - f.c3;
    f.total = usable ? f.c3 + n3 : 0u;
    return f;
}
__device__ __forceinline__ unsigned fast_pos(const FastFlat& f, unsigned jj) {
    return jj + (jj < f.c1 ? f.o0 : (jj < f.c2 ? f.o1 : (jj < f.c3 ? f.o2 : f.o3)));
}

// Stage 3: candidates [base, base + B) as B INDEPENDENT 16-byte loads (one memory round trip for almost every query
// instead of one per row and per pair of points) ...
template <int B>
__device__ __forceinline__ void fast_load(const float4* __restrict__ pts, const FastFlat& f, unsigned base,
                                          float4 (&cand)[B]) {
#pragma unroll
    for (int j = 0; j < B; ++j) cand[j] = pts[fast_pos(f, min(base + j, f.total - 1))];
}
// ... and their evaluation, branch-free: the (distance, index)-lexicographic minimum is the minimum of the 64-bit key
// (distance bits : index) — distances are non-negative floats, whose bit patterns order like the values; a NaN distance
// orders above every number and never wins, as with `d < best`. `bj` tracks the winner's number in the list.
__device__ __forceinline__ unsigned long long nn_key(float d2, int idx) {
    return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)idx;
}
template <int B>
__device__ __forceinline__ void fast_eval(const FastFlat& f, unsigned base, const float4 (&cand)[B], float qx, float qy,
                                          float qz, unsigned long long& best_key, unsigned& bj) {
#pragma unroll
    for (int j = 0; j < B; ++j) {
        const float d = dist2(qx, qy, qz, cand[j].x, cand[j].y, cand[j].z);
        unsigned long long key = nn_key(d, __float_as_int(cand[j].w));
        key = (base + j < f.total) ? key : ~0ull;
        const bool better = key < best_key;
        best_key = better ? key : best_key;
        bj = better ? base + j : bj;
    }
}

// Returns true when the answer in `best` is proven exact.
// `bound2` (default: none): the caller only wants the nearest point if its squared distance is BELOW bound2 — a registration
// rejects correspondences beyond max_correspondence_distance anyway. The search then starts from that bound instead of
// FLT_MAX: a query with nothing that near is "proven" (no point, idx = -1, d2 = bound2) as soon as the scanned block covers
// the bound's ball, instead of walking rings of cells out to a neighbour it would discard. With partial overlap (the
// reference's example: 84 % of the source points have no correspondence) that walk was most of an iteration.
// `seed` (optional): a REAL point of the grid with its distance to the query — an upper bound better than any constant. The
// registration hands in a source point's previous correspondence when its reuse certificate has failed: the new nearest
// neighbour is at most that far away, which proves most answers inside the first block and prunes the rows of the later stages
// to the seed's ball instead of the caller's bound. The seed wins unless a point is nearer (or as near with a lower index).
__device__ __forceinline__ bool grid_nn1_fast(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                              const GridDesc& g, float qx, float qy, float qz, Nearest& best,
                                              float bound2 = FLT_MAX, const Nearest* seed = nullptr) {
    best.d2 = bound2; best.idx = -1; best.pos = 0; best.x = best.y = best.z = 0.0f;
    const bool usable = isfinite(qx) && isfinite(qy) && isfinite(qz) && g.n != 0;
    if (!usable) return true;
    if (seed) best = *seed;
    unsigned s[4], e[4];
    float cov_cells;
    fast_extents(start, g, qx, qy, qz, s, e, cov_cells);
    const FastFlat f = fast_flatten(s, e, true);
    const unsigned long long none = nn_key(best.d2, best.idx);  // nothing better found: {bound2, -1}, or the seed
    unsigned long long key = none;
    unsigned bj = 0;
    // Candidates per trip. The 2x2x2 block of the registration's target grid (0.5 points a cell) holds 4 points on average;
    // with 8 a trip, 2 % of the queries — so three waves in four — came back for a second, dependent trip; with 12 it is one
    // wave in fifty (slots past the end re-read the last candidate: the line the lane has just fetched). Measured on one box,
    // results bit-identical: searching launches 1-4 % shorter (GICP 1 M until converged 0.260 against 0.265 ms, point-to-distribution
    // 47.2 against 47.7 us per step); 16 a trip spills in the 1024-lane kernels (+9 us on every searching launch).
#ifndef SP_FAST_B
#define SP_FAST_B 12
#endif
    for (unsigned base = 0; base < f.total; base += SP_FAST_B) {
        float4 cand[SP_FAST_B];
        fast_load<SP_FAST_B>(pts, f, base, cand);
        fast_eval<SP_FAST_B>(f, base, cand, qx, qy, qz, key, bj);
    }
    if (key != none) {
        best.pos = fast_pos(f, bj);
        const float4 w = pts[best.pos];  // the winner again (an L1 hit): cheaper than carrying x,y,z through every compare
        best.x = w.x; best.y = w.y; best.z = w.z;
        best.idx = __float_as_int(w.w);
        best.d2 = __uint_as_float((unsigned)(key >> 32));
    }
    const float cov = fmaxf(cov_cells * g.h - g.eps, 0.0f);
    return best.d2 < cov * cov;  // strict: an unseen point at exactly this distance could win a tie
}

// Second stage for k = 1: a block of 4 cells per axis around the query (own cell, one more on the farther side, two
// on the nearer side: every face of the block is >= 1.5 h away), searched as two groups of 8 x-rows. A row is skipped
// when its (y,z) box distance exceeds the current best, and its x-range is trimmed to the cells the ball of radius
// sqrt(best) can reach; the surviving extents are fetched as one batch of independent loads, then all candidates in
// batches of BATCH (same flattening as the first stage). `best` must hold a valid upper bound or {FLT_MAX, -1}.
template <int BATCH>
__device__ __forceinline__ void grid_scan_rows8(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                                const GridDesc& g, float qx, float qy, float qz, int xlo, int xhi,
                                                int ylo, int zA, int zB, Nearest& best, int ymax = 0x7fffffff,
                                                int zmax = 0x7fffffff) {
    unsigned off[8], c[9];
    c[0] = 0;
    const float fxq = (qx - g.ox) * g.inv_h;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int y = ylo + (r & 3), z = (r & 4) ? zB : zA;
        bool ok = y >= 0 && y < g.ny && z >= 0 && z < g.nz && y <= ymax && z <= zmax;
        const int yc = min(max(y, 0), g.ny - 1), zc = min(max(z, 0), g.nz - 1);
        const float dyz2 = gap2(qz, g.oz + zc * g.h, g.oz + (zc + 1) * g.h, g.eps) +
                           gap2(qy, g.oy + yc * g.h, g.oy + (yc + 1) * g.h, g.eps);
        ok = ok && !(dyz2 > best.d2);
        // cells of this row that the ball can reach: |x - qx| <= rad, widened by 2 eps for the rounding of the cell
        // assignment and of this expression
        const float rad = (sqrtf(fmaxf(best.d2 - dyz2, 0.0f)) * 1.000001f + 2.0f * g.eps) * g.inv_h;
        const int xa = max(xlo, (int)fmaxf(floorf(fxq - rad), (float)xlo));
        const int xb = min(xhi, (int)fminf(floorf(fxq + rad), (float)xhi));
        ok = ok && xa <= xb;
        const unsigned row = ((unsigned)zc * g.ny + yc) * g.nx;
        unsigned s0 = 0, e0 = 0;
        if (ok) { s0 = start[row + xa]; e0 = start[row + xb + 1]; }
        off[r] = s0 - c[r];
        c[r + 1] = c[r] + (e0 - s0);
    }
    const unsigned total = c[8];
    for (unsigned base = 0; base < total; base += BATCH) {
        float4 cand[BATCH];
        unsigned cpos[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const unsigned jj = min(base + j, total - 1);
            unsigned pos = off[0] + jj;
#pragma unroll
            for (int r = 1; r < 8; ++r) pos = (jj >= c[r]) ? off[r] + jj : pos;
            cpos[j] = pos;
            cand[j] = pts[pos];
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const float d = dist2(qx, qy, qz, cand[j].x, cand[j].y, cand[j].z);
            const int pi = __float_as_int(cand[j].w);
            const bool valid = base + j < total;
            if (valid && (d < best.d2 || (d == best.d2 && pi < best.idx))) {
                best.d2 = d; best.idx = pi; best.pos = cpos[j]; best.x = cand[j].x; best.y = cand[j].y; best.z = cand[j].z;
            }
        }
    }
}

// Returns true when `best` is proven exact after the 4x4x4 block.
__device__ __forceinline__ bool grid_nn1_block4(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                                const GridDesc& g, float qx, float qy, float qz, Nearest& best) {
    const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
    const int cx = (int)fminf(fmaxf(floorf(fx), 0.0f), (float)(g.nx - 1));
    const int cy = (int)fminf(fmaxf(floorf(fy), 0.0f), (float)(g.ny - 1));
    const int cz = (int)fminf(fmaxf(floorf(fz), 0.0f), (float)(g.nz - 1));
    const int ux = (fx - (float)cx < 0.5f) ? 0 : 1, uy = (fy - (float)cy < 0.5f) ? 0 : 1,
              uz = (fz - (float)cz < 0.5f) ? 0 : 1;
    const int xlo = max(cx - 2 + ux, 0), xhi = min(cx + 1 + ux, g.nx - 1);
    const int ylo = cy - 2 + uy, zlo = cz - 2 + uz;  // rows outside the grid are skipped inside
    // nearer two z-layers first (they tighten the bound for the outer two)
    grid_scan_rows8<4>(pts, start, g, qx, qy, qz, xlo, xhi, ylo, zlo + 1, zlo + 2, best);
    grid_scan_rows8<4>(pts, start, g, qx, qy, qz, xlo, xhi, ylo, zlo, zlo + 3, best);
    const int yhi = min(ylo + 3, g.ny - 1), zhi = min(zlo + 3, g.nz - 1);
    const int ya = max(ylo, 0), za = max(zlo, 0);
    float cov = FLT_MAX;
    if (xlo > 0) cov = fminf(cov, qx - (g.ox + xlo * g.h));
    if (xhi < g.nx - 1) cov = fminf(cov, (g.ox + (xhi + 1) * g.h) - qx);
    if (ya > 0) cov = fminf(cov, qy - (g.oy + ya * g.h));
    if (yhi < g.ny - 1) cov = fminf(cov, (g.oy + (yhi + 1) * g.h) - qy);
    if (za > 0) cov = fminf(cov, qz - (g.oz + za * g.h));
    if (zhi < g.nz - 1) cov = fminf(cov, (g.oz + (zhi + 1) * g.h) - qz);
    if (cov == FLT_MAX) return true;
    cov = fmaxf(cov - g.eps, 0.0f);
    return best.d2 < cov * cov;
}

// Third stage, when a candidate exists but is not proven nearest (queries that are far from every point: outside the
// cloud's bounding box, or in a hole): every x-row of cells that the ball of radius sqrt(best) reaches, in groups of 4 x 2
// rows through the same batched row scan. Exact by construction (the bound is a real point, every cell that intersects the
// ball is either scanned or pruned against a bound that only shrinks) — and, unlike the ring walk, it never visits the
// rows of a shell that the ball does not touch: for a query d outside the box the ball cuts only a thin cap of cells.
// Returns false when the ball spans more than 16 rows in y or z (left to the ring walk).
__device__ __forceinline__ bool grid_nn1_ball(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                              const GridDesc& g, float qx, float qy, float qz, Nearest& best) {
    const float rad = (sqrtf(best.d2) * 1.000001f + 2.0f * g.eps) * g.inv_h;  // in cells, widened like the row trimming
    const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
    auto lo_cell = [](float v, int n) { return (int)fminf(fmaxf(floorf(v), 0.0f), (float)(n - 1)); };
    const int x0 = lo_cell(fx - rad, g.nx), x1 = lo_cell(fx + rad, g.nx);
    const int y0 = lo_cell(fy - rad, g.ny), y1 = lo_cell(fy + rad, g.ny);
    const int z0 = lo_cell(fz - rad, g.nz), z1 = lo_cell(fz + rad, g.nz);
    if (y1 - y0 >= 16 || z1 - z0 >= 16) return false;
    for (int z = z0; z <= z1; z += 2)
        for (int y = y0; y <= y1; y += 4) {
            const int yb = min(y + 3, y1), zb = min(z + 1, z1);
            const float d = gap2(qy, g.oy + y * g.h, g.oy + (yb + 1) * g.h, g.eps) +
                            gap2(qz, g.oz + z * g.h, g.oz + (zb + 1) * g.h, g.eps);
            if (d > best.d2) continue;  // the whole 4 x 2 group of rows is out of reach
            grid_scan_rows8<4>(pts, start, g, qx, qy, qz, x0, x1, y, z, z + 1, best, y1, z1);
        }
    return true;
}

// The third stage with the WHOLE WAVE on one query at a time (all 64 lanes must call it; `open` marks the lanes whose query
// is still unproven after the 4x4x4 block, `best` their upper bound). A lane that scans its ball alone walks every row and
// every candidate serially — on a cloud of surfaces (tens of points per cell) or with a bound of several cells (a source point
// with nothing within max_correspondence_distance) that is thousands of dependent steps, and the wave, the workgroup and, in
// the device-resident optimiser loop, the whole step wait for it. Here the rows of the ball are dealt to the lanes (row r to
// lane r mod 64): each lane prunes and trims its rows against the query's bound, scans their candidates, and a wave-wide
// minimum of (distance, index) picks the winner. Exact like grid_nn1_ball: every cell the ball intersects is scanned or pruned
// against a valid upper bound; the result is the (distance, index)-lexicographic minimum, whoever finds it.
// Any number of rows (no fall-back to the ring walk).
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o, 64);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ void grid_nn1_ball_wave(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                                   const GridDesc& g, bool open, float qx, float qy, float qz, Nearest& best) {
    unsigned long long todo = __ballot(open);
    const int lane = (int)(threadIdx.x & 63u);
    // (distance bits : index + 2^31): orders like `d < best || (d == best && idx < best_idx)` with idx = -1 below every index
    auto key_of = [](float d2, int idx) { return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)(idx + 0x80000000); };
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const float ux = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), src));
        const float uy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), src));
        const float uz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), src));
        const float ub = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best.d2), src));
        const int ui = __builtin_amdgcn_readlane(best.idx, src);
        const float rad = (sqrtf(ub) * 1.000001f + 2.0f * g.eps) * g.inv_h;  // in cells, widened like the row trimming
        const float fx = (ux - g.ox) * g.inv_h, fy = (uy - g.oy) * g.inv_h, fz = (uz - g.oz) * g.inv_h;
        auto lo_cell = [](float v, int n) { return (int)fminf(fmaxf(floorf(v), 0.0f), (float)(n - 1)); };
        const int x0 = lo_cell(fx - rad, g.nx), x1 = lo_cell(fx + rad, g.nx);
        const int y0 = lo_cell(fy - rad, g.ny), y1 = lo_cell(fy + rad, g.ny);
        const int z0 = lo_cell(fz - rad, g.nz), z1 = lo_cell(fz + rad, g.nz);
        const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        const unsigned long long key0 = key_of(ub, ui);
        unsigned long long key = key0;
        unsigned bpos = 0;
        for (int r = lane; r < nrows; r += 64) {
            const int y = y0 + r % ny, z = z0 + r / ny;
            const float dyz2 = gap2(uy, g.oy + y * g.h, g.oy + (y + 1) * g.h, g.eps) + gap2(uz, g.oz + z * g.h, g.oz + (z + 1) * g.h, g.eps);
            if (dyz2 > ub) continue;
            const float radx = (sqrtf(fmaxf(ub - dyz2, 0.0f)) * 1.000001f + 2.0f * g.eps) * g.inv_h;
            const int xa = max(x0, (int)fmaxf(floorf(fx - radx), (float)x0));
            const int xb = min(x1, (int)fminf(floorf(fx + radx), (float)x1));
            if (xa > xb) continue;
            const unsigned row = ((unsigned)z * g.ny + y) * g.nx;
            const unsigned s = start[row + xa], e = start[row + xb + 1];
            for (unsigned i = s; i < e; i += 4) {
                float4 c[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = pts[min(i + j, e - 1)];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long kj = key_of(dist2(ux, uy, uz, c[j].x, c[j].y, c[j].z), __float_as_int(c[j].w));
                    const bool better = i + j < e && kj < key;
                    key = better ? kj : key;
                    bpos = better ? i + j : bpos;
                }
            }
        }
        const unsigned long long m = wave_min_u64(key);
        if (m != key0) {  // something beat the bound (uniform over the wave)
            const int wl = __ffsll((long long)__ballot(key == m)) - 1;
            const unsigned wpos = (unsigned)__builtin_amdgcn_readlane((int)bpos, wl);
            if (lane == src) {
                const float4 w = pts[wpos];
                best.d2 = __uint_as_float((unsigned)(m >> 32));
                best.idx = __float_as_int(w.w);
                best.pos = wpos;
                best.x = w.x; best.y = w.y; best.z = w.z;
            }
        }
    }
}

// ONE query, the whole wave (all 64 lanes call it with the SAME query and the same `best`: a bound — {d2, idx = -1}: only points
// nearer than d2 count — or a real point, e.g. the previous winner). The ball of the bound is scanned like grid_nn1_ball_wave
// scans it, but here the CANDIDATES are dealt to the lanes: lane r finds the extent of row r of the ball (pruned and trimmed
// against the bound), then the rows are taken four at a time and lane l evaluates candidate l of each — for a bound of a
// fraction of a cell that is one round trip for the extents and one for the points, whatever the cells hold (a lane scanning
// its own 2x2x2 block walks every candidate in batches of eight: on a cloud of surfaces, tens of points per cell, that chain
// is what a small alignment waits for). `best` comes back uniform: the (distance, index)-lexicographic minimum over the ball,
// or unchanged when nothing beats it. Exact for the reason grid_nn1_ball is.
__device__ __forceinline__ void grid_nn1_query_wave(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                                    const GridDesc& g, float ux, float uy, float uz, float ub, Nearest& best,
                                                    float& second2) {
    // ub: the squared radius of the ball that is scanned (every target with d^2 <= ub lies in a scanned cell).
    // best (in): a real point inside the ball — the previous winner, best.d2 <= ub — or {ub, -1}: only points nearer than ub count.
    // best (out): the (distance, index)-lexicographic minimum over the ball, or unchanged when nothing qualifies.
    // second2: a lower bound of the squared distance of every OTHER target: the runner-up of the scan, or ub when the scan saw no
    // second point nearer than its own radius (what lies outside the ball is farther than that). The margin certificate of the
    // wave-per-point path is built on it (registration_device.h, fused_query_wave).
    const int lane = (int)(threadIdx.x & 63u);
    auto key_of = [](float d2, int idx) { return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)(idx + 0x80000000); };
    const float rad = (sqrtf(ub) * 1.000001f + 2.0f * g.eps) * g.inv_h;  // in cells, widened like the row trimming
    const float fx = (ux - g.ox) * g.inv_h, fy = (uy - g.oy) * g.inv_h, fz = (uz - g.oz) * g.inv_h;
    auto lo_cell = [](float v, int n) { return (int)fminf(fmaxf(floorf(v), 0.0f), (float)(n - 1)); };
    const int x0 = lo_cell(fx - rad, g.nx), x1 = lo_cell(fx + rad, g.nx);
    const int y0 = lo_cell(fy - rad, g.ny), y1 = lo_cell(fy + rad, g.ny);
    const int z0 = lo_cell(fz - rad, g.nz), z1 = lo_cell(fz + rad, g.nz);
    const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
    constexpr unsigned long long kNone = ~0ull;
    unsigned long long k1 = kNone, k2 = kNone;  // the lane's two nearest candidates
    unsigned bpos = 0;
    for (int r0 = 0; r0 < nrows; r0 += 64) {
        const int r = r0 + lane;
        unsigned s = 0, e = 0;
        if (r < nrows) {
            const int y = y0 + r % ny, z = z0 + r / ny;
            const float dyz2 = gap2(uy, g.oy + y * g.h, g.oy + (y + 1) * g.h, g.eps) + gap2(uz, g.oz + z * g.h, g.oz + (z + 1) * g.h, g.eps);
            if (!(dyz2 > ub)) {
                const float radx = (sqrtf(fmaxf(ub - dyz2, 0.0f)) * 1.000001f + 2.0f * g.eps) * g.inv_h;
                const int xa = max(x0, (int)fmaxf(floorf(fx - radx), (float)x0));
                const int xb = min(x1, (int)fminf(floorf(fx + radx), (float)x1));
                if (xa <= xb) {
                    const unsigned row = ((unsigned)z * g.ny + y) * g.nx;
                    s = start[row + xa];
                    e = start[row + xb + 1];
                }
            }
        }
        unsigned long long todo = __ballot(e > s);
        while (todo) {  // (uniform) four rows at a time, candidate `lane` (+ 64, ...) of each
            unsigned sj[4], ej[4], len = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sj[j] = ej[j] = 0u;
                if (todo) {
                    const int l = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    sj[j] = (unsigned)__builtin_amdgcn_readlane((int)s, l);
                    ej[j] = (unsigned)__builtin_amdgcn_readlane((int)e, l);
                    len = max(len, ej[j] - sj[j]);
                }
            }
            for (unsigned base = 0; base < len; base += 64) {
                float4 c[4];
                unsigned pj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pj[j] = sj[j] + base + (unsigned)lane;
                    c[j] = pts[pj[j] < ej[j] ? pj[j] : sj[j]];  // (sj = 0 for a slot without a row: a valid address, never used)
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long kj = key_of(dist2(ux, uy, uz, c[j].x, c[j].y, c[j].z), __float_as_int(c[j].w));
                    const bool valid = pj[j] < ej[j];
                    const bool first = valid && kj < k1, runner = valid && !first && kj < k2;
                    k2 = first ? k1 : (runner ? kj : k2);
                    k1 = first ? kj : k1;
                    bpos = first ? pj[j] : bpos;
                }
            }
        }
    }
    const unsigned long long m = wave_min_u64(k1);
    // the key to beat: the point handed in (it lies inside the ball, so the scan met it itself), or the bare bound
    const unsigned long long key_in = best.idx >= 0 ? key_of(best.d2, best.idx) : key_of(ub, -1);
    const unsigned long long winner = m < key_in ? m : (best.idx >= 0 ? key_in : kNone);  // kNone: nothing qualifies
    const unsigned long long runner_up = wave_min_u64(k1 == winner ? k2 : k1);  // (each target once: the winner's own entry is left out)
    second2 = ub;
    if (winner != kNone && runner_up != kNone) second2 = fminf(ub, __uint_as_float((unsigned)(runner_up >> 32)));
    if (m < key_in) {
        const int wl = __ffsll((long long)__ballot(k1 == m)) - 1;
        const unsigned wpos = (unsigned)__builtin_amdgcn_readlane((int)bpos, wl);
        const float4 w = pts[wpos];
        best.d2 = __uint_as_float((unsigned)(m >> 32));
        best.idx = __float_as_int(w.w);
        best.pos = wpos;
        best.x = w.x; best.y = w.y; best.z = w.z;
    }
}

// Stages after an inexact fast result (`best` holds its upper bound or {FLT_MAX, -1}).
__device__ __forceinline__ void grid_nn1_later_stages(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                                      const GridDesc& g, float qx, float qy, float qz, Nearest& best) {
    if (grid_nn1_block4(pts, start, g, qx, qy, qz, best)) return;
    if (best.d2 < FLT_MAX && grid_nn1_ball(pts, start, g, qx, qy, qz, best)) return;  // a point's or the caller's bound
    // the block covers the rings r <= 1 of the own cell completely
    const Nearest seed = best;
    best = grid_nn1(pts, start, g, qx, qy, qz, &seed, 2);
}

__device__ __forceinline__ Nearest grid_nn1_auto(const float4* __restrict__ pts, const unsigned* __restrict__ start,
                                                 const GridDesc& g, float qx, float qy, float qz, float bound2 = FLT_MAX,
                                                 const Nearest* seed = nullptr) {
    Nearest best;
    if (!grid_nn1_fast(pts, start, g, qx, qy, qz, best, bound2, seed)) grid_nn1_later_stages(pts, start, g, qx, qy, qz, best);
    return best;
}

}  // namespace sp

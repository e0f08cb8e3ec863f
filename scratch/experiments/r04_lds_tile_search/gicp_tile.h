// Exact nearest-neighbour search of a workgroup's queries against an LDS-staged tile of the target grid — the search of the
// per-iteration GICP kernel while many correspondences still change (the first launches of an alignment).
// Replaces, for those launches, the per-lane walk through global memory (grid_device.h: 2x2x2 block -> 4x4x4 block -> ball ->
// rings; the reference: KDTree::knn_search_async k = 1, algorithms/knn/kdtree.hpp:463-553, from Registration::align,
// algorithms/registration/registration.hpp:229-234).
//
// Why: by the counters a searching launch was bound by vector-instruction issue and by scattered 16-byte loads (44 address
// cycles per wave load, ~39 loads and ~14 dependent round trips per 64 queries), and whole waves walked the later stages for
// the 41 % of their lanes the first block did not prove (profiles/README.md, r03_search_launch_experiments_not_kept.txt).
// Here the 1024 queries of a workgroup pass are spatial neighbours (sp_gicp_source_prepare puts the source into blocks of
// target cells), so the cells any of them can need form one small box:
//   1. bounding box of the queries' cells (+ kTileMargin cells)  ->  its x-rows are contiguous runs of the cell-ordered
//      target: cell extents and points are copied to LDS with coalesced loads, ONE round trip each for the whole workgroup;
//   2. every query scans the 3x3x3 cells around its own cell in LDS (ds_read latency instead of an HBM/L2 round trip per
//      step): exact when the winner is nearer than the block's nearest inner face (88 % of uniform queries at 0.5 points
//      per cell);
//   3. the unproven queries are COMPACTED into a queue and scanned over 5x5x5 cells by groups of G lanes (G = 1..8 chosen so
//      that all 16 waves work): no wave walks a later stage for a handful of its lanes;
//   4. what is still unproven (a query with no target within two cells: holes, outside the cloud) finishes with the
//      ring walk through global memory, seeded with the bound found so far.
// A pass whose box does not fit the tile is split into halves of its lanes (down to 128); a part that still does not fit
// searches through global memory as before. The answer is the exact nearest neighbour with ties to the lowest original index
// in every case, so it is bit-identical to grid_nn1_auto / the brute-force search (tests/test_gpu_tile_search.py).
#pragma once
#include "grid_device.h"

namespace sp {

constexpr int kTileThreads = 1024;
constexpr int kTilePts = 4608;      // target points a tile can hold (72 KB)
constexpr int kTileStarts = 8192;   // cell-extent entries: rows * (cells per row + 1) (32 KB)
constexpr int kTileRows = 1024;     // x-rows per tile: one lane each in the offset scan
constexpr int kTileMargin = 2;      // cells around the queries' own cells: the 5x5x5 block of every query lies inside
constexpr int kTileMinSeg = 128;    // smallest part of a pass that gets a tile of its own

struct TileLds {
    float4 pts[kTilePts];             // staged target points, row after row; w = original index bits
    unsigned start[kTileStarts];      // [row][x]: LDS position of the first point of the cell
    unsigned rowoff[kTileRows + 1];   // LDS position of a row's first point (exclusive scan of the row lengths; [rows] = total)
    unsigned rowdelta[kTileRows];     // LDS position = global position + rowdelta[row] (mod 2^32)
    float4 q[kTileThreads];           // queries of the queued stages, by owner lane
    unsigned long long key[kTileThreads];  // best (distance, index) of an owner's query so far
    unsigned gpos[kTileThreads];      // its position in the grid-ordered target
    unsigned queue[kTileThreads];     // owner lanes whose query is to be scanned in the current stage
    int bbox[8];                      // min x, y, z and min of the negated max x, y, z
    unsigned wsum[kTileThreads / kWave];
    unsigned qcount[2];
};

// wave-wide minimum delivered to lane 63 on the DPP path (the same tree as wave_sum_to_lane63)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_intmax(int x) {
    return __builtin_amdgcn_update_dpp(0x7fffffff, x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_min_to_lane63(int v) {
    v = min(v, dpp_or_intmax<0x111, 0xf>(v));
    v = min(v, dpp_or_intmax<0x112, 0xf>(v));
    v = min(v, dpp_or_intmax<0x114, 0xf>(v));
    v = min(v, dpp_or_intmax<0x118, 0xf>(v));
    v = min(v, dpp_or_intmax<0x142, 0xa>(v));
    v = min(v, dpp_or_intmax<0x143, 0xc>(v));
    return v;
}
// inclusive prefix sum over the wave (every lane), same DPP sequence
__device__ __forceinline__ unsigned wave_inclusive_scan_u32(unsigned v) {
    v += (unsigned)dpp_or_zero<0x111, 0xf>((int)v);
    v += (unsigned)dpp_or_zero<0x112, 0xf>((int)v);
    v += (unsigned)dpp_or_zero<0x114, 0xf>((int)v);
    v += (unsigned)dpp_or_zero<0x118, 0xf>((int)v);
    v += (unsigned)dpp_or_zero<0x142, 0xa>((int)v);
    v += (unsigned)dpp_or_zero<0x143, 0xc>((int)v);
    return v;
}
// a / d for a < 2^21, d >= 1 with inv_d = 1.0f / d: (a + 0.5) / d is at least 0.5 / d away from every integer, the float
// product is off by less than a * 2^-22
__device__ __forceinline__ unsigned udiv_small(unsigned a, float inv_d) { return (unsigned)(((float)a + 0.5f) * inv_d); }

struct TileBox {  // the staged box of cells (uniform over the workgroup)
    int x0, y0, z0;    // first cell
    int X, Y, Z;       // cells per axis
};

// The block of half-width R cells around the query's own cell has been scanned completely: is `d2` proven minimal?
// (the same coverage rule as grid_nn1's ring loop: the nearest face of the block that is not the grid's boundary)
__device__ __forceinline__ bool tile_block_proves(const GridDesc& g, float qx, float qy, float qz, int cx, int cy, int cz,
                                                  int R, float d2) {
    float cov = FLT_MAX;
    if (cx - R > 0) cov = fminf(cov, qx - (g.ox + (cx - R) * g.h));
    if (cx + R < g.nx - 1) cov = fminf(cov, (g.ox + (cx + R + 1) * g.h) - qx);
    if (cy - R > 0) cov = fminf(cov, qy - (g.oy + (cy - R) * g.h));
    if (cy + R < g.ny - 1) cov = fminf(cov, (g.oy + (cy + R + 1) * g.h) - qy);
    if (cz - R > 0) cov = fminf(cov, qz - (g.oz + (cz - R) * g.h));
    if (cz + R < g.nz - 1) cov = fminf(cov, (g.oz + (cz + R + 1) * g.h) - qz);
    if (cov == FLT_MAX) return true;  // the block is the whole grid
    cov = fmaxf(cov - g.eps, 0.0f);
    return d2 < cov * cov;  // strict: an unseen point at exactly this distance could win a tie
}

// One stage: the `count` owners in L.queue are scanned over the block of half-width R around their own cell, G lanes per
// query (G a power of two <= 32; the rows of the block are dealt to the G lanes). Results go to L.key / L.gpos of
// the owner. Uniform control flow; the caller brackets it with barriers.
template <int R>
__device__ __forceinline__ void tile_scan_stage(TileLds& L, const GridDesc& g, const TileBox& B, unsigned count, unsigned G) {
    constexpr int W = 2 * R + 1, NROWS = W * W;
    const unsigned gshift = 31u - (unsigned)__builtin_clz(G);
    const unsigned work = count << gshift;
    for (unsigned w0 = 0; w0 < work; w0 += kTileThreads) {
        const unsigned w = w0 + threadIdx.x;
        const unsigned entry = w >> gshift, sub = w & (G - 1u);
        const bool active = entry < count;
        unsigned owner = 0;
        unsigned long long key = ~0ull;
        unsigned gpos = 0;
        if (active) {
            owner = L.queue[entry];
            const float4 q = L.q[owner];
            key = L.key[owner];
            gpos = L.gpos[owner];
            const int cx = cell_coord(q.x, g.ox, g.inv_h, g.nx), cy = cell_coord(q.y, g.oy, g.inv_h, g.ny),
                      cz = cell_coord(q.z, g.oz, g.inv_h, g.nz);
            const int xa = max(cx - R, 0), xb = min(cx + R, g.nx - 1);
            for (int ri = (int)sub; ri < NROWS; ri += (int)G) {
                const int dz = (R == 1) ? ((ri * 11) >> 5) : ((ri * 13) >> 6);  // ri / W for ri < W * W
                const int dy = ri - dz * W;
                const int y = cy + dy - R, z = cz + dz - R;
                if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
                const float dyz2 = gap2(q.z, g.oz + z * g.h, g.oz + (z + 1) * g.h, g.eps) +
                                   gap2(q.y, g.oy + y * g.h, g.oy + (y + 1) * g.h, g.eps);
                if (dyz2 > __uint_as_float((unsigned)(key >> 32))) continue;  // the row cannot hold a nearer point
                const unsigned r = (unsigned)((z - B.z0) * B.Y + (y - B.y0));
                const unsigned base = r * (unsigned)(B.X + 1) - (unsigned)B.x0;
                const unsigned delta = L.rowdelta[r];
                const unsigned s = L.start[base + xa], e = L.start[base + xb + 1];
                for (unsigned p = s; p < e; p += 2) {
                    const float4 c0 = L.pts[p];
                    const float4 c1 = L.pts[min(p + 1, e - 1)];
                    const unsigned long long k0 = nn_key(dist2(q.x, q.y, q.z, c0.x, c0.y, c0.z), __float_as_int(c0.w));
                    unsigned long long k1 = nn_key(dist2(q.x, q.y, q.z, c1.x, c1.y, c1.z), __float_as_int(c1.w));
                    k1 = (p + 1 < e) ? k1 : ~0ull;
                    const bool b0 = k0 < key;
                    key = b0 ? k0 : key;
                    gpos = b0 ? p - delta : gpos;
                    const bool b1 = k1 < key;
                    key = b1 ? k1 : key;
                    gpos = b1 ? p + 1 - delta : gpos;
                }
            }
        }
        // minimum over the G lanes of a group (groups are aligned runs of lanes inside a wave; idle lanes hold ~0)
        for (unsigned m = 1; m < G; m <<= 1) {
            const unsigned ohi = (unsigned)__shfl_xor((int)(key >> 32), (int)m, 64);
            const unsigned olo = (unsigned)__shfl_xor((int)(unsigned)key, (int)m, 64);
            const unsigned opos = (unsigned)__shfl_xor((int)gpos, (int)m, 64);
            const unsigned long long ok = ((unsigned long long)ohi << 32) | olo;
            const bool b = ok < key;
            key = b ? ok : key;
            gpos = b ? opos : gpos;
        }
        if (active && sub == 0) {
            L.key[owner] = key;
            L.gpos[owner] = gpos;
        }
    }
}

// Stage 1 when most lanes of the workgroup have a query: every lane scans the 3x3x3 cells around its own query's cell itself
// (no queue, no owner indirection). All 27 extent / offset reads are issued together; a row outside the grid is an empty range.
__device__ __forceinline__ void tile_scan_own3(const TileLds& L, const GridDesc& g, const TileBox& B, float qx, float qy, float qz,
                                               int cx, int cy, int cz, unsigned long long& key, unsigned& gpos) {
    const int w1 = B.X + 1;
    const int xa = max(cx - 1, 0) - B.x0, xb1 = min(cx + 1, g.nx - 1) + 1 - B.x0;
    const int rc = (cz - B.z0) * B.Y + (cy - B.y0);
    unsigned s[9], e[9], d[9];
#pragma unroll
    for (int ri = 0; ri < 9; ++ri) {
        const int dz = ri / 3 - 1, dy = ri % 3 - 1;
        const bool ok = (unsigned)(cy + dy) < (unsigned)g.ny && (unsigned)(cz + dz) < (unsigned)g.nz;
        const int r = ok ? rc + dz * B.Y + dy : rc;
        s[ri] = L.start[r * w1 + xa];
        e[ri] = L.start[r * w1 + xb1];
        d[ri] = L.rowdelta[r];
        e[ri] = ok ? e[ri] : s[ri];
    }
    unsigned lbest = 0xffffffffu, dbest = 0u;
#pragma unroll
    for (int ri = 0; ri < 9; ++ri) {
        const unsigned before = lbest;
        for (unsigned p = s[ri]; p < e[ri]; ++p) {
            const float4 c = L.pts[p];
            const unsigned long long k = nn_key(dist2(qx, qy, qz, c.x, c.y, c.z), __float_as_int(c.w));
            const bool b = k < key;
            key = b ? k : key;
            lbest = b ? p : lbest;
        }
        dbest = lbest != before ? d[ri] : dbest;
    }
    if (lbest != 0xffffffffu) gpos = lbest - dbest;
}

// Slot in L.queue for every lane with `push` set: one LDS atomic per wave. Returns nothing; L.qcount[which] counts.
__device__ __forceinline__ void tile_queue_push(TileLds& L, int which, bool push) {
    const unsigned long long m = __ballot(push);
    if (m == 0ull) return;  // wave-uniform
    const unsigned lane = threadIdx.x & (kWave - 1);
    const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    unsigned base = 0;
    if (lane == (unsigned)__builtin_ctzll(m)) base = atomicAdd(&L.qcount[which], (unsigned)__builtin_popcountll(m));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
    if (push) L.queue[base + before] = threadIdx.x;
}

// lanes per query: as many as keep all 16 waves busy, at most `cap` (a power of two <= 32: a group never leaves its wave)
__device__ __forceinline__ unsigned tile_group_width(unsigned count, unsigned cap) {
    unsigned G = 1;
    while (G < cap && count * (G << 1) <= (unsigned)kTileThreads) G <<= 1;
    return G;
}

// The search of one workgroup pass. Every lane of the workgroup calls it (barriers inside); `need`: this lane has a query
// (finite coordinates) to be answered. On return, for a lane with `need`:
//   final == true : key / gpos hold the exact answer (key = nn_key(bound2, -1): nothing nearer than the bound)
//   final == false: key / gpos hold the best candidate found so far (or the bound), `rings_done` says which rings of cells
//                   around the query's own cell have been covered completely (-1: none): the caller finishes through global
//                   memory.
// `seg_hint` (in/out, uniform): the part size that fitted last time.
#ifdef SP_TILE_DEBUG
#define TILE_STAMP(slot) do { if (dbg && tid == 0 && blockIdx.x == 3) { const unsigned long long t_ = wall_clock64(); atomicAdd(dbg + 8 + (slot), (unsigned)(t_ - t_prev)); t_prev = t_; } } while (0)
#else
#define TILE_STAMP(slot) do {} while (0)
#endif
__device__ __forceinline__ void tile_search_pass(TileLds& L, const float4* __restrict__ tpts, const unsigned* __restrict__ tstart,
                                                 const GridDesc& g, bool need, float qx, float qy, float qz, float bound2,
                                                 unsigned& seg_hint, unsigned long long& key, unsigned& gpos, bool& final,
                                                 int& rings_done, unsigned* dbg = nullptr) {
    const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const unsigned long long none = nn_key(bound2, -1);
    key = none;
    gpos = 0;
    final = !need;
    rings_done = -1;
    int cx = 0, cy = 0, cz = 0;
    if (need) {
        cx = cell_coord(qx, g.ox, g.inv_h, g.nx);
        cy = cell_coord(qy, g.oy, g.inv_h, g.ny);
        cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
    }
    unsigned lo = 0, len = seg_hint;
#ifdef SP_TILE_DEBUG
    unsigned long long t_prev = wall_clock64();
#endif
    while (lo < (unsigned)kTileThreads) {  // uniform
        const bool mine = need && tid >= lo && tid < lo + len;
        // ---- bounding box of the part's cells
        if (tid < 8) L.bbox[tid] = 0x7fffffff;
        if (tid == 8) { L.qcount[0] = 0u; L.qcount[1] = 0u; }
        __syncthreads();
        {
            const int big = 0x7fffffff;
            const int m0 = wave_min_to_lane63(mine ? cx : big), m1 = wave_min_to_lane63(mine ? cy : big),
                      m2 = wave_min_to_lane63(mine ? cz : big), m3 = wave_min_to_lane63(mine ? -cx : big),
                      m4 = wave_min_to_lane63(mine ? -cy : big), m5 = wave_min_to_lane63(mine ? -cz : big);
            if (lane == kWave - 1 && m0 != big) {
                atomicMin(&L.bbox[0], m0); atomicMin(&L.bbox[1], m1); atomicMin(&L.bbox[2], m2);
                atomicMin(&L.bbox[3], m3); atomicMin(&L.bbox[4], m4); atomicMin(&L.bbox[5], m5);
            }
        }
        __syncthreads();
        TILE_STAMP(0);
        if (L.bbox[0] == 0x7fffffff) {  // nobody in this part has a query
            lo += len;
            continue;
        }
        TileBox B;
        B.x0 = max(L.bbox[0] - kTileMargin, 0);
        B.y0 = max(L.bbox[1] - kTileMargin, 0);
        B.z0 = max(L.bbox[2] - kTileMargin, 0);
        B.X = min(-L.bbox[3] + kTileMargin, g.nx - 1) - B.x0 + 1;
        B.Y = min(-L.bbox[4] + kTileMargin, g.ny - 1) - B.y0 + 1;
        B.Z = min(-L.bbox[5] + kTileMargin, g.nz - 1) - B.z0 + 1;
        const unsigned long long rows64 = (unsigned long long)B.Y * (unsigned long long)B.Z;
        const unsigned long long nstarts64 = rows64 * (unsigned long long)(B.X + 1);
        bool fits = rows64 <= (unsigned long long)kTileRows && nstarts64 <= (unsigned long long)kTileStarts;
        const unsigned rows = (unsigned)rows64, nstarts = (unsigned)nstarts64;
        if (fits) {
            // ---- first round trip: every row's extent in the cell table (one lane per row) AND the box's cell extents (rows
            // padded to a power of two of entries so that (row, x) come out of the lane number by a shift), all in flight together
            const float inv_y = 1.0f / (float)B.Y;
            const unsigned cell0 = (unsigned)(((size_t)B.z0 * g.ny + (size_t)B.y0) * g.nx + (size_t)B.x0);
            const unsigned zstride = (unsigned)g.ny * (unsigned)g.nx;
            auto row_cell = [&](unsigned r) {  // cell-table index of the first cell of tile row r
                const unsigned zi = udiv_small(r, inv_y), yi = r - zi * (unsigned)B.Y;
                return cell0 + zi * zstride + yi * (unsigned)g.nx;
            };
            unsigned gs = 0, rl = 0;
            if (tid < rows) {
                const unsigned rb = row_cell(tid);
                gs = tstart[rb];
                rl = tstart[rb + (unsigned)B.X] - gs;
            }
            const unsigned xs = 32u - (unsigned)__builtin_clz((unsigned)B.X);  // 2^xs >= X + 1
            const unsigned xmask = (1u << xs) - 1u, slots = rows << xs;
            constexpr unsigned kSB = 10;  // extent loads per lane in the first batch (10240 slots)
            unsigned v[kSB];
#pragma unroll
            for (unsigned k = 0; k < kSB; ++k) {  // (unconditional loads, clamped: a conditional one would be waited for at once)
                const unsigned sl = tid + k * kTileThreads, r = min(sl >> xs, rows - 1u), xi = min(sl & xmask, (unsigned)B.X);
                v[k] = tstart[row_cell(r) + xi];
            }
            const unsigned incl = wave_inclusive_scan_u32(rl);
            if (lane == kWave - 1) L.wsum[wave] = incl;
            __syncthreads();
            TILE_STAMP(1);
            unsigned before = 0, total = 0;
#pragma unroll
            for (unsigned w2 = 0; w2 < (unsigned)kTileThreads / kWave; ++w2) {
                const unsigned wv = L.wsum[w2];
                before += w2 < wave ? wv : 0u;
                total += wv;
            }
            fits = total <= (unsigned)kTilePts;
            if (fits) {
                if (tid < rows) {
                    L.rowoff[tid] = before + incl - rl;
                    L.rowdelta[tid] = (before + incl - rl) - gs;
                }
                if (tid == 0) L.rowoff[rows] = total;
                __syncthreads();
                TILE_STAMP(2);
                // ---- second round trip: the points, 16 lanes along every row (no search for the row of a point)
                const unsigned pslots = rows << 4;
                for (unsigned sl0 = 0; sl0 < pslots; sl0 += 5u * kTileThreads) {
                    float4 pv[5];
                    unsigned dst[5];
#pragma unroll
                    for (unsigned j = 0; j < 5; ++j) {  // (unconditional loads: an idle lane re-reads its row's first point)
                        const unsigned sl = sl0 + tid + j * kTileThreads, r = min(sl >> 4, rows - 1u), i = sl & 15u;
                        const unsigned off = L.rowoff[r], cnt = L.rowoff[r + 1] - off, del = L.rowdelta[r];
                        const bool ok = sl < pslots && i < cnt;
                        dst[j] = ok ? off + i : 0xffffffffu;
                        pv[j] = tpts[min(off + (ok ? i : 0u), total - 1u) - del];
                    }
#pragma unroll
                    for (unsigned j = 0; j < 5; ++j)
                        if (dst[j] != 0xffffffffu) L.pts[dst[j]] = pv[j];
                    TILE_STAMP(6);
#pragma unroll 1
                    for (unsigned j = 0; j < 5; ++j) {  // (rows of more than 16 points)
                        const unsigned sl = sl0 + tid + j * kTileThreads, r = sl >> 4;
                        if (sl < pslots) {
                            const unsigned off = L.rowoff[r], cnt = L.rowoff[r + 1] - off, del = L.rowdelta[r];
                            for (unsigned i = (sl & 15u) + 16u; i < cnt; i += 16u) L.pts[off + i] = tpts[off + i - del];
                        }
                    }
                }
                TILE_STAMP(7);
                // ---- the cell extents as LDS positions
#pragma unroll
                for (unsigned k = 0; k < kSB; ++k) {
                    const unsigned sl = tid + k * kTileThreads, r = sl >> xs, xi = sl & xmask;
                    if (sl < slots && xi <= (unsigned)B.X) L.start[r * (unsigned)(B.X + 1) + xi] = v[k] + L.rowdelta[r];
                }
                for (unsigned sl0 = kSB * kTileThreads; sl0 < slots; sl0 += kTileThreads) {  // (wide, flat boxes)
                    const unsigned sl = sl0 + tid, r = sl >> xs, xi = sl & xmask;
                    if (sl < slots && xi <= (unsigned)B.X)
                        L.start[r * (unsigned)(B.X + 1) + xi] = tstart[row_cell(r) + xi] + L.rowdelta[r];
                }
            }
        }
#ifdef SP_TILE_DEBUG
        if (dbg && tid == 0) { atomicAdd(dbg + 0, 1u); if (!fits) atomicAdd(dbg + 1, 1u); atomicAdd(dbg + 6, rows); atomicAdd(dbg + 7, nstarts); }
#endif
        if (!fits) {  // uniform
            if (len > (unsigned)kTileMinSeg) {
                len >>= 1;
                __syncthreads();  // (L.bbox / L.wsum are rewritten at the top)
                continue;
            }
            lo += len;  // this part searches through global memory (final stays false, rings_done = -1)
            __syncthreads();
            continue;
        }
        seg_hint = len;
        // ---- stage 1: every query of the part over the 3x3x3 cells around its own cell
        if (mine) {
            L.q[tid] = make_float4(qx, qy, qz, 0.0f);
            L.key[tid] = none;
            L.gpos[tid] = 0u;
        }
        TILE_STAMP(8);
        tile_queue_push(L, 0, mine);
        __syncthreads();  // (the tile is complete, too)
        TILE_STAMP(3);
        const unsigned q1 = L.qcount[0];
        const bool direct = q1 > (unsigned)kTileThreads / 2;  // most lanes have a query: no queue
        if (direct) {
            if (mine) tile_scan_own3(L, g, B, qx, qy, qz, cx, cy, cz, key, gpos);
        } else {
            tile_scan_stage<1>(L, g, B, q1, tile_group_width(q1, 8u));
            __syncthreads();
            if (mine) { key = L.key[tid]; gpos = L.gpos[tid]; }
        }
        TILE_STAMP(4);
        bool open = false;
        if (mine) {
            open = !tile_block_proves(g, qx, qy, qz, cx, cy, cz, 1, __uint_as_float((unsigned)(key >> 32)));
            if (open && direct) { L.key[tid] = key; L.gpos[tid] = gpos; }
        }
        // ---- stage 2: the unproven ones, compacted, over 5x5x5 cells
        __syncthreads();  // (every stage-1 worker is done with L.queue)
        tile_queue_push(L, 1, open);
        __syncthreads();
        const unsigned q2 = L.qcount[1];
#ifdef SP_TILE_DEBUG
        if (dbg && tid == 0) { atomicAdd(dbg + 2, q1); atomicAdd(dbg + 3, q2); if (len < 1024u) atomicAdd(dbg + 5, 1u); }
#endif
        if (q2) {
            tile_scan_stage<2>(L, g, B, q2, tile_group_width(q2, 32u));
            __syncthreads();
        }
        if (mine) {
            if (open) { key = L.key[tid]; gpos = L.gpos[tid]; }
            final = !open || tile_block_proves(g, qx, qy, qz, cx, cy, cz, 2, __uint_as_float((unsigned)(key >> 32)));
            rings_done = 2;
#ifdef SP_TILE_DEBUG
            if (dbg && !final) atomicAdd(dbg + 4, 1u);
#endif
        }
        lo += len;
        __syncthreads();  // the tile is rewritten by the next part
        TILE_STAMP(5);
    }
}

}  // namespace sp

// K1 — brute-force kNN for gfx950 (replaces algorithms/knn/bruteforce.hpp:24-96).
//
// Layout of the work: a 2-D grid, x = tiles of 256*QPT queries, y = contiguous chunks of the target cloud.
// Every workgroup stages its target chunk through LDS in tiles of 1024 float4 (coalesced 16-byte loads, 16 KiB),
// then all 256 lanes walk the tile with wave-uniform (broadcast) ds_read_b128 and keep a sorted top-k per query
// in registers. Chunks write partial top-k lists; a second tiny kernel merges them in chunk order, which keeps
// the reference's tie rule (strict '<': the lowest target index wins) exactly.
// The bound is fp32 VALU, not HBM: 9 VALU ops per (query, target) pair, HBM traffic is O(nq + nt).
// k = 1 has its own packed-fp32 kernel (3.5 instructions per pair); k > 1 on >= 16 tiles of targets runs it as pass A of a
// bound-then-collect scheme: chunk minima -> a bound per query -> only the (query, chunk) pairs that can hold a neighbour
// are scanned again (a fifth of them at k = 20), every target within the bound becomes a candidate, a wave sorts each
// query's candidates (8.5 ms single pass -> 3.6 ms bounded lists in round 2 -> see knn_bf_collect_kernel).
#include "sp_common.h"
#include "sp_math.h"
#include "sp_wave_select.h"

namespace sp {
namespace {

constexpr int kTile = 1024;

// Sorted insertion into the first k slots (strict '<', later arrivals go after equal distances), written as a
// fully unrolled carry chain so the arrays stay in registers.
template <int KCAP>
__device__ __forceinline__ void topk_insert(float (&bd)[KCAP], int (&bi)[KCAP], int k, float d, int idx, float& kth) {
    if (KCAP == 1) {
        const bool better = d < bd[0];
        bi[0] = better ? idx : bi[0];
        bd[0] = better ? d : bd[0];
        kth = bd[0];
        return;
    }
    float cd = d;
    int ci = idx;
    bool shifting = false;
#pragma unroll
    for (int i = 0; i < KCAP; ++i) {
        if (i < k) {
            const bool sw = shifting || (cd < bd[i]);
            const float td = bd[i];
            const int ti = bi[i];
            const float nd = sw ? cd : td;
            bd[i] = nd;
            bi[i] = sw ? ci : ti;
            cd = sw ? td : cd;
            ci = sw ? ti : ci;
            shifting = sw;
            kth = nd;  // after the last executed iteration (i == k-1) this is bestK[k-1].dist_sq
        }
    }
}

template <int KCAP, int QPT>
__global__ __launch_bounds__(kBlock) void knn_bf_kernel(const float4* __restrict__ queries, unsigned nq,
                                                        const float4* __restrict__ targets, unsigned nt, int k,
                                                        unsigned chunk, int32_t* __restrict__ idx_out,
                                                        float* __restrict__ d2_out) {
    __shared__ float4 tile[kTile];
    const unsigned split = blockIdx.y;
    const unsigned t_begin = split * chunk;
    const unsigned t_end = min(nt, t_begin + chunk);

    float qx[QPT], qy[QPT], qz[QPT];
    unsigned qid[QPT];
    float bd[QPT][KCAP];
    int bi[QPT][KCAP];
    float kth[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        qid[u] = (blockIdx.x * QPT + u) * kBlock + threadIdx.x;
        const float4 q = queries[min(qid[u], nq - 1)];
        qx[u] = q.x; qy[u] = q.y; qz[u] = q.z;
#pragma unroll
        for (int i = 0; i < KCAP; ++i) { bd[u][i] = FLT_MAX; bi[u][i] = -1; }
        kth[u] = FLT_MAX;
    }

    for (unsigned base = t_begin; base < t_end; base += kTile) {
        const unsigned cnt = min((unsigned)kTile, t_end - base);
        __syncthreads();
        for (unsigned i = threadIdx.x; i < cnt; i += kBlock) tile[i] = targets[base + i];
        __syncthreads();
#pragma unroll 4
        for (unsigned j = 0; j < cnt; ++j) {
            const float4 p = tile[j];  // same address in every lane: one broadcast LDS read
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const float d = dist2(qx[u], qy[u], qz[u], p.x, p.y, p.z);
                if (d < kth[u]) {
                    topk_insert<KCAP>(bd[u], bi[u], k, d, (int)(base + j), kth[u]);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        if (qid[u] < nq) {
            const size_t o = ((size_t)split * nq + qid[u]) * (size_t)k;
#pragma unroll
            for (int i = 0; i < KCAP; ++i)
                if (i < k) { d2_out[o + i] = bd[u][i]; idx_out[o + i] = bi[u][i]; }
        }
    }
}

// k = 1 in 3.5 VALU instructions per (query, target) pair instead of 9:
//   * eight queries per lane as four packed pairs: v_pk_add / v_pk_mul / v_pk_fma_f32 evaluate two distances per instruction
//     (each component is the same IEEE operation chain as dist2, so the values are bit-identical);
//   * the scan keeps only the running MINIMUM (v_pk_min_f32, no compare + two selects per pair) and, once per sub-block
//     of 64 targets, which sub-block produced it (strict '<': the earliest one);
//   * the winner's index is recovered afterwards by re-evaluating that one sub-block in order and taking the first target
//     whose distance equals the minimum — the lowest index among equal distances, the reference's tie rule. NaN
//     distances never win either form (minNum semantics / `d < best` false).
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kSub = 64;   // targets per tracked sub-block (kTile % kSub == 0)
constexpr int kQ1 = 8;     // queries per lane (four packed pairs: four independent dependency chains per target)

// The target's coordinates arrive as the two aligned register pairs of one ds_read_b128, so each broadcast is an
// op_sel on the packed instruction instead of a v_mov.
__device__ __forceinline__ v2f pk_dist2(v2f qx, v2f qy, v2f qz, v2f pxy, v2f pzw) {
    const v2f dx = qx - __builtin_shufflevector(pxy, pxy, 0, 0), dy = qy - __builtin_shufflevector(pxy, pxy, 1, 1),
              dz = qz - __builtin_shufflevector(pzw, pzw, 0, 0);
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}

// MIN_ONLY: the chunk minimum alone (pass A of the two-pass k > 1 path below), no index recovery.
template <bool MIN_ONLY = false>
__global__ __launch_bounds__(kBlock) void knn_bf_k1_kernel(const float4* __restrict__ queries, unsigned nq,
                                                           const float4* __restrict__ targets, unsigned nt,
                                                           unsigned chunk, int32_t* __restrict__ idx_out,
                                                           float* __restrict__ d2_out) {
    __shared__ float4 tile[kTile];
    const unsigned split = blockIdx.y;
    const unsigned t_begin = split * chunk;
    const unsigned t_end = min(nt, t_begin + chunk);
    unsigned qid[kQ1];
    v2f qx[kQ1 / 2], qy[kQ1 / 2], qz[kQ1 / 2], best[kQ1 / 2];
    unsigned bsub[kQ1];
#pragma unroll
    for (int u = 0; u < kQ1; ++u) {
        qid[u] = (blockIdx.x * kQ1 + u) * kBlock + threadIdx.x;
        const float4 q = queries[min(qid[u], nq - 1)];
        qx[u >> 1][u & 1] = q.x; qy[u >> 1][u & 1] = q.y; qz[u >> 1][u & 1] = q.z;
        best[u >> 1][u & 1] = FLT_MAX;
        bsub[u] = 0xFFFFFFFFu;
    }
    for (unsigned base = t_begin; base < t_end; base += kTile) {
        const unsigned cnt = min((unsigned)kTile, t_end - base);
        __syncthreads();
        // pad the tile with copies of its first point: a duplicate cannot change the minimum
        for (unsigned i = threadIdx.x; i < kTile; i += kBlock) tile[i] = targets[base + (i < cnt ? i : 0u)];
        __syncthreads();
        const unsigned nsub = (cnt + kSub - 1) / kSub;
        for (unsigned sb = 0; sb < nsub; ++sb) {
            v2f m[kQ1 / 2];
#pragma unroll
            for (int h = 0; h < kQ1 / 2; ++h) m[h] = v2f{FLT_MAX, FLT_MAX};
#pragma unroll 4
            for (int j = 0; j < kSub; ++j) {
                const float4 p = tile[sb * kSub + j];  // same address in every lane: one broadcast LDS read
                const v2f pxy = {p.x, p.y}, pzw = {p.z, p.w};
#pragma unroll
                for (int h = 0; h < kQ1 / 2; ++h)
                    m[h] = __builtin_elementwise_min(m[h], pk_dist2(qx[h], qy[h], qz[h], pxy, pzw));
            }
            const unsigned where = base + sb * kSub;
#pragma unroll
            for (int u = 0; u < kQ1; ++u) {
                const float mu = m[u >> 1][u & 1];
                const bool better = mu < best[u >> 1][u & 1];
                best[u >> 1][u & 1] = better ? mu : best[u >> 1][u & 1];
                bsub[u] = better ? where : bsub[u];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < kQ1; ++u) {
        if (qid[u] >= nq) continue;
        const float b = best[u >> 1][u & 1];
        if (MIN_ONLY) { d2_out[(size_t)split * nq + qid[u]] = b; continue; }
        int idx = -1;
        if (bsub[u] != 0xFFFFFFFFu) {
            const float x = qx[u >> 1][u & 1], y = qy[u >> 1][u & 1], z = qz[u >> 1][u & 1];
            const unsigned e = min(t_end, bsub[u] + kSub);
            unsigned first = 0xFFFFFFFFu;  // lowest matching index: branch-free, 8 independent loads per step
            for (unsigned j0 = bsub[u]; j0 < e; j0 += 8) {
                float4 pp[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) pp[t] = targets[min(j0 + t, e - 1)];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const bool hit = (j0 + t < e) && dist2(x, y, z, pp[t].x, pp[t].y, pp[t].z) == b;
                    first = min(first, hit ? j0 + t : 0xFFFFFFFFu);
                }
            }
            idx = first == 0xFFFFFFFFu ? -1 : (int)first;
        }
        const size_t o = (size_t)split * nq + qid[u];
        d2_out[o] = b;
        idx_out[o] = idx;
    }
}

// Merge nsplit sorted partial lists per query, in chunk order (ascending target index at equal distance).
template <int KCAP>
__global__ __launch_bounds__(kBlock) void knn_bf_merge_kernel(const int32_t* __restrict__ pidx,
                                                              const float* __restrict__ pd2, unsigned nq, int k,
                                                              unsigned nsplit, int32_t* __restrict__ idx_out,
                                                              float* __restrict__ d2_out) {
    const unsigned q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= nq) return;
    float bd[KCAP];
    int bi[KCAP];
#pragma unroll
    for (int i = 0; i < KCAP; ++i) { bd[i] = FLT_MAX; bi[i] = -1; }
    float kth = FLT_MAX;
    for (unsigned s = 0; s < nsplit; ++s) {
        const size_t o = ((size_t)s * nq + q) * (size_t)k;
        for (int i = 0; i < k; ++i) {
            const float d = pd2[o + i];
            if (!(d < kth)) break;  // list is ascending: nothing further in this chunk can enter
            topk_insert<KCAP>(bd, bi, k, d, pidx[o + i], kth);
        }
    }
    const size_t o = (size_t)q * (size_t)k;
#pragma unroll
    for (int i = 0; i < KCAP; ++i)
        if (i < k) { d2_out[o + i] = bd[i]; idx_out[o + i] = bi[i]; }
}

// ---- k > 1 on large problems: bound first, then collect ------------------------------------------------------------
// The sorted insertion costs ~5 instructions per list slot and, in a stream of n targets, a query inserts about
// k ln(n / k) times (170 at k = 20, n = 100 k): with 64 lanes x 2 queries per wave almost a quarter of the targets send the
// wave through the insertion code. So nothing is inserted while scanning:
//   pass A   the k = 1 kernel over G >= k chunks of the targets: G chunk minima per query. Their k-th smallest, tau, is the
//            distance of a real target and at least k targets are within it (one per chunk), so the k-th neighbour is too.
//   lists    a query needs chunk c again only if that chunk's minimum is within tau — k of the G chunks unless minima tie
//            (20 of 98 at 100 k targets). Every chunk gets the list of the queries that need it (one atomic per wave and
//            chunk reserves the wave's slots; the order inside a list does not reach the results).
//   pass B   one workgroup per (chunk, 512 queries of its list): distances again at packed-fp32 rate, four targets folded
//            with v_pk_min and tested against the bound once; a target within the bound is appended to its query's
//            candidate array as the 64-bit key (distance bits : index). About k + G/k candidates per query.
//   select   one wave per query sorts the candidates (64-lane bitonic network) and writes the first k: ordered by
//            (distance, index) — the reference's strict '<' in ascending target order. A query with more than kCandCap
//            candidates (many targets at the same distance, or fewer than k finite chunk minima) is rescanned over all
//            targets by its wave with the sorted-insertion form. Lists are bit-identical to the single-pass kernel's.
constexpr int kCandCap = 64;

constexpr int kMaxChunks = 256;     // G: the targets are cut into at most this many chunks of whole LDS tiles
constexpr int kMaxNeeded = 32;      // chunks one query may be listed for; beyond that (tied minima) its wave rescans everything

// ---- pass A in 2 instructions per pair: chunk minima of an APPROXIMATE distance with a proven error bound -----------------
// Pass A only has to bound the k-th neighbour's distance, so it need not evaluate the reference's expression. With both
// clouds centred on the targets' bounding box (q^ = q - c, p^ = p - c) the tile holds (-2p^x, -2p^y, -2p^z, |p^|^2) and
//     d~ = fma(q^x, -2p^x, fma(q^y, -2p^y, fma(q^z, -2p^z, |p^|^2))) + |q^|^2
// costs 3 packed FMAs + 1 packed min per two pairs (the |q^|^2 is added to the minimum afterwards) against 7 for the exact
// form. Error against the reference's fp32 value d_ref (u = 2^-24, Q = |q^|, P = |p^| <= pmax):
//     |d~ - |q^-p^|^2|        <= 7 u (Q+P)^2     (3 roundings in |p^|^2, 3 in the chain, 3 in |q^|^2, 1 in the final add)
//     ||q^-p^|^2 - |q-p|^2|   <= 2.1 u (Q+P)^2   (the two centring subtractions round)
//     |d_ref - |q-p|^2|       <= 5.1 u (Q+P)^2   (5 roundings, relative)
// so |d~ - d_ref| <= E = 20 u (Q + pmax)^2 with room to spare. The bound kernel turns the k-th smallest approximate chunk
// minimum m_k into tau = m_k + E (at least k targets have d_ref <= tau) and lists a query for every chunk whose
// approximate minimum is <= tau + E (no chunk holding a target with d_ref <= tau is missed). Candidates and results come
// from the exact expression in the collect kernel, as before: lists stay bit-identical. Centred coordinates make E / spacing^2
// a function of the target COUNT alone for a roughly uniform cloud (0.008 at 10^5 targets, 0.04 at 10^6): the host uses this
// pass up to kApproxMaxTargets and the exact k = 1 kernel beyond. A cloud with overflowing or non-finite extents makes E
// non-finite; such queries go to the select kernel's full rescan.
constexpr size_t kApproxMaxTargets = 2u << 20;

struct BfFrame {
    float cx, cy, cz;  // centre of the finite targets' bounding box
    float pmax;        // >= |p - c| for every finite target
};
__device__ __forceinline__ unsigned enc_ordered(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(unsigned e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}
__global__ void knn_bf_box_init_kernel(unsigned* box) {
    if (threadIdx.x < 6) box[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
}
__global__ __launch_bounds__(kBlock) void knn_bf_box_kernel(const float4* __restrict__ targets, unsigned nt, unsigned* box) {
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < nt; i += gridDim.x * kBlock) {
        const float4 p = targets[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            const unsigned e[3] = {enc_ordered(p.x), enc_ordered(p.y), enc_ordered(p.z)};
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
    __shared__ unsigned red[kBlock / kWave][6];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[threadIdx.x >> 6][a] = mn[a]; red[threadIdx.x >> 6][3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {  // one atomic per workgroup and bound (64 workgroups: the six words share a cache line)
        unsigned v = red[0][threadIdx.x];
        for (int w = 1; w < kBlock / kWave; ++w) v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&box[threadIdx.x], v);
        else atomicMax(&box[threadIdx.x], v);
    }
}
__global__ void knn_bf_frame_kernel(const unsigned* __restrict__ box, BfFrame* __restrict__ frame) {
    if (threadIdx.x != 0) return;
    BfFrame f{0.0f, 0.0f, 0.0f, 0.0f};
    if (box[0] != 0xffffffffu) {  // (no finite target: every distance is non-finite, nothing is ever a candidate)
        float h2 = 0.0f;
        float c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = dec_ordered(box[a]), hi = dec_ordered(box[3 + a]);
            c[a] = 0.5f * lo + 0.5f * hi;
            const float h = fmaxf(hi - c[a], c[a] - lo);
            h2 += h * h;
        }
        f.cx = c[0]; f.cy = c[1]; f.cz = c[2];
        f.pmax = sqrtf(h2) * 1.0001f;  // (the roundings above are parts in 10^7)
    }
    *frame = f;
}

__global__ __launch_bounds__(kBlock) void knn_bf_chunkmin_kernel(const float4* __restrict__ queries, unsigned nq,
                                                                 const float4* __restrict__ targets, unsigned nt,
                                                                 unsigned chunk, const BfFrame* __restrict__ frame,
                                                                 float* __restrict__ amin, unsigned il = 0) {
    __shared__ float4 tile[kTile];
    const unsigned split = blockIdx.y;
    // (il != 0: interleaved chunks, see chunk_target — slot s of chunk c is target s * il + c)
    const unsigned t_begin = il ? 0u : split * chunk;
    const unsigned t_end = il ? (split < nt ? (nt - split + il - 1) / il : 0u) : min(nt, t_begin + chunk);
    const float cx = frame->cx, cy = frame->cy, cz = frame->cz;
    unsigned qid[kQ1];
    v2f qx[kQ1 / 2], qy[kQ1 / 2], qz[kQ1 / 2], m[kQ1 / 2];
    float qq[kQ1];
#pragma unroll
    for (int u = 0; u < kQ1; ++u) {
        qid[u] = (blockIdx.x * kQ1 + u) * kBlock + threadIdx.x;
        const float4 q = queries[min(qid[u], nq - 1)];
        const float x = q.x - cx, y = q.y - cy, z = q.z - cz;
        qx[u >> 1][u & 1] = x; qy[u >> 1][u & 1] = y; qz[u >> 1][u & 1] = z;
        qq[u] = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
        m[u >> 1][u & 1] = FLT_MAX;
    }
    for (unsigned base = t_begin; base < t_end; base += kTile) {
        const unsigned cnt = min((unsigned)kTile, t_end - base);
        __syncthreads();
        // (the tile is padded to a multiple of 4 with copies of its first point: a duplicate cannot change the minimum)
        const unsigned padded = (cnt + 3u) & ~3u;
        for (unsigned i = threadIdx.x; i < padded; i += kBlock) {
            const unsigned slot = base + (i < cnt ? i : 0u);
            const float4 p = targets[il ? slot * il + split : slot];
            const float x = p.x - cx, y = p.y - cy, z = p.z - cz;
            tile[i] = make_float4(-2.0f * x, -2.0f * y, -2.0f * z, __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
        }
        __syncthreads();
#pragma unroll 4
        for (unsigned j = 0; j < padded; ++j) {
            const float4 t = tile[j];  // same address in every lane: one broadcast LDS read
            const v2f axy = {t.x, t.y}, czw = {t.z, t.w};
#pragma unroll
            for (int h = 0; h < kQ1 / 2; ++h) {
                v2f s = __builtin_elementwise_fma(qz[h], __builtin_shufflevector(czw, czw, 0, 0), __builtin_shufflevector(czw, czw, 1, 1));
                s = __builtin_elementwise_fma(qy[h], __builtin_shufflevector(axy, axy, 1, 1), s);
                s = __builtin_elementwise_fma(qx[h], __builtin_shufflevector(axy, axy, 0, 0), s);
                m[h] = __builtin_elementwise_min(m[h], s);  // (minNum: a NaN from a non-finite target never wins)
            }
        }
    }
#pragma unroll
    for (int u = 0; u < kQ1; ++u)
        if (qid[u] < nq) {
            const float mu = m[u >> 1][u & 1];
            amin[(size_t)split * nq + qid[u]] = mu < FLT_MAX ? mu + qq[u] : FLT_MAX;
        }
}

// ---- pass A on the matrix cores ---------------------------------------------------------------------------------------
// |p^|^2 - 2 q^.p^ over (chunk of targets) x (queries) is a GEMM with K = 4; in bf16 pieces it is one
// v_mfma_f32_32x32x16_bf16 per 32 targets x 32 queries: 1024 pairs in 32 cycles of a SIMD, against 2 packed VALU
// instructions per pair (8 pairs per cycle). Every centred coordinate is split exactly, x = x1 + x2 + x3 with bf16 pieces
// (|x - x1 - x2| <= 2^-16 |x|), and the 16 k-slots carry, per coordinate, the four products of the two leading pieces
//     A (target row)   -2p1  -2p1  -2p2  -2p2          B (query column)   q1  q2  q1  q2
// then |p^|^2 in three pieces against 1, 1, 1, and one empty slot. Products of bf16 pieces are exact in fp32; what is lost:
//     the dropped piece products                     <= 2^-14 |q_x p_x| per coordinate, <= 2^-16 (Q+P)^2 in total
//     the matrix core's 16 accumulations             <= 2u each of the running magnitude (taken as truncating): 33 u (Q+P)^2
//     |p^|^2, |q^|^2, the final add, centring, d_ref  <= 14.2 u (Q+P)^2 as in the VALU form above
// = 1.81e-5 (Q+P)^2; the bound kernel uses kMfmaErr = 2.0e-5. At 10^5 uniform targets that is a quarter of the nearest
// neighbour's squared distance (a few per cent of the 20th's): some more (query, chunk) pairs for the exact collect pass,
// which stays a small part. D[target][query] has the query on the lane and 16 targets in the registers, so the running
// minimum is 8 v_min3_f32 per MFMA on the lane's own registers — no cross-lane work until the chunk ends.
// Rows of non-finite targets and the padding up to whole chunks are (0, ..., |p|^2 = 3.4e38): never a minimum.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float kValuErr = 20.0f * 5.9604645e-8f;
constexpr float kMfmaErr = 2.0e-5f;
constexpr int kQTiles = 8;                          // 32-query tiles per wave: their B fragments stay in registers
constexpr unsigned kMfmaQueries = 4 * kQTiles * 32;  // queries per workgroup of 4 waves

__device__ __forceinline__ unsigned bf16_rne(float f) {  // bits of the nearest bf16 (finite f)
    const unsigned u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_val(unsigned b) { return __uint_as_float(b << 16); }
// x = v(a) + v(b) + v(c) exactly (each remainder is exactly representable in fp32)
__device__ __forceinline__ void bf16_split3(float x, unsigned& a, unsigned& b, unsigned& c) {
    a = bf16_rne(x);
    const float r1 = x - bf16_val(a);
    b = bf16_rne(r1);
    c = bf16_rne(r1 - bf16_val(b));
}
__device__ __forceinline__ unsigned pack2(unsigned lo, unsigned hi) { return (lo & 0xffffu) | (hi << 16); }

__global__ __launch_bounds__(kBlock) void knn_bf_prep_targets_kernel(const float4* __restrict__ targets, unsigned nt,
                                                                     unsigned rows, const BfFrame* __restrict__ frame,
                                                                     uint4* __restrict__ tA, unsigned chunk = 0, unsigned il = 0) {
    const unsigned t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= rows) return;
    uint4 lo = make_uint4(0u, 0u, 0u, 0u), hi = make_uint4(0u, 0u, 0x7f7fu, 0u);  // (k12 = 3.39e38, everything else 0)
    // row t = slot t % chunk of chunk t / chunk; with interleaved chunks (il != 0) that is target slot * il + chunk index
    const unsigned ti = il ? (t % chunk) * il + t / chunk : t;
    if (ti < nt) {
        const float4 p = targets[ti];
        const float x = p.x - frame->cx, y = p.y - frame->cy, z = p.z - frame->cz;
        const float w = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
        if (isfinite(w)) {
            unsigned x1, x2, x3, y1, y2, y3, z1, z2, z3, w1, w2, w3;
            bf16_split3(-2.0f * x, x1, x2, x3);  // (the factor is a power of two: the pieces of -2x are -2 times those of x)
            bf16_split3(-2.0f * y, y1, y2, y3);
            bf16_split3(-2.0f * z, z1, z2, z3);
            bf16_split3(w, w1, w2, w3);
            lo = make_uint4(pack2(x1, x1), pack2(x2, x2), pack2(y1, y1), pack2(y2, y2));
            hi = make_uint4(pack2(z1, z1), pack2(z2, z2), pack2(w1, w2), pack2(w3, 0u));
        }
    }
    tA[2 * (size_t)t] = lo;
    tA[2 * (size_t)t + 1] = hi;
}
__global__ __launch_bounds__(kBlock) void knn_bf_prep_queries_kernel(const float4* __restrict__ queries, unsigned nq,
                                                                     unsigned cols, const BfFrame* __restrict__ frame,
                                                                     uint4* __restrict__ qB, float* __restrict__ qq) {
    const unsigned q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= cols) return;
    uint4 lo = make_uint4(0u, 0u, 0u, 0u), hi = lo;
    float n2 = 0.0f;
    if (q < nq) {
        const float4 p = queries[q];
        const float x = p.x - frame->cx, y = p.y - frame->cy, z = p.z - frame->cz;
        n2 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
        unsigned x1, x2, x3, y1, y2, y3, z1, z2, z3;
        bf16_split3(x, x1, x2, x3);
        bf16_split3(y, y1, y2, y3);
        bf16_split3(z, z1, z2, z3);
        lo = make_uint4(pack2(x1, x2), pack2(x1, x2), pack2(y1, y2), pack2(y1, y2));
        hi = make_uint4(pack2(z1, z2), pack2(z1, z2), pack2(0x3f80u, 0x3f80u), pack2(0x3f80u, 0u));
    }
    qB[2 * (size_t)q] = lo;
    qB[2 * (size_t)q + 1] = hi;
    qq[q] = n2;
}

__device__ __forceinline__ float min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

// grid: x = groups of kMfmaQueries queries, y = groups of `chunks_per_wg` chunks. `chunk` is a multiple of 32; tA holds
// nchunks * chunk rows, qB / qq a multiple of kMfmaQueries columns.
__global__ __launch_bounds__(kBlock) void knn_bf_chunkmin_mfma_kernel(const uint4* __restrict__ tA, const uint4* __restrict__ qB,
                                                                      const float* __restrict__ qq, unsigned nq, unsigned chunk,
                                                                      unsigned nchunks, unsigned chunks_per_wg,
                                                                      float* __restrict__ amin) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned r = lane & 31, h = lane >> 5;
    const unsigned q0 = (blockIdx.x * 4 + wave) * (kQTiles * 32);
    bf16x8 b[kQTiles];
    float myqq[kQTiles];
#pragma unroll
    for (int t = 0; t < kQTiles; ++t) {
        const unsigned qi = q0 + t * 32 + r;
        const uint4 v = qB[2 * (size_t)qi + h];
        b[t] = __builtin_bit_cast(bf16x8, v);
        myqq[t] = qq[qi];
    }
    const f32x16 zero = {};
    const unsigned steps = chunk / 32;
    const unsigned c_end = min(nchunks, (blockIdx.y + 1) * chunks_per_wg);
    for (unsigned c = blockIdx.y * chunks_per_wg; c < c_end; ++c) {
        float m[kQTiles];
#pragma unroll
        for (int t = 0; t < kQTiles; ++t) m[t] = FLT_MAX;
        const uint4* a_ptr = tA + 2 * ((size_t)c * chunk + r) + h;  // lane (r, h): k = 8h .. 8h+7 of target row r
        uint4 a_next = a_ptr[0];
        for (unsigned s = 0; s < steps; ++s) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, a_next);
            if (s + 1 < steps) a_next = a_ptr[(size_t)(s + 1) * 64];  // (in flight while this step's MFMAs run)
            // (measured: the SIMD is vector-issue-bound here — 8 v_min3 + the MFMA's own 8 issue cycles per 32-cycle product —
            // and reaches 82 % of that bound whichever way the products and minima are ordered in the source)
#pragma unroll
            for (int t = 0; t < kQTiles; ++t) {
                const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[t], zero, 0, 0, 0);
                float v = min3(m[t], d[0], d[1]);
                v = min3(v, d[2], d[3]);
                v = min3(v, d[4], d[5]);
                v = min3(v, d[6], d[7]);
                v = min3(v, d[8], d[9]);
                v = min3(v, d[10], d[11]);
                v = min3(v, d[12], d[13]);
                m[t] = min3(v, d[14], d[15]);
            }
        }
#pragma unroll
        for (int t = 0; t < kQTiles; ++t) {
            const float v = fminf(m[t], __shfl_xor(m[t], 32, 64));  // the two lane halves hold different targets of one query
            const unsigned qi = q0 + t * 32 + r;
            if (h == 0 && qi < nq) amin[(size_t)c * nq + qi] = v < 1e37f ? v + myqq[t] : FLT_MAX;
        }
    }
}

// tau per query, and wave_cnt[c][wave] = how many of the wave's 64 queries need chunk c. (No atomics: a counter per chunk,
// bumped by every wave, put 150 k atomics on four cache lines — 1.3 ms at 100 k queries; counts + scans are 20 us.)
__global__ __launch_bounds__(kBlock) void knn_bf_bound_kernel(const float* __restrict__ chunk_min, unsigned nq, int k,
                                                              unsigned nchunks, const float4* __restrict__ queries,
                                                              const BfFrame* __restrict__ frame /* nullptr: exact minima */,
                                                              float err_coeff, float* __restrict__ bound, uint4* __restrict__ need_mask,
                                                              unsigned* __restrict__ wave_cnt, unsigned nwaves,
                                                              unsigned* __restrict__ cand_cnt) {
    const unsigned q = blockIdx.x * kBlock + threadIdx.x;
    const bool live = q < nq;
    const unsigned qc = live ? q : nq - 1;
    float bd[20];
    int bi[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) { bd[i] = FLT_MAX; bi[i] = -1; }
    float kth = FLT_MAX;
    // (a hundred thousand queries are 1.5 waves per SIMD: nothing hides a load's latency, so they go out eight at a time)
    for (unsigned c0 = 0; c0 < nchunks; c0 += 8) {
        float d[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = c0 + j < nchunks ? chunk_min[(size_t)(c0 + j) * nq + qc] : FLT_MAX;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (d[j] < kth) topk_insert<20>(bd, bi, k, d[j], 0, kth);
    }
    float E = 0.0f;  // |approximate - reference| distance, see knn_bf_chunkmin_kernel
    if (frame) {
        const float4 qq = queries[qc];
        const float x = qq.x - frame->cx, y = qq.y - frame->cy, z = qq.z - frame->cz;
        const float r = sqrtf(x * x + y * y + z * z) * 1.0001f + frame->pmax;
        E = err_coeff * 1.0001f * r * r;
    }
    // `d < bound` must admit d == tau; a query without k finite chunk minima (NaN / overflowing coordinates) is not bounded
    static_assert(kMaxChunks == 256, "the need mask is 8 words");
    float b = FLT_MAX, bn = FLT_MAX;
    bool rescan = !(E < FLT_MAX);  // non-finite frame or query: no usable bound (and the minima may be meaningless)
    if (!rescan && kth < FLT_MAX) {
        const float tau = E > 0.0f ? nextafterf(kth + E, FLT_MAX) : kth;
        b = fminf(nextafterf(tau, FLT_MAX), FLT_MAX);
        bn = E > 0.0f ? fminf(nextafterf(nextafterf(tau + E, FLT_MAX), FLT_MAX), FLT_MAX) : b;
    }
    // which chunks can hold a candidate: one bit per chunk (G <= 256), kept in registers for the counts below and stored for
    // the list kernel, so the minima are read twice, not four times
    unsigned mask[kMaxChunks / 32];
#pragma unroll
    for (int i = 0; i < kMaxChunks / 32; ++i) mask[i] = 0u;
    unsigned needed = 0;
#pragma unroll
    for (int i = 0; i < kMaxChunks / 32; ++i) {
        if ((unsigned)i * 32u < nchunks) {  // (uniform)
#pragma unroll
            for (int c0 = 0; c0 < 32; c0 += 8) {
                float d[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned c = i * 32 + c0 + j;
                    d[j] = c < nchunks ? chunk_min[(size_t)c * nq + qc] : FLT_MAX;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) mask[i] |= (d[j] < bn ? 1u : 0u) << (c0 + j);
            }
            needed += __builtin_popcount(mask[i]);
        }
    }
    rescan = rescan || needed > (unsigned)kMaxNeeded;
    if (rescan) {  // listed nowhere: the select kernel scans all targets for it
        b = -1.0f;
#pragma unroll
        for (int i = 0; i < kMaxChunks / 32; ++i) mask[i] = 0u;
    }
    if (live) {
        bound[q] = b;
        need_mask[2 * (size_t)q] = make_uint4(mask[0], mask[1], mask[2], mask[3]);
        need_mask[2 * (size_t)q + 1] = make_uint4(mask[4], mask[5], mask[6], mask[7]);
        cand_cnt[q] = rescan ? (unsigned)kCandCap + 1u : 0u;
    }
    const unsigned wave = q >> 6, lane = threadIdx.x & 63;
    if (wave >= nwaves) return;  // (the last workgroup may hold an idle wave)
#pragma unroll
    for (int i = 0; i < kMaxChunks / 32; ++i) {
        if ((unsigned)i * 32u >= nchunks) break;  // (uniform)
        const unsigned mi = live ? mask[i] : 0u;
        for (unsigned c0 = 0; c0 < 32 && i * 32 + c0 < nchunks; ++c0) {
            const unsigned long long m = __ballot((mi >> c0) & 1u);
            if (lane == 0) wave_cnt[(size_t)(i * 32 + c0) * nwaves + wave] = (unsigned)__builtin_popcountll(m);
        }
    }
}
// One workgroup per chunk: wave_cnt[c][.] becomes its exclusive scan, chunk_cnt[c] the total.
__global__ __launch_bounds__(kBlock) void knn_bf_wave_offsets_kernel(unsigned* __restrict__ wave_cnt, unsigned nwaves,
                                                                     unsigned* __restrict__ chunk_cnt) {
    __shared__ unsigned s[kBlock];
    unsigned* row = wave_cnt + (size_t)blockIdx.x * nwaves;
    const unsigned t = threadIdx.x;
    unsigned carry = 0;
    for (unsigned base = 0; base < nwaves; base += kBlock) {
        const unsigned v = base + t < nwaves ? row[base + t] : 0u;
        s[t] = v;
        __syncthreads();
        for (unsigned d = 1; d < (unsigned)kBlock; d <<= 1) {
            const unsigned o = t >= d ? s[t - d] : 0u;
            __syncthreads();
            s[t] += o;
            __syncthreads();
        }
        if (base + t < nwaves) row[base + t] = carry + s[t] - v;
        carry += s[kBlock - 1];
        __syncthreads();
    }
    if (t == 0) chunk_cnt[blockIdx.x] = carry;
}
// chunk_off = exclusive scan of chunk_cnt (<= 256 values); blk_off = exclusive scan of the collect kernel's work items per
// chunk (blocks of 2 * kBlock listed queries, times `subs` sub-ranges of the chunk's targets), blk_off[nchunks] = their number.
__global__ __launch_bounds__(kMaxChunks) void knn_bf_offsets_kernel(const unsigned* __restrict__ chunk_cnt, unsigned nchunks,
                                                                    unsigned subs, unsigned* __restrict__ chunk_off,
                                                                    unsigned* __restrict__ blk_off) {
    __shared__ unsigned s[kMaxChunks], sb[kMaxChunks];
    const unsigned t = threadIdx.x;
    const unsigned cnt = t < nchunks ? chunk_cnt[t] : 0u;
    const unsigned blocks = (cnt + 2 * kBlock - 1) / (2 * kBlock) * subs;
    s[t] = cnt;
    sb[t] = blocks;
    __syncthreads();
    for (unsigned d = 1; d < (unsigned)kMaxChunks; d <<= 1) {
        const unsigned v = t >= d ? s[t - d] : 0u, vb = t >= d ? sb[t - d] : 0u;
        __syncthreads();
        s[t] += v;
        sb[t] += vb;
        __syncthreads();
    }
    if (t < nchunks) {
        chunk_off[t] = s[t] - cnt;
        blk_off[t] = sb[t] - blocks;
    }
    if (t == nchunks - 1) blk_off[nchunks] = sb[t];
}
// The lists themselves: chunk c's queries, ascending, at chunk_list[chunk_off[c] ...].
__global__ __launch_bounds__(kBlock) void knn_bf_lists_kernel(const uint4* __restrict__ need_mask, unsigned nq, unsigned nchunks,
                                                              const unsigned* __restrict__ chunk_off,
                                                              const unsigned* __restrict__ wave_off, unsigned nwaves,
                                                              unsigned* __restrict__ chunk_list) {
    const unsigned q = blockIdx.x * kBlock + threadIdx.x;
    const bool live = q < nq;
    const unsigned wave = q >> 6, lane = threadIdx.x & 63;
    unsigned mask[kMaxChunks / 32] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (live) {
        const uint4 a = need_mask[2 * (size_t)q], b = need_mask[2 * (size_t)q + 1];
        mask[0] = a.x; mask[1] = a.y; mask[2] = a.z; mask[3] = a.w;
        mask[4] = b.x; mask[5] = b.y; mask[6] = b.z; mask[7] = b.w;
    }
    // where this wave's entries start in every chunk's list: fetched by all lanes at once (chunk c in lane c % 64), then
    // broadcast per chunk — 196 dependent scalar loads in a row were most of this kernel at 1.5 waves per SIMD
    unsigned start[kMaxChunks / 64];
#pragma unroll
    for (int j = 0; j < kMaxChunks / 64; ++j) {
        const unsigned c = j * 64 + lane;
        start[j] = (c < nchunks && wave < nwaves) ? chunk_off[c] + wave_off[(size_t)c * nwaves + wave] : 0u;
    }
#pragma unroll
    for (int i = 0; i < kMaxChunks / 32; ++i) {
        if ((unsigned)i * 32u >= nchunks) break;  // (uniform)
        for (unsigned c0 = 0; c0 < 32 && i * 32 + c0 < nchunks; ++c0) {
            const bool need = (mask[i] >> c0) & 1u;
            const unsigned long long m = __ballot(need);
            const unsigned base = (unsigned)__builtin_amdgcn_readlane((int)start[i / 2], (i & 1) * 32 + c0);
            if (need) chunk_list[base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = q;
        }
    }
}

constexpr int kGroup = 4;  // targets folded per bound test (kTile % kGroup == 0)

__global__ __launch_bounds__(kBlock) void knn_bf_collect_kernel(const float4* __restrict__ queries, unsigned nq,
                                                                const float4* __restrict__ targets, unsigned nt,
                                                                unsigned chunk, const float* __restrict__ bound,
                                                                const unsigned* __restrict__ chunk_cnt,
                                                                const unsigned* __restrict__ chunk_off,
                                                                const unsigned* __restrict__ blk_off, unsigned nchunks,
                                                                unsigned subs, const unsigned* __restrict__ chunk_list,
                                                                unsigned* __restrict__ cand_cnt,
                                                                unsigned long long* __restrict__ cand, unsigned il = 0) {
    __shared__ float4 tile[kTile];
    const unsigned items = blk_off[nchunks];
    for (unsigned item = blockIdx.x; item < items; item += gridDim.x) {  // (workgroup-uniform)
    unsigned lo = 0, hi = nchunks;  // the chunk of this work item: last c with blk_off[c] <= item
    while (hi - lo > 1) {
        const unsigned mid = (lo + hi) >> 1;
        if (blk_off[mid] <= item) lo = mid;
        else hi = mid;
    }
    const unsigned c = lo;
    const unsigned local = item - blk_off[c];
    const unsigned listed = chunk_cnt[c];
    const unsigned first = (local / subs) * 2 * kBlock;
    const unsigned sub = local % subs;
    const unsigned* const list = chunk_list + chunk_off[c];
    const unsigned sub_len = chunk / subs;  // (a multiple of kGroup: chunks are multiples of 256, subs is 1, 2 or 4)
    // the targets of this work item: a range of the chunk's slots; slot s of chunk c is target c * chunk + s, or, with
    // interleaved chunks (il != 0: every chunk a uniform sample of the cloud, plan_bounded), target s * il + c
    const unsigned n_c = il ? (c < nt ? (nt - c + il - 1) / il : 0u) : 0u;  // slots of an interleaved chunk
    const unsigned t_begin = il ? min(n_c, sub * sub_len) : min(nt, c * chunk + sub * sub_len);
    const unsigned t_end = il ? min(n_c, t_begin + sub_len) : min(nt, t_begin + sub_len);
    unsigned qid[2];
    v2f qx, qy, qz;
    float cap[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned slot = first + u * kBlock + threadIdx.x;
        qid[u] = list[min(slot, listed - 1)];
        const float4 q = queries[qid[u]];
        qx[u] = q.x; qy[u] = q.y; qz[u] = q.z;
        cap[u] = slot < listed ? bound[qid[u]] : -1.0f;  // (no distance is below -1: an idle slot collects nothing)
    }
    for (unsigned base = t_begin; base < t_end; base += kTile) {
        const unsigned cnt = min((unsigned)kTile, t_end - base);
        __syncthreads();
        // the tail of the last tile is padded with a point 1e18 away (finite distance, beyond every real bound; the
        // exact path below checks the index as well)
        const unsigned ngroups = (cnt + kGroup - 1) / kGroup;
        for (unsigned i = threadIdx.x; i < ngroups * kGroup; i += kBlock)
            tile[i] = i < cnt ? targets[il ? (base + i) * il + c : base + i] : make_float4(1e18f, 1e18f, 1e18f, 0.0f);
        __syncthreads();
#pragma unroll 2
        for (unsigned g = 0; g < ngroups; ++g) {
            v2f d[kGroup];
#pragma unroll
            for (int t = 0; t < kGroup; ++t) {
                const float4 p = tile[g * kGroup + t];  // same address in every lane: one broadcast LDS read
                d[t] = pk_dist2(qx, qy, qz, v2f{p.x, p.y}, v2f{p.z, p.w});
            }
            const v2f m = __builtin_elementwise_min(__builtin_elementwise_min(d[0], d[1]),
                                                    __builtin_elementwise_min(d[2], d[3]));
            if (m[0] < cap[0] || m[1] < cap[1]) {
#pragma unroll
                for (int t = 0; t < kGroup; ++t) {
                    const unsigned j = g * kGroup + t;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (j < cnt && d[t][u] < cap[u]) {
                            const unsigned s = atomicAdd(&cand_cnt[qid[u]], 1u);
                            if (s < (unsigned)kCandCap)
                                cand[(size_t)qid[u] * kCandCap + s] = cand_key(d[t][u], (int)(il ? (base + j) * il + c : base + j));
                        }
                    }
                }
            }
        }
    }
    }
}

// One wave per query: sort its candidates, or (overflow) scan every target with the sorted top-k across the lanes.
__global__ __launch_bounds__(kBlock) void knn_bf_select_kernel(const float4* __restrict__ queries, unsigned nq,
                                                               const float4* __restrict__ targets, unsigned nt, int k,
                                                               const unsigned* __restrict__ cand_cnt,
                                                               const unsigned long long* __restrict__ cand,
                                                               int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
    const unsigned q = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (q >= nq) return;  // (wave-uniform)
    const unsigned lane = threadIdx.x & 63;
    const unsigned cnt = cand_cnt[q];
    Cand best;
    best.pos = 0;
    if (cnt <= (unsigned)kCandCap) {
        Cand c;
        c.key = lane < cnt ? cand[(size_t)q * kCandCap + lane] : kNoCand;
        c.pos = 0;
        best = bitonic_sort64(c, lane);
    } else {
        const float4 qq = queries[q];
        const unsigned long long kmask = (1ull << k) - 1ull;  // k <= 20
        best.key = kNoCand;
        unsigned long long kth = kNoCand;
        for (unsigned base = 0; base < nt; base += 64) {
            const unsigned j = base + lane;
            const float4 p = targets[min(j, nt - 1)];
            Cand c;
            c.key = j < nt ? cand_key(dist2(qq.x, qq.y, qq.z, p.x, p.y, p.z), (int)j) : kNoCand;
            c.pos = 0;
            if (base == 0) {
                best = bitonic_sort64(c, lane);
                kth = bcast_k(best.key, k - 1);
            } else {
                insert_candidates(c, best, kth, k, kmask, lane);
            }
        }
    }
    if (lane < (unsigned)k) {
        // an empty slot, or a distance that `d < FLT_MAX` would not have admitted (inf / NaN bit patterns order above it)
        const bool none = best.key >= kNoCand;
        idx_out[(size_t)q * k + lane] = none ? -1 : key_idx(best.key);
        d2_out[(size_t)q * k + lane] = none ? FLT_MAX : key_d2(best.key);
    }
}

// ---- small target clouds (up to kSmallMaxTargets points: a voxel-downsampled scan) ----------------------------------------------
// The whole target cloud sits in LDS (x | y | z planes, 12 bytes a point) of every workgroup and ONE launch answers every query:
// a wave takes kSmallQ queries at a time and its 64 lanes deal the targets among themselves (lane l: targets l, l + 64, ...).
//   pass 1  every lane's nearest target per query; the k-th smallest of the 64 lane minima (64-lane bitonic sort) is an upper
//           bound tau of the k-th neighbour's distance — k different targets are that near
//   pass 2  the targets with d <= tau (k of them, plus whatever shares a lane with a nearer one: a handful) go to a list in LDS
//           by ballot + prefix count; 64 entries a query, more is the overflow path below
//   select  one candidate per lane, the 64-lane sort by (distance, index), the first k are the answer — the brute-force rule
//           (ties to the lower index); overflow: every target through insert_candidates (knn_bf_select_kernel's loop)
// Same expression for the distance as everywhere (dist2), evaluated twice instead of stored. The bounded pipeline above is a
// dozen launches (boxes, frames, pass A on the matrix cores, offsets, lists, collect, select): ~110 us for two 6 k-point clouds
// with k = 10 whatever the work; this is one launch of ~10 k wave instructions per four queries.
constexpr unsigned kSmallMinTargets = 256, kSmallMaxTargets = 12032;
constexpr int kSmallBlock = 512, kSmallQMax = 4;  // lanes a workgroup; queries a wave scans together (1 .. 4: template parameter)
constexpr unsigned kSmallListCap = 64;
constexpr size_t small_lds_bytes(unsigned nt) {
    return (size_t)3 * ((nt + 63u) & ~63u) * sizeof(float) + (size_t)(kSmallBlock / 64) * kSmallQMax * kSmallListCap * sizeof(unsigned long long);
}
template <int kSmallQ>
__global__ __launch_bounds__(kSmallBlock) void knn_bf_small_kernel(const float4* __restrict__ queries, unsigned nq,
                                                                   const float4* __restrict__ targets, unsigned nt, int k,
                                                                   int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
    extern __shared__ float small_lds[];
    const unsigned ntp = (nt + 63u) & ~63u;
    float* const tx = small_lds;
    float* const ty = small_lds + ntp;
    float* const tz = small_lds + 2 * (size_t)ntp;
    unsigned long long* const lists = reinterpret_cast<unsigned long long*>(small_lds + 3 * (size_t)ntp);
    for (unsigned j = threadIdx.x; j < ntp; j += kSmallBlock) {
        const float4 p = targets[min(j, nt - 1)];
        tx[j] = p.x; ty[j] = p.y; tz[j] = p.z;
    }
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr unsigned kWaves = kSmallBlock / 64;
    unsigned long long* const my_lists = lists + (size_t)wave * kSmallQ * kSmallListCap;
    const unsigned groups = (nq + kSmallQ - 1) / kSmallQ;
    const float inf = __builtin_inff();
    for (unsigned g = blockIdx.x * kWaves + wave; g < groups; g += gridDim.x * kWaves) {
        float qx[kSmallQ], qy[kSmallQ], qz[kSmallQ], mn[kSmallQ], tau[kSmallQ];
#pragma unroll
        for (int u = 0; u < kSmallQ; ++u) {
            const float4 q = queries[min(g * kSmallQ + u, nq - 1)];
            qx[u] = q.x; qy[u] = q.y; qz[u] = q.z;
            mn[u] = inf;
        }
#pragma unroll 4
        for (unsigned j = lane; j < ntp; j += 64) {
            const float x = tx[j], y = ty[j], z = tz[j];
            const bool valid = j < nt;
#pragma unroll
            for (int u = 0; u < kSmallQ; ++u) {
                const float d = dist2(qx[u], qy[u], qz[u], x, y, z);
                mn[u] = (valid && d < mn[u]) ? d : mn[u];
            }
        }
#pragma unroll
        for (int u = 0; u < kSmallQ; ++u) {
            Cand c;
            c.key = cand_key(mn[u], (int)lane);
            c.pos = 0;
            c = bitonic_sort64(c, lane);
            tau[u] = key_d2(bcast_k(c.key, k - 1));
        }
        unsigned cnt[kSmallQ];
#pragma unroll
        for (int u = 0; u < kSmallQ; ++u) cnt[u] = 0;
        // four targets a lane per trip: their twelve LDS reads and the distances are independent of the (rare) list stores, which
        // the one-target form put between every two reads — the loop is bound by latency, not by issue (two busy waves a SIMD)
        constexpr int kT = 4;
        for (unsigned j0 = lane; j0 < ntp; j0 += 64 * kT) {
            float x[kT], y[kT], z[kT];
#pragma unroll
            for (int t = 0; t < kT; ++t) {
                const unsigned j = min(j0 + 64u * t, ntp - 64u + lane);  // (past the end: the last trip's own target again, not counted)
                x[t] = tx[j]; y[t] = ty[j]; z[t] = tz[j];
            }
            float d[kT][kSmallQ];
            unsigned long long any = 0ull;
#pragma unroll
            for (int t = 0; t < kT; ++t) {
                const bool valid = j0 + 64u * t < nt;
#pragma unroll
                for (int u = 0; u < kSmallQ; ++u) {
                    d[t][u] = dist2(qx[u], qy[u], qz[u], x[t], y[t], z[t]);
                    d[t][u] = valid ? d[t][u] : inf;  // (tau is finite or inf; inf <= inf is caught by `valid` below)
                    any |= __ballot(valid && d[t][u] <= tau[u]);
                }
            }
            if (any) {  // (uniform; one trip in ten: k of nt targets are hits)
#pragma unroll
                for (int t = 0; t < kT; ++t) {
                    const unsigned j = j0 + 64u * t;
                    const bool valid = j < nt;
#pragma unroll
                    for (int u = 0; u < kSmallQ; ++u) {
                        const bool hit = valid && d[t][u] <= tau[u];
                        const unsigned long long m = __ballot(hit);
                        if (m) {  // (uniform)
                            const unsigned pos = cnt[u] + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                            if (hit && pos < kSmallListCap) my_lists[u * kSmallListCap + pos] = cand_key(d[t][u], (int)j);
                            cnt[u] += (unsigned)__builtin_popcountll(m);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < kSmallQ; ++u) {
            const unsigned q = g * kSmallQ + u;
            if (q >= nq) continue;  // (uniform)
            Cand best;
            best.pos = 0;
            if (cnt[u] <= kSmallListCap) {
                Cand c;
                c.key = lane < cnt[u] ? my_lists[u * kSmallListCap + lane] : kNoCand;
                c.pos = 0;
                best = bitonic_sort64(c, lane);
            } else {  // more than 64 targets within tau (equal distances, duplicates): every target against the sorted top k
                const unsigned long long kmask = (1ull << k) - 1ull;  // k <= 20
                best.key = kNoCand;
                unsigned long long kth = kNoCand;
                for (unsigned base = 0; base < nt; base += 64) {
                    const unsigned j = base + lane;
                    Cand c;
                    c.key = j < nt ? cand_key(dist2(qx[u], qy[u], qz[u], tx[j], ty[j], tz[j]), (int)j) : kNoCand;
                    c.pos = 0;
                    if (base == 0) {
                        best = bitonic_sort64(c, lane);
                        kth = bcast_k(best.key, k - 1);
                    } else {
                        insert_candidates(c, best, kth, k, kmask, lane);
                    }
                }
            }
            if (lane < (unsigned)k) {
                const bool none = best.key >= kNoCand;  // (an empty slot, or a distance `d < FLT_MAX` would not have admitted)
                idx_out[(size_t)q * k + lane] = none ? -1 : key_idx(best.key);
                d2_out[(size_t)q * k + lane] = none ? FLT_MAX : key_d2(best.key);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
bool small_applies(size_t nq, size_t nt) {
    // (measured, self-search: 6 k x 6 k, k = 10: 30 us against 108 for the bounded pipeline; 12 k x 12 k: 145 against 113 — the
    // matrix cores' bounding pass wins from about 10^8 pairs on)
    return nt >= kSmallMinTargets && nt <= kSmallMaxTargets && nq * nt <= (size_t)80 * 1000 * 1000;
}
int run_small(const float* queries, size_t nq, const float* targets, size_t nt, size_t k, int32_t* idx_out, float* d2_out,
              hipStream_t st) {
    auto set_lds = [](const void* f) {
        return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes(kSmallMaxTargets)) == hipSuccess;
    };
    static const bool attr_ok = set_lds(reinterpret_cast<const void*>(knn_bf_small_kernel<1>)) &&
                                set_lds(reinterpret_cast<const void*>(knn_bf_small_kernel<2>)) &&
                                set_lds(reinterpret_cast<const void*>(knn_bf_small_kernel<3>)) &&
                                set_lds(reinterpret_cast<const void*>(knn_bf_small_kernel<4>));
    if (!attr_ok) return SP_ERR_HIP;
    // queries a wave scans together: as few as give every one of the device's 256 x 8 waves a group (a group's time is its scan's
    // latency, and a wave with nothing to do is a quarter of a CU idle: 6 k queries by fours leave a quarter of the waves empty)
    constexpr unsigned kWavesAll = 256u * (kSmallBlock / 64);
    const int q = (int)std::min<size_t>(kSmallQMax, std::max<size_t>(1, (nq + kWavesAll - 1) / kWavesAll));
    const unsigned groups = (unsigned)((nq + q - 1) / q);
    const unsigned grid = std::min(256u, (groups + kSmallBlock / 64 - 1) / (kSmallBlock / 64));
    const float4* const qp = reinterpret_cast<const float4*>(queries);
    const float4* const tp = reinterpret_cast<const float4*>(targets);
    const size_t lds = small_lds_bytes((unsigned)nt);
    switch (q) {
        case 1: knn_bf_small_kernel<1><<<grid, kSmallBlock, lds, st>>>(qp, (unsigned)nq, tp, (unsigned)nt, (int)k, idx_out, d2_out); break;
        case 2: knn_bf_small_kernel<2><<<grid, kSmallBlock, lds, st>>>(qp, (unsigned)nq, tp, (unsigned)nt, (int)k, idx_out, d2_out); break;
        case 3: knn_bf_small_kernel<3><<<grid, kSmallBlock, lds, st>>>(qp, (unsigned)nq, tp, (unsigned)nt, (int)k, idx_out, d2_out); break;
        default: knn_bf_small_kernel<4><<<grid, kSmallBlock, lds, st>>>(qp, (unsigned)nq, tp, (unsigned)nt, (int)k, idx_out, d2_out); break;
    }
    return launch_status();
}

// Which approximate pass A runs (sp_knn_bruteforce_set_pass_a: 0 = matrix cores (default), 1 = packed VALU): both are kept
// because both are measured in bench.py's stages block; the lists do not depend on the choice.
int g_pass_a_valu = 0;
// (sp_knn_bruteforce_set_pass_a(2 / 3): the one-launch path for small target clouds off / on — tests compare the two)
int g_small_path = 1;

struct BfPlan {
    unsigned qblocks, nsplit, chunk;
    int qpt;
};

BfPlan plan(size_t nq, size_t nt, size_t k) {
    BfPlan p;
    p.qpt = (k == 1) ? kQ1 : 1;
    p.qblocks = div_up(nq, (size_t)kBlock * p.qpt);
    const unsigned max_split = div_up(nt, kTile);
    unsigned want = div_up((size_t)kNumCU * 4, p.qblocks);  // >= 4 workgroups per CU
    if (want < 1) want = 1;
    p.nsplit = want > max_split ? max_split : want;
    if (p.nsplit < 1) p.nsplit = 1;
    if (p.nsplit > 65535) p.nsplit = 65535;
    unsigned chunk = div_up(nt, p.nsplit);
    chunk = div_up(chunk, kTile) * kTile;  // whole LDS tiles per chunk
    p.chunk = chunk ? chunk : kTile;
    p.nsplit = div_up(nt, p.chunk);
    if (p.nsplit < 1) p.nsplit = 1;
    return p;
}

template <int KCAP, int QPT>
int run(const float* q, size_t nq, const float* t, size_t nt, size_t k, int32_t* idx, float* d2, void* ws,
        const BfPlan& p, hipStream_t st) {
    int32_t* pidx = idx;
    float* pd2 = d2;
    if (p.nsplit > 1) {
        pidx = static_cast<int32_t*>(ws);
        pd2 = reinterpret_cast<float*>(pidx + (size_t)p.nsplit * nq * k);
    }
    knn_bf_kernel<KCAP, QPT><<<dim3(p.qblocks, p.nsplit), kBlock, 0, st>>>(
        reinterpret_cast<const float4*>(q), (unsigned)nq, reinterpret_cast<const float4*>(t), (unsigned)nt, (int)k,
        p.chunk, pidx, pd2);
    if (p.nsplit > 1)
        knn_bf_merge_kernel<KCAP><<<div_up(nq, kBlock), kBlock, 0, st>>>(pidx, pd2, (unsigned)nq, (int)k, p.nsplit,
                                                                          idx, d2);
    return launch_status();
}

int run_k1(const float* q, size_t nq, const float* t, size_t nt, int32_t* idx, float* d2, void* ws, const BfPlan& p,
           hipStream_t st) {
    int32_t* pidx = idx;
    float* pd2 = d2;
    if (p.nsplit > 1) {
        pidx = static_cast<int32_t*>(ws);
        pd2 = reinterpret_cast<float*>(pidx + (size_t)p.nsplit * nq);
    }
    knn_bf_k1_kernel<false><<<dim3(p.qblocks, p.nsplit), kBlock, 0, st>>>(reinterpret_cast<const float4*>(q), (unsigned)nq,
                                                                   reinterpret_cast<const float4*>(t), (unsigned)nt,
                                                                   p.chunk, pidx, pd2);
    if (p.nsplit > 1)
        knn_bf_merge_kernel<1><<<div_up(nq, kBlock), kBlock, 0, st>>>(pidx, pd2, (unsigned)nq, 1, p.nsplit, idx, d2);
    return launch_status();
}

// Plans and workspace of the two-pass path (k > 1). Used when the targets split into at least k chunks of whole LDS tiles.
struct BoundedPlan {
    bool use;
    BfPlan a, b;                                       // pass A = the k = 1 kernel over G chunks; pass B = two queries per lane
    size_t off_min, off_bound, off_counts, off_chunk_list, off_cand, off_operands, bytes;  // workspace layout
};

// From this many targets on the two-pass path is taken (when they split into at least k chunks). It used to start at 16 K; on
// the 1 k .. 16 k-point clouds below that (the reference example's downsampled scans) the single-pass kernel's sorted insertion
// made k = 10 / 20 cost 0.4 / 1.1 ms whatever the size, where the two passes need 0.1 ms (scratch/bf_small.py).
constexpr size_t kBoundedMinTargets = 2048;
constexpr size_t kInterleaveMaxTargets = 65536;  // interleaved chunks up to here (run_bounded)
BoundedPlan plan_bounded(size_t nq, size_t nt, size_t k) {
    BoundedPlan P{};
    if (k < 1 || nt < (size_t)kBoundedMinTargets) return P;
    if (k == 1 && nt > kApproxMaxTargets) return P;  // (exact minima only: the k = 1 kernel finishes the job itself)
    P.a.qpt = kQ1;
    P.a.qblocks = div_up(nq, (size_t)kBlock * kQ1);
    P.a.chunk = div_up(div_up(nt, (size_t)kMaxChunks), (size_t)256) * 256;  // as many chunks as the need mask has bits
    P.a.nsplit = div_up(nt, P.a.chunk);
    if (P.a.nsplit < k) return P;
    P.b.qpt = 2;
    P.b.qblocks = div_up(nq, (size_t)kBlock * 2);  // (upper bound: workgroups beyond a chunk's list return at once)
    P.b.chunk = P.a.chunk;
    P.b.nsplit = P.a.nsplit;
    P.use = true;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    P.off_min = 0;
    P.off_bound = up(P.off_min + (size_t)P.a.nsplit * nq * 4);
    // chunk list lengths | offsets | work-item offsets (kMaxChunks (+1) each) | candidates per query | per-wave counts, chunk-major
    P.off_counts = up(P.off_bound + nq * 4 + 64 + nq * 32);  // (bound | box + frame | need mask, 8 words per query)
    P.off_chunk_list = up(P.off_counts + (3 * (size_t)kMaxChunks + 64 + nq + (size_t)P.a.nsplit * div_up(nq, (size_t)64)) * 4);
    P.off_cand = up(P.off_chunk_list + nq * (size_t)kMaxNeeded * 4);
    P.off_operands = up(P.off_cand + nq * (size_t)kCandCap * 8);  // bf16 rows of the targets | columns of the queries | |q^|^2
    const size_t rows = (size_t)P.a.nsplit * P.a.chunk, cols = div_up(nq, (size_t)kMfmaQueries) * (size_t)kMfmaQueries;
    P.bytes = P.off_operands + rows * 32 + cols * 36;
    return P;
}

int run_bounded(const float* q, size_t nq, const float* t, size_t nt, size_t k, int32_t* idx, float* d2, void* ws,
                const BoundedPlan& P, hipStream_t st) {
    char* w = static_cast<char*>(ws);
    float* amin = reinterpret_cast<float*>(w + P.off_min);
    float* bound = reinterpret_cast<float*>(w + P.off_bound);
    unsigned* box = reinterpret_cast<unsigned*>(bound + nq);
    BfFrame* frame = reinterpret_cast<BfFrame*>(box + 8);
    uint4* need_mask = reinterpret_cast<uint4*>(w + ((P.off_bound + nq * 4 + 64 + 15) & ~(size_t)15));
    unsigned* chunk_cnt = reinterpret_cast<unsigned*>(w + P.off_counts);
    unsigned* chunk_off = chunk_cnt + kMaxChunks;
    unsigned* blk_off = chunk_cnt + 2 * kMaxChunks;
    unsigned* cand_cnt = chunk_cnt + 3 * kMaxChunks + 64;
    unsigned* wave_cnt = cand_cnt + nq;
    unsigned* chunk_list = reinterpret_cast<unsigned*>(w + P.off_chunk_list);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(w + P.off_cand);
    const float4* q4 = reinterpret_cast<const float4*>(q);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    const unsigned G = P.a.nsplit, nwaves = div_up(nq, (size_t)64);
    const bool approx = nt <= kApproxMaxTargets;
    // Chunks are INTERLEAVED for small target clouds (chunk c = targets c, c + G, c + 2G, ...): the bound is the k-th smallest
    // chunk minimum, which is tight when every chunk is a uniform sample of the cloud and useless when chunks are spatial
    // clumps — a cloud stored in voxel order (what downsampling yields) made contiguous chunks list hundreds of candidates per
    // query and sent most queries to the overflow rescan (0.21 ms per 6 k-point cloud, k = 10, against 0.04). The strided
    // reads this costs stay inside L2 for clouds of this size; larger clouds keep contiguous chunks.
    const unsigned il = (approx && nt <= kInterleaveMaxTargets) ? G : 0u;
    if (approx) {
        const unsigned rows = G * P.a.chunk, cols = div_up(nq, (size_t)kMfmaQueries) * kMfmaQueries;
        uint4* tA = reinterpret_cast<uint4*>(w + P.off_operands);
        uint4* qB = tA + 2 * (size_t)rows;
        float* qq = reinterpret_cast<float*>(qB + 2 * (size_t)cols);
        knn_bf_box_init_kernel<<<1, 64, 0, st>>>(box);
        knn_bf_box_kernel<<<std::min(div_up(nt, kBlock), 64u), kBlock, 0, st>>>(t4, (unsigned)nt, box);
        knn_bf_frame_kernel<<<1, 64, 0, st>>>(box, frame);
        if (g_pass_a_valu) {
            knn_bf_chunkmin_kernel<<<dim3(P.a.qblocks, G), kBlock, 0, st>>>(q4, (unsigned)nq, t4, (unsigned)nt, P.a.chunk, frame, amin, il);
        } else {
            knn_bf_prep_targets_kernel<<<div_up(rows, kBlock), kBlock, 0, st>>>(t4, (unsigned)nt, rows, frame, tA, P.a.chunk, il);
            knn_bf_prep_queries_kernel<<<div_up(cols, kBlock), kBlock, 0, st>>>(q4, (unsigned)nq, cols, frame, qB, qq);
            const unsigned cpw = 2;  // chunks per workgroup (1 .. 14 and 2 / 4 / 8 tiles per wave all within 10 %: scratch/bf_sweep.sh)
            knn_bf_chunkmin_mfma_kernel<<<dim3(cols / kMfmaQueries, div_up(G, cpw)), kBlock, 0, st>>>(tA, qB, qq, (unsigned)nq, P.a.chunk,
                                                                                                      G, cpw, amin);
        }
    } else {
        knn_bf_k1_kernel<true><<<dim3(P.a.qblocks, G), kBlock, 0, st>>>(q4, (unsigned)nq, t4, (unsigned)nt, P.a.chunk, nullptr, amin);
    }
    knn_bf_bound_kernel<<<div_up(nq, kBlock), kBlock, 0, st>>>(amin, (unsigned)nq, (int)k, G, q4, approx ? frame : nullptr,
                                                               g_pass_a_valu ? kValuErr : kMfmaErr, bound, need_mask, wave_cnt, nwaves,
                                                               cand_cnt);
    knn_bf_wave_offsets_kernel<<<G, kBlock, 0, st>>>(wave_cnt, nwaves, chunk_cnt);
    // few (query, chunk) pairs at small k: more, shorter work items (sub-ranges of a chunk's targets) keep every CU busy
    const unsigned subs = k <= 4 ? 4u : (k <= 10 ? 2u : 1u);
    knn_bf_offsets_kernel<<<1, kMaxChunks, 0, st>>>(chunk_cnt, G, subs, chunk_off, blk_off);
    knn_bf_lists_kernel<<<div_up(nq, kBlock), kBlock, 0, st>>>(need_mask, (unsigned)nq, G, chunk_off, wave_cnt, nwaves, chunk_list);
    knn_bf_collect_kernel<<<kNumCU * 8, kBlock, 0, st>>>(q4, (unsigned)nq, t4, (unsigned)nt, P.b.chunk, bound, chunk_cnt, chunk_off,
                                                         blk_off, G, subs, chunk_list, cand_cnt, cand, il);
    knn_bf_select_kernel<<<div_up(nq * 64, kBlock), kBlock, 0, st>>>(q4, (unsigned)nq, t4, (unsigned)nt, (int)k, cand_cnt, cand, idx, d2);
    return launch_status();
}

__global__ void fill_empty_kernel(int32_t* idx, float* d2, size_t n) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) { idx[i] = -1; d2[i] = FLT_MAX; }
}

}  // namespace
}  // namespace sp

extern "C" int sp_knn_bruteforce_set_pass_a(int valu) {
    if (valu == 2 || valu == 3) sp::g_small_path = valu == 3 ? 1 : 0;
    else sp::g_pass_a_valu = valu ? 1 : 0;
    return SP_OK;
}

extern "C" size_t sp_knn_bruteforce_workspace_bytes(size_t nq, size_t nt, size_t k) {
    if (nq == 0 || nt == 0 || k == 0) return 0;
    const sp::BoundedPlan bp = sp::plan_bounded(nq, nt, k);
    if (bp.use) return bp.bytes;
    const sp::BfPlan p = sp::plan(nq, nt, k);
    return p.nsplit > 1 ? (size_t)p.nsplit * nq * k * 8 : 0;
}

void sp_set_error(const char* msg);

extern "C" int sp_knn_bruteforce(const float* queries, size_t nq, const float* targets, size_t nt, size_t k,
                                 int32_t* idx_out, float* d2_out, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    using namespace sp;
    if (k == 0 || k > 20) {
        sp_set_error("[knn_search_bruteforce] k must be in [1, 20] (MAX_K = 20, bruteforce.hpp:26)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (nq >= (1ull << 31) || nt >= (1ull << 31)) {
        sp_set_error("[knn_search_bruteforce] more than 2^31 points: indices are int32");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (nq == 0) return SP_OK;
    hipStream_t st = as_stream(stream);
    if (nt == 0) {  // no targets: every slot keeps its initial -1 / FLT_MAX (knn/result.hpp:21-27)
        fill_empty_kernel<<<div_up(nq * k, kBlock), kBlock, 0, st>>>(idx_out, d2_out, nq * k);
        return launch_status();
    }
    if (g_small_path && small_applies(nq, nt)) return run_small(queries, nq, targets, nt, k, idx_out, d2_out, st);
    const BoundedPlan bp = plan_bounded(nq, nt, k);
    if (bp.use) {
        if (workspace == nullptr || workspace_bytes < bp.bytes) {
            sp_set_error("[knn_search_bruteforce] workspace too small (sp_knn_bruteforce_workspace_bytes)");
            return SP_ERR_INVALID_ARGUMENT;
        }
        return run_bounded(queries, nq, targets, nt, k, idx_out, d2_out, workspace, bp, st);
    }
    const BfPlan p = plan(nq, nt, k);
    if (p.nsplit > 1 && (workspace == nullptr || workspace_bytes < (size_t)p.nsplit * nq * k * 8)) {
        sp_set_error("[knn_search_bruteforce] workspace too small (sp_knn_bruteforce_workspace_bytes)");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (k == 1) return run_k1(queries, nq, targets, nt, idx_out, d2_out, workspace, p, st);
    if (k <= 5) return run<5, 1>(queries, nq, targets, nt, k, idx_out, d2_out, workspace, p, st);
    if (k <= 10) return run<10, 1>(queries, nq, targets, nt, k, idx_out, d2_out, workspace, p, st);
    return run<20, 1>(queries, nq, targets, nt, k, idx_out, d2_out, workspace, p, st);
}

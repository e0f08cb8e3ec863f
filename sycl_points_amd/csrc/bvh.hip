// Device-built bounding-volume hierarchy for exact kNN on clouds of ANY density profile
// (the job of KDTree::build + knn_search_async, algorithms/knn/kdtree.hpp:292-413, 424-562, without the host build).
//
// Why a second structure: GridKNN (grid.hip) sizes its cells for the cloud's AVERAGE density. On the reference's bundled
// LiDAR scan (points on surfaces, density 5000 : 1 between cells) a query near the sensor scans thousands of candidates and
// the KD-tree is 3-10x faster (profiles/r02_h_real_cloud_grid_vs_kdtree.txt) — but the reference's KD-tree is built on the
// host by recursive nth_element: 30 ms per 1M points on 16 threads, every frame (pipeline/submapping.hpp:197). This tree
// follows the points wherever they are and is built by the device in a fraction of a millisecond (Karras 2012, "Maximizing
// parallelism in the construction of BVHs, octrees and k-d trees"):
//   build   bounding box (integer atomics) -> 48-bit Morton key of every point (box read from device memory: no host round
//           trip) -> the library's radix sort -> points gathered in Morton order -> one lane per internal node finds its key
//           range and split from the longest common prefixes of neighbouring keys (equal keys: the index breaks the tie), so a
//           node is an octree cell and siblings are DISJOINT -> boxes bottom-up, one lane per point walking towards the root;
//           the second child to arrive at a node carries on (write-through stores + an agent-scope ticket, no fence).
//           A node keeps BOTH children's boxes (64 bytes: one line per visit).
//   search  one lane per query, depth-first, per-lane stack in LDS. A subtree of <= 16 points is a leaf: its points are
//           consecutive and come as batches of independent loads. Exact: a subtree is skipped only when its box lies strictly
//           beyond the current k-th distance, and lists are (distance, index)-lexicographic like brute force, so results are
//           bit-identical to knn_search_bruteforce, ties included. (A query that would overflow the stack scans all points.)
//   self    the cloud's own points are the queries, taken in Morton order (neighbouring lanes walk the same subtrees:
//           their node and leaf loads share cache lines) and written back by original index.
#include "grid_device.h"
#include "radix_sort.h"
#include "sp_internal.h"
#include "sp_wave_select.h"

void sp_set_error(const char* msg);

struct sp_bvh {
    size_t n = 0;
    float4* pts = nullptr;    // n points in Morton order, w = original index bits
    float4* node = nullptr;   // 4 x float4 per internal node (n - 1): (Llo, first) (Lhi, split) (Rlo, last) (Rhi, -)
    float4* obox = nullptr;   // 2 x float4 per internal node: its own box (re-tested when a stacked node is taken up again);
                              // lo.w = the lowest original index below the node (ties, see bvh_search_kernel)
    unsigned* bbox = nullptr; // the cloud's bounding box as the build found it (6 order-preserving words): external queries are
                              // sorted along the same curve before a search
    mutable sp::StreamSet streams;
    int sort_queries = 1;     // external queries (>= 400 k of them) searched in the order of the tree's curve (0: as given)
    int self_heap = 1;        // searches by bvh_heap_kernel (0: the sorted-insertion kernel alone; tests, comparisons)
};

namespace sp {
namespace {

constexpr int kBvhLeaf = 16;
constexpr int kBvhStack = 48;
constexpr uint64_t kBvhInvalidKey = (1ull << 48) - 1ull;  // cell (65535, 65535, 65535): no valid point gets it

__device__ __forceinline__ unsigned enc_f(float f) {  // order-preserving float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void bvh_bbox_init_kernel(unsigned* bbox) {
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0u;
}
// bbox[0..2] = min xyz, bbox[3..5] = max xyz (encoded), over the finite points; four loads in flight per lane.
__global__ __launch_bounds__(kBlock) void bvh_bbox_kernel(const float4* __restrict__ pts, unsigned n, unsigned* bbox) {
    __shared__ unsigned red[kBlock / kWave][6];
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    const unsigned stride = gridDim.x * kBlock;
    for (unsigned i0 = blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += 4 * stride) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = pts[min(i0 + u * stride, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * stride < n && isfinite(p[u].x) && isfinite(p[u].y) && isfinite(p[u].z)) {
                const unsigned e[3] = {enc_f(p[u].x), enc_f(p[u].y), enc_f(p[u].z)};
#pragma unroll
                for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], e[a]); mx[a] = max(mx[a], e[a]); }
            }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = min(mn[a], (unsigned)__shfl_xor((int)mn[a], o, 64));
            mx[a] = max(mx[a], (unsigned)__shfl_xor((int)mx[a], o, 64));
        }
    const unsigned wave = threadIdx.x / kWave;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[wave][a] = mn[a]; red[wave][3 + a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned v = red[0][threadIdx.x];
        for (int w = 1; w < kBlock / kWave; ++w) v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&bbox[threadIdx.x], v);
        else atomicMax(&bbox[threadIdx.x], v);
    }
}

__device__ __forceinline__ uint64_t spread21(uint64_t x) {  // up to 21 bits -> every third bit
    x = (x | (x << 32)) & 0x001f00000000ffffull;
    x = (x | (x << 16)) & 0x001f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__global__ __launch_bounds__(kBlock) void bvh_key_kernel(const float4* __restrict__ pts, unsigned n,
                                                         const unsigned* __restrict__ bbox, uint64_t* __restrict__ keys,
                                                         unsigned* __restrict__ vals) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    uint64_t key = kBvhInvalidKey;  // non-finite points: behind all others, in no box, never a neighbour (their distance is NaN)
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        const float v[3] = {p.x, p.y, p.z};
        uint64_t c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = dec_f(bbox[a]), hi = dec_f(bbox[3 + a]);
            const float ext = hi - lo;
            const float t = ext > 0.0f ? (v[a] - lo) / ext * 65535.0f : 0.0f;
            c[a] = (uint64_t)fminf(fmaxf(t, 0.0f), 65534.0f);
        }
        key = spread21(c[0]) | (spread21(c[1]) << 1) | (spread21(c[2]) << 2);
    }
    keys[i] = key;
    vals[i] = i;
}
__global__ __launch_bounds__(kBlock) void bvh_gather_kernel(const float4* __restrict__ pts, const unsigned* __restrict__ order,
                                                            unsigned n, float4* __restrict__ out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned src = order[i];
    float4 p = pts[src];
    p.w = __uint_as_float(src);
    out[i] = p;
}

// Length of the common prefix of the (key, position) pairs i and j; -1 outside the array.
__device__ __forceinline__ int bvh_delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    return a == b ? 64 + __clz((unsigned)(i ^ j)) : __clzll((long long)(a ^ b));
}
// Karras 2012, section 4: internal node i covers the keys [first, last] and splits them behind `split`; its children are
// node `split` (a point when first == split) and node `split + 1` (a point when split + 1 == last).
__global__ __launch_bounds__(kBlock) void bvh_hierarchy_kernel(const uint64_t* __restrict__ keys, int n, float4* __restrict__ node,
                                                               int* __restrict__ parent, int* __restrict__ leaf_parent,
                                                               unsigned* __restrict__ tickets) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (bvh_delta(keys, n, i, i + 1) - bvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = bvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (bvh_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (bvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = bvh_delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (bvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int split = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    // (the boxes arrive bottom-up; the three integers ride in the w slots)
    node[4 * (size_t)i] = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(first));
    node[4 * (size_t)i + 1] = make_float4(-INFINITY, -INFINITY, -INFINITY, __int_as_float(split));
    node[4 * (size_t)i + 2] = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(last));
    node[4 * (size_t)i + 3] = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    tickets[i] = 0u;
    if (first == split) leaf_parent[split] = i; else parent[split] = i;
    if (split + 1 == last) leaf_parent[last] = i; else parent[split + 1] = i;
    if (i == 0) parent[0] = -1;
}
// One lane per point, walking towards the root: it writes the box of the subtree it has finished into its parent's slot for
// that child and takes the parent's ticket; the first to arrive leaves, the second reads its sibling's box and carries the
// union on. Cross-workgroup hand-off without a fence (MI355X_MICROARCH.md, the counter form: every handed-off word is stored
// write-through (sc1) by ONE lane, that lane drains its stores before it adds to the agent-scope counter, and the lane whose
// add came second reads them with sc1 loads).
__global__ __launch_bounds__(kBlock) void bvh_box_kernel(const float4* __restrict__ spts, int n, float4* __restrict__ node,
                                                         float4* __restrict__ obox, const int* __restrict__ parent,
                                                         const int* __restrict__ leaf_parent, unsigned* __restrict__ tickets,
                                                         int* __restrict__ child_min) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = spts[i];
    int min_idx = __float_as_int(p.w);  // lowest original index of the finished subtree
    const bool ok = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
    float lo[3] = {ok ? p.x : INFINITY, ok ? p.y : INFINITY, ok ? p.z : INFINITY};
    float hi[3] = {ok ? p.x : -INFINITY, ok ? p.y : -INFINITY, ok ? p.z : -INFINITY};
    int child = i;
    bool child_is_point = true;
    int cur = leaf_parent[i];
    while (cur >= 0) {
        float* const rec = reinterpret_cast<float*>(node + 4 * (size_t)cur);
        // (first and split were written by the hierarchy launch: plain loads)
        const int first = __float_as_int(rec[3]), split = __float_as_int(rec[7]);
        // the left child of `cur` is number `split` — the POINT `split` when the left range is that single point
        const bool left = child_is_point ? (child == split && first == split) : (child == split && first != split);
        float* const mine = rec + (left ? 0 : 8);
        float* const other = rec + (left ? 8 : 0);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            __hip_atomic_store(mine + a, lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 4 + a, hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(child_min + 2 * (size_t)cur + (left ? 0 : 1), min_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (__hip_atomic_fetch_add(tickets + cur, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // first to arrive
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], __hip_atomic_load(other + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            hi[a] = fmaxf(hi[a], __hip_atomic_load(other + 4 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        min_idx = min(min_idx, __hip_atomic_load(child_min + 2 * (size_t)cur + (left ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        obox[2 * (size_t)cur] = make_float4(lo[0], lo[1], lo[2], __int_as_float(min_idx));
        obox[2 * (size_t)cur + 1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
        child = cur;
        child_is_point = false;
        cur = parent[cur];
    }
}

__device__ __forceinline__ float box_d2(float lx, float ly, float lz, float hx, float hy, float hz, float qx, float qy, float qz) {
    const float dx = fmaxf(fmaxf(lx - qx, qx - hx), 0.0f), dy = fmaxf(fmaxf(ly - qy, qy - hy), 0.0f),
                dz = fmaxf(fmaxf(lz - qz, qz - hz), 0.0f);
    // never above the true squared distance to any point inside (per axis the box's gap is below the point's; the factor
    // covers the different rounding of the two sums): a box is skipped only when it lies strictly beyond the k-th distance
    return fmaf(dx, dx, fmaf(dy, dy, dz * dz)) * 0.999999f;
}

// RADIUS (KDTree::radius_search_async, kdtree.hpp:574-719): the `k` nearest among the points within radius_sq (inclusive) —
// a candidate must lie inside the ball AND beat the current k-th; a subtree is skipped when its box lies beyond the smaller
// of the two bounds.
template <int KCAP, bool RADIUS = false>
__global__ __launch_bounds__(kBlock) void bvh_search_kernel(const float4* __restrict__ node, const float4* __restrict__ obox,
                                                            const float4* __restrict__ spts, unsigned n,
                                                            const float4* __restrict__ queries, unsigned nq, int k, Mat4Arg T_val,
                                                            const float* __restrict__ T_dev, int32_t* __restrict__ idx_out,
                                                            float* __restrict__ d2_out, float radius_sq = 0.0f,
                                                            const unsigned* __restrict__ todo = nullptr,
                                                            const unsigned* __restrict__ todo_count = nullptr) {
    __shared__ unsigned st_node[kBvhStack][kBlock];
    unsigned qi = blockIdx.x * kBlock + threadIdx.x;
    if (todo) {  // only the listed queries (what bvh_self_heap_kernel handed on)
        if (qi >= *todo_count) return;
        qi = todo[qi];
    }
    if (qi >= nq) return;  // no barrier below
    const unsigned lane = threadIdx.x;
    float qx, qy, qz;
    size_t row;
    if (queries) {
        const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
        const float4 q4 = queries[qi];
        transform_point(T, q4.x, q4.y, q4.z, qx, qy, qz);
        row = qi;
    } else {  // self-kNN: the cloud's own points in Morton order, results by original index
        const float4 q4 = spts[qi];
        qx = q4.x; qy = q4.y; qz = q4.z;
        row = __float_as_uint(q4.w);
        if (__float_as_int(q4.w) < 0) return;  // a removed point (sp_bvh_remove_by_flags): no row of its own any more
    }
    float bd[KCAP];
    int bi[KCAP];
#pragma unroll
    for (int i = 0; i < KCAP; ++i) { bd[i] = FLT_MAX; bi[i] = -1; }
    float kth = FLT_MAX;
    int kth_idx = -1;
    // Self-kNN: the query's neighbours in Morton order are mostly its neighbours in space, so they go into the list BEFORE the
    // descent — the k-th of them bounds the search from the first node on, instead of the bound staying infinite until k
    // points have been met on the way down. The descent then skips those positions [w0, w1] (a point must not enter twice).
    unsigned w0 = 1u, w1 = 0u;  // (empty)
    // the points [first, last], batches of eight independent loads
    auto scan = [&](unsigned first, unsigned last, bool skip_window) {
#pragma unroll 1
        for (unsigned b = first; b <= last; b += 8) {
            float4 slot[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) slot[s] = spts[min(b + s, last)];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float d = dist2(qx, qy, qz, slot[s].x, slot[s].y, slot[s].z);
                const int pi = __float_as_int(slot[s].w);
                const bool seen = skip_window && b + s >= w0 && b + s <= w1;
                // (a non-finite point gives a NaN distance: every comparison fails, it is never taken)
                if (b + s <= last && !seen && (!RADIUS || d <= radius_sq) && (d < kth || (d == kth && pi < kth_idx)))
                    lex_insert<KCAP>(bd, bi, k, d, pi, kth, kth_idx);
            }
        }
    };
    if (!queries && KCAP > 1 && n > (unsigned)kBvhLeaf && isfinite(qx) && isfinite(qy) && isfinite(qz)) {
        const unsigned half = (unsigned)(KCAP <= 10 ? 8 : 16);
        w0 = qi >= half ? qi - half : 0u;
        w1 = min(qi + half, n - 1u);
        scan(w0, w1, false);
    }
    if (isfinite(qx) && isfinite(qy) && isfinite(qz) && n != 0u) {
        if (n <= (unsigned)kBvhLeaf) {
            scan(0u, n - 1u, false);
        } else {
            int sp_top = 0;
            bool overflow = false;
            unsigned cur = 0;  // the root is always entered
            for (;;) {
                const float4 r0 = node[4 * (size_t)cur], r1 = node[4 * (size_t)cur + 1], r2 = node[4 * (size_t)cur + 2],
                             r3 = node[4 * (size_t)cur + 3];
                const unsigned first = __float_as_uint(r0.w), split = __float_as_uint(r1.w), last = __float_as_uint(r2.w);
                const float dl = box_d2(r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, qx, qy, qz);
                const float dr = box_d2(r2.x, r2.y, r2.z, r3.x, r3.y, r3.z, qx, qy, qz);
                const bool l_near = !(dr < dl);
                // nearer child first; a child of <= kBvhLeaf points is scanned on the spot, a larger one is entered / stacked
                unsigned next = ~0u;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const bool take_left = (pass == 0) == l_near;
                    const float dc = take_left ? dl : dr;
                    const unsigned cf = take_left ? first : split + 1u, cl = take_left ? split : last;
                    if (dc > kth || (RADIUS && dc > radius_sq)) continue;  // (an empty box is +inf away)
                    // A box exactly AT the k-th distance can only matter through a point at that very distance with a lower
                    // index than the k-th neighbour's (lists are (distance, index)-lexicographic). Clouds with thousands of
                    // copies of one point (invalid returns of a scan) would otherwise visit every copy from every copy.
                    if (dc == kth && cl > cf && __float_as_int(obox[2 * (size_t)(take_left ? split : split + 1u)].w) > kth_idx) continue;
                    if (cl - cf < (unsigned)kBvhLeaf) {
                        scan(cf, cl, true);
                    } else {
                        const unsigned child = take_left ? split : split + 1u;
                        if (next == ~0u) {
                            next = child;  // entered next (pass 1: the nearer child was a leaf or out of reach)
                        } else if (sp_top < kBvhStack) {
                            st_node[sp_top][lane] = child;
                            ++sp_top;
                        } else {
                            overflow = true;
                        }
                    }
                }
                if (overflow) break;
                if (next != ~0u) { cur = next; continue; }
                // take up the most recent stacked node that is still within reach
                bool found = false;
                while (sp_top > 0) {
                    --sp_top;
                    const unsigned c = st_node[sp_top][lane];
                    const float4 o0 = obox[2 * (size_t)c], o1 = obox[2 * (size_t)c + 1];
                    const float dc = box_d2(o0.x, o0.y, o0.z, o1.x, o1.y, o1.z, qx, qy, qz);
                    if (dc > kth || (RADIUS && dc > radius_sq) || (dc == kth && __float_as_int(o0.w) > kth_idx)) continue;
                    cur = c;
                    found = true;
                    break;
                }
                if (!found) break;
            }
            if (overflow) {  // a tree deeper than the stack (degenerate clouds): start over and look at every point
#pragma unroll
                for (int i = 0; i < KCAP; ++i) { bd[i] = FLT_MAX; bi[i] = -1; }
                kth = FLT_MAX; kth_idx = -1;
                scan(0u, n - 1u, false);
            }
        }
    }
    const size_t o = row * (size_t)k;
#pragma unroll
    for (int i = 0; i < KCAP; ++i)
        if (i < k) { d2_out[o + i] = bd[i]; idx_out[o + i] = bi[i]; }
}

// ---------------------------------------------------------------------------------------------------------------
// Searches (lists of up to 32 entries) with the running k best of a lane in a 4-ary MAX-HEAP of (distance, index) keys: the
// root and its four children on registers, the sixteen grandchildren in the lane's LDS column. Replacing the root costs two
// sift steps (the largest of four, twice) whatever k is, where the sorted list of bvh_search_kernel is a k-step select chain
// that the whole wave walks whenever ANY of its 64 lanes accepts a candidate — with lanes in different leaves that is
// nearly every candidate (107 k vector wave-instructions per 64 queries at k = 20, profiles/r03_bvh_vs_kdtree_vs_grid.txt).
// The walk is restructured as well: the lanes of a wave first all find a leaf, then all scan one (bvh_walk), instead of
// scanning inline in the descent with a few lanes active. The bound of the walk starts at the k-th smallest distance to the
// lane's 33 neighbours in Morton order when the cloud's own points are the queries (bisection on the bit patterns of the 33
// distances, which stay on registers), at the radius for a radius search, at infinity otherwise, and follows the root once k
// candidates are in. At the end the k keys are
// ranked by counting. Exact and bit-identical to the sorted-insertion kernel: keys are (distance, index)-lexicographic, a
// subtree is skipped only when its box lies strictly beyond the bound or exactly at it without a lower index inside.
// A non-finite query point or a tree deeper than the stack goes to that kernel through a list.
constexpr int kSelHalf = 16;   // Morton neighbours either side
constexpr int kHeapStack = 20; // stacked far siblings per lane in LDS (with the heap: 52 KB per workgroup, three per CU) ...
constexpr int kHeapStackDeep = 28;  // ... and behind them in HBM, in a slot of a shared arena the lane claims when it first gets
                                    // there (an atomic counter; arena full: handed on): a handful of queries per million do
                                    // (a Morton neighbourhood across a jump of the curve starts them with a wide bound). Without
                                    // it they were handed on, and the other kernel's answer to ITS full stack is a scan of all
                                    // points by one lane: 1.5 ms of the 3.4 ms of the non-uniform 1 M cloud for 11 queries.

// Depth-first walk of the hierarchy from the root for the query (qx, qy, qz): leaf_fn(first, last) for every run of <= 32
// consecutive points (a leaf, or two sibling leaves) whose box is not beyond (bound, bound_idx), both read anew at every
// test (leaf_fn lowers them). Returns false when the per-lane stack overflowed.
template <int STACK, class LeafFn>
__device__ __forceinline__ bool bvh_walk(const float4* __restrict__ node, const float4* __restrict__ obox,
                                         unsigned (*st_node)[kBlock], unsigned* __restrict__ arena,
                                         unsigned* __restrict__ arena_count, unsigned arena_slots,
                                         unsigned lane, float qx, float qy, float qz,
                                         const float& bound, const int& bound_idx, bool active, LeafFn&& leaf_fn) {
    int sp_top = 0;
    bool ok = true;
    unsigned* deep = nullptr;  // this lane's kHeapStackDeep words of the arena, once claimed
    unsigned cur = 0;
    bool have_cur = active;
    unsigned pf = 1u, pl = 0u;  // pending run of points (empty)
    // a box exactly AT the bound matters only through a point at that very distance with a lower index than the k-th's
    auto tie_skip = [&](float dc, unsigned child) {
        return dc == bound && __float_as_int(obox[2 * (size_t)child].w) > bound_idx;
    };
    while (have_cur || pl >= pf) {
        while (have_cur && pl < pf) {
            const float4 r0 = node[4 * (size_t)cur], r1 = node[4 * (size_t)cur + 1], r2 = node[4 * (size_t)cur + 2],
                         r3 = node[4 * (size_t)cur + 3];
            const unsigned first = __float_as_uint(r0.w), split = __float_as_uint(r1.w), last = __float_as_uint(r2.w);
            const float dl = box_d2(r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, qx, qy, qz);
            const float dr = box_d2(r2.x, r2.y, r2.z, r3.x, r3.y, r3.z, qx, qy, qz);
            const bool ll = split - first < (unsigned)kBvhLeaf, rl = last - (split + 1u) < (unsigned)kBvhLeaf;
            bool al = !(dl > bound), ar = !(dr > bound);  // (an empty box is +inf away)
            if (al && !ll && split > first && tie_skip(dl, split)) al = false;
            if (ar && !rl && last > split + 1u && tie_skip(dr, split + 1u)) ar = false;
            if (al && ll) { pf = first; pl = split; }
            if (ar && rl) { pf = (al && ll) ? first : split + 1u; pl = last; }
            const bool el = al && !ll, er = ar && !rl;
            if (el && er) {  // the nearer child next, the other one stacked
                const bool l_near = !(dr < dl);
                const unsigned far = l_near ? split + 1u : split;
                if (sp_top < STACK) {
                    st_node[sp_top][lane] = far;
                    ++sp_top;
                } else if (sp_top < STACK + kHeapStackDeep) {
                    if (deep == nullptr) {
                        const unsigned slot = atomicAdd(arena_count, 1u);
                        if (slot < arena_slots) deep = arena + (size_t)slot * kHeapStackDeep;
                    }
                    if (deep != nullptr) {
                        deep[sp_top - STACK] = far;
                        ++sp_top;
                    } else {
                        ok = false;  // (the arena is full: a cloud that sends thousands of lanes this deep)
                    }
                } else {
                    ok = false;
                }
                cur = l_near ? split : split + 1u;
            } else if (el) {
                cur = split;
            } else if (er) {
                cur = split + 1u;
            } else {  // the most recent stacked node that is still within reach
                have_cur = false;
                while (sp_top > 0) {
                    --sp_top;
                    const unsigned c = sp_top < STACK ? st_node[sp_top][lane] : deep[sp_top - STACK];
                    const float4 o0 = obox[2 * (size_t)c], o1 = obox[2 * (size_t)c + 1];
                    const float dc = box_d2(o0.x, o0.y, o0.z, o1.x, o1.y, o1.z, qx, qy, qz);
                    if (dc > bound || (dc == bound && __float_as_int(o0.w) > bound_idx)) continue;
                    cur = c;
                    have_cur = true;
                    break;
                }
            }
            if (!ok) { have_cur = false; pf = 1u; pl = 0u; sp_top = 0; }
        }
        if (pl >= pf) {
            leaf_fn(pf, pl);
            pf = 1u; pl = 0u;
        }
    }
    return ok;
}

// External queries in the order of the tree's own curve: neighbouring lanes then walk the same subtrees and their node and
// leaf loads share cache lines, as they do for the cloud's own points (1 M queries in arbitrary order, k = 20: 5.8 ms on the
// non-uniform cloud against 3.4 ms for the same points in tree order). Key = the build's 48-bit Morton code of the transformed
// query in the tree's bounding box (clamped; non-finite queries last), sorted by the library's radix sort.
__global__ __launch_bounds__(kBlock) void bvh_query_key_kernel(const float4* __restrict__ queries, unsigned nq, Mat4Arg T_val,
                                                               const float* __restrict__ T_dev, const unsigned* __restrict__ bbox,
                                                               uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nq) return;
    const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
    const float4 q4 = queries[i];
    float v[3];
    transform_point(T, q4.x, q4.y, q4.z, v[0], v[1], v[2]);
    uint64_t key = kBvhInvalidKey;
    if (isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2])) {  // (the key of bvh_key_kernel, clamped to the tree's box)
        uint64_t c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = dec_f(bbox[a]), hi = dec_f(bbox[3 + a]);
            const float ext = hi - lo;
            const float t = ext > 0.0f ? (v[a] - lo) / ext * 65535.0f : 0.0f;
            c[a] = (uint64_t)fminf(fmaxf(t, 0.0f), 65534.0f);
        }
        key = spread21(c[0]) | (spread21(c[1]) << 1) | (spread21(c[2]) << 2);
    }
    keys[i] = key;
    vals[i] = i;
}

// MODE 0: the cloud's own points in Morton order (rows by original index); 1: external queries searched at T * q;
// 2: the same within radius_sq (inclusive; KDTree::radius_search_async).
// KCAP 5 (root + four children, no LDS level), 10, 21 (sixteen grandchildren) or 32 (EIGHT children on registers and up to four
// grandchildren below each: 23 slots in LDS, two workgroups per CU — still two sift steps; the sorted list at k = 24 runs
// 12 ms per 1 M queries on 190 registers where this runs 3)
template <int KCAP, int MODE>
__global__ __launch_bounds__(kBlock) void bvh_heap_kernel(const float4* __restrict__ node, const float4* __restrict__ obox,
                                                          const float4* __restrict__ spts, unsigned n,
                                                          const float4* __restrict__ queries, unsigned nq, int k, Mat4Arg T_val,
                                                          const float* __restrict__ T_dev, float radius_sq,
                                                          int32_t* __restrict__ idx_out, float* __restrict__ d2_out,
                                                          unsigned* __restrict__ todo, unsigned* __restrict__ todo_count,
                                                          const unsigned* __restrict__ order, unsigned* __restrict__ arena,
                                                          unsigned arena_slots) {
    constexpr int F1 = KCAP > 21 ? 8 : 4;                     // children of the root (slots 1 .. F1)
    constexpr int kDeep = KCAP > 1 + F1 ? KCAP - 1 - F1 : 1;  // grandchildren (slots 1 + F1 ..)
    __shared__ unsigned st_node[kHeapStack][kBlock];
    __shared__ unsigned long long heap2[kDeep][kBlock];
    const unsigned gi = blockIdx.x * kBlock + threadIdx.x;
    if (gi >= nq) return;  // no barrier below
    const unsigned qi = (MODE != 0 && order) ? order[gi] : gi;  // (external queries: in the order of the tree's curve)
    const unsigned lane = threadIdx.x;
    float qx, qy, qz;
    size_t o;
    if (MODE == 0) {
        const float4 q4 = spts[qi];
        qx = q4.x; qy = q4.y; qz = q4.z;
        if (__float_as_int(q4.w) < 0) return;  // a removed point: no row of its own
        o = (size_t)__float_as_uint(q4.w) * (size_t)k;
    } else {
        const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
        const float4 q4 = queries[qi];
        transform_point(T, q4.x, q4.y, q4.z, qx, qy, qz);
        o = (size_t)qi * (size_t)k;
    }
    const bool mine = isfinite(qx) && isfinite(qy) && isfinite(qz);  // false: handed on to bvh_search_kernel

    // Where the walk starts to prune: the k-th smallest distance to 33 consecutive points of the Morton order around the query
    // (when k of them are valid) — the query's own neighbours on the curve for a self-search, for an external query those of
    // the leaf a greedy descent (nearer child, no stack) ends in; a radius search starts from the radius if that is smaller.
    // Nothing is inserted for it, so no point can enter twice. (Without it an external search ran twice as long as the
    // self-search of the same points: the bound stays infinite until k candidates have been met.)
    float top = MODE == 2 ? radius_sq : FLT_MAX;
    if (mine) {
        unsigned centre = qi;
        if (MODE != 0) {
            unsigned cur = 0;
            for (;;) {
                const float4 r0 = node[4 * (size_t)cur], r1 = node[4 * (size_t)cur + 1], r2 = node[4 * (size_t)cur + 2],
                             r3 = node[4 * (size_t)cur + 3];
                const unsigned first = __float_as_uint(r0.w), split = __float_as_uint(r1.w), last = __float_as_uint(r2.w);
                const float dl = box_d2(r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, qx, qy, qz);
                const float dr = box_d2(r2.x, r2.y, r2.z, r3.x, r3.y, r3.z, qx, qy, qz);
                const bool left = !(dr < dl);
                const unsigned cf = left ? first : split + 1u, cl = left ? split : last;
                if (cl - cf < (unsigned)kBvhLeaf) {
                    centre = cf + ((cl - cf) >> 1);
                    break;
                }
                cur = left ? split : split + 1u;
            }
        }
        unsigned wb[2 * kSelHalf + 1];
        unsigned hi = 0u, finite = 0u;
#pragma unroll
        for (int u = 0; u <= 2 * kSelHalf; ++u) {
            const long long pos = (long long)centre - kSelHalf + u;
            const bool inside = pos >= 0 && pos < (long long)n;
            const float4 p = spts[inside ? (size_t)pos : (size_t)centre];
            const float d = dist2(qx, qy, qz, p.x, p.y, p.z);
            const unsigned b = __float_as_uint(d);
            const bool valid = inside && b <= 0x7f7fffffu;  // (d >= 0: the bit pattern orders like the value; NaN and inf are above)
            wb[u] = valid ? b : 0xffffffffu;
            hi = valid ? max(hi, b) : hi;
            finite += valid ? 1u : 0u;
        }
        unsigned lo = 0u;  // count(<= hi) >= k throughout
#pragma unroll 1
        for (int it = 0; it < 20; ++it) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            unsigned c = 0;
#pragma unroll
            for (int u = 0; u <= 2 * kSelHalf; ++u) c += wb[u] <= mid ? 1u : 0u;
            if (c >= (unsigned)k) hi = mid;
            else lo = mid + 1u;
        }
        if (finite >= (unsigned)k) top = fminf(top, __uint_as_float(hi));
    }

    // slots below k start as (FLT_MAX, INT_MAX) — above every candidate, so the first k candidates replace them and the root is
    // a real key from then on; slots from k on hold 0, which is never the largest child and never moves
    unsigned long long h0 = kNoCand, hh[F1];
#pragma unroll
    for (int j = 0; j < F1; ++j) hh[j] = 1 + j < k ? kNoCand : 0ull;
    if (KCAP > 1 + F1)
#pragma unroll
        for (int j = 0; j < kDeep; ++j) heap2[j][lane] = 1 + F1 + j < k ? kNoCand : 0ull;
    float bound = top;
    int bound_idx = 0x7fffffff;
    const bool ok = bvh_walk<kHeapStack>(node, obox, st_node, arena, todo_count + 1, arena_slots, lane, qx, qy, qz, bound, bound_idx, mine, [&](unsigned first, unsigned last) {
#pragma unroll 1
        for (unsigned b = first; b <= last; b += 8) {
            float4 slot[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) slot[s] = spts[min(b + s, last)];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float d = dist2(qx, qy, qz, slot[s].x, slot[s].y, slot[s].z);
                const unsigned long long nk = cand_key(d, __float_as_int(slot[s].w));
                // (a removed / non-finite point: NaN, the comparison with `top` fails)
                if (b + s <= last && d <= top && nk < h0) {
                    // the root leaves; the newcomer sinks from its place past every larger child
                    unsigned long long m = hh[0];
                    int mi = 0;
#pragma unroll
                    for (int j = 1; j < F1; ++j)
                        if (hh[j] > m) { m = hh[j]; mi = j; }
                    if (m > nk) {
                        h0 = m;
                        unsigned long long put = nk;  // what child mi receives
                        if (KCAP > 1 + F1) {
                            const int base = 4 * mi;
                            unsigned long long c2[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) c2[j] = base + j < kDeep ? heap2[min(base + j, kDeep - 1)][lane] : 0ull;
                            unsigned long long m2 = c2[0];
                            int m2j = 0;
#pragma unroll
                            for (int j = 1; j < 4; ++j)
                                if (c2[j] > m2) { m2 = c2[j]; m2j = j; }
                            if (m2 > nk) {
                                heap2[base + m2j][lane] = nk;
                                put = m2;
                            }
                        }
#pragma unroll
                        for (int j = 0; j < F1; ++j) hh[j] = mi == j ? put : hh[j];
                    } else {
                        h0 = nk;
                    }
                }
            }
        }
        // once k candidates are in, the root bounds the walk
        if (h0 != kNoCand) {
            bound = key_d2(h0);
            bound_idx = key_idx(h0);
        }
    });
    if (!mine || !ok) {  // (a non-finite query gets the empty list from the other kernel)
        todo[atomicAdd(todo_count, 1u)] = qi;
        return;
    }
    // the k keys ranked by counting (equal keys can only be empty slots: they keep their order)
    unsigned long long key[KCAP];
    key[0] = h0;
#pragma unroll
    for (int j = 0; j < F1; ++j)
        if (1 + j < KCAP) key[1 + j] = hh[j];
#pragma unroll
    for (int i = 1 + F1; i < KCAP; ++i) key[i] = heap2[i - 1 - F1][lane];
#pragma unroll
    for (int i = 0; i < KCAP; ++i) {
        if (i < k) {
            unsigned rank = 0;
#pragma unroll
            for (int j = 0; j < KCAP; ++j) rank += (j < k && (key[j] < key[i] || (key[j] == key[i] && j < i))) ? 1u : 0u;
            const bool empty = key[i] == kNoCand;
            idx_out[o + rank] = empty ? -1 : key_idx(key[i]);
            d2_out[o + rank] = empty ? FLT_MAX : key_d2(key[i]);
        }
    }
}

__global__ __launch_bounds__(kBlock) void bvh_export_kernel(const float4* __restrict__ spts, unsigned n, float4* __restrict__ out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 p = spts[i];
    const unsigned o = __float_as_uint(p.w);
    p.w = 1.0f;
    out[o] = p;
}

// Lazy delete (KDTree::remove_nodes_by_flags, kdtree.hpp:282-284, 721-765): a removed point stays in its leaf with NaN
// coordinates (its distance is NaN, no comparison takes it) and index -1; a kept one is relabelled. The boxes stay as they are
// (conservative). `kept_before` = exclusive prefix sum of the flags.
__global__ __launch_bounds__(kBlock) void bvh_remove_points_kernel(float4* __restrict__ spts, unsigned n,
                                                                   const uint8_t* __restrict__ flags,
                                                                   const int32_t* __restrict__ new_indices, unsigned n_flags) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 p = spts[i];
    const int o = __float_as_int(p.w);
    if (o < 0 || (unsigned)o >= n_flags) return;  // removed earlier / outside the flag array: left alone
    if (flags[o]) {
        p.w = __int_as_float(new_indices[o]);
    } else {
        const float nan = __int_as_float(0x7fc00000);
        p = make_float4(nan, nan, nan, __int_as_float(-1));
    }
    spts[i] = p;
}
// The lowest index below a node (obox lo.w: what lets a box exactly at the k-th distance be skipped when it cannot hold a
// lower index) becomes a LOWER BOUND of the lowest NEW index below it: the number of kept points before the old lowest index.
// Exact for an order-preserving relabelling (the reference's: FilterByFlags::calculate_indices, a running count); any bound
// from below keeps the search exact.
__global__ __launch_bounds__(kBlock) void bvh_remove_minidx_kernel(float4* __restrict__ obox, unsigned n_internal,
                                                                   const uint32_t* __restrict__ kept_before, unsigned n_flags) {
    const unsigned c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= n_internal) return;
    const int old_min = __float_as_int(obox[2 * (size_t)c].w);
    if (old_min >= 0 && (unsigned)old_min < n_flags) obox[2 * (size_t)c].w = __int_as_float((int)kept_before[old_min]);
}
__global__ __launch_bounds__(kBlock) void bvh_flags_to_u32_kernel(const uint8_t* __restrict__ flags, unsigned n,
                                                                  uint32_t* __restrict__ out) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = flags[i] ? 1u : 0u;
}

template <int KCAP>
void launch_bvh(const sp_bvh* b, const float4* q, unsigned nq, int k, const Mat4Arg& Tv, const float* T_dev, int32_t* idx, float* d2,
                hipStream_t st, float radius_sq) {
    if (radius_sq >= 0.0f)
        bvh_search_kernel<KCAP, true><<<div_up(nq, kBlock), kBlock, 0, st>>>(b->node, b->obox, b->pts, (unsigned)b->n, q, nq, k, Tv,
                                                                           T_dev, idx, d2, radius_sq);
    else
        bvh_search_kernel<KCAP><<<div_up(nq, kBlock), kBlock, 0, st>>>(b->node, b->obox, b->pts, (unsigned)b->n, q, nq, k, Tv, T_dev,
                                                                     idx, d2);
}

// radius_sq < 0: plain kNN
int bvh_dispatch(const sp_bvh* b, const float4* q, unsigned nq, size_t k, const float* transT, int transT_on_device,
                 int32_t* idx_out, float* d2_out, hipStream_t st, float radius_sq = -1.0f) {
    if (k == 0 || k > 32) {
        sp_set_error("[BVH] k must be in 1..32");
        return SP_ERR_INVALID_ARGUMENT;
    }
    b->streams.note(st);
    Mat4Arg Tv;
    for (int i = 0; i < 16; ++i) Tv.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    if (transT && !transT_on_device)
        for (int i = 0; i < 16; ++i) Tv.m[i] = transT[i];
    const float* T_dev = transT_on_device ? transT : nullptr;
    const int kk = (int)k;
    if (k >= (q ? 1u : 2u) && b->n > (size_t)kBvhLeaf && b->self_heap) {
        // heap kernel; what it hands on (non-finite queries, a tree deeper than its stack) is finished by the sorted-insertion kernel
        unsigned* todo = nullptr;
        // (+ the arena for the deep part of the walks' stacks: a slot of kHeapStackDeep words per lane that gets there, claimed
        // through the counter behind todo_count — 64 K slots at most, 7 MB, where a column per query was 116 B a query)
        const unsigned arena_slots = (unsigned)std::min<size_t>(nq, 65536);
        if (pooled_alloc(&todo, ((size_t)nq + 2 + (size_t)arena_slots * kHeapStackDeep) * sizeof(unsigned), st) != hipSuccess) return SP_ERR_HIP;
        unsigned* const todo_count = todo + nq;  // [0] the list's length, [1] arena slots claimed
        unsigned* const deep = todo + nq + 2;
        int rc = zero_async(todo_count, 8, st);
        // external queries: sorted along the tree's curve first (0.15 ms per million: 10-20 % off a search of a million queries, a
        // loss below a few hundred thousand)
        uint32_t* sortbuf = nullptr;
        const unsigned* order = nullptr;
        if (rc == SP_OK && q && nq >= 400000u && b->sort_queries) {
            const size_t wsb = radix_sort_u64_workspace_bytes(nq);
            const size_t words = 6 * (size_t)nq + (wsb + 3) / 4 + 2;
            if (pooled_alloc(&sortbuf, words * sizeof(uint32_t), st) != hipSuccess) { rc = SP_ERR_HIP; }
            else {
                uint64_t *ka = reinterpret_cast<uint64_t*>(sortbuf), *kb = ka + nq;
                uint32_t *va = reinterpret_cast<uint32_t*>(kb + nq), *vb = va + nq;
                void* const tmp = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(vb + nq) + 7) & ~uintptr_t(7));
                bvh_query_key_kernel<<<div_up(nq, kBlock), kBlock, 0, st>>>(q, nq, Tv, T_dev, b->bbox, ka, va);
                bool in_b = false;
                rc = radix_sort_pairs_u64(ka, kb, va, vb, nq, 48, tmp, wsb, &in_b, st);
                order = in_b ? vb : va;
            }
        }
        if (rc == SP_OK) {
            const unsigned g = div_up(nq, kBlock);
            const unsigned n32 = (unsigned)b->n;
            const float r2 = radius_sq;
#define SP_BVH_HEAP(KC, OLDK)                                                                                                          \
    do {                                                                                                                                \
        if (!q) {                                                                                                                       \
            bvh_heap_kernel<KC, 0><<<g, kBlock, 0, st>>>(b->node, b->obox, b->pts, n32, q, nq, kk, Tv, T_dev, r2, idx_out, d2_out, todo, \
                                                        todo_count, order, deep, arena_slots);                                                             \
            bvh_search_kernel<OLDK><<<g, kBlock, 0, st>>>(b->node, b->obox, b->pts, n32, q, nq, kk, Tv, T_dev, idx_out, d2_out, 0.0f,    \
                                                         todo, todo_count);                                                            \
        } else if (r2 < 0.0f) {                                                                                                         \
            bvh_heap_kernel<KC, 1><<<g, kBlock, 0, st>>>(b->node, b->obox, b->pts, n32, q, nq, kk, Tv, T_dev, r2, idx_out, d2_out, todo, \
                                                        todo_count, order, deep, arena_slots);                                                             \
            bvh_search_kernel<OLDK><<<g, kBlock, 0, st>>>(b->node, b->obox, b->pts, n32, q, nq, kk, Tv, T_dev, idx_out, d2_out, 0.0f,    \
                                                         todo, todo_count);                                                            \
        } else {                                                                                                                        \
            bvh_heap_kernel<KC, 2><<<g, kBlock, 0, st>>>(b->node, b->obox, b->pts, n32, q, nq, kk, Tv, T_dev, r2, idx_out, d2_out, todo, \
                                                        todo_count, order, deep, arena_slots);                                                             \
            bvh_search_kernel<OLDK, true><<<g, kBlock, 0, st>>>(b->node, b->obox, b->pts, n32, q, nq, kk, Tv, T_dev, idx_out, d2_out,    \
                                                               r2, todo, todo_count);                                                  \
        }                                                                                                                               \
    } while (0)
            if (k <= 5) SP_BVH_HEAP(5, 10);
            else if (k <= 10) SP_BVH_HEAP(10, 10);
            else if (k <= 20) SP_BVH_HEAP(21, 20);
            else if (k == 21) SP_BVH_HEAP(21, 32);
            else SP_BVH_HEAP(32, 32);
#undef SP_BVH_HEAP
            rc = launch_status();
        }
        StreamSet used;
        used.note(st);
        pooled_free_after(todo, used);
        if (sortbuf) pooled_free_after(sortbuf, used);
        return rc;
    }
    if (k == 1) launch_bvh<1>(b, q, nq, kk, Tv, T_dev, idx_out, d2_out, st, radius_sq);
    else if (k <= 10) launch_bvh<10>(b, q, nq, kk, Tv, T_dev, idx_out, d2_out, st, radius_sq);
    else if (k <= 20) launch_bvh<20>(b, q, nq, kk, Tv, T_dev, idx_out, d2_out, st, radius_sq);
    else launch_bvh<32>(b, q, nq, kk, Tv, T_dev, idx_out, d2_out, st, radius_sq);
    return launch_status();
}

}  // namespace
}  // namespace sp

extern "C" void sp_bvh_destroy(sp_bvh* b) {
    if (!b) return;
    sp::pooled_free_after(b->pts, b->streams);
    sp::pooled_free_after(b->node, b->streams);
    sp::pooled_free_after(b->obox, b->streams);
    sp::pooled_free_after(b->bbox, b->streams);
    delete b;
}

extern "C" int sp_bvh_create(const float* points, size_t n, void* stream, sp_bvh** out) {
    using namespace sp;
    if (!out || (n && !points)) return SP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (n >= (1ull << 30)) {
        sp_set_error("[BVH] more than 2^30 points");
        return SP_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = as_stream(stream);
    sp_bvh* b = new sp_bvh();
    b->n = n;
    b->streams.note(st);
    const size_t ni = n > 1 ? n - 1 : 1;  // internal nodes
    ScratchBuf b_kin, b_kout, b_vin, b_vout, b_tmp, b_parent, b_lparent, b_tickets, b_cmin;
    const size_t tmp_bytes = radix_sort_u64_workspace_bytes(n ? n : 1);
    hipError_t e = pooled_alloc(&b->pts, (n ? n : 1) * sizeof(float4), st);
    if (e == hipSuccess) e = pooled_alloc(&b->node, 4 * ni * sizeof(float4), st);
    if (e == hipSuccess) e = pooled_alloc(&b->obox, 2 * ni * sizeof(float4), st);
    if (e == hipSuccess) e = pooled_alloc(&b->bbox, 8 * sizeof(unsigned), st);
    if (e == hipSuccess && n) e = b_kin.get(n * 8, st);
    if (e == hipSuccess && n) e = b_kout.get(n * 8, st);
    if (e == hipSuccess && n) e = b_vin.get(n * 4, st);
    if (e == hipSuccess && n) e = b_vout.get(n * 4, st);
    if (e == hipSuccess && n) e = b_tmp.get(tmp_bytes, st);
    if (e == hipSuccess && n) e = b_parent.get(ni * 4, st);
    if (e == hipSuccess && n) e = b_lparent.get(n * 4, st);
    if (e == hipSuccess && n) e = b_tickets.get(ni * 4, st);
    if (e == hipSuccess && n) e = b_cmin.get(2 * ni * 4, st);
    auto fail = [&](const char* msg) {
        sp_set_error(msg);
        (void)hipStreamSynchronize(st);
        sp_bvh_destroy(b);
        return SP_ERR_HIP;
    };
    if (e != hipSuccess) return fail(hipGetErrorString(e));
    const float4* pts = reinterpret_cast<const float4*>(points);
    if (n) {
        unsigned* const bbox = b->bbox;
        uint64_t *kin = b_kin.as<uint64_t>(), *kout = b_kout.as<uint64_t>();
        unsigned *vin = b_vin.as<unsigned>(), *vout = b_vout.as<unsigned>();
        bvh_bbox_init_kernel<<<1, 64, 0, st>>>(bbox);
        unsigned g = div_up(n, (size_t)kBlock * 16);
        bvh_bbox_kernel<<<g > 256u ? 256u : (g ? g : 1u), kBlock, 0, st>>>(pts, (unsigned)n, bbox);
        bvh_key_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(pts, (unsigned)n, bbox, kin, vin);
        bool in_b = false;
        if (radix_sort_pairs_u64(kin, kout, vin, vout, n, 48, b_tmp.p, tmp_bytes, &in_b, st) != SP_OK) return fail("[BVH] sort failed");
        if (!in_b) { kout = kin; vout = vin; }
        bvh_gather_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(pts, vout, (unsigned)n, b->pts);
        if (n > 1) {
            bvh_hierarchy_kernel<<<div_up(n - 1, kBlock), kBlock, 0, st>>>(kout, (int)n, b->node, b_parent.as<int>(),
                                                                          b_lparent.as<int>(), b_tickets.as<unsigned>());
            bvh_box_kernel<<<div_up(n, kBlock), kBlock, 0, st>>>(b->pts, (int)n, b->node, b->obox, b_parent.as<int>(),
                                                                b_lparent.as<int>(), b_tickets.as<unsigned>(), b_cmin.as<int>());
        }
    }
    if (launch_status() != SP_OK || hipStreamSynchronize(st) != hipSuccess) return fail("[BVH] build failed");  // scratch idle from here
    *out = b;
    return SP_OK;
}

extern "C" int sp_internal_bvh_option(sp_bvh* b, int option, int value) {
    if (!b) return SP_ERR_INVALID_ARGUMENT;
    if (option == SP_INTERNAL_BVH_SELF_HEAP) b->self_heap = value;
    else if (option == SP_INTERNAL_BVH_SORT_QUERIES) b->sort_queries = value;
    else return SP_ERR_INVALID_ARGUMENT;
    return SP_OK;
}

extern "C" size_t sp_bvh_size(const sp_bvh* b) { return b ? b->n : 0; }

extern "C" int sp_bvh_search(const sp_bvh* bvh, const float* queries, size_t nq, size_t k, const float* transT,
                             int transT_on_device, int32_t* idx_out, float* d2_out, void* stream) {
    if (!bvh || !idx_out || !d2_out || (!queries && nq)) return SP_ERR_INVALID_ARGUMENT;
    if (nq == 0) return SP_OK;
    if (nq >= (1ull << 32)) return SP_ERR_INVALID_ARGUMENT;
    return sp::bvh_dispatch(bvh, reinterpret_cast<const float4*>(queries), (unsigned)nq, k, transT, transT_on_device, idx_out, d2_out,
                            sp::as_stream(stream));
}

extern "C" int sp_bvh_radius_search(const sp_bvh* bvh, const float* queries, size_t nq, size_t max_k, float radius,
                                    const float* transT, int transT_on_device, int32_t* idx_out, float* d2_out, void* stream) {
    if (!bvh || !idx_out || !d2_out || (!queries && nq)) return SP_ERR_INVALID_ARGUMENT;
    if (nq == 0) return SP_OK;
    if (nq >= (1ull << 32) || !(radius >= 0.0f)) return SP_ERR_INVALID_ARGUMENT;
    return sp::bvh_dispatch(bvh, reinterpret_cast<const float4*>(queries), (unsigned)nq, max_k, transT, transT_on_device, idx_out,
                            d2_out, sp::as_stream(stream), radius * radius);
}

extern "C" int sp_bvh_remove_by_flags(sp_bvh* bvh, const uint8_t* flags, const int32_t* new_indices, size_t n_flags, void* stream) {
    using namespace sp;
    if (!bvh || (n_flags && (!flags || !new_indices))) return SP_ERR_INVALID_ARGUMENT;
    if (bvh->n == 0 || n_flags == 0) return SP_OK;
    if (n_flags >= (1ull << 30)) return SP_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    bvh->streams.note(st);
    ScratchBuf b_f, b_pre, b_ws;
    const size_t wsb = exclusive_scan_u32_workspace_bytes(n_flags);
    hipError_t e = b_f.get(n_flags * 4, st);
    if (e == hipSuccess) e = b_pre.get(n_flags * 4, st);
    if (e == hipSuccess) e = b_ws.get(wsb ? wsb : 16, st);
    if (e != hipSuccess) { sp_set_error(hipGetErrorString(e)); return SP_ERR_HIP; }
    bvh_flags_to_u32_kernel<<<div_up(n_flags, kBlock), kBlock, 0, st>>>(flags, (unsigned)n_flags, b_f.as<uint32_t>());
    if (exclusive_scan_u32(b_f.as<uint32_t>(), b_pre.as<uint32_t>(), n_flags, nullptr, b_ws.p, wsb, st) != SP_OK) {
        sp_set_error("[BVH::remove_nodes_by_flags] scan failed");
        return SP_ERR_HIP;
    }
    if (bvh->n > 1)
        bvh_remove_minidx_kernel<<<div_up(bvh->n - 1, kBlock), kBlock, 0, st>>>(bvh->obox, (unsigned)(bvh->n - 1), b_pre.as<uint32_t>(),
                                                                              (unsigned)n_flags);
    bvh_remove_points_kernel<<<div_up(bvh->n, kBlock), kBlock, 0, st>>>(bvh->pts, (unsigned)bvh->n, flags, new_indices,
                                                                      (unsigned)n_flags);
    const int rc = launch_status();
    if (hipStreamSynchronize(st) != hipSuccess) return SP_ERR_HIP;  // the scratch is idle again
    return rc;
}

extern "C" int sp_bvh_self_knn(const sp_bvh* bvh, size_t k, int32_t* idx_out, float* d2_out, void* stream) {
    if (!bvh || !idx_out || !d2_out) return SP_ERR_INVALID_ARGUMENT;
    if (bvh->n == 0) return SP_OK;
    return sp::bvh_dispatch(bvh, nullptr, (unsigned)bvh->n, k, nullptr, 0, idx_out, d2_out, sp::as_stream(stream));
}

extern "C" int sp_bvh_export_points(const sp_bvh* bvh, float* points_out, void* stream) {
    if (!bvh || (!points_out && bvh->n)) return SP_ERR_INVALID_ARGUMENT;
    if (bvh->n == 0) return SP_OK;
    hipStream_t st = sp::as_stream(stream);
    bvh->streams.note(st);
    sp::bvh_export_kernel<<<sp::div_up(bvh->n, sp::kBlock), sp::kBlock, 0, st>>>(bvh->pts, (unsigned)bvh->n,
                                                                               reinterpret_cast<float4*>(points_out));
    return sp::launch_status();
}

// K2/K3/K4 — KD-tree kNN, radius search and lazy delete for gfx950
// (replaces algorithms/knn/kdtree.hpp:142-766).
//
// Build: on the host, the reference's rule verbatim in behaviour (kdtree.hpp:62-91, 292-413): split axis = largest
// range over a <=100-sample sweep, median by std::nth_element on the index array, leaves of <= leaf_threshold
// points. The topology and the order of points inside a leaf are therefore the reference's (the subtrees below the top
// levels are built by several host threads, build_parallel below); only the storage differs, and is chosen for the GPU:
//   * internal nodes: 32 B {x,y,z, idx | left, right, axis, -}, two 16-byte loads;
//   * leaf points: float4 {x,y,z, idx-bits} in fixed-stride blocks of `leaf_threshold` slots (pad slots idx=-1),
//     so a child code names its leaf block directly — no leaf-descriptor load on the dependent chain — and every
//     lane of a wave scans the same number of slots; half the bytes of the reference's 32-byte leaf nodes.
//   child code: >= 0 internal node, -1 none, <= -2 leaf block -(code+2).
// Search: one query per lane, the reference's traversal (near child first, far child pushed with its plane
// distance, far stack capped at 16 entries with silent drop, strict '<' first-visited tie rule). The far stack
// lives in LDS ([slot][lane], bank-conflict free); the near "stack" of the reference never holds more than one
// entry, so it is a register. A point is valid iff idx >= 0 (see sp_kdtree_remove_by_flags in the C ABI header).
#include <algorithm>
#include <future>
#include <numeric>
#include <thread>
#include <vector>

#include "sp_common.h"
#include "sp_math.h"

void sp_set_error(const char* msg);

struct sp_kdtree {
    size_t n_points = 0;
    unsigned n_internal = 0;
    unsigned n_leaves = 0;
    unsigned stride = 16;
    int root = -1;
    float4* d_internal = nullptr;  // 2 x float4 per node
    float4* d_leaf = nullptr;
    mutable sp::StreamSet streams;  // every stream the arrays were used on (sp_common.h: event-tagged pool, no sync in destroy)
};

namespace sp {
namespace {

constexpr int kFarDepth = 16;  // MAX_DEPTH / 2, kdtree.hpp:206,437
constexpr int kNone = -1;

struct HostInternal {
    float x, y, z;
    int32_t idx;
    int32_t left, right;
    uint32_t axis, pad;
};
static_assert(sizeof(HostInternal) == 32, "internal node is 32 bytes");

uint8_t find_axis_range(const float* pts, const std::vector<uint32_t>& gi, uint32_t start, uint32_t end) {
    const int64_t size = (int64_t)end - (int64_t)start + 1;
    if (size <= 1) return 0;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const size_t step = (size_t)std::max<int64_t>(size / 100, 1);
    for (size_t i = start; i <= end; i += step) {
        const float* p = pts + 4 * (size_t)gi[i];
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], p[a]);
            mx[a] = std::max(mx[a], p[a]);
        }
    }
    const float r0 = mx[0] - mn[0], r1 = mx[1] - mn[1], r2 = mx[2] - mn[2];
    if (r0 >= r1 && r0 >= r2) return 0;
    if (r1 >= r0 && r1 >= r2) return 1;
    return 2;
}

// One subtree over gi[start .. end] into `internal` / `leaves` (appended; child codes are indices into THESE vectors),
// iteratively, as the reference's build loop does (kdtree.hpp:292-413). Returns the code of the subtree's root.
int build_range(const float* pts, std::vector<uint32_t>& gi, uint32_t start, uint32_t end, size_t leaf_threshold,
                std::vector<HostInternal>& internal, std::vector<float4>& leaves) {
    const size_t stride = leaf_threshold;
    int root = kNone;
    // child slots point into `internal`, which may reallocate: keep (node, side) instead of raw pointers
    struct Ref { int node; int side; uint32_t start, end; };
    std::vector<Ref> stack;
    stack.push_back({-1, 0, start, end});
    auto set_child = [&](const Ref& r, int code) {
        if (r.node < 0) root = code;
        else if (r.side == 0) internal[r.node].left = code;
        else internal[r.node].right = code;
    };
    while (!stack.empty()) {
        const Ref task = stack.back();
        stack.pop_back();
        const uint32_t size = task.end - task.start + 1;
        if (size <= leaf_threshold) {
            const int leaf_id = (int)(leaves.size() / stride);
            for (uint32_t i = 0; i < (uint32_t)stride; ++i) {
                float4 s;
                if (i < size) {
                    const uint32_t p = gi[task.start + i];
                    s.x = pts[4 * (size_t)p]; s.y = pts[4 * (size_t)p + 1]; s.z = pts[4 * (size_t)p + 2];
                    s.w = __builtin_bit_cast(float, (int32_t)p);
                } else {
                    s.x = s.y = s.z = 0.0f;
                    s.w = __builtin_bit_cast(float, (int32_t)-1);
                }
                leaves.push_back(s);
            }
            set_child(task, -(leaf_id + 2));
            continue;
        }
        const uint8_t axis = find_axis_range(pts, gi, task.start, task.end);
        const uint32_t median = task.start + size / 2;
        std::nth_element(gi.begin() + task.start, gi.begin() + median, gi.begin() + task.end + 1,
                         [&](uint32_t a, uint32_t b) { return pts[4 * (size_t)a + axis] < pts[4 * (size_t)b + axis]; });
        const uint32_t p = gi[median];
        HostInternal node;
        node.x = pts[4 * (size_t)p]; node.y = pts[4 * (size_t)p + 1]; node.z = pts[4 * (size_t)p + 2];
        node.idx = (int32_t)p;
        node.left = kNone; node.right = kNone;
        node.axis = axis; node.pad = 0;
        const int id = (int)internal.size();
        internal.push_back(node);
        set_child(task, id);
        if (task.start < median) stack.push_back({id, 0, task.start, median - 1});
        if (median < task.end) stack.push_back({id, 1, median + 1, task.end});
    }
    return root;
}

// The reference builds on one host thread, and that build "dominates preprocessing at 1 M points" (139 ms here). The split of
// a range depends only on the range, so the two halves below a node are independent: the top levels hand their left half
// to another thread, every subtree is built into its own vectors and spliced behind its parent with its codes shifted.
// Same nodes, same children, same order of points in every leaf — the topology (and therefore every search result, the
// first-visited tie rule included) is the single-threaded build's; only the numbering of the nodes differs.
struct SubTree {
    std::vector<HostInternal> internal;
    std::vector<float4> leaves;
    int root = kNone;
};

int shifted(int code, int d_internal, int d_leaf) {
    if (code >= 0) return code + d_internal;
    if (code <= -2) return code - d_leaf;
    return code;
}
void splice(SubTree& into, SubTree&& from, size_t stride, int* root_code_out) {
    const int d_internal = (int)into.internal.size(), d_leaf = (int)(into.leaves.size() / stride);
    for (HostInternal& nd : from.internal) {
        nd.left = shifted(nd.left, d_internal, d_leaf);
        nd.right = shifted(nd.right, d_internal, d_leaf);
    }
    *root_code_out = shifted(from.root, d_internal, d_leaf);
    into.internal.insert(into.internal.end(), from.internal.begin(), from.internal.end());
    into.leaves.insert(into.leaves.end(), from.leaves.begin(), from.leaves.end());
}

SubTree build_parallel(const float* pts, std::vector<uint32_t>& gi, uint32_t start, uint32_t end, size_t leaf_threshold,
                       int spawn_levels) {
    SubTree t;
    const uint32_t size = end - start + 1;
    if (spawn_levels <= 0 || size <= leaf_threshold || size < 2048u) {  // (a thread costs ~30 us to start: worth it from ~0.1 ms of work)
        t.root = build_range(pts, gi, start, end, leaf_threshold, t.internal, t.leaves);
        return t;
    }
    const uint8_t axis = find_axis_range(pts, gi, start, end);
    const uint32_t median = start + size / 2;
    std::nth_element(gi.begin() + start, gi.begin() + median, gi.begin() + end + 1,
                     [&](uint32_t a, uint32_t b) { return pts[4 * (size_t)a + axis] < pts[4 * (size_t)b + axis]; });
    const uint32_t p = gi[median];
    HostInternal node;
    node.x = pts[4 * (size_t)p]; node.y = pts[4 * (size_t)p + 1]; node.z = pts[4 * (size_t)p + 2];
    node.idx = (int32_t)p;
    node.left = kNone; node.right = kNone;
    node.axis = axis; node.pad = 0;
    t.internal.push_back(node);
    t.root = 0;
    // size >= 2048 > 2: both halves exist
    std::future<SubTree> left = std::async(std::launch::async, [&, start, median, spawn_levels] {
        return build_parallel(pts, gi, start, median - 1, leaf_threshold, spawn_levels - 1);
    });
    SubTree right = build_parallel(pts, gi, median + 1, end, leaf_threshold, spawn_levels - 1);
    SubTree l = left.get();
    int code;
    splice(t, std::move(l), leaf_threshold, &code);
    t.internal[0].left = code;
    splice(t, std::move(right), leaf_threshold, &code);
    t.internal[0].right = code;
    return t;
}

void build_host(const float* pts, size_t n, size_t leaf_threshold, std::vector<HostInternal>& internal,
                std::vector<float4>& leaves, int& root) {
    internal.clear();
    leaves.clear();
    root = kNone;
    if (n == 0) return;
    std::vector<uint32_t> gi(n);
    std::iota(gi.begin(), gi.end(), 0u);
    unsigned threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 1;
    if (threads > 16) threads = 16;  // a fair share of a multi-GPU host
    int levels = 0;
    while ((1u << levels) < threads) ++levels;  // 2^levels subtrees in flight
    SubTree t = build_parallel(pts, gi, 0u, (uint32_t)(n - 1), leaf_threshold, levels);
    internal = std::move(t.internal);
    leaves = std::move(t.leaves);
    root = t.root;
}

// Sorted insertion into the first k slots, strict '<' (kdtree.hpp:119-137): the first visited wins ties.
template <int KCAP>
__device__ __forceinline__ void best_insert(float (&bd)[KCAP], int (&bi)[KCAP], int k, float d, int idx, float& kth) {
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

template <int KCAP, bool RADIUS, bool BATCH_LEAF>
__global__ __launch_bounds__(kBlock) void kdtree_search_kernel(const float4* __restrict__ internal,
                                                               const float4* __restrict__ leaf, int root,
                                                               unsigned stride, const float4* __restrict__ queries,
                                                               unsigned nq, int k, float radius_sq, Mat4Arg T_val,
                                                               const float* __restrict__ T_dev,
                                                               int32_t* __restrict__ idx_out,
                                                               float* __restrict__ d2_out) {
    __shared__ float far_dist[kFarDepth][kBlock];
    __shared__ int far_node[kFarDepth][kBlock];
    const unsigned qi = blockIdx.x * kBlock + threadIdx.x;
    if (qi >= nq) return;  // no barrier below: safe
    const unsigned lane = threadIdx.x;

    const Rigid T = load_rigid_colmajor(T_dev ? T_dev : T_val.m);
    const float4 q4 = queries[qi];
    float qx, qy, qz;
    transform_point(T, q4.x, q4.y, q4.z, qx, qy, qz);

    float bd[KCAP];
    int bi[KCAP];
#pragma unroll
    for (int i = 0; i < KCAP; ++i) { bd[i] = FLT_MAX; bi[i] = -1; }
    float kth = FLT_MAX;  // bestK[k-1].dist_sq

    int far_ptr = 0;
    int cur = root;
    float cur_dist = 0.0f;
    bool have_near = true;
    for (;;) {
        if (!have_near) {
            if (far_ptr == 0) break;
            --far_ptr;
            cur = far_node[far_ptr][lane];
            cur_dist = far_dist[far_ptr][lane];
        }
        have_near = false;
        const float limit = RADIUS ? fminf(kth, radius_sq) : kth;
        if (cur_dist > limit) continue;
        if (cur == kNone) continue;
        if (cur <= -2) {  // leaf block
            const float4* blk = leaf + (size_t)(-(cur + 2)) * stride;
            auto visit = [&](const float4 p) {
                const int pidx = __builtin_bit_cast(int, p.w);
                const bool valid = pidx >= 0;
                const float d = valid ? dist2(qx, qy, qz, p.x, p.y, p.z) : FLT_MAX;
                const bool take = RADIUS ? (valid && d <= radius_sq && d < kth) : (d < kth);
                if (take) {
                    best_insert<KCAP>(bd, bi, k, d, pidx, kth);
                }
            };
            if (KCAP == 1 && stride == 16) {
                // the default leaf size: fetch the whole 256-byte block with 16 independent loads, then visit in order
                float4 slot[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) slot[s] = blk[s];
#pragma unroll
                for (int s = 0; s < 16; ++s) visit(slot[s]);
            } else if (KCAP == 1) {
#pragma unroll 4
                for (unsigned s = 0; s < stride; ++s) visit(blk[s]);
            } else if (BATCH_LEAF && KCAP <= 20 && stride == 16) {
                // sorted lists, few queries (the launch is a latency chain, not a throughput problem: 69 k queries k = 10 0.77 ->
                // 0.59 ms; at 1 M queries the eight copies of the insertion chain cost 6 %): fetch the block as independent loads,
                // two batches of eight — one round trip each instead of sixteen behind sixteen data-dependent branches
#pragma unroll 1
                for (int half = 0; half < 2; ++half) {
                    float4 slot[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) slot[s] = blk[half * 8 + s];
#pragma unroll
                    for (int s = 0; s < 8; ++s) visit(slot[s]);
                }
            } else {  // long lists / other leaf sizes: the insertion chain is long, keep one copy of it
#pragma unroll 1
                for (unsigned s = 0; s < stride; ++s) visit(blk[s]);
            }
            continue;
        }
        const float4 n0 = internal[2 * (size_t)cur];
        const float4 n1 = internal[2 * (size_t)cur + 1];
        const int nidx = __builtin_bit_cast(int, n0.w);
        const int left = __builtin_bit_cast(int, n1.x), right = __builtin_bit_cast(int, n1.y);
        const int axis = __builtin_bit_cast(int, n1.z);
        const float dx = qx - n0.x, dy = qy - n0.y, dz = qz - n0.z;
        const bool valid = nidx >= 0;
        const float d = valid ? chain3(dx, dx, dy, dy, dz, dz) : FLT_MAX;
        const bool take = RADIUS ? (valid && d <= radius_sq && d < kth) : (d < kth);
        if (take) {
            best_insert<KCAP>(bd, bi, k, d, nidx, kth);
        }
        const float axis_dist = axis == 0 ? dx : (axis == 1 ? dy : dz);
        const int nearer = (axis_dist <= 0) ? left : right;
        const int further = (axis_dist <= 0) ? right : left;
        const float split = axis_dist * axis_dist;
        // kNN: strict '<' against the k-th best after this node's insertion (kdtree.hpp:535);
        // radius: '<=' against the limit read when the node was popped (kdtree.hpp:645,692).
        const bool search_further = RADIUS ? (split <= limit) : (split < kth);
        if (search_further && further != kNone && far_ptr < kFarDepth) {
            far_node[far_ptr][lane] = further;
            far_dist[far_ptr][lane] = split;
            ++far_ptr;
        }
        if (nearer != kNone) {
            cur = nearer;
            cur_dist = 0.0f;
            have_near = true;
        }
    }
    const size_t o = (size_t)qi * (size_t)k;
#pragma unroll
    for (int i = 0; i < KCAP; ++i)
        if (i < k) { d2_out[o + i] = bd[i]; idx_out[o + i] = bi[i]; }
}

__global__ void fill_empty_kernel(int32_t* idx, float* d2, size_t n) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) { idx[i] = -1; d2[i] = FLT_MAX; }
}

// K4: rewrite idx of every stored point (kdtree.hpp:743-755). valid <=> idx >= 0.
__global__ void remove_internal_kernel(float4* internal, unsigned n_internal, const uint8_t* flags,
                                       const int32_t* new_idx, unsigned n_flags) {
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_internal) return;
    float4 n0 = internal[2 * (size_t)i];
    const int p = __builtin_bit_cast(int, n0.w);
    if (p < 0 || (unsigned)p >= n_flags) return;
    const int ni = flags[p] ? new_idx[p] : -1;
    n0.w = __builtin_bit_cast(float, ni);
    internal[2 * (size_t)i] = n0;
}
__global__ void remove_leaf_kernel(float4* leaf, size_t n_slots, const uint8_t* flags, const int32_t* new_idx,
                                   unsigned n_flags) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_slots) return;
    float4 s = leaf[i];
    const int p = __builtin_bit_cast(int, s.w);
    if (p < 0 || (unsigned)p >= n_flags) return;
    const int ni = flags[p] ? new_idx[p] : -1;
    s.w = __builtin_bit_cast(float, ni);
    leaf[i] = s;
}

template <int KCAP, bool RADIUS>
int launch_search(const sp_kdtree* t, const float* q, size_t nq, size_t k, float radius_sq, const float* T,
                  int T_dev, int32_t* idx, float* d2, hipStream_t st) {
    Mat4Arg tv;
    for (int i = 0; i < 16; ++i) tv.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    if (T && !T_dev)
        for (int i = 0; i < 16; ++i) tv.m[i] = T[i];
    if (KCAP > 1 && KCAP <= 20 && nq < 200000)
        kdtree_search_kernel<KCAP, RADIUS, true><<<div_up(nq, kBlock), kBlock, 0, st>>>(
            t->d_internal, t->d_leaf, t->root, t->stride, reinterpret_cast<const float4*>(q), (unsigned)nq, (int)k,
            radius_sq, tv, T_dev ? T : nullptr, idx, d2);
    else
        kdtree_search_kernel<KCAP, RADIUS, false><<<div_up(nq, kBlock), kBlock, 0, st>>>(
            t->d_internal, t->d_leaf, t->root, t->stride, reinterpret_cast<const float4*>(q), (unsigned)nq, (int)k,
            radius_sq, tv, T_dev ? T : nullptr, idx, d2);
    return launch_status();
}

template <bool RADIUS>
int dispatch_search(const sp_kdtree* t, const float* q, size_t nq, size_t k, float radius_sq, const float* T, int T_dev,
                    int32_t* idx, float* d2, hipStream_t st) {
    if (k == 1) return launch_search<1, RADIUS>(t, q, nq, k, radius_sq, T, T_dev, idx, d2, st);
    if (k <= 10) return launch_search<10, RADIUS>(t, q, nq, k, radius_sq, T, T_dev, idx, d2, st);
    if (k <= 20) return launch_search<20, RADIUS>(t, q, nq, k, radius_sq, T, T_dev, idx, d2, st);
    if (k <= 50) return launch_search<50, RADIUS>(t, q, nq, k, radius_sq, T, T_dev, idx, d2, st);
    return launch_search<100, RADIUS>(t, q, nq, k, radius_sq, T, T_dev, idx, d2, st);
}

}  // namespace
}  // namespace sp

extern "C" int sp_kdtree_create(const float* points_host, size_t n, size_t leaf_threshold, void* stream,
                                sp_kdtree** out) {
    using namespace sp;
    if (!out) return SP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (leaf_threshold == 0 || leaf_threshold > 4096) {
        sp_set_error("[KDTree::build] leaf_threshold must be in [1, 4096]");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (n >= (1ull << 31)) {
        sp_set_error("[KDTree::build] more than 2^31 points: indices are int32");
        return SP_ERR_INVALID_ARGUMENT;
    }
    std::vector<HostInternal> internal;
    std::vector<float4> leaves;
    int root = kNone;
    build_host(points_host, n, leaf_threshold, internal, leaves, root);
    sp_kdtree* t = new sp_kdtree();
    t->n_points = n;
    t->n_internal = (unsigned)internal.size();
    t->stride = (unsigned)leaf_threshold;
    t->n_leaves = (unsigned)(leaves.size() / leaf_threshold);
    t->root = root;
    hipStream_t st = as_stream(stream);
    t->streams.note(st);
    hipError_t e = hipSuccess;
    if (!internal.empty()) {
        e = pooled_alloc(&t->d_internal, internal.size() * 32, st);  // (hipMalloc + hipFree are ~0.1-0.2 ms apiece on this runtime)
        if (e == hipSuccess) e = hipMemcpyAsync(t->d_internal, internal.data(), internal.size() * 32, hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess && !leaves.empty()) {
        e = pooled_alloc(&t->d_leaf, leaves.size() * 16, st);
        if (e == hipSuccess) e = hipMemcpyAsync(t->d_leaf, leaves.data(), leaves.size() * 16, hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // the host vectors die at return
    if (e != hipSuccess) {
        sp_set_error(hipGetErrorString(e));
        sp_kdtree_destroy(t);
        return SP_ERR_HIP;
    }
    *out = t;
    return SP_OK;
}

extern "C" void sp_kdtree_destroy(sp_kdtree* t) {
    if (!t) return;
    sp::pooled_free_after(t->d_internal, t->streams);
    sp::pooled_free_after(t->d_leaf, t->streams);
    delete t;
}

extern "C" size_t sp_kdtree_size(const sp_kdtree* t) { return t ? t->n_points : 0; }

extern "C" int sp_kdtree_search(const sp_kdtree* tree, const float* queries, size_t nq, size_t k, const float* transT,
                                int transT_on_device, int32_t* idx_out, float* d2_out, void* stream) {
    using namespace sp;
    if (!tree || k == 0) {
        sp_set_error("[KDTree::knn_search_async] null tree or k == 0");
        return SP_ERR_INVALID_ARGUMENT;
    }
    if (k > 100) {
        sp_set_error("[KDTree::knn_search_async] `k` is too large. not support.");
        return SP_ERR_RUNTIME;
    }
    if (nq == 0) return SP_OK;
    hipStream_t st = as_stream(stream);
    tree->streams.note(st);
    if (tree->root == kNone) {
        fill_empty_kernel<<<div_up(nq * k, kBlock), kBlock, 0, st>>>(idx_out, d2_out, nq * k);
        return launch_status();
    }
    return dispatch_search<false>(tree, queries, nq, k, -1.0f, transT, transT_on_device, idx_out, d2_out, st);
}

extern "C" int sp_kdtree_radius_search(const sp_kdtree* tree, const float* queries, size_t nq, size_t max_k,
                                       float radius, const float* transT, int transT_on_device, int32_t* idx_out,
                                       float* d2_out, void* stream) {
    using namespace sp;
    if (!tree) return SP_ERR_INVALID_ARGUMENT;
    if (max_k > 100) {
        sp_set_error("[KDTree::radius_search_async] `max_k` is too large. not support.");
        return SP_ERR_RUNTIME;
    }
    if (nq == 0 || max_k == 0) return SP_OK;
    hipStream_t st = as_stream(stream);
    tree->streams.note(st);
    if (tree->root == kNone) {
        fill_empty_kernel<<<div_up(nq * max_k, kBlock), kBlock, 0, st>>>(idx_out, d2_out, nq * max_k);
        return launch_status();
    }
    return dispatch_search<true>(tree, queries, nq, max_k, radius * radius, transT, transT_on_device, idx_out, d2_out,
                                 st);
}

extern "C" int sp_kdtree_remove_by_flags(sp_kdtree* tree, const uint8_t* flags, const int32_t* new_indices,
                                         size_t n_flags, void* stream) {
    using namespace sp;
    if (!tree) return SP_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    tree->streams.note(st);
    if (tree->n_internal)
        remove_internal_kernel<<<div_up(tree->n_internal, kBlock), kBlock, 0, st>>>(tree->d_internal, tree->n_internal,
                                                                                    flags, new_indices, (unsigned)n_flags);
    const size_t slots = (size_t)tree->n_leaves * tree->stride;
    if (slots)
        remove_leaf_kernel<<<div_up(slots, kBlock), kBlock, 0, st>>>(tree->d_leaf, slots, flags, new_indices,
                                                                    (unsigned)n_flags);
    return launch_status();
}

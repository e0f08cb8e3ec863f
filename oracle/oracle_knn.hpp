// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
// CPU restatement of the reference's KNN path:
//   brute force      /root/reference/cpp/include/sycl_points/algorithms/knn/bruteforce.hpp:24-96
//   KD-tree build    .../knn/kdtree.hpp:62-91, 292-413
//   KD-tree kNN      .../knn/kdtree.hpp:119-137, 424-562
//   radius search    .../knn/kdtree.hpp:574-719
//   lazy delete      .../knn/kdtree.hpp:721-765
//   query transform  .../common/transform.hpp:31-37
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

#include "oracle_math.hpp"

namespace oracle {

// knn/kdtree.hpp:34-45 — 32-byte node, same field order.
struct FlatKDNode {
    float pt[4];
    int32_t idx;
    int32_t left;
    int32_t right;
    uint8_t axis;
    uint8_t is_leaf;
    uint8_t valid;
    uint8_t pad;
};
static_assert(sizeof(FlatKDNode) == 32, "FlatKDNode must be 32 bytes");

struct NodeEntry {  // knn/kdtree.hpp:56-59
    int32_t nodeIdx;
    float dist_sq;
};

// common/transform.hpp:31-37: T (given column-major, Eigen layout) times homogeneous point, each row an
// fma chain over columns 0..3 starting from 0 (eigen_utils.hpp:113-127).
inline void transform_point(const float* p, float* out, const float* T_colmajor) {
    for (int i = 0; i < 4; ++i) {
        float sum = 0.0f;
        for (int j = 0; j < 4; ++j) sum = std::fmaf(T_colmajor[j * 4 + i], p[j], sum);
        out[i] = sum;
    }
}

// Squared distance as the KD-tree computes it: subtract<4,1> then dot<4> (knn/kdtree.hpp:509-511).
inline float dist4(const float* q, const float* t) {
    float r = 0.0f;
    for (int i = 0; i < 4; ++i) {
        const float d = q[i] - t[i];
        r = std::fmaf(d, d, r);
    }
    return r;
}

// knn/bruteforce.hpp:63-68.  sycl::dot(float4,float4) is implementation-defined; it is pinned here to the
// same chain the KD-tree uses (fma over x,y,z then the w term, which is 0*0): brute force and KD-tree then
// produce bit-identical distances, so brute force is an exact oracle for every other KNN structure.
inline float dist_bruteforce(const float* q, const float* t) {
    const float dx = q[0] - t[0], dy = q[1] - t[1], dz = q[2] - t[2], dw = 0.0f;
    float r = 0.0f;
    r = std::fmaf(dx, dx, r);
    r = std::fmaf(dy, dy, r);
    r = std::fmaf(dz, dz, r);
    r = std::fmaf(dw, dw, r);
    return r;
}

// knn/bruteforce.hpp:24-96 — exact kNN, strict '<' so the lowest index wins ties; k <= 20.
inline void knn_bruteforce(const float* queries, size_t nq, const float* targets, size_t nt, size_t k, int32_t* idx_out,
                           float* d2_out) {
    constexpr size_t MAX_K = 20;
    if (k == 0 || k > MAX_K) return;  // the reference has no check and overruns kD/kI; callers (pyoracle) raise instead
#pragma omp parallel for schedule(static)
    for (long long qi = 0; qi < (long long)nq; ++qi) {
        const float* query = queries + 4 * qi;
        float kD[MAX_K];
        int32_t kI[MAX_K];
        for (size_t i = 0; i < k; ++i) {
            kD[i] = std::numeric_limits<float>::max();
            kI[i] = -1;
        }
        for (size_t j = 0; j < nt; ++j) {
            const float dist = dist_bruteforce(query, targets + 4 * j);
            if (dist < kD[k - 1]) {
                int32_t pos = (int32_t)k - 1;
                while (pos > 0 && dist < kD[pos - 1]) {
                    kD[pos] = kD[pos - 1];
                    kI[pos] = kI[pos - 1];
                    --pos;
                }
                kD[pos] = dist;
                kI[pos] = (int32_t)j;
            }
        }
        for (size_t i = 0; i < k; ++i) {
            d2_out[qi * k + i] = kD[i];
            idx_out[qi * k + i] = kI[i];
        }
    }
}

// knn/kdtree.hpp:62-91
inline uint8_t find_axis_range(const float* points, const std::vector<uint32_t>& indices, uint32_t start, uint32_t end) {
    const int64_t size = (int64_t)end - (int64_t)start + 1;
    if (size <= 1) return 0;
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float mx[3] = {std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(),
                   std::numeric_limits<float>::lowest()};
    const size_t step = (size_t)std::max(size / 100, (int64_t)1);
    for (size_t i = start; i <= end; i += step) {
        const uint32_t idx = indices[i];
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], points[4 * (size_t)idx + a]);
            mx[a] = std::max(mx[a], points[4 * (size_t)idx + a]);
        }
    }
    const float r0 = mx[0] - mn[0], r1 = mx[1] - mn[1], r2 = mx[2] - mn[2];
    if (r0 >= r1 && r0 >= r2) return 0;
    if (r1 >= r0 && r1 >= r2) return 1;
    return 2;
}

// knn/kdtree.hpp:292-413 — explicit-stack median-split build; leaf blocks of <= leaf_threshold points.
inline std::vector<FlatKDNode> kdtree_build(const float* points, size_t n, size_t leaf_threshold = 16) {
    std::vector<FlatKDNode> tree;
    if (n == 0) return tree;
    FlatKDNode blank;
    std::memset(&blank, 0, sizeof(blank));
    blank.left = -1;
    blank.right = -1;
    blank.valid = 1;
    tree.assign(n * 2, blank);

    std::vector<uint32_t> gidx(n);
    std::iota(gidx.begin(), gidx.end(), 0u);
    struct Task { uint32_t node, start, end; };
    std::vector<Task> stack;
    stack.reserve(n);
    stack.push_back({0u, 0u, (uint32_t)(n - 1)});
    uint32_t next = 1;
    while (!stack.empty()) {
        const Task task = stack.back();
        stack.pop_back();
        const uint32_t size = task.end - task.start + 1;
        if (task.start > task.end || size == 0) continue;
        FlatKDNode& node = tree[task.node];
        if (size <= leaf_threshold) {
            const uint32_t leafStart = next;
            next += size;
            for (uint32_t i = 0; i < size; ++i) {
                const uint32_t pidx = gidx[task.start + i];
                FlatKDNode& m = tree[leafStart + i];
                std::memcpy(m.pt, points + 4 * (size_t)pidx, 16);
                m.idx = (int32_t)pidx;
                m.is_leaf = 1;
                m.axis = 0;
                m.left = -1;
                m.right = -1;
                m.valid = 1;
            }
            node.is_leaf = 1;
            node.valid = 1;
            node.idx = -1;
            node.axis = 0;
            node.left = (int32_t)leafStart;
            node.right = (int32_t)size;
            continue;
        }
        node.is_leaf = 0;
        node.valid = 1;
        const uint8_t axis = find_axis_range(points, gidx, task.start, task.end);
        const uint32_t median = task.start + size / 2;
        std::nth_element(gidx.begin() + task.start, gidx.begin() + median, gidx.begin() + task.end + 1,
                         [&](uint32_t a, uint32_t b) { return points[4 * (size_t)a + axis] < points[4 * (size_t)b + axis]; });
        const uint32_t pidx = gidx[median];
        std::memcpy(node.pt, points + 4 * (size_t)pidx, 16);
        node.idx = (int32_t)pidx;
        node.axis = axis;
        if (task.start < median) {
            const uint32_t l = next++;
            node.left = (int32_t)l;
            stack.push_back({l, task.start, median - 1});
        }
        if (median < task.end) {
            const uint32_t r = next++;
            node.right = (int32_t)r;
            stack.push_back({r, median + 1, task.end});
        }
    }
    tree.resize(next);
    return tree;
}

// knn/kdtree.hpp:119-137
inline void insert_to_bestK(NodeEntry* bestK, float dist_sq, int32_t nodeIdx, size_t k, size_t found_num, size_t MAX_K) {
    if (MAX_K == 1) {
        bestK[0].nodeIdx = dist_sq < bestK[0].dist_sq ? nodeIdx : bestK[0].nodeIdx;
        bestK[0].dist_sq = dist_sq < bestK[0].dist_sq ? dist_sq : bestK[0].dist_sq;
        return;
    }
    if (dist_sq >= bestK[k - 1].dist_sq) return;
    size_t pos = std::min(found_num - 1, k - 1);
    while (pos > 0 && dist_sq < bestK[pos - 1].dist_sq) {
        bestK[pos] = bestK[pos - 1];
        --pos;
    }
    bestK[pos].nodeIdx = nodeIdx;
    bestK[pos].dist_sq = dist_sq;
}

inline size_t kdtree_max_k_class(size_t k) {  // knn/kdtree.hpp:207-223
    if (k == 1) return 1;
    for (size_t c : {10, 20, 30, 40, 50, 100})
        if (k <= c) return c;
    return 0;
}

// knn/kdtree.hpp:424-562 (radius_sq < 0: plain kNN) and :574-719 (radius_sq >= 0: radius search).
// Two stacks of MAX_DEPTH/2 = 16 entries; pushes beyond that are silently dropped, as in the reference.
inline void kdtree_search(const FlatKDNode* tree, size_t treeSize, const float* queries, size_t nq, size_t k,
                          const float* T_colmajor, int32_t* idx_out, float* d2_out, float radius_sq = -1.0f) {
    constexpr size_t MAX_DEPTH_HALF = 16;
    const size_t MAX_K = kdtree_max_k_class(k);
    const bool radius_mode = radius_sq >= 0.0f;
#pragma omp parallel for schedule(dynamic, 256)
    for (long long qi = 0; qi < (long long)nq; ++qi) {
        float query[4];
        transform_point(queries + 4 * qi, query, T_colmajor);
        NodeEntry bestK[100];
        for (size_t i = 0; i < MAX_K; ++i) bestK[i] = NodeEntry{-1, std::numeric_limits<float>::max()};
        NodeEntry nearStack[MAX_DEPTH_HALF], farStack[MAX_DEPTH_HALF];
        size_t nearPtr = 0, farPtr = 0;
        nearStack[nearPtr++] = {0, 0.0f};
        size_t found_num = 0;
        while (nearPtr > 0 || farPtr > 0) {
            const NodeEntry cur = nearPtr > 0 ? nearStack[--nearPtr] : farStack[--farPtr];
            const int32_t nodeIdx = cur.nodeIdx;
            const float limit = radius_mode ? std::fmin(bestK[k - 1].dist_sq, radius_sq) : bestK[k - 1].dist_sq;
            if (cur.dist_sq > limit) continue;
            if (nodeIdx == -1 || (size_t)nodeIdx >= treeSize) continue;
            const FlatKDNode node = tree[nodeIdx];
            if (node.is_leaf != 0) {
                for (int32_t li = 0; li < node.right; ++li) {
                    const FlatKDNode& m = tree[node.left + li];
                    const bool is_valid = (m.valid == 1);
                    const float d = is_valid ? dist4(query, m.pt) : std::numeric_limits<float>::max();
                    if (radius_mode) {
                        if (is_valid && d <= radius_sq) {
                            ++found_num;
                            insert_to_bestK(bestK, d, m.idx, k, found_num, MAX_K);
                        }
                    } else {
                        found_num = is_valid ? found_num + 1 : found_num;
                        insert_to_bestK(bestK, d, m.idx, k, found_num, MAX_K);
                    }
                }
                continue;
            }
            const bool is_valid = (node.valid == 1);
            float diff[4];
            for (int i = 0; i < 4; ++i) diff[i] = query[i] - node.pt[i];
            float dsq = 0.0f;
            for (int i = 0; i < 4; ++i) dsq = std::fmaf(diff[i], diff[i], dsq);
            const float d = is_valid ? dsq : std::numeric_limits<float>::max();
            if (radius_mode) {
                if (is_valid && d <= radius_sq) {
                    ++found_num;
                    insert_to_bestK(bestK, d, node.idx, k, found_num, MAX_K);
                }
            } else {
                found_num = is_valid ? found_num + 1 : found_num;
                insert_to_bestK(bestK, d, node.idx, k, found_num, MAX_K);
            }
            const float axisDistance = diff[node.axis];
            const int32_t nearer = (axisDistance <= 0) ? node.left : node.right;
            const int32_t further = (axisDistance <= 0) ? node.right : node.left;
            const float splitDistSq = axisDistance * axisDistance;
            // kNN: strict '<' against the (updated) k-th best; radius: '<=' against the limit read at pop time.
            const bool searchFurther = radius_mode ? (splitDistSq <= limit) : (splitDistSq < bestK[k - 1].dist_sq);
            if (searchFurther && further != -1 && farPtr < MAX_DEPTH_HALF) farStack[farPtr++] = {further, splitDistSq};
            if (nearer != -1 && nearPtr < MAX_DEPTH_HALF) nearStack[nearPtr++] = {nearer, 0.0f};
        }
        for (size_t i = 0; i < k; ++i) {
            d2_out[qi * k + i] = bestK[i].dist_sq;
            idx_out[qi * k + i] = bestK[i].nodeIdx;
        }
    }
}

// knn/kdtree.hpp:721-765
inline void kdtree_remove_by_flags(FlatKDNode* tree, size_t n_nodes, const uint8_t* flags, const int32_t* new_indices,
                                   size_t flags_size) {
    for (size_t i = 0; i < n_nodes; ++i) {
        const int32_t p = tree[i].idx;
        if (p < 0 || (size_t)p >= flags_size) continue;
        tree[i].valid = flags[p];
        tree[i].idx = new_indices[p];
    }
}

}  // namespace oracle

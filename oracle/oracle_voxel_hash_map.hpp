// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of algorithms/mapping/voxel_hash_map.hpp (VoxelHashMap): a double-hashed open-addressing table keyed by
// compute_voxel_bit, per-voxel sums of the map-frame points / log-Euclidean covariances / colours / intensities, staleness
// removal, rehash on load, bounding-box export. Sequential: points are inserted in input order, so sums are in input
// order (the reference accumulates with relaxed float atomics after a work-group-local pre-reduction: its order is
// unspecified, results agree to rounding) and the export runs in slot order.
// Pinned by the known answers of cpp/tests/test_voxel_hash_map.cpp (tests/test_oracle_pins.py).
#pragma once
#include <array>
#include <cstdint>
#include <vector>

#include "oracle_features.hpp"
#include "oracle_knn.hpp"
#include "oracle_math.hpp"

namespace oracle {

// utils/eigen_utils.hpp:646-659
inline Mat3 log_spd_3x3(const Mat3& A, float min_eigenvalue = 1e-6f) {
    Vec3 ev;
    Mat3 V;
    symmetric_eigen_decomposition_3x3(A, ev, V);
    Vec3 lg;
    for (int i = 0; i < 3; ++i) lg[i] = std::log(sycl_max(ev[i], min_eigenvalue));  // sycl::fmax on non-NaN values
    return ensure_symmetric<3>(matmul<3, 3, 3>(matmul<3, 3, 3>(V, as_diagonal<3>(lg)), transpose<3, 3>(V)));
}
// utils/eigen_utils.hpp:664-677
inline Mat3 exp_spd_3x3(const Mat3& A) {
    Vec3 ev;
    Mat3 V;
    symmetric_eigen_decomposition_3x3(A, ev, V);
    Vec3 ex;
    for (int i = 0; i < 3; ++i) ex[i] = std::exp(ev[i]);
    return ensure_symmetric<3>(matmul<3, 3, 3>(matmul<3, 3, 3>(V, as_diagonal<3>(ex)), transpose<3, 3>(V)));
}

class VoxelHashMap {
public:
    static constexpr std::array<size_t, 11> kCapacityCandidates = {30029,  60013,   120011,  240007,   480013,  960017,
                                                                   1920001, 3840007, 7680017, 15360013, 30720007};
    static constexpr size_t kMaxProbe = 100;  // voxel_hash_map.hpp:505

    struct Core { float sx = 0, sy = 0, sz = 0; uint32_t count = 0; };
    struct Cov { float xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0; };
    struct Color { float r = 0, g = 0, b = 0, a = 0; };

    float voxel_size = 0, voxel_size_inv = 0;
    uint32_t max_staleness = 100, remove_old_data_cycle = 10, min_num_point = 1, staleness_counter = 0;
    float rehash_threshold = 0.7f;
    size_t capacity = kCapacityCandidates[0], voxel_num = 0;
    bool has_cov = false, has_rgb = false, has_intensity = false;
    std::vector<uint64_t> key;
    std::vector<Core> core;
    std::vector<Cov> cov;
    std::vector<Color> color;
    std::vector<float> intensity;
    std::vector<uint32_t> last_update;

    explicit VoxelHashMap(float vs) {  // voxel_hash_map.hpp:28-48 (the caller rejects vs <= 0)
        voxel_size = vs;
        voxel_size_inv = 1.0f / vs;
        clear();
    }
    void clear() {  // :83-113
        capacity = kCapacityCandidates[0];
        voxel_num = 0;
        staleness_counter = 0;
        has_cov = has_rgb = has_intensity = false;
        allocate(capacity);
    }
    // :587-592
    static uint64_t hash2(uint64_t h, size_t cap) { return (cap - 2) - (h % (cap - 2)); }
    static size_t slot_id(uint64_t h, size_t probe, size_t cap) { return (size_t)((h + probe * hash2(h, cap)) % cap); }

    // :117-141
    void add_point_cloud(const float* pts, const float* covs, const float* rgb, const float* inten, size_t n, const float* pose16) {
        if (rehash_threshold < (float)voxel_num / (float)capacity) {
            const size_t next = next_capacity();
            if (next > capacity) rehash(next);
        }
        if (n > 0) add_impl(pts, covs, rgb, inten, n, pose16);
        if (remove_old_data_cycle > 0 && (staleness_counter % remove_old_data_cycle) == 0) remove_old_data();
        ++staleness_counter;
    }

    // :146-190, 933-1068 (slot order; the reference's order is that of an atomic counter). Returns the voxel count.
    size_t downsampling(const float* center3, float distance, float* pts_out, float* cov_out, float* rgb_out,
                        float* inten_out, uint64_t* keys_out) const {
        if (voxel_num == 0) return 0;
        const float mnx = center3[0] - distance, mny = center3[1] - distance, mnz = center3[2] - distance;
        const float mxx = center3[0] + distance, mxy = center3[1] + distance, mxz = center3[2] + distance;
        size_t out = 0;
        for (size_t i = 0; i < capacity; ++i) {
            const Core& c = core[i];
            if (key[i] == VOXEL_INVALID || c.count < min_num_point || c.count == 0U) continue;
            const float inv = 1.0f / (float)c.count;
            const float cx = c.sx * inv, cy = c.sy * inv, cz = c.sz * inv;
            if (!((cx >= mnx && cx <= mxx) && (cy >= mny && cy <= mxy) && (cz >= mnz && cz <= mxz))) continue;
            pts_out[4 * out + 0] = cx; pts_out[4 * out + 1] = cy; pts_out[4 * out + 2] = cz; pts_out[4 * out + 3] = 1.0f;
            if (cov_out && has_cov) {  // :345-361: the mean in log space, mapped back
                Mat3 m;
                m(0, 0) = cov[i].xx * inv; m(0, 1) = m(1, 0) = cov[i].xy * inv; m(0, 2) = m(2, 0) = cov[i].xz * inv;
                m(1, 1) = cov[i].yy * inv; m(1, 2) = m(2, 1) = cov[i].yz * inv; m(2, 2) = cov[i].zz * inv;
                const Mat3 e = exp_spd_3x3(m);
                float* o = cov_out + 16 * out;
                for (int k = 0; k < 16; ++k) o[k] = 0.0f;
                for (int col = 0; col < 3; ++col)
                    for (int row = 0; row < 3; ++row) o[col * 4 + row] = e(row, col);
            }
            if (rgb_out && has_rgb) {
                rgb_out[4 * out + 0] = color[i].r * inv; rgb_out[4 * out + 1] = color[i].g * inv;
                rgb_out[4 * out + 2] = color[i].b * inv; rgb_out[4 * out + 3] = color[i].a * inv;
            }
            if (inten_out && has_intensity) inten_out[out] = intensity[i] * inv;
            if (keys_out) keys_out[out] = key[i];
            ++out;
        }
        return out;
    }

    // :196-246
    float overlap_ratio(const float* pts, size_t n, const float* pose16) const {
        if (n == 0 || voxel_num == 0) return 0.0f;
        uint32_t hits = 0;
        for (size_t i = 0; i < n; ++i) {
            float w[4];
            transform_point(pts + 4 * i, w, pose16);
            const uint64_t h = compute_voxel_bit(w, voxel_size_inv);
            if (h == VOXEL_INVALID) continue;
            for (size_t p = 0; p < kMaxProbe; ++p) {
                const size_t s = slot_id(h, p, capacity);
                if (key[s] == h) { if (core[s].count >= min_num_point) ++hits; break; }
                if (key[s] == VOXEL_INVALID) break;
            }
        }
        return (float)hits / (float)n;
    }

    // :788-843
    void remove_old_data() {
        if (staleness_counter <= max_staleness) return;
        const uint32_t remove_staleness = staleness_counter - max_staleness;
        size_t kept = 0;
        for (size_t i = 0; i < capacity; ++i) {
            if (key[i] == VOXEL_INVALID) continue;
            if (last_update[i] >= remove_staleness) { ++kept; continue; }
            key[i] = VOXEL_INVALID; core[i] = Core{}; cov[i] = Cov{}; color[i] = Color{}; intensity[i] = 0.0f; last_update[i] = 0;
        }
        set_voxel_num(kept);
    }

private:
    void allocate(size_t cap) {
        key.assign(cap, VOXEL_INVALID);
        core.assign(cap, Core{});
        cov.assign(cap, Cov{});
        color.assign(cap, Color{});
        intensity.assign(cap, 0.0f);
        last_update.assign(cap, 0U);
        capacity = cap;
    }
    size_t next_capacity() const {
        for (const size_t c : kCapacityCandidates)
            if (c > capacity) return c;
        return capacity;
    }
    void set_voxel_num(size_t n) {  // :519-526
        voxel_num = n;
        if (n == 0) has_cov = has_rgb = has_intensity = false;
    }
    // global_reduction, :549-585: first free or matching slot within kMaxProbe probes; otherwise the entry is dropped
    void insert(uint64_t h, const Core& c, const Cov& cv, const Color& cl, float it, uint32_t stamp, size_t& num) {
        if (h == VOXEL_INVALID) return;
        for (size_t p = 0; p < kMaxProbe; ++p) {
            const size_t s = slot_id(h, p, capacity);
            if (key[s] == VOXEL_INVALID) { key[s] = h; ++num; }
            else if (key[s] != h) continue;
            core[s].sx += c.sx; core[s].sy += c.sy; core[s].sz += c.sz; core[s].count += c.count;
            if (has_cov) { cov[s].xx += cv.xx; cov[s].xy += cv.xy; cov[s].xz += cv.xz; cov[s].yy += cv.yy; cov[s].yz += cv.yz; cov[s].zz += cv.zz; }
            if (has_rgb) { color[s].r += cl.r; color[s].g += cl.g; color[s].b += cl.b; color[s].a += cl.a; }
            if (has_intensity) intensity[s] += it;
            last_update[s] = stamp;
            return;
        }
    }
    // rotate_covariance_upper_triangle (:420-458): R C R^T of the 3x3 block, in the reference's fma order
    static Cov rotate_cov(const float* c16, const float* T) {
        const float cxx = c16[0], cxy = c16[4], cxz = c16[8], cyy = c16[5], cyz = c16[9], czz = c16[10];
        const float r00 = T[0], r01 = T[4], r02 = T[8], r10 = T[1], r11 = T[5], r12 = T[9], r20 = T[2], r21 = T[6], r22 = T[10];
        auto f3 = [](float a, float b, float c, float d, float e, float f) { return std::fmaf(a, b, std::fmaf(c, d, e * f)); };
        const float a00 = f3(r02, cxz, r01, cxy, r00, cxx), a01 = f3(r02, cyz, r01, cyy, r00, cxy), a02 = f3(r02, czz, r01, cyz, r00, cxz);
        const float a10 = f3(r12, cxz, r11, cxy, r10, cxx), a11 = f3(r12, cyz, r11, cyy, r10, cxy), a12 = f3(r12, czz, r11, cyz, r10, cxz);
        const float a20 = f3(r22, cxz, r21, cxy, r20, cxx), a21 = f3(r22, cyz, r21, cyy, r20, cxy), a22 = f3(r22, czz, r21, cyz, r20, cxz);
        (void)a20;
        Cov o;
        o.xx = f3(a02, r02, a01, r01, a00, r00); o.xy = f3(a02, r12, a01, r11, a00, r10); o.xz = f3(a02, r22, a01, r21, a00, r20);
        o.yy = f3(a12, r12, a11, r11, a10, r10); o.yz = f3(a12, r22, a11, r21, a10, r20); o.zz = f3(a22, r22, a21, r21, a20, r20);
        return o;
    }
    // add_point_cloud_impl, :594-786 (load_entry + global_reduction per point)
    void add_impl(const float* pts, const float* covs, const float* rgb, const float* inten, size_t n, const float* pose16) {
        has_cov |= covs != nullptr;
        has_rgb |= rgb != nullptr;
        has_intensity |= inten != nullptr;
        size_t num = voxel_num;
        for (size_t i = 0; i < n; ++i) {
            float w[4];
            transform_point(pts + 4 * i, w, pose16);
            const uint64_t h = compute_voxel_bit(w, voxel_size_inv);
            Core c; c.sx = w[0]; c.sy = w[1]; c.sz = w[2]; c.count = 1;
            Cov cv; Color cl; float it = 0.0f;
            if (covs) {
                cv = rotate_cov(covs + 16 * i, pose16);
                Mat3 m;  // encode_covariance_for_aggregation, :460-480
                m(0, 0) = cv.xx; m(0, 1) = m(1, 0) = cv.xy; m(0, 2) = m(2, 0) = cv.xz; m(1, 1) = cv.yy; m(1, 2) = m(2, 1) = cv.yz; m(2, 2) = cv.zz;
                const Mat3 l = log_spd_3x3(m);
                cv.xx = l(0, 0); cv.xy = l(0, 1); cv.xz = l(0, 2); cv.yy = l(1, 1); cv.yz = l(1, 2); cv.zz = l(2, 2);
            }
            if (rgb) { cl.r = rgb[4 * i]; cl.g = rgb[4 * i + 1]; cl.b = rgb[4 * i + 2]; cl.a = rgb[4 * i + 3]; }
            if (inten) it = inten[i];
            insert(h, c, cv, cl, it, staleness_counter, num);
        }
        voxel_num = num;
    }
    // :845-931
    void rehash(size_t new_cap) {
        if (capacity >= new_cap) return;
        const auto okey = key; const auto ocore = core; const auto ocov = cov; const auto ocol = color;
        const auto oint = intensity; const auto olast = last_update;
        const size_t old_cap = capacity;
        allocate(new_cap);
        size_t num = 0;
        for (size_t i = 0; i < old_cap; ++i)
            if (okey[i] != VOXEL_INVALID) insert(okey[i], ocore[i], ocov[i], ocol[i], oint[i], olast[i], num);
        set_voxel_num(num);
    }
};

}  // namespace oracle

// VoxelHashMap for gfx950 — submap accumulation either side of the hot path (replaces
// algorithms/mapping/voxel_hash_map.hpp:22-1072).
//
// The table lives in HBM as six parallel arrays (key u64 | core {sum xyz, count} 16 B | log-covariance sums 24 B | colour
// sums 16 B | intensity sum | last update), open addressing with the reference's double hashing (compute_slot_id, :587-592,
// 100 probes). One lane per input point: transform into the map frame (the reference's fma chain), compute_voxel_bit (the
// K9 key), rotate + log-map the covariance, claim or find the slot with a 64-bit compare-and-swap and add with relaxed
// device-scope atomics (global_atomic_add_f32 executes at the memory side on gfx950; ~1.3 TB/s of added bytes chip-wide,
// MI355X_MICROARCH.md "Global float atomics"). The reference pre-reduces inside a work-group with a bitonic sort before its
// atomics; here that step is dropped: 60 B of atomics per point at 1M points is ~50 us, less than the sort.
// As in the reference the accumulation order is unspecified (float sums agree to rounding, counts exactly).
// Export (downsampling) is deterministic: slot order via flags + exclusive scan (the reference's NVIDIA path, :947-985).
// Host-side control flow (rehash schedule, staleness counter, has_* flags) follows :117-141 line by line; like the
// reference, add_point_cloud waits for its kernel and reads the voxel count back.

#include "radix_sort.h"
#include "sp_common.h"
#include "sp_math.h"

void sp_set_error(const char* msg);

namespace sp {
namespace {

constexpr uint64_t kInvalidKey = ~0ull;  // VoxelConstants::invalid_coord
constexpr unsigned kMaxProbe = 100;      // voxel_hash_map.hpp:505
constexpr size_t kCapacityCandidates[11] = {30029,  60013,   120011,  240007,   480013,  960017,
                                            1920001, 3840007, 7680017, 15360013, 30720007};  // :486-487

struct CovSum { float xx, xy, xz, yy, yz, zz; };

struct Table {
    uint64_t* key;
    float4* core;        // sum_x, sum_y, sum_z, count (uint32 bits)
    CovSum* cov;         // sums of log(C) (upper triangle)
    float4* color;
    float* intensity;
    uint32_t* last_update;
    unsigned long long capacity;
};

// filter::kernel::compute_voxel_bit (voxel_constants.hpp:36-62) — the same arithmetic as voxel.hip's K9
__device__ __forceinline__ uint64_t voxel_key3(float x, float y, float z, float inv) {
    constexpr int64_t mask = (1 << 21) - 1;
    constexpr int64_t offset = 1 << 20;
    if (!isfinite(x) || !isfinite(y) || !isfinite(z)) return kInvalidKey;
    const int64_t c0 = (int64_t)floorf(x * inv) + offset;
    const int64_t c1 = (int64_t)floorf(y * inv) + offset;
    const int64_t c2 = (int64_t)floorf(z * inv) + offset;
    if (c0 < 0 || mask < c0 || c1 < 0 || mask < c1 || c2 < 0 || mask < c2) return kInvalidKey;
    return ((uint64_t)(c0 & mask)) | ((uint64_t)(c1 & mask) << 21) | ((uint64_t)(c2 & mask) << 42);
}

// :587-592
__device__ __forceinline__ unsigned long long slot_id(uint64_t h, unsigned long long probe, unsigned long long cap) {
    const unsigned long long h2 = (cap - 2) - (h % (cap - 2));
    return (h + probe * h2) % cap;
}

// V diag(f(ev)) V^T, symmetrised (eigen_utils.hpp:646-677): LOG = log(max(ev, 1e-6)), else exp(ev)
template <bool LOG>
__device__ __forceinline__ Mat3 spd_map(const Mat3& A) {
    float ev[3];
    Mat3 V;
    symmetric_eigen3(A, ev, V);
    float f[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) f[i] = LOG ? logf(sycl_max(ev[i], 1e-6f)) : expf(ev[i]);
    Mat3 VD;  // multiply<3,3,3>(V, diag): per element an fma chain over k with two zero terms
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) s = fmaf(V.m[i][k], (k == j) ? f[k] : 0.0f, s);
            VD.m[i][j] = s;
        }
    const Mat3 P = matmul_bt(VD, V);  // (V D) V^T
    Mat3 S;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) S.m[i][j] = (i == j) ? P.m[i][j] : (P.m[i][j] + P.m[j][i]) * 0.5f;
    return S;
}

// rotate_covariance_upper_triangle (:420-458), the reference's fma order
__device__ __forceinline__ CovSum rotate_cov(const float4* __restrict__ c, const Rigid& T) {
    const float4 c0 = c[0], c1 = c[1], c2 = c[2];  // columns of the 4x4
    const float cxx = c0.x, cxy = c1.x, cxz = c2.x, cyy = c1.y, cyz = c2.y, czz = c2.z;
    const float(&R)[3][3] = T.R;
    auto f3 = [](float a, float b, float c_, float d, float e, float f) { return fmaf(a, b, fmaf(c_, d, e * f)); };
    const float a00 = f3(R[0][2], cxz, R[0][1], cxy, R[0][0], cxx), a01 = f3(R[0][2], cyz, R[0][1], cyy, R[0][0], cxy),
                a02 = f3(R[0][2], czz, R[0][1], cyz, R[0][0], cxz);
    const float a10 = f3(R[1][2], cxz, R[1][1], cxy, R[1][0], cxx), a11 = f3(R[1][2], cyz, R[1][1], cyy, R[1][0], cxy),
                a12 = f3(R[1][2], czz, R[1][1], cyz, R[1][0], cxz);
    const float a20 = f3(R[2][2], cxz, R[2][1], cxy, R[2][0], cxx), a21 = f3(R[2][2], cyz, R[2][1], cyy, R[2][0], cxy),
                a22 = f3(R[2][2], czz, R[2][1], cyz, R[2][0], cxz);
    CovSum o;
    o.xx = f3(a02, R[0][2], a01, R[0][1], a00, R[0][0]);
    o.xy = f3(a02, R[1][2], a01, R[1][1], a00, R[1][0]);
    o.xz = f3(a02, R[2][2], a01, R[2][1], a00, R[2][0]);
    o.yy = f3(a12, R[1][2], a11, R[1][1], a10, R[1][0]);
    o.yz = f3(a12, R[2][2], a11, R[2][1], a10, R[2][0]);
    o.zz = f3(a22, R[2][2], a21, R[2][1], a20, R[2][0]);
    return o;
}

// atomic_ref<float, relaxed, device>::fetch_add with the result unused (:254-285): the hardware's no-return
// global_atomic_add_f32 (plain atomicAdd(float*) compiles to a compare-and-swap loop without -munsafe-fp-atomics; the
// table is hipMalloc memory, where the hardware form is valid)
__device__ __forceinline__ void fadd(float* p, float v) { unsafeAtomicAdd(p, v); }

// global_reduction (:549-585): claim the first free slot or find the key's slot within kMaxProbe probes, then add.
// An entry that finds neither is dropped, as in the reference.
__device__ __forceinline__ void insert(const Table& t, uint64_t h, float sx, float sy, float sz, unsigned count,
                                       const CovSum& cv, bool has_cov, const float4 col, bool has_rgb, float inten,
                                       bool has_intensity, uint32_t stamp, unsigned* __restrict__ voxel_num) {
    if (h == kInvalidKey) return;
    for (unsigned p = 0; p < kMaxProbe; ++p) {
        const unsigned long long s = slot_id(h, p, t.capacity);
        const unsigned long long seen = atomicCAS(reinterpret_cast<unsigned long long*>(t.key + s), kInvalidKey, h);
        if (seen == kInvalidKey) atomicAdd(voxel_num, 1u);
        else if (seen != h) continue;
        float* core = reinterpret_cast<float*>(t.core + s);
        fadd(core + 0, sx);
        fadd(core + 1, sy);
        fadd(core + 2, sz);
        atomicAdd(reinterpret_cast<unsigned*>(core + 3), count);
        if (has_cov) {
            float* c = reinterpret_cast<float*>(t.cov + s);
            fadd(c + 0, cv.xx); fadd(c + 1, cv.xy); fadd(c + 2, cv.xz);
            fadd(c + 3, cv.yy); fadd(c + 4, cv.yz); fadd(c + 5, cv.zz);
        }
        if (has_rgb) {
            float* c = reinterpret_cast<float*>(t.color + s);
            fadd(c + 0, col.x); fadd(c + 1, col.y); fadd(c + 2, col.z); fadd(c + 3, col.w);
        }
        if (has_intensity) fadd(t.intensity + s, inten);
        __hip_atomic_store(t.last_update + s, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // :325-328
        return;
    }
}

// add_point_cloud_impl (:594-786): load_entry + global_reduction, one lane per point
__global__ __launch_bounds__(kBlock) void vhm_add_kernel(Table t, const float4* __restrict__ pts,
                                                         const float4* __restrict__ covs, const float4* __restrict__ rgb,
                                                         const float* __restrict__ inten, unsigned n, Mat4Arg pose,
                                                         float inv, bool map_has_cov, bool map_has_rgb,
                                                         bool map_has_intensity, uint32_t stamp,
                                                         unsigned* __restrict__ voxel_num) {
    const Rigid T = load_rigid_colmajor(pose.m);
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 p = pts[i];
        float wx, wy, wz;
        transform_point(T, p.x, p.y, p.z, wx, wy, wz);
        const uint64_t h = voxel_key3(wx, wy, wz, inv);
        CovSum cv{0, 0, 0, 0, 0, 0};
        if (covs && h != kInvalidKey) {  // rotate into the map frame, then log-Euclidean encoding (:460-480)
            const CovSum r = rotate_cov(covs + 4 * (size_t)i, T);
            Mat3 m;
            m.m[0][0] = r.xx; m.m[0][1] = m.m[1][0] = r.xy; m.m[0][2] = m.m[2][0] = r.xz;
            m.m[1][1] = r.yy; m.m[1][2] = m.m[2][1] = r.yz; m.m[2][2] = r.zz;
            const Mat3 l = spd_map<true>(m);
            cv = CovSum{l.m[0][0], l.m[0][1], l.m[0][2], l.m[1][1], l.m[1][2], l.m[2][2]};
        }
        const float4 col = rgb ? rgb[i] : make_float4(0, 0, 0, 0);
        insert(t, h, wx, wy, wz, 1u, cv, map_has_cov, col, map_has_rgb, inten ? inten[i] : 0.0f, map_has_intensity, stamp,
               voxel_num);
    }
}

// rehash (:845-931): every live slot of the old table re-enters the new one with its own time stamp
__global__ __launch_bounds__(kBlock) void vhm_rehash_kernel(Table old_t, Table new_t, bool has_cov, bool has_rgb,
                                                            bool has_intensity, unsigned* __restrict__ voxel_num) {
    const unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= old_t.capacity) return;
    const uint64_t k = old_t.key[i];
    if (k == kInvalidKey) return;
    const float4 c = old_t.core[i];
    insert(new_t, k, c.x, c.y, c.z, __float_as_uint(c.w), has_cov ? old_t.cov[i] : CovSum{0, 0, 0, 0, 0, 0}, has_cov,
           has_rgb ? old_t.color[i] : make_float4(0, 0, 0, 0), has_rgb, has_intensity ? old_t.intensity[i] : 0.0f,
           has_intensity, old_t.last_update[i], voxel_num);
}

// remove_old_data_impl (:788-843)
__global__ __launch_bounds__(kBlock) void vhm_remove_kernel(Table t, uint32_t remove_staleness,
                                                            unsigned* __restrict__ voxel_num) {
    const unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= t.capacity) return;
    if (t.key[i] == kInvalidKey) return;
    if (t.last_update[i] >= remove_staleness) { atomicAdd(voxel_num, 1u); return; }
    t.key[i] = kInvalidKey;
    t.core[i] = make_float4(0, 0, 0, 0);
    t.cov[i] = CovSum{0, 0, 0, 0, 0, 0};
    t.color[i] = make_float4(0, 0, 0, 0);
    t.intensity[i] = 0.0f;
    t.last_update[i] = 0;
}

// should_include_voxel (:402-418)
__global__ __launch_bounds__(kBlock) void vhm_flag_kernel(Table t, uint32_t min_num_point, float mnx, float mny, float mnz,
                                                          float mxx, float mxy, float mxz, unsigned* __restrict__ flags) {
    const unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= t.capacity) return;
    const float4 c = t.core[i];
    const unsigned count = __float_as_uint(c.w);
    bool keep = t.key[i] != kInvalidKey && count >= min_num_point && count != 0u;
    if (keep) {
        const float inv = 1.0f / (float)count;
        const float cx = c.x * inv, cy = c.y * inv, cz = c.z * inv;
        keep = (cx >= mnx && cx <= mxx) && (cy >= mny && cy <= mxy) && (cz >= mnz && cz <= mxz);
    }
    flags[i] = keep ? 1u : 0u;
}

// compute_averaged_attributes (:330-386) into the compacted outputs (slot order)
__global__ __launch_bounds__(kBlock) void vhm_export_kernel(Table t, const unsigned* __restrict__ flags,
                                                            const unsigned* __restrict__ pos, unsigned out_capacity,
                                                            float4* __restrict__ pts_out, float4* __restrict__ cov_out,
                                                            float4* __restrict__ rgb_out, float* __restrict__ inten_out,
                                                            uint64_t* __restrict__ keys_out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= t.capacity || !flags[i]) return;
    const unsigned o = pos[i];
    if (o >= out_capacity) return;
    const float4 c = t.core[i];
    const float inv = 1.0f / (float)__float_as_uint(c.w);
    pts_out[o] = make_float4(c.x * inv, c.y * inv, c.z * inv, 1.0f);
    if (cov_out) {
        const CovSum s = t.cov[i];
        Mat3 m;
        m.m[0][0] = s.xx * inv; m.m[0][1] = m.m[1][0] = s.xy * inv; m.m[0][2] = m.m[2][0] = s.xz * inv;
        m.m[1][1] = s.yy * inv; m.m[1][2] = m.m[2][1] = s.yz * inv; m.m[2][2] = s.zz * inv;
        const Mat3 e = spd_map<false>(m);
        float4* o4 = cov_out + 4 * (size_t)o;  // column-major 4x4, 3x3 block used
        o4[0] = make_float4(e.m[0][0], e.m[1][0], e.m[2][0], 0.0f);
        o4[1] = make_float4(e.m[0][1], e.m[1][1], e.m[2][1], 0.0f);
        o4[2] = make_float4(e.m[0][2], e.m[1][2], e.m[2][2], 0.0f);
        o4[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (rgb_out) {
        const float4 k = t.color[i];
        rgb_out[o] = make_float4(k.x * inv, k.y * inv, k.z * inv, k.w * inv);
    }
    if (inten_out) inten_out[o] = t.intensity[i] * inv;
    if (keys_out) keys_out[o] = t.key[i];
}

// compute_overlap_ratio (:196-246)
__global__ __launch_bounds__(kBlock) void vhm_overlap_kernel(Table t, const float4* __restrict__ pts, unsigned n,
                                                             Mat4Arg pose, float inv, uint32_t min_num_point,
                                                             unsigned* __restrict__ hits) {
    const Rigid T = load_rigid_colmajor(pose.m);
    unsigned mine = 0;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 p = pts[i];
        float wx, wy, wz;
        transform_point(T, p.x, p.y, p.z, wx, wy, wz);
        const uint64_t h = voxel_key3(wx, wy, wz, inv);
        if (h == kInvalidKey) continue;
        for (unsigned pr = 0; pr < kMaxProbe; ++pr) {
            const unsigned long long s = slot_id(h, pr, t.capacity);
            const uint64_t k = t.key[s];
            if (k == h) { if (__float_as_uint(t.core[s].w) >= min_num_point) ++mine; break; }
            if (k == kInvalidKey) break;
        }
    }
    mine = wave_sum_u32(mine);
    if ((threadIdx.x & (kWave - 1)) == 0 && mine) atomicAdd(hits, mine);  // integer: exact, order-independent
}

__global__ void vhm_fill_keys_kernel(uint64_t* keys, unsigned long long n) {
    const unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) keys[i] = kInvalidKey;
}

}  // namespace
}  // namespace sp

struct sp_voxel_hash_map {
    float voxel_size = 0.0f, voxel_size_inv = 0.0f;
    uint32_t max_staleness = 100, remove_old_data_cycle = 10, min_num_point = 1, staleness_counter = 0;
    float rehash_threshold = 0.7f;
    size_t voxel_num = 0;
    bool has_cov = false, has_rgb = false, has_intensity = false;
    sp::Table t{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    unsigned* counter = nullptr;        // device: voxel count / hit count of the running call
    unsigned *flags = nullptr, *pos = nullptr;  // export scratch, sized to the capacity
    size_t scratch_cap = 0;
    void* scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
};

namespace sp {
namespace {

void free_table(Table& t) {
    (void)hipFree(t.key); (void)hipFree(t.core); (void)hipFree(t.cov); (void)hipFree(t.color); (void)hipFree(t.intensity);
    (void)hipFree(t.last_update);
    t = Table{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
}

// allocate_storage (:528-544): keys invalid, everything else zero
int alloc_table(Table& t, size_t cap, hipStream_t st) {
    hipError_t e = hipMalloc(&t.key, cap * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&t.core, cap * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&t.cov, cap * sizeof(CovSum));
    if (e == hipSuccess) e = hipMalloc(&t.color, cap * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&t.intensity, cap * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&t.last_update, cap * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(t.core, 0, cap * sizeof(float4), st);
    if (e == hipSuccess) e = hipMemsetAsync(t.cov, 0, cap * sizeof(CovSum), st);
    if (e == hipSuccess) e = hipMemsetAsync(t.color, 0, cap * sizeof(float4), st);
    if (e == hipSuccess) e = hipMemsetAsync(t.intensity, 0, cap * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(t.last_update, 0, cap * sizeof(uint32_t), st);
    if (e != hipSuccess) { sp_set_error(hipGetErrorString(e)); free_table(t); return SP_ERR_HIP; }
    t.capacity = cap;
    vhm_fill_keys_kernel<<<div_up(cap, kBlock), kBlock, 0, st>>>(t.key, cap);
    return launch_status();
}

void set_voxel_num(sp_voxel_hash_map* m, size_t n) {  // update_voxel_num_and_flags (:519-526)
    m->voxel_num = n;
    if (n == 0) m->has_cov = m->has_rgb = m->has_intensity = false;
}

int read_counter(sp_voxel_hash_map* m, hipStream_t st, unsigned* out) {
    if (hipMemcpyAsync(out, m->counter, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess) return SP_ERR_HIP;
    return hip_status(hipStreamSynchronize(st));  // the reference waits here too (wait_and_throw + shared read)
}
__global__ void vhm_seed_counter_kernel(unsigned* counter, unsigned v) { *counter = v; }
int write_counter(sp_voxel_hash_map* m, hipStream_t st, unsigned v) {
    // the value travels in the kernarg segment: nothing is read from a host variable after this returns
    vhm_seed_counter_kernel<<<1, 1, 0, st>>>(m->counter, v);
    return launch_status();
}

int rehash(sp_voxel_hash_map* m, size_t new_cap, hipStream_t st) {
    if (m->t.capacity >= new_cap) return SP_OK;
    Table old_t = m->t, new_t{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    int rc = alloc_table(new_t, new_cap, st);
    if (rc != SP_OK) return rc;
    if ((rc = write_counter(m, st, 0)) != SP_OK) { free_table(new_t); return rc; }
    vhm_rehash_kernel<<<div_up(old_t.capacity, kBlock), kBlock, 0, st>>>(old_t, new_t, m->has_cov, m->has_rgb,
                                                                        m->has_intensity, m->counter);
    unsigned cnt = 0;
    rc = launch_status();
    if (rc == SP_OK) rc = read_counter(m, st, &cnt);
    if (rc != SP_OK) { free_table(new_t); return rc; }
    m->t = new_t;
    free_table(old_t);
    set_voxel_num(m, cnt);
    return SP_OK;
}

int remove_old(sp_voxel_hash_map* m, hipStream_t st) {
    if (m->staleness_counter <= m->max_staleness) return SP_OK;
    int rc = write_counter(m, st, 0);
    if (rc != SP_OK) return rc;
    vhm_remove_kernel<<<div_up(m->t.capacity, kBlock), kBlock, 0, st>>>(m->t, m->staleness_counter - m->max_staleness,
                                                                      m->counter);
    unsigned cnt = 0;
    rc = launch_status();
    if (rc == SP_OK) rc = read_counter(m, st, &cnt);
    if (rc == SP_OK) set_voxel_num(m, cnt);
    return rc;
}

Mat4Arg pose_arg(const float* pose16) {
    Mat4Arg a;
    for (int i = 0; i < 16; ++i) a.m[i] = pose16 ? pose16[i] : ((i % 5 == 0) ? 1.0f : 0.0f);
    return a;
}

}  // namespace
}  // namespace sp

extern "C" int sp_vhm_create(float voxel_size, void* stream, sp_voxel_hash_map** out) {
    using namespace sp;
    if (!out) return SP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!(voxel_size > 0.0f)) {
        sp_set_error("voxel_size must be positive.");  // voxel_hash_map.hpp:41-43
        return SP_ERR_INVALID_ARGUMENT;
    }
    sp_voxel_hash_map* m = new sp_voxel_hash_map();
    m->voxel_size = voxel_size;
    m->voxel_size_inv = 1.0f / voxel_size;
    hipStream_t st = as_stream(stream);
    int rc = hip_status(hipMalloc(&m->counter, sizeof(unsigned)));
    if (rc == SP_OK) rc = alloc_table(m->t, kCapacityCandidates[0], st);
    if (rc == SP_OK) rc = hip_status(hipStreamSynchronize(st));
    if (rc != SP_OK) { sp_vhm_destroy(m); return rc; }
    *out = m;
    return SP_OK;
}

extern "C" void sp_vhm_destroy(sp_voxel_hash_map* m) {
    if (!m) return;
    sp::free_table(m->t);
    (void)hipFree(m->counter); (void)hipFree(m->flags); (void)hipFree(m->pos); (void)hipFree(m->scan_tmp);
    delete m;
}

extern "C" int sp_vhm_clear(sp_voxel_hash_map* m, void* stream) {  // :83-113
    using namespace sp;
    if (!m) return SP_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    Table fresh{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    int rc = alloc_table(fresh, kCapacityCandidates[0], st);
    if (rc == SP_OK) rc = hip_status(hipStreamSynchronize(st));
    if (rc != SP_OK) return rc;
    free_table(m->t);
    m->t = fresh;
    m->voxel_num = 0;
    m->staleness_counter = 0;
    m->has_cov = m->has_rgb = m->has_intensity = false;
    return SP_OK;
}

extern "C" int sp_vhm_set(sp_voxel_hash_map* m, int param, float value) {
    if (!m) return SP_ERR_INVALID_ARGUMENT;
    switch (param) {
        case SP_VHM_VOXEL_SIZE:
            if (!(value > 0.0f)) { sp_set_error("voxel_size must be positive."); return SP_ERR_INVALID_ARGUMENT; }
            m->voxel_size = value;
            m->voxel_size_inv = 1.0f / value;
            return SP_OK;
        case SP_VHM_MAX_STALENESS: m->max_staleness = (uint32_t)value; return SP_OK;
        case SP_VHM_REMOVE_OLD_DATA_CYCLE: m->remove_old_data_cycle = (uint32_t)value; return SP_OK;
        case SP_VHM_REHASH_THRESHOLD: m->rehash_threshold = value; return SP_OK;
        case SP_VHM_MIN_NUM_POINT: m->min_num_point = (uint32_t)value; return SP_OK;
    }
    return SP_ERR_INVALID_ARGUMENT;
}
extern "C" float sp_vhm_get(const sp_voxel_hash_map* m, int param) {
    if (!m) return 0.0f;
    switch (param) {
        case SP_VHM_VOXEL_SIZE: return m->voxel_size;
        case SP_VHM_MAX_STALENESS: return (float)m->max_staleness;
        case SP_VHM_REMOVE_OLD_DATA_CYCLE: return (float)m->remove_old_data_cycle;
        case SP_VHM_REHASH_THRESHOLD: return m->rehash_threshold;
        case SP_VHM_MIN_NUM_POINT: return (float)m->min_num_point;
    }
    return 0.0f;
}
extern "C" size_t sp_vhm_info(const sp_voxel_hash_map* m, int what) {
    if (!m) return 0;
    switch (what) {
        case SP_VHM_INFO_VOXEL_NUM: return m->voxel_num;
        case SP_VHM_INFO_CAPACITY: return (size_t)m->t.capacity;
        case SP_VHM_INFO_STALENESS_COUNTER: return m->staleness_counter;
        case SP_VHM_INFO_HAS_COV: return m->has_cov;
        case SP_VHM_INFO_HAS_RGB: return m->has_rgb;
        case SP_VHM_INFO_HAS_INTENSITY: return m->has_intensity;
    }
    return 0;
}

extern "C" int sp_vhm_remove_old_data(sp_voxel_hash_map* m, void* stream) {
    if (!m) return SP_ERR_INVALID_ARGUMENT;
    return sp::remove_old(m, sp::as_stream(stream));
}

// add_point_cloud (:117-141)
extern "C" int sp_vhm_add_point_cloud(sp_voxel_hash_map* m, const float* points, const float* covs, const float* rgb,
                                      const float* intensities, size_t n, const float* sensor_pose_host16, void* stream) {
    using namespace sp;
    if (!m || (n && !points)) return SP_ERR_INVALID_ARGUMENT;
    if (n >= (1ull << 32)) { sp_set_error("[VoxelHashMap] more than 2^32 points"); return SP_ERR_INVALID_ARGUMENT; }
    hipStream_t st = as_stream(stream);
    int rc = SP_OK;
    if (m->rehash_threshold < (float)m->voxel_num / (float)m->t.capacity) {
        size_t next = (size_t)m->t.capacity;
        for (const size_t c : kCapacityCandidates)
            if (c > m->t.capacity) { next = c; break; }
        if (next > m->t.capacity && (rc = rehash(m, next, st)) != SP_OK) return rc;
    }
    if (n > 0) {
        m->has_cov |= covs != nullptr;
        m->has_rgb |= rgb != nullptr;
        m->has_intensity |= intensities != nullptr;
        if ((rc = write_counter(m, st, (unsigned)m->voxel_num)) != SP_OK) return rc;
        vhm_add_kernel<<<stream_grid(n), kBlock, 0, st>>>(
            m->t, reinterpret_cast<const float4*>(points), reinterpret_cast<const float4*>(covs),
            reinterpret_cast<const float4*>(rgb), intensities, (unsigned)n, pose_arg(sensor_pose_host16), m->voxel_size_inv,
            m->has_cov, m->has_rgb, m->has_intensity, m->staleness_counter, m->counter);
        unsigned cnt = 0;
        rc = launch_status();
        if (rc == SP_OK) rc = read_counter(m, st, &cnt);
        if (rc != SP_OK) return rc;
        m->voxel_num = cnt;
    }
    if (m->remove_old_data_cycle > 0 && (m->staleness_counter % m->remove_old_data_cycle) == 0)
        if ((rc = remove_old(m, st)) != SP_OK) return rc;
    ++m->staleness_counter;
    return SP_OK;
}

// downsampling (:146-190) + downsampling_impl (:933-1068)
extern "C" int sp_vhm_downsampling(sp_voxel_hash_map* m, const float* center_host3, float distance, float* points_out,
                                   float* covs_out, float* rgb_out, float* intensities_out, uint64_t* keys_out_opt,
                                   size_t out_capacity, size_t* n_out_host, void* stream) {
    using namespace sp;
    if (!m || !center_host3 || !n_out_host) return SP_ERR_INVALID_ARGUMENT;
    *n_out_host = 0;
    if (m->voxel_num == 0) return SP_OK;
    if (!points_out || out_capacity < m->voxel_num) {
        sp_set_error("[VoxelHashMap::downsampling] output arrays must hold sp_vhm_info(SP_VHM_INFO_VOXEL_NUM) entries");
        return SP_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = as_stream(stream);
    const size_t cap = (size_t)m->t.capacity;
    if (m->scratch_cap < cap) {
        (void)hipFree(m->flags); (void)hipFree(m->pos); (void)hipFree(m->scan_tmp);
        m->flags = m->pos = nullptr; m->scan_tmp = nullptr; m->scratch_cap = 0;
        const size_t tmp = exclusive_scan_u32_workspace_bytes(cap + 1);
        hipError_t e = hipMalloc(&m->flags, (cap + 1) * sizeof(unsigned));
        if (e == hipSuccess) e = hipMalloc(&m->pos, (cap + 1) * sizeof(unsigned));
        if (e == hipSuccess) e = hipMalloc(&m->scan_tmp, tmp ? tmp : 16);
        if (e != hipSuccess) { sp_set_error(hipGetErrorString(e)); return SP_ERR_HIP; }
        m->scan_tmp_bytes = tmp;
        m->scratch_cap = cap;
    }
    vhm_flag_kernel<<<div_up(cap, kBlock), kBlock, 0, st>>>(m->t, m->min_num_point, center_host3[0] - distance,
                                                           center_host3[1] - distance, center_host3[2] - distance,
                                                           center_host3[0] + distance, center_host3[1] + distance,
                                                           center_host3[2] + distance, m->flags);
    if (hipMemsetAsync(m->flags + cap, 0, sizeof(unsigned), st) != hipSuccess) return SP_ERR_HIP;
    if (exclusive_scan_u32(m->flags, m->pos, cap + 1, nullptr, m->scan_tmp, m->scan_tmp_bytes, st) != SP_OK) {
        sp_set_error("[VoxelHashMap::downsampling] scan failed");
        return SP_ERR_HIP;
    }
    vhm_export_kernel<<<div_up(cap, kBlock), kBlock, 0, st>>>(
        m->t, m->flags, m->pos, (unsigned)out_capacity, reinterpret_cast<float4*>(points_out),
        m->has_cov ? reinterpret_cast<float4*>(covs_out) : nullptr, m->has_rgb ? reinterpret_cast<float4*>(rgb_out) : nullptr,
        m->has_intensity ? intensities_out : nullptr, keys_out_opt);
    int rc = launch_status();
    unsigned total = 0;
    if (rc == SP_OK && hipMemcpyAsync(&total, m->pos + cap, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess) rc = SP_ERR_HIP;
    if (rc == SP_OK) rc = hip_status(hipStreamSynchronize(st));
    if (rc == SP_OK) *n_out_host = total;
    return rc;
}

extern "C" int sp_vhm_overlap_ratio(const sp_voxel_hash_map* m, const float* points, size_t n,
                                    const float* sensor_pose_host16, float* ratio_out_host, void* stream) {
    using namespace sp;
    if (!m || !ratio_out_host) return SP_ERR_INVALID_ARGUMENT;
    *ratio_out_host = 0.0f;
    if (n == 0 || !points || m->voxel_num == 0) return SP_OK;
    hipStream_t st = as_stream(stream);
    sp_voxel_hash_map* mm = const_cast<sp_voxel_hash_map*>(m);  // the counter scratch only
    int rc = write_counter(mm, st, 0);
    if (rc != SP_OK) return rc;
    vhm_overlap_kernel<<<stream_grid(n), kBlock, 0, st>>>(m->t, reinterpret_cast<const float4*>(points), (unsigned)n,
                                                          pose_arg(sensor_pose_host16), m->voxel_size_inv, m->min_num_point,
                                                          m->counter);
    unsigned hits = 0;
    rc = launch_status();
    if (rc == SP_OK) rc = read_counter(mm, st, &hits);
    if (rc == SP_OK) *ratio_out_host = (float)hits / (float)n;
    return rc;
}

// sycl_points facade for MI355X — core types.
//
// Source-compatible stand-ins for the reference's L1/L3 layers:
//   utils/sycl_utils.hpp  : sycl_utils::DeviceQueue, sycl_utils::events, shared_vector<T>, shared_vector_ptr<T>
//   points/types.hpp      : PointType, Covariance, Normal, RGBType, TransformMatrix, containers
//   points/point_cloud.hpp: PointCloudCPU, PointCloudShared
// The reference keeps every attribute in USM-shared std::vectors and lets pages migrate. On MI355X (XNACK off) managed
// pages would be served over PCIe, so shared_vector<T> here is a host std::vector<T> plus an explicit HBM mirror with
// dirty tracking: host accessors behave like std::vector, kernels take device_data(), and copies happen only when the
// side being read is stale — at the same points where the reference calls set_accessed_by_host / _device
// (utils/sycl_utils.hpp:572-625).
#pragma once
#include <hip/hip_runtime.h>
#include <sycl_points_amd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <map>
#include <memory>
#include <new>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#include <Eigen/Geometry>
#else
// Minimal Eigen subset (the container image has no Eigen): fixed-size float matrices, column-major like Eigen.
namespace Eigen {
template <typename T, int R, int C>
struct Matrix {
    static_assert(std::is_same<T, float>::value, "eigen_lite: float only");
    T d[R * C];
    Matrix() { for (int i = 0; i < R * C; ++i) d[i] = T(0); }
    Matrix(T x, T y, T z) { static_assert(R * C == 3, "size"); d[0] = x; d[1] = y; d[2] = z; }
    Matrix(T x, T y, T z, T w) { static_assert(R * C == 4, "size"); d[0] = x; d[1] = y; d[2] = z; d[3] = w; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Identity() { Matrix m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1); return m; }
    T& operator()(int i, int j) { return d[j * R + i]; }
    const T& operator()(int i, int j) const { return d[j * R + i]; }
    T& operator()(int i) { return d[i]; }
    const T& operator()(int i) const { return d[i]; }
    T& operator[](int i) { return d[i]; }
    const T& operator[](int i) const { return d[i]; }
    T* data() { return d; }
    const T* data() const { return d; }
    T& x() { return d[0]; } T& y() { return d[1]; } T& z() { return d[2]; } T& w() { return d[3]; }
    const T& x() const { return d[0]; } const T& y() const { return d[1]; } const T& z() const { return d[2]; } const T& w() const { return d[3]; }
    static constexpr int rows() { return R; }
    static constexpr int cols() { return C; }
    void setZero() { for (int i = 0; i < R * C; ++i) d[i] = T(0); }
    void setIdentity() { *this = Identity(); }
    Matrix<T, C, R> transpose() const { Matrix<T, C, R> t; for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) t(j, i) = (*this)(i, j); return t; }
    Matrix operator+(const Matrix& o) const { Matrix r; for (int i = 0; i < R * C; ++i) r.d[i] = d[i] + o.d[i]; return r; }
    Matrix operator-(const Matrix& o) const { Matrix r; for (int i = 0; i < R * C; ++i) r.d[i] = d[i] - o.d[i]; return r; }
    Matrix operator-() const { Matrix r; for (int i = 0; i < R * C; ++i) r.d[i] = -d[i]; return r; }
    Matrix operator*(T s) const { Matrix r; for (int i = 0; i < R * C; ++i) r.d[i] = d[i] * s; return r; }
    Matrix operator/(T s) const { Matrix r; for (int i = 0; i < R * C; ++i) r.d[i] = d[i] / s; return r; }
    Matrix& operator+=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d[i] += o.d[i]; return *this; }
    template <int K>
    Matrix<T, R, K> operator*(const Matrix<T, C, K>& o) const {
        Matrix<T, R, K> r;
        for (int i = 0; i < R; ++i) for (int j = 0; j < K; ++j) { T s = T(0); for (int k = 0; k < C; ++k) s += (*this)(i, k) * o(k, j); r(i, j) = s; }
        return r;
    }
    T norm() const { T s = T(0); for (int i = 0; i < R * C; ++i) s += d[i] * d[i]; return std::sqrt(s); }
    T squaredNorm() const { T s = T(0); for (int i = 0; i < R * C; ++i) s += d[i] * d[i]; return s; }
    bool operator==(const Matrix& o) const { return std::memcmp(d, o.d, sizeof d) == 0; }
};
template <typename T, int N>
using Vector = Matrix<T, N, 1>;
using Vector3f = Matrix<float, 3, 1>;
using Vector4f = Matrix<float, 4, 1>;
using Matrix3f = Matrix<float, 3, 3>;
using Matrix4f = Matrix<float, 4, 4>;
template <typename T> struct aligned_allocator : std::allocator<T> {
    template <typename U> struct rebind { using other = aligned_allocator<U>; };
};
enum TransformTraits { Isometry = 1 };
// Transform<float,3,Isometry> subset: a 4x4 whose last row is 0 0 0 1.
struct Isometry3f {
    Matrix4f m = Matrix4f::Identity();
    Isometry3f() = default;
    explicit Isometry3f(const Matrix4f& mm) : m(mm) {}
    static Isometry3f Identity() { return Isometry3f(); }
    Matrix4f& matrix() { return m; }
    const Matrix4f& matrix() const { return m; }
    Matrix3f linear() const { Matrix3f r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = m(i, j); return r; }
    Vector3f translation() const { return Vector3f(m(0, 3), m(1, 3), m(2, 3)); }
    Isometry3f operator*(const Isometry3f& o) const { Isometry3f r; sp_rigid_mul_host(m.data(), o.m.data(), r.m.data()); return r; }
    Isometry3f inverse() const {
        Isometry3f r;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m(i, j) = m(j, i);
        for (int i = 0; i < 3; ++i) r.m(i, 3) = -(r.m(i, 0) * m(0, 3) + r.m(i, 1) * m(1, 3) + r.m(i, 2) * m(2, 3));
        return r;
    }
};
}  // namespace Eigen
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif

namespace sycl_points {

// ---------------------------------------------------------------------------------------------- errors
inline void throw_on_error(int rc) {
    if (rc == SP_OK) return;
    const std::string msg = sp_last_error();
    if (rc == SP_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}
inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

namespace detail {
// The HBM mirrors of the containers (shared_vector) and the per-call scratch of the algorithms come from a small cache of
// device buffers: hipMalloc / hipFree cost 0.1-0.2 ms apiece on this runtime, and a pipeline that builds a few point clouds
// per frame (the reference's example: twenty attribute vectors per loop) spent more time there than in its kernels.
// A released buffer is tagged with an event recorded on the stream its owner used (no device-wide wait in a destructor —
// round 2 called hipDeviceSynchronize() for every buffer a shared_vector dropped) AND on every other live DeviceQueue stream
// of that device: a container is read by kernels of other queues too (a tree's queue searching a query cloud bound to another
// queue, a registration reading both clouds), and those kernels are not behind the owner's event. Usually there is one queue
// and one event; a program with q queues pays q events per released buffer, never a device-wide wait. The buffer is handed
// out again only once all its events have completed; buffers are keyed by the device they were allocated on. At most 48
// buffers / 2 GiB are kept. Streams the facade did not create (a raw stream handed to the C ABI with a container's pointer)
// are the caller's to order against the container's lifetime.
// (hipEventQuery is not capture-safe: do not create or destroy containers while a stream of the process is being captured.)
struct QueueStreams {  // the live DeviceQueue streams of the process
    static void add(hipStream_t s, int device) {
        std::lock_guard<std::mutex> lock(mutex());
        list().push_back({s, device});
    }
    static void remove(hipStream_t s) {
        std::lock_guard<std::mutex> lock(mutex());
        auto& l = list();
        for (size_t i = 0; i < l.size(); ++i)
            if (l[i].first == s) { l.erase(l.begin() + (std::ptrdiff_t)i); break; }
    }
    /// no live queue other than the one that owns `s`: nothing but work on `s` itself can touch that queue's containers
    static bool alone(hipStream_t s) {
        std::lock_guard<std::mutex> lock(mutex());
        for (const auto& e : list())
            if (e.first != s) return false;
        return true;
    }
    static bool is_live(hipStream_t s) {
        std::lock_guard<std::mutex> lock(mutex());
        for (const auto& e : list())
            if (e.first == s) return true;
        return false;
    }
    static std::vector<hipStream_t> others(hipStream_t own, int device) {
        std::lock_guard<std::mutex> lock(mutex());
        std::vector<hipStream_t> out;
        for (const auto& e : list())
            if (e.second == device && e.first != own) out.push_back(e.first);
        return out;
    }

private:
    static std::mutex& mutex() { static std::mutex m; return m; }
    static std::vector<std::pair<hipStream_t, int>>& list() {
        static auto* l = new std::vector<std::pair<hipStream_t, int>>();
        return *l;
    }
};
struct DeviceBufferCache {
    struct Entry {
        void* p;
        size_t bytes;
        int device;
        // owner != nullptr: every use of the buffer was enqueued on this stream and no other queue exists that could have read
        // it — work enqueued on the SAME stream later needs no wait at all (round 5: the common case, one queue per process; an
        // event per released buffer was 2 us of host time, fifty times per loop of the reference's example)
        hipStream_t owner;
        std::vector<hipEvent_t> ready;  // otherwise: one event per stream that may still use it; empty: idle already
    };
    /// `stream`: the stream the caller will use the buffer on (nullptr: unknown — any stream, or the host).
    static void* acquire(size_t bytes, size_t* got, hipStream_t stream = nullptr) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::vector<hipEvent_t> wait_for;
        hipStream_t wait_stream = nullptr;
        void* taken = nullptr;
        {
            std::lock_guard<std::mutex> lock(mutex());
            auto& pool = buffers();
            size_t best = pool.size(), best_busy = pool.size();
            for (size_t i = 0; i < pool.size(); ++i) {
                Entry& e = pool[i];
                if (e.device != dev || e.bytes < bytes || e.bytes > 2 * bytes + 4096) continue;
                const bool free_now = (e.owner != nullptr && stream != nullptr && e.owner == stream) || settle(e);
                if (free_now) {
                    if (best == pool.size() || e.bytes < pool[best].bytes) best = i;
                } else if (best_busy == pool.size() || e.bytes < pool[best_busy].bytes) {
                    best_busy = i;  // its last user is still running
                }
            }
            // A buffer of the right size whose last user has not finished yet is still the better deal: waiting for that event
            // (a kernel or two of the caller's own stream, usually done by the time we look) costs microseconds, hipMalloc
            // 0.1-0.2 ms. A frame loop frees and re-creates its containers back to back: without this it allocated every time.
            const size_t pick = best != pool.size() ? best : best_busy;
            if (pick != pool.size()) {
                taken = pool[pick].p;
                *got = pool[pick].bytes;
                wait_for.swap(pool[pick].ready);
                if (pool[pick].owner != nullptr && pool[pick].owner != stream) wait_stream = pool[pick].owner;
                total() -= pool[pick].bytes;
                pool.erase(pool.begin() + (std::ptrdiff_t)pick);
            }
        }
        if (taken) {
            if (wait_stream) (void)hipStreamSynchronize(wait_stream);  // (another stream, or the host, takes over: wait for the owner's work)
            for (hipEvent_t ev : wait_for) {
                (void)hipEventSynchronize(ev);
                (void)hipEventDestroy(ev);
            }
            return taken;
        }
        void* p = nullptr;
        hip_check(hipMalloc(&p, bytes), "hipMalloc");
        *got = bytes;
        return p;
    }
    /// `stream`: every kernel / copy that touched the buffer was enqueued on it (or has completed). idle = true: the caller
    /// knows the device is done with the buffer (it synchronised).
    static void release(void* p, size_t bytes, hipStream_t stream, bool idle = false) {
        if (!p) return;
        Entry e{p, bytes, 0, nullptr, {}};
        (void)hipGetDevice(&e.device);
        if (!idle) {
            std::vector<hipStream_t> streams = QueueStreams::others(stream, e.device);
            if (streams.empty() && stream != nullptr && QueueStreams::is_live(stream)) {  // (a queue's own stream: its end is announced)
                e.owner = stream;  // one queue: stream order is all the protection the buffer needs
            } else {
                streams.insert(streams.begin(), stream);
                for (hipStream_t s : streams) {
                    hipEvent_t ev = nullptr;
                    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, s) != hipSuccess) {
                        if (ev) (void)hipEventDestroy(ev);
                        for (hipEvent_t done : e.ready) (void)hipEventDestroy(done);
                        e.ready.clear();
                        (void)hipGetLastError();
                        (void)hipDeviceSynchronize();  // no event to be had: wait, as hipFree would
                        break;
                    }
                    e.ready.push_back(ev);
                }
            }
        }
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> lock(mutex());
            auto& pool = buffers();
            pool.push_back(e);
            total() += bytes;
            for (size_t i = 0; i < pool.size() && (pool.size() > 48 || total() > (size_t(2) << 30));) {  // oldest settled first
                if (pool[i].device == e.device && settle(pool[i])) {
                    drop.push_back(pool[i].p);
                    total() -= pool[i].bytes;
                    pool.erase(pool.begin() + (std::ptrdiff_t)i);
                } else {
                    ++i;
                }
            }
        }
        for (void* d : drop) (void)hipFree(d);
    }
    /// A queue's stream is about to be destroyed: buffers tagged with it wait for it once and are idle from then on.
    static void forget_stream(hipStream_t s) {
        bool any = false;
        {
            std::lock_guard<std::mutex> lock(mutex());
            for (Entry& e : buffers()) any = any || e.owner == s;
        }
        if (!any) return;
        (void)hipStreamSynchronize(s);
        std::lock_guard<std::mutex> lock(mutex());
        for (Entry& e : buffers())
            if (e.owner == s) e.owner = nullptr;
    }

private:
    static bool settle(Entry& e) {  // true once nothing on the device uses the buffer any more
        if (e.owner != nullptr) {
            // (a stream that is no longer registered has been destroyed — which waited for its work; never query a dead handle)
            if (QueueStreams::is_live(e.owner) && hipStreamQuery(e.owner) != hipSuccess) { (void)hipGetLastError(); return false; }
            e.owner = nullptr;
        }
        while (!e.ready.empty()) {
            if (hipEventQuery(e.ready.back()) != hipSuccess) { (void)hipGetLastError(); return false; }
            (void)hipEventDestroy(e.ready.back());
            e.ready.pop_back();
        }
        return true;
    }
    static std::mutex& mutex() { static std::mutex m; return m; }
    static size_t& total() { static size_t t = 0; return t; }
    static std::vector<Entry>& buffers() {
        static auto* c = new std::vector<Entry>();  // never destroyed: the HIP runtime may be gone by then
        return *c;
    }
};
// A box known to hold every point of a container, remembered by whoever produced the container (voxel downsampling knows its
// voxels' key box) for whoever builds a spatial structure on it next (GridKNN::build -> sp_grid_create_bounded: no bounding-box
// kernel, no read-back). Keyed by the container's generation — process-wide unique and changed by every modification — so a
// stale entry can never be taken for a live one. A ring of 32: the hint is used within the frame that made it, or not at all.
struct BoundsHints {
    static void put(uint64_t generation, const float* min_max6) {
        std::lock_guard<std::mutex> lock(mutex());
        Entry& e = ring()[next()++ % kEntries];
        e.generation = generation;
        for (int a = 0; a < 6; ++a) e.b[a] = min_max6[a];
    }
    static bool get(uint64_t generation, float* min_max6) {
        std::lock_guard<std::mutex> lock(mutex());
        for (const Entry& e : ring())
            if (e.generation == generation && generation != 0) {
                for (int a = 0; a < 6; ++a) min_max6[a] = e.b[a];
                return true;
            }
        return false;
    }

private:
    static constexpr size_t kEntries = 32;
    struct Entry { uint64_t generation = 0; float b[6] = {0, 0, 0, 0, 0, 0}; };
    static std::mutex& mutex() { static std::mutex m; return m; }
    static std::array<Entry, kEntries>& ring() { static std::array<Entry, kEntries> r; return r; }
    static size_t& next() { static size_t n = 0; return n; }
};
// Host memory of large containers: PINNED blocks from a process-wide pool (round 5). A copy between pageable memory and the
// device is staged — by the runtime at 3-5 GB/s, or through StagedCopy's pinned buffers below at the host's memcpy speed: for
// the reference example's two 1.1 MB scans that memcpy was 45 us apiece, a tenth of the loop. A container whose host vector
// already lives in pinned memory is copied by the DMA engine directly. hipHostMalloc costs a millisecond, so blocks (powers of
// two from 256 KB) go back to a free list, never to the system; beyond kMaxTotal, or when pinned memory is not to be had (no
// device, limits), the allocator hands out ordinary heap memory and StagedCopy serves the copies as before.
struct PinnedPool {
    static constexpr size_t kMinBytes = size_t(256) << 10, kMaxTotal = size_t(2) << 30;
    static void* acquire(size_t bytes) {
        size_t cls = kMinBytes;
        while (cls < bytes) cls <<= 1;
        State& s = state();
        {
            std::lock_guard<std::mutex> lock(s.m);
            auto it = s.free.find(cls);
            if (it != s.free.end() && !it->second.empty()) {
                void* p = it->second.back();
                it->second.pop_back();
                s.live[p] = cls;
                return p;
            }
            if (s.failed || s.total + cls > kMaxTotal) return nullptr;
            s.total += cls;
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, cls, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            std::lock_guard<std::mutex> lock(s.m);
            s.total -= cls;
            s.failed = true;  // (no device, or the limit of locked memory: do not try again for every container)
            return nullptr;
        }
        std::lock_guard<std::mutex> lock(s.m);
        s.live[p] = cls;
        return p;
    }
    /// true: p was a pooled block and is back on the free list
    static bool release(void* p) {
        State& s = state();
        std::lock_guard<std::mutex> lock(s.m);
        auto it = s.live.find(p);
        if (it == s.live.end()) return false;
        s.free[it->second].push_back(p);
        s.live.erase(it);
        return true;
    }
    /// [p, p + bytes) lies inside a live pooled block
    static bool owns(const void* p, size_t bytes) {
        State& s = state();
        std::lock_guard<std::mutex> lock(s.m);
        auto it = s.live.upper_bound(const_cast<void*>(p));
        if (it == s.live.begin()) return false;
        --it;
        const char* b = static_cast<const char*>(it->first);
        return static_cast<const char*>(p) >= b && static_cast<const char*>(p) + bytes <= b + it->second;
    }

private:
    struct State {
        std::mutex m;
        std::map<void*, size_t> live;
        std::map<size_t, std::vector<void*>> free;
        size_t total = 0;
        bool failed = false;
    };
    static State& state() {
        static auto* s = new State();  // never destroyed: containers with static lifetime may outlive any static of ours
        return *s;
    }
};
/// std::allocator with PinnedPool behind allocations of 256 KB and more (the reference's shared_vector is a std::vector with a
/// USM allocator, utils/sycl_utils.hpp:630-635: a custom allocator is what its callers already see).
template <typename T>
struct host_allocator {
    using value_type = T;
    host_allocator() = default;
    template <class U> host_allocator(const host_allocator<U>&) {}
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes >= PinnedPool::kMinBytes)
            if (void* p = PinnedPool::acquire(bytes)) return static_cast<T*>(p);
        return static_cast<T*>(::operator new(bytes));
    }
    void deallocate(T* p, size_t n) {
        if (n * sizeof(T) >= PinnedPool::kMinBytes && PinnedPool::release(p)) return;
        ::operator delete(p);
    }
    template <class U> bool operator==(const host_allocator<U>&) const { return true; }
    template <class U> bool operator!=(const host_allocator<U>&) const { return false; }
};
// Large copies between a container's host vector and its HBM mirror go through two pinned 8 MB buffers (one filled /
// drained by the host while the other is in flight): a hipMemcpy from or to pageable memory is staged by the runtime at
// 3-5 GB/s on this stack — 16 MB of points took 3-5 ms to upload, an 80 MB neighbour list 25 ms to read back — where the
// DMA engine does 25+ GB/s from pinned memory and the host's own memcpy 10+. Process-wide, behind a mutex (a container's
// upload / download is a synchronous step anyway); small copies take the runtime's path.
struct StagedCopy {
    static constexpr size_t kChunk = size_t(8) << 20, kMin = size_t(1) << 20;
    static void h2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
        Buffers* b = (bytes >= kMin && !PinnedPool::owns(src, bytes)) ? buffers(st) : nullptr;  // (pinned already: the DMA engine reads it in place)
        if (!b) {
            hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st), "H2D");
            hip_check(hipStreamSynchronize(st), "H2D sync");
            return;
        }
        std::lock_guard<std::mutex> lock(b->m);
        size_t off = 0;
        for (int i = 0; off < bytes; i ^= 1) {
            const size_t len = std::min(kChunk, bytes - off);
            hip_check(hipEventSynchronize(b->ev[i]), "staging");  // the copy that last used this buffer has left it
            std::memcpy(b->p[i], static_cast<const char*>(src) + off, len);
            hip_check(hipMemcpyAsync(static_cast<char*>(dst) + off, b->p[i], len, hipMemcpyHostToDevice, st), "H2D");
            hip_check(hipEventRecord(b->ev[i], st), "staging");
            off += len;
        }
        hip_check(hipStreamSynchronize(st), "H2D sync");
    }
    /// (the caller has synchronised the stream the device data was produced on)
    static void d2h(void* dst, const void* src, size_t bytes, hipStream_t st) {
        Buffers* b = (bytes >= kMin && !PinnedPool::owns(dst, bytes)) ? buffers(st) : nullptr;
        if (!b) {
            hip_check(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), "D2H");
            return;
        }
        std::lock_guard<std::mutex> lock(b->m);
        const size_t chunks = (bytes + kChunk - 1) / kChunk;
        auto len_of = [&](size_t c) { return std::min(kChunk, bytes - c * kChunk); };
        auto issue = [&](size_t c) {
            hip_check(hipMemcpyAsync(b->p[c & 1], static_cast<const char*>(src) + c * kChunk, len_of(c), hipMemcpyDeviceToHost, st), "D2H");
            hip_check(hipEventRecord(b->ev[c & 1], st), "staging");
        };
        auto drain = [&](size_t c) {
            hip_check(hipEventSynchronize(b->ev[c & 1]), "staging");
            std::memcpy(static_cast<char*>(dst) + c * kChunk, b->p[c & 1], len_of(c));
        };
        // chunk c is in flight into one buffer while chunk c - 1 is copied out of the other
        issue(0);
        for (size_t c = 1; c < chunks; ++c) {
            issue(c);
            drain(c - 1);
        }
        drain(chunks - 1);
    }

private:
    struct Buffers {
        void* p[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        std::mutex m;
    };
    // One pair of pinned buffers AND events per device: an event belongs to the device that was current when it was
    // created, and recording it on a stream of another device is an error — a process with DeviceQueue(0) and DeviceQueue(1)
    // would have thrown on every large copy of the second one. nullptr when pinned memory is not to be had (or the device
    // ordinal is beyond the table): the runtime's own path is taken.
    static Buffers* buffers(hipStream_t st) {
        constexpr int kMaxDevices = 64;
        static Buffers* table[kMaxDevices] = {};
        static bool tried[kMaxDevices] = {};
        static std::mutex table_mutex;
        int dev = 0, current = 0;
        hipDevice_t sdev = 0;
        if (hipGetDevice(&current) != hipSuccess) return nullptr;
        dev = current;
        if (st != nullptr && hipStreamGetDevice(st, &sdev) == hipSuccess) dev = (int)sdev;  // the STREAM's device, not the current one
        else (void)hipGetLastError();
        if (dev < 0 || dev >= kMaxDevices) return nullptr;
        std::lock_guard<std::mutex> lock(table_mutex);
        if (!tried[dev]) {
            tried[dev] = true;
            auto* nb = new Buffers();  // never destroyed: the HIP runtime may be gone by then
            bool ok = dev == current || hipSetDevice(dev) == hipSuccess;  // (events are created on the current device)
            for (int i = 0; i < 2 && ok; ++i)
                ok = hipHostMalloc(&nb->p[i], kChunk, hipHostMallocPortable) == hipSuccess &&
                     hipEventCreateWithFlags(&nb->ev[i], hipEventDisableTiming) == hipSuccess;
            if (dev != current) (void)hipSetDevice(current);
            if (!ok) (void)hipGetLastError();
            table[dev] = ok ? nb : nullptr;
        }
        return table[dev];
    }
};
}  // namespace detail

namespace sycl_utils {

/// utils/sycl_utils.hpp:491-626 — queue handle. Here: a device ordinal and an in-order HIP stream.
struct DeviceQueue {
    using Ptr = std::shared_ptr<DeviceQueue>;
    struct StreamHolder {
        hipStream_t stream = nullptr;
        int device = 0;
        ~StreamHolder() {
            if (stream) {
                detail::QueueStreams::remove(stream);
                detail::DeviceBufferCache::forget_stream(stream);
                sp_stream_retired(stream);  // (the library's own pool tags buffers with the stream too)
                (void)hipStreamDestroy(stream);
            }
        }
    };
    std::shared_ptr<StreamHolder> ptr;  // the reference exposes `ptr` (a sycl::queue); kept as the stream holder

    DeviceQueue() : DeviceQueue(0) {}
    explicit DeviceQueue(int device) {
        ptr = std::make_shared<StreamHolder>();
        ptr->device = device;
        throw_on_error(sp_set_device(device));
        hip_check(hipStreamCreateWithFlags(&ptr->stream, hipStreamNonBlocking), "hipStreamCreate");
        detail::QueueStreams::add(ptr->stream, device);
    }
    hipStream_t stream() const { return ptr->stream; }
    void wait() const { hip_check(hipStreamSynchronize(ptr->stream), "hipStreamSynchronize"); }
    size_t get_work_group_size() const { return 256; }
    size_t get_global_size(size_t n) const { return (n + 255) / 256 * 256; }
    void print_device_info() const {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, ptr->device) == hipSuccess)
            printf("Device: %s (%s), %d CUs\n", p.name, p.gcnArchName, p.multiProcessorCount);
    }
    // mem-advise hooks of the reference (utils/sycl_utils.hpp:572-625): residency is explicit here, so they are no-ops.
    template <typename T> void set_accessed_by_device(const T*, size_t) const {}
    template <typename T> void clear_accessed_by_device(const T*, size_t) const {}
    template <typename T> void set_accessed_by_host(const T*, size_t) const {}
    template <typename T> void clear_accessed_by_host(const T*, size_t) const {}
};

/// MI355X extension: RAII handle of the library's RCCL communicator (sp_comm), one process per GPU. Rank 0
/// makes the 128-byte id (Communicator::unique_id) and hands it to the other ranks through the application's own channel
/// (MPI_Bcast, a file, ...); every rank then constructs Communicator(id, rank, world) — a collective call.
class Communicator {
public:
    static std::vector<unsigned char> unique_id() {
        std::vector<unsigned char> id(SP_COMM_ID_BYTES);
        throw_on_error(sp_comm_unique_id(id.data()));
        return id;
    }
    Communicator(const std::vector<unsigned char>& id, int rank, int world) {
        if (id.size() != SP_COMM_ID_BYTES) throw std::invalid_argument("[Communicator] the id must be SP_COMM_ID_BYTES bytes");
        throw_on_error(sp_comm_create(id.data(), rank, world, &h_));
    }
    ~Communicator() { sp_comm_destroy(h_); }
    Communicator(const Communicator&) = delete;
    Communicator& operator=(const Communicator&) = delete;
    sp_comm* handle() const { return h_; }
    int rank() const { return sp_comm_rank(h_); }
    int world() const { return sp_comm_world(h_); }

private:
    sp_comm* h_ = nullptr;
};

/// MI355X extension: RAII handle of the direct exchange (sp_xchg): every rank owns a small slot buffer the
/// other ranks map through hipIpc and store their per-iteration 128-byte row into — no collective launch per iteration.
/// Construct on every rank, pass `handle_bytes()` to all ranks through the application's own channel, then `connect()` with
/// the world's handles concatenated in rank order. A peer's row that does not arrive within `set_timeout_ms` (2 s) ends the
/// alignment with SP_ERR_RUNTIME instead of hanging.
class Exchange {
public:
    Exchange(int rank, int world) { throw_on_error(sp_xchg_create(rank, world, &h_)); }
    ~Exchange() { sp_xchg_destroy(h_); }
    Exchange(const Exchange&) = delete;
    Exchange& operator=(const Exchange&) = delete;
    std::vector<unsigned char> handle_bytes() const {
        std::vector<unsigned char> b(SP_XCHG_HANDLE_BYTES);
        throw_on_error(sp_xchg_handle(h_, b.data()));
        return b;
    }
    void connect(const std::vector<unsigned char>& all_handles_in_rank_order) {
        if (all_handles_in_rank_order.size() != size_t(SP_XCHG_HANDLE_BYTES) * size_t(world()))
            throw std::invalid_argument("[Exchange] connect needs world x SP_XCHG_HANDLE_BYTES bytes");
        throw_on_error(sp_xchg_connect(h_, all_handles_in_rank_order.data()));
    }
    void set_timeout_ms(unsigned ms) { throw_on_error(sp_xchg_set_timeout_ms(h_, ms)); }
    sp_xchg* handle() const { return h_; }
    int rank() const { return sp_xchg_rank(h_); }
    int world() const { return sp_xchg_world(h_); }

private:
    sp_xchg* h_ = nullptr;
};

/// The reference's `sycl::event`: here the stream the work was enqueued on (all work of a cloud shares one in-order
/// stream, so "depends on these events" is already implied by enqueue order; the type exists for source compatibility).
struct event {
    hipStream_t stream = nullptr;
};
/// utils/sycl_utils.hpp:234-280 — a set of events.
struct events {
    std::vector<event> evs;
    events() = default;
    explicit events(hipStream_t s) { evs.push_back(event{s}); }
    events& operator+=(const events& o) { evs.insert(evs.end(), o.evs.begin(), o.evs.end()); return *this; }
    void wait() const { for (const auto& e : evs) hip_check(hipStreamSynchronize(e.stream), "hipStreamSynchronize"); }
    void wait_and_throw() const { wait(); hip_check(hipGetLastError(), "device error"); }
};

}  // namespace sycl_utils

// ---------------------------------------------------------------------------------------------- shared_vector
/// utils/sycl_utils.hpp:630-635. std::vector semantics on the host + an HBM mirror.
///
/// Which copy is current is tracked per container, not per element: every NON-CONST accessor (operator[], at, data, begin /
/// end) must assume the caller writes and marks the host copy newer, so the next device use uploads the whole vector again.
/// Read through a const reference, `host()` or `std::as_const(v)` when only reading — nothing is marked then — and write
/// in bulk where possible. (The reference's USM-shared vector pays per page instead; with XNACK off on this part managed
/// memory would be served over PCIe)
template <typename T>
class shared_vector {
public:
    using value_type = T;
    using host_vector = std::vector<T, detail::host_allocator<T>>;
    using iterator = typename host_vector::iterator;
    using const_iterator = typename host_vector::const_iterator;

    shared_vector() = default;
    explicit shared_vector(const sycl_utils::DeviceQueue& q) : queue_(q.ptr) {}
    shared_vector(size_t n, const sycl_utils::DeviceQueue& q) : host_(n), queue_(q.ptr) {}
    shared_vector(size_t n, const T& v, const sycl_utils::DeviceQueue& q) : host_(n, v), queue_(q.ptr) {}
    // the reference passes `*queue.ptr` (a sycl::queue) to the allocator; accept the stream holder the same way
    shared_vector(size_t n, const sycl_utils::DeviceQueue::StreamHolder& h) : host_(n) { bind(h); }
    shared_vector(size_t n, const T& v, const sycl_utils::DeviceQueue::StreamHolder& h) : host_(n, v) { bind(h); }
    shared_vector(const shared_vector& o) : host_(o.host()), queue_(o.queue_), stream_(o.stream_) {}
    shared_vector& operator=(const shared_vector& o) {
        if (this != &o) { wait_upload(); host_ = o.host(); host_dirty_ = true; dev_dirty_ = false; generation_ = next_generation(); }
        return *this;
    }
    ~shared_vector() {
        wait_upload();  // (the host vector's pinned block goes back to its pool: no DMA may still be reading it)
        if (up_ev_) (void)hipEventDestroy(up_ev_);
        if (dev_) detail::DeviceBufferCache::release(dev_, dev_bytes_, stream());
    }

    // ---- host side (std::vector surface)
    size_t size() const { return size_override_ ? dev_size_ : host_.size(); }
    bool empty() const { return size() == 0; }
    void resize(size_t n) { sync_host(); host_.resize(n); touch(); }
    void resize(size_t n, const T& v) { sync_host(); host_.resize(n, v); touch(); }
    void reserve(size_t n) { wait_upload(); host_.reserve(n); }
    void clear() { wait_upload(); size_override_ = false; dev_dirty_ = false; host_.clear(); touch(); }
    void assign(size_t n, const T& v) { wait_upload(); size_override_ = false; dev_dirty_ = false; host_.assign(n, v); touch(); }
    /// the n elements at p become the contents (one pass over fresh memory; resize + copy would touch every page twice)
    void assign(const T* p, size_t n) { wait_upload(); size_override_ = false; dev_dirty_ = false; host_.assign(p, p + n); touch(); }
    void push_back(const T& v) { sync_host(); host_.push_back(v); touch(); }
    template <class... A> void emplace_back(A&&... a) { sync_host(); host_.emplace_back(std::forward<A>(a)...); touch(); }
    T& operator[](size_t i) { sync_host(); touch(); return host_[i]; }
    const T& operator[](size_t i) const { sync_host(); return host_[i]; }
    T& at(size_t i) { sync_host(); touch(); return host_.at(i); }
    const T& at(size_t i) const { sync_host(); return host_.at(i); }
    T* data() { sync_host(); touch(); return host_.data(); }
    const T* data() const { sync_host(); return host_.data(); }
    iterator begin() { sync_host(); touch(); return host_.begin(); }
    iterator end() { sync_host(); touch(); return host_.end(); }
    const_iterator begin() const { sync_host(); return host_.begin(); }
    const_iterator end() const { sync_host(); return host_.end(); }
    const host_vector& host() const { sync_host(); return host_; }
    /// std::vector::insert / erase on the host copy (PointCloudShared::extend / erase, points/point_cloud.hpp:319-368)
    template <class It>
    iterator insert(const_iterator pos, It first, It last) {
        sync_host();
        touch();
        return host_.insert(pos, first, last);
    }
    iterator erase(const_iterator first, const_iterator last) {
        sync_host();
        touch();
        return host_.erase(first, last);
    }
    /// the elements of `o` appended (its host copy is brought up to date first)
    void append(const shared_vector& o) {
        const host_vector& src = o.host();
        sync_host();
        host_.insert(host_.end(), src.begin(), src.end());
        touch();
    }

    // ---- device side (what the kernels get)
    /// Read-only device pointer; uploads first if the host copy is newer.
    const T* device_data() const { sync_device(); return dev_; }
    /// For a kernel that reads the vector ONCE, front to back (the box filter over a fresh scan): the device copy when it is
    /// current, else the pinned host copy itself (*in_place = true) — pinned blocks are mapped into the device's address
    /// space, the kernel streams them over PCIe at the rate the DMA engine would have copied them, and there is no copy to
    /// wait for and no device buffer for a cloud that is dropped right after. The caller launches its kernel and then calls
    /// host_read_enqueued(that stream): the host copy may not change (or go back to its pool) before that launch has finished.
    const T* device_readable_once(bool* in_place) const {
        *in_place = false;
        if (!dev_dirty_ && (host_dirty_ || dev_ == nullptr || dev_size_ != host_.size()) && !host_.empty() &&
            detail::PinnedPool::owns(host_.data(), host_.size() * sizeof(T))) {
            wait_upload();
            *in_place = true;
            return host_.data();
        }
        return device_data();
    }
    void host_read_enqueued(hipStream_t launched_on) const {
        if (up_ev_ == nullptr) hip_check(hipEventCreateWithFlags(&up_ev_, hipEventDisableTiming), "event");
        hip_check(hipEventRecord(up_ev_, launched_on), "event");
        up_pending_ = true;
    }
    /// Writable device pointer for `n` elements (kernel output): the host copy becomes stale, nothing is uploaded.
    T* device_data_for_write(size_t n) {
        ensure_capacity(n);
        dev_size_ = n; size_override_ = true; dev_dirty_ = true; host_dirty_ = false;
        generation_ = next_generation();
        return dev_;
    }
    /// `n` elements that live on the device only for now — a kernel is about to write all of them (device_data_for_write(n)),
    /// or `fill_bits` (a 32-bit pattern, for 4-byte T) is put there by a device fill. No host storage is touched until the host
    /// reads the vector: a 20 M-entry neighbour list is 80 MB that std::vector would allocate and initialise for nothing
    /// (25 ms per list on the host clock of the config-4 harness, examples/bench_registration.cpp).
    void resize_on_device(size_t n, const uint32_t* fill_bits = nullptr) {
        static_assert(sizeof(T) % 4 == 0, "device fill works on 32-bit words");
        wait_upload();
        host_vector().swap(host_);
        ensure_capacity(n);
        if (fill_bits && n) hip_check(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dev_), (int)*fill_bits, n * (sizeof(T) / 4), stream()), "fill");
        dev_size_ = n; size_override_ = true; dev_dirty_ = true; host_dirty_ = false;
        generation_ = next_generation();
    }
    /// Read-write device pointer (in-place kernels).
    T* device_data_rw() { sync_device(); dev_dirty_ = true; dev_size_ = host_.size(); size_override_ = true; generation_ = next_generation(); return dev_; }
    /// After a kernel produced fewer rows than reserved (compaction, downsampling).
    void set_device_size(size_t n) { dev_size_ = n; size_override_ = true; dev_dirty_ = true; generation_ = next_generation(); }
    /// Changes whenever the contents may have changed (any non-const access counts): a structure built on the container can
    /// tell whether it still describes it.
    uint64_t generation() const { return generation_; }
    hipStream_t stream() const { return queue_ ? queue_->stream : stream_; }

private:
    void touch() { host_dirty_ = true; generation_ = next_generation(); }
    void bind(const sycl_utils::DeviceQueue::StreamHolder& h) { stream_ = h.stream; }
    void ensure_capacity(size_t n) const {
        if (n <= dev_cap_) return;
        size_t got = 0;
        T* nd = static_cast<T*>(detail::DeviceBufferCache::acquire(std::max<size_t>(n, 1) * sizeof(T), &got, stream()));
        if (dev_) {
            if (dev_dirty_ && dev_size_)  // (in stream order behind the kernels that wrote the old buffer)
                hip_check(hipMemcpyAsync(nd, dev_, std::min(dev_size_, n) * sizeof(T), hipMemcpyDeviceToDevice, stream()), "hipMemcpy");
            detail::DeviceBufferCache::release(dev_, dev_bytes_, stream());
        }
        dev_ = nd;
        dev_bytes_ = got;
        dev_cap_ = got / sizeof(T);
    }
    void sync_device() const {
        if (dev_dirty_) return;  // device is the newest copy
        if (host_dirty_ || dev_ == nullptr || dev_size_ != host_.size()) {
            ensure_capacity(host_.size());
            const size_t bytes = host_.size() * sizeof(T);
            if (!host_.empty() && detail::PinnedPool::owns(host_.data(), bytes)) {
                // A pinned host vector: the DMA engine reads it in place and the host does NOT wait for it — kernels enqueued next
                // are ordered behind the copy by the stream, and whatever writes the host vector (or frees it) waits for the
                // copy's event first (wait_upload). The wait this replaces was the 25 us the DMA takes for a 1 MB scan, during
                // which the host could already be enqueuing the scan's first kernels.
                hip_check(hipMemcpyAsync(dev_, host_.data(), bytes, hipMemcpyHostToDevice, stream()), "H2D");
                if (detail::QueueStreams::alone(stream())) {
                    if (up_ev_ == nullptr) hip_check(hipEventCreateWithFlags(&up_ev_, hipEventDisableTiming), "event");
                    hip_check(hipEventRecord(up_ev_, stream()), "event");
                    up_pending_ = true;
                } else {
                    // (another queue exists: its kernels may read this container and are not ordered behind a copy on this
                    // queue's stream — the copy is waited for, as every upload was before round 5)
                    hip_check(hipStreamSynchronize(stream()), "H2D sync");
                }
            } else if (!host_.empty()) {
                detail::StagedCopy::h2d(dev_, host_.data(), bytes, stream());  // (synchronous: the host vector may be modified right after)
            } else {
                hip_check(hipStreamSynchronize(stream()), "H2D sync");
            }
            dev_size_ = host_.size();
            host_dirty_ = false;
        }
    }
    /// An upload of the (pinned) host vector may still be in flight: whoever is about to write or free the host vector waits.
    void wait_upload() const {
        if (!up_pending_) return;
        hip_check(hipEventSynchronize(up_ev_), "upload");
        up_pending_ = false;
    }
    void sync_host() const {
        wait_upload();
        if (!dev_dirty_) return;
        hip_check(hipStreamSynchronize(stream()), "sync");
        host_.resize(dev_size_);
        if (dev_size_) detail::StagedCopy::d2h(host_.data(), dev_, dev_size_ * sizeof(T), stream());
        dev_dirty_ = false;
        size_override_ = false;
        host_dirty_ = false;
    }

    mutable host_vector host_;
    mutable hipEvent_t up_ev_ = nullptr;  // behind the latest upload of a pinned host vector
    mutable bool up_pending_ = false;
    mutable T* dev_ = nullptr;
    mutable size_t dev_cap_ = 0, dev_size_ = 0, dev_bytes_ = 0;
    mutable bool host_dirty_ = true, dev_dirty_ = false, size_override_ = false;
    // Drawn from ONE process-wide counter: a cache keyed by (container address, generation) — Registration's
    // prepared target rows — must not match a DIFFERENT container that the allocator placed at a recycled address; with
    // per-container counts starting at 0 two containers could agree on both.
    static uint64_t next_generation() {
        static std::atomic<uint64_t> counter{1};
        return counter.fetch_add(1, std::memory_order_relaxed);
    }
    uint64_t generation_ = next_generation();
    std::shared_ptr<sycl_utils::DeviceQueue::StreamHolder> queue_;
    hipStream_t stream_ = nullptr;
};
template <typename T>
using shared_vector_ptr = std::shared_ptr<shared_vector<T>>;

// ---------------------------------------------------------------------------------------------- points/types.hpp
using PointType = Eigen::Vector4f;
using Covariance = Eigen::Matrix4f;
using Normal = Eigen::Vector4f;
using RGBType = Eigen::Vector4f;
using TransformMatrix = Eigen::Matrix4f;
using TimestampOffset = float;
using PointContainerCPU = std::vector<PointType, Eigen::aligned_allocator<PointType>>;
using CovarianceContainerCPU = std::vector<Covariance, Eigen::aligned_allocator<Covariance>>;
using NormalContainerCPU = std::vector<Normal, Eigen::aligned_allocator<Normal>>;
using RGBContainerCPU = std::vector<RGBType, Eigen::aligned_allocator<RGBType>>;
using IntensityContainerCPU = std::vector<float>;
using TimestampContainerCPU = std::vector<TimestampOffset>;
using PointContainerShared = shared_vector<PointType>;
using CovarianceContainerShared = shared_vector<Covariance>;
using NormalContainerShared = shared_vector<Normal>;
using RGBContainerShared = shared_vector<RGBType>;
using IntensityContainerShared = shared_vector<float>;
using TimestampContainerShared = shared_vector<TimestampOffset>;
static_assert(sizeof(PointType) == 16 && sizeof(Covariance) == 64, "API layouts: 16-byte points, 64-byte covariances");

// ---------------------------------------------------------------------------------------------- points/point_cloud.hpp
/// points/point_cloud.hpp:12-70
struct PointCloudCPU {
    using Ptr = std::shared_ptr<PointCloudCPU>;
    std::shared_ptr<PointContainerCPU> points = std::make_shared<PointContainerCPU>();
    std::shared_ptr<CovarianceContainerCPU> covs = std::make_shared<CovarianceContainerCPU>();
    std::shared_ptr<NormalContainerCPU> normals = std::make_shared<NormalContainerCPU>();
    std::shared_ptr<RGBContainerCPU> rgb = std::make_shared<RGBContainerCPU>();
    std::shared_ptr<IntensityContainerCPU> intensities = std::make_shared<IntensityContainerCPU>();
    std::shared_ptr<TimestampContainerCPU> timestamp_offsets = std::make_shared<TimestampContainerCPU>();
    double start_time_ms = 0.0, end_time_ms = 0.0;
    size_t size() const { return points->size(); }
    bool has_cov() const { return covs && covs->size() == points->size(); }
    bool has_normal() const { return normals && normals->size() == points->size(); }
    bool has_rgb() const { return rgb && rgb->size() == points->size(); }
    bool has_intensity() const { return intensities && intensities->size() == points->size(); }
    bool has_timestamps() const { return timestamp_offsets && timestamp_offsets->size() == points->size() && !timestamp_offsets->empty(); }
};

/// points/point_cloud.hpp:73-476 — six attribute containers + the queue; deep copy constructor, shallow assignment.
struct PointCloudShared {
    using Ptr = std::shared_ptr<PointCloudShared>;
    sycl_utils::DeviceQueue queue;
    std::shared_ptr<PointContainerShared> points;
    std::shared_ptr<CovarianceContainerShared> covs;
    std::shared_ptr<NormalContainerShared> normals;
    std::shared_ptr<RGBContainerShared> rgb;
    std::shared_ptr<IntensityContainerShared> intensities;
    std::shared_ptr<TimestampContainerShared> timestamp_offsets;
    double start_time_ms = 0.0, end_time_ms = 0.0;

    explicit PointCloudShared(const sycl_utils::DeviceQueue& q) : queue(q) { alloc(); }
    PointCloudShared(const sycl_utils::DeviceQueue& q, const PointCloudCPU& cpu) : queue(q) {
        alloc();
        copy_in(*points, *cpu.points);
        if (cpu.has_cov()) copy_in(*covs, *cpu.covs);
        if (cpu.has_normal()) copy_in(*normals, *cpu.normals);
        if (cpu.has_rgb()) copy_in(*rgb, *cpu.rgb);
        if (cpu.has_intensity()) copy_in(*intensities, *cpu.intensities);
        if (cpu.has_timestamps()) copy_in(*timestamp_offsets, *cpu.timestamp_offsets);
        start_time_ms = cpu.start_time_ms;
        end_time_ms = cpu.end_time_ms;
    }
    PointCloudShared(const PointCloudShared& o) : queue(o.queue), start_time_ms(o.start_time_ms), end_time_ms(o.end_time_ms) {
        points = std::make_shared<PointContainerShared>(*o.points);  // deep copy (point_cloud.hpp:202-236)
        covs = std::make_shared<CovarianceContainerShared>(*o.covs);
        normals = std::make_shared<NormalContainerShared>(*o.normals);
        rgb = std::make_shared<RGBContainerShared>(*o.rgb);
        intensities = std::make_shared<IntensityContainerShared>(*o.intensities);
        timestamp_offsets = std::make_shared<TimestampContainerShared>(*o.timestamp_offsets);
    }
    PointCloudShared& operator=(const PointCloudShared&) = default;  // shallow, as the reference's implicit operator=

    size_t size() const { return points->size(); }
    bool has_cov() const { return covs && covs->size() == points->size() && points->size() > 0; }
    bool has_normal() const { return normals && normals->size() == points->size() && points->size() > 0; }
    bool has_rgb() const { return rgb && rgb->size() == points->size() && points->size() > 0; }
    bool has_intensity() const { return intensities && intensities->size() == points->size() && points->size() > 0; }
    bool has_timestamps() const { return timestamp_offsets && timestamp_offsets->size() == points->size() && points->size() > 0; }
    PointType* points_ptr() const { return points->data(); }
    Covariance* covs_ptr() const { return covs->data(); }
    Normal* normals_ptr() const { return normals->data(); }
    void resize_points(size_t n) const { points->resize(n); }
    void resize_covs(size_t n) const { covs->resize(n); }
    void resize_normals(size_t n) const { normals->resize(n); }
    void resize_rgb(size_t n) const { rgb->resize(n); }
    void resize_intensities(size_t n) const { intensities->resize(n); }
    void resize_timestamps(size_t n) const { timestamp_offsets->resize(n); }
    void reserve_points(size_t n) const { points->reserve(n); }
    void reserve_covs(size_t n) const { covs->reserve(n); }
    void reserve_normals(size_t n) const { normals->reserve(n); }
    void reserve_rgb(size_t n) const { rgb->reserve(n); }
    void reserve_intensities(size_t n) const { intensities->reserve(n); }
    void reserve_timestamps(size_t n) const { timestamp_offsets->reserve(n); }
    /// points/point_cloud.hpp:307-317 (the timestamp base goes with the offsets)
    void clear() {
        points->clear(); covs->clear(); normals->clear(); rgb->clear(); intensities->clear(); timestamp_offsets->clear();
        start_time_ms = 0.0; end_time_ms = 0.0;
    }
    /// points/point_cloud.hpp:319-338: the points of `other` appended; an attribute survives when both clouds have it
    void extend(const PointCloudShared& other) {
        const size_t org_size = size();
        const bool cov = has_cov() && other.has_cov(), nrm = has_normal() && other.has_normal(), col = has_rgb() && other.has_rgb(),
                   inten = has_intensity() && other.has_intensity();
        if (cov) covs->append(*other.covs);
        if (nrm) normals->append(*other.normals);
        if (col) rgb->append(*other.rgb);
        if (inten) intensities->append(*other.intensities);
        points->append(*other.points);
        merge_timestamp_offsets(other, org_size);
    }
    /// points/point_cloud.hpp:340-366: the points [start_idx, end_idx) and their attributes removed
    void erase(size_t start_idx, size_t end_idx) {
        auto cut = [&](auto& v) { v->erase(std::as_const(*v).begin() + (std::ptrdiff_t)start_idx, std::as_const(*v).begin() + (std::ptrdiff_t)end_idx); };
        const bool cov = has_cov(), nrm = has_normal(), col = has_rgb(), inten = has_intensity(), ts = has_timestamps();
        if (cov) cut(covs);
        if (nrm) cut(normals);
        if (col) cut(rgb);
        if (inten) cut(intensities);
        if (ts) {
            cut(timestamp_offsets);
            if (timestamp_offsets->empty()) {
                start_time_ms = 0.0;
                end_time_ms = 0.0;
            } else {
                const auto& h = timestamp_offsets->host();
                end_time_ms = start_time_ms + static_cast<double>(*std::max_element(h.begin(), h.end()));
            }
        }
        cut(points);
    }
    void operator+=(const PointCloudShared& pc) { extend(pc); }

private:
    /// points/point_cloud.hpp:393-397
    void invalidate_timestamps() {
        timestamp_offsets->clear();
        start_time_ms = 0.0;
        end_time_ms = 0.0;
    }
    /// points/point_cloud.hpp:399-423: move the timestamp base to an earlier time (later: offsets would go negative)
    void shift_timestamp_base(double new_start_time_ms) {
        if (!has_timestamps() || new_start_time_ms >= start_time_ms) {
            if (new_start_time_ms > start_time_ms) invalidate_timestamps();
            return;
        }
        const double delta_ms = start_time_ms - new_start_time_ms;
        const double max_value = static_cast<double>(std::numeric_limits<TimestampOffset>::max());
        for (auto& offset : *timestamp_offsets) {
            const double adjusted = static_cast<double>(offset) + delta_ms;
            if (adjusted > max_value)
                throw std::runtime_error("[PointCloudShared::shift_timestamp_base] Timestamp offset overflow while shifting base");
            offset = static_cast<TimestampOffset>(adjusted);
        }
        start_time_ms = new_start_time_ms;
    }
    /// points/point_cloud.hpp:425-474, statement for statement. extend() calls it AFTER the points have been appended, as the
    /// reference does, so `has_timestamps()` (offsets as many as points) is false for this cloud whenever `other` is not
    /// empty: the merged cloud keeps timestamps only when this cloud was empty (it adopts the other's); two timestamped
    /// clouds merge into one without valid timestamps. That is the reference's behaviour and is kept.
    void merge_timestamp_offsets(const PointCloudShared& other, size_t original_size) {
        if (other.size() == 0) return;
        if (!other.has_timestamps()) {
            if (has_timestamps()) invalidate_timestamps();
            return;
        }
        if (!has_timestamps()) {
            if (original_size == 0) {  // this cloud was empty: adopt the other cloud's timestamps
                timestamp_offsets->append(*other.timestamp_offsets);
                start_time_ms = other.start_time_ms;
                end_time_ms = other.end_time_ms;
            }
            return;  // points without (valid) timestamps: the other cloud's are dropped
        }
        const double new_start_ms = std::min(start_time_ms, other.start_time_ms);
        if (new_start_ms < start_time_ms) shift_timestamp_base(new_start_ms);
        const double base_delta_ms = other.start_time_ms - new_start_ms;
        const double max_value = static_cast<double>(std::numeric_limits<TimestampOffset>::max());
        if (base_delta_ms > max_value)
            throw std::runtime_error("[PointCloudShared::merge_timestamp_offsets] Timestamp base delta exceeds representable offset range");
        const auto& src = other.timestamp_offsets->host();
        std::vector<TimestampOffset> shifted;
        shifted.reserve(src.size());
        for (const auto offset : src) {
            const double adjusted = static_cast<double>(offset) + base_delta_ms;
            if (adjusted > max_value)
                throw std::runtime_error("[PointCloudShared::merge_timestamp_offsets] Timestamp offset overflow while merging clouds");
            shifted.push_back(static_cast<TimestampOffset>(adjusted));
        }
        timestamp_offsets->insert(std::as_const(*timestamp_offsets).end(), shifted.begin(), shifted.end());
        start_time_ms = new_start_ms;
        end_time_ms = std::max(end_time_ms, other.end_time_ms);
    }

public:

    // device views for the kernels
    const float* points_device() const { return reinterpret_cast<const float*>(points->device_data()); }
    const float* covs_device() const { return has_cov() ? reinterpret_cast<const float*>(covs->device_data()) : nullptr; }
    const float* normals_device() const { return has_normal() ? reinterpret_cast<const float*>(normals->device_data()) : nullptr; }

private:
    void alloc() {
        points = std::make_shared<PointContainerShared>(queue);
        covs = std::make_shared<CovarianceContainerShared>(queue);
        normals = std::make_shared<NormalContainerShared>(queue);
        rgb = std::make_shared<RGBContainerShared>(queue);
        intensities = std::make_shared<IntensityContainerShared>(queue);
        timestamp_offsets = std::make_shared<TimestampContainerShared>(queue);
    }
    template <class S, class V>
    static void copy_in(S& dst, const V& src) {
        static_assert(sizeof(dst[0]) == sizeof(src[0]), "same element layout");
        dst.assign(reinterpret_cast<const typename S::value_type*>(src.data()), src.size());
    }
};

}  // namespace sycl_points

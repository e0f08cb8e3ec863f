// C-ABI housekeeping: error message slot, device selection, host twins of the solver arithmetic.
#include <mutex>
#include <string>
#include <vector>

#include "sp_common.h"
#include "sp_math.h"

namespace {
thread_local std::string g_last_error;
}

void sp_set_error(const char* msg) { g_last_error = msg ? msg : ""; }

namespace sp {
namespace {
std::once_flag g_err_once;
unsigned* g_err_host = nullptr;  // pinned, mapped
unsigned* g_err_dev = nullptr;
}  // namespace
void* pinned_mailbox() {
    // 256 bytes of pinned host memory per host thread, for the few words a build reads back (bounding box, counts): a copy into
    // pageable memory is staged inside the runtime call (20 us for 24 bytes on this stack); nullptr when pinned memory is not to
    // be had (the caller then reads into its own variable). Never freed: the HIP runtime may be gone when a thread ends.
    thread_local void* p = [] {
        void* q = nullptr;
        if (hipHostMalloc(&q, 256, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); q = nullptr; }
        return q;
    }();
    return p;
}
unsigned* device_error_word() {
    std::call_once(g_err_once, [] {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return; }  // (kernels on any device of the process raise it)
        *static_cast<volatile unsigned*>(h) = 0u;
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
        g_err_host = static_cast<unsigned*>(h);
        g_err_dev = static_cast<unsigned*>(d);
    });
    return g_err_dev;
}
int launch_status() {
    const int rc = hip_status(hipGetLastError());
    if (g_err_host) {
        const unsigned code = *static_cast<volatile unsigned*>(g_err_host);
        if (code != 0u) {
            *static_cast<volatile unsigned*>(g_err_host) = 0u;
            sp_set_error(code == kDevErrLookback
                             ? "a kernel of an EARLIER call gave up waiting for another workgroup's prefix (exclusive scan look-back): "
                               "the offsets that call produced (compaction, removal, grid units) are wrong"
                             : code == kDevErrBounds
                                   ? "a kernel of an EARLIER call (sp_grid_create_bounded) met a finite point outside the bounds its caller "
                                     "vouched for: that grid's searches are not exact"
                                   : "a kernel of an earlier call reported a device-side failure");
            return SP_ERR_HIP;
        }
    }
    return rc;
}
}  // namespace sp

namespace sp {
namespace {
struct PoolEntry {
    void* p; size_t bytes; int device; bool busy;
    std::vector<hipEvent_t> pending;  // work that may still touch p (scratch_release_after); empty: idle
    // released behind the work of exactly ONE stream: no event is recorded (2 us of host time per buffer) — a caller that will
    // use the buffer on the same stream takes it at once (stream order), anyone else once the stream has drained
    hipStream_t owner = nullptr;
};
// true once every pending event has completed (they are destroyed then) / the owner's stream has drained; never blocks
bool entry_idle(PoolEntry& e) {
    // (an owner's stream is never queried: the caller may have destroyed it since — a query of a dead handle crashes inside the
    // runtime. Such an entry waits for its own stream to ask again, for sp_stream_retired, or for the sweep in scratch_acquire)
    if (e.owner != nullptr) return false;
    while (!e.pending.empty()) {
        if (hipEventQuery(e.pending.back()) != hipSuccess) { (void)hipGetLastError(); return false; }
        (void)hipEventDestroy(e.pending.back());
        e.pending.pop_back();
    }
    return true;
}
std::mutex g_pool_mutex;
std::vector<PoolEntry> g_pool;
std::vector<void*> g_to_free;  // surplus buffers: freed by the next scratch_acquire (an allocating call), never by a release
constexpr size_t kKeep = 24;
constexpr size_t kForeignMax = 32;
constexpr size_t kKeepBytes = size_t(4) << 30;  // idle bytes kept per device
}  // namespace

namespace {
__global__ void zero_words_kernel(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ __launch_bounds__(256) void zero_quads_kernel(uint4* p, size_t n) {  // 16 bytes per lane: the streaming-store rate
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4(0u, 0u, 0u, 0u);
}
}  // namespace
int zero_async(void* p, size_t bytes, hipStream_t st) {
    if (!p || bytes == 0) return SP_OK;
    size_t n = (bytes + 3) / 4;
    uint32_t* w = static_cast<uint32_t*>(p);
    if (n >= 4096 && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {  // the bulk as 16-byte stores, the tail as words
        const size_t q = n / 4;
        const unsigned blocks = (unsigned)((q + 255) / 256 > 2048 ? 2048 : (q + 255) / 256);
        zero_quads_kernel<<<blocks, 256, 0, st>>>(static_cast<uint4*>(p), q);
        w += 4 * q;
        n -= 4 * q;
        if (n == 0) return launch_status();
    }
    const unsigned blocks = n <= 256 ? 1u : (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    zero_words_kernel<<<blocks, n <= 64 ? 64 : 256, 0, st>>>(w, n);
    return launch_status();
}

hipError_t scratch_acquire(void** ptr, size_t bytes, hipStream_t st) {
    *ptr = nullptr;
    if (bytes < 256) bytes = 256;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::vector<void*> surplus;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        surplus.swap(g_to_free);
    }
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        // Settle every released buffer of this device whose events have completed (an entry that is never the best fit would
        // otherwise keep its events, and the pool would grow past kKeep with nothing known to be idle), then trim the idle ones
        // to kKeep buffers / kKeepBytes: smallest first, freed just below (this call allocates and may synchronise anyway).
        size_t idle = 0, idle_bytes = 0;
        if (g_pool.size() > kKeep / 2)  // (a handful of entries: nothing to trim, and the best-fit scan below settles what it looks at)
            for (auto& e : g_pool)
                if (!e.busy && e.device == dev && entry_idle(e)) { ++idle; idle_bytes += e.bytes; }
        while (idle > kKeep || idle_bytes > kKeepBytes) {
            size_t k = g_pool.size();
            for (size_t i = 0; i < g_pool.size(); ++i) {
                const PoolEntry& e = g_pool[i];
                if (e.busy || e.device != dev || !e.pending.empty() || e.owner != nullptr) continue;
                const bool smaller = k == g_pool.size() || e.bytes < g_pool[k].bytes;
                // over the byte budget only: drop the LARGEST idle buffer (the smallest would not bring the total down)
                const bool larger = k == g_pool.size() || e.bytes > g_pool[k].bytes;
                if (idle > kKeep ? smaller : larger) k = i;
            }
            if (k == g_pool.size()) break;
            surplus.push_back(g_pool[k].p);
            --idle;
            idle_bytes -= g_pool[k].bytes;
            g_pool.erase(g_pool.begin() + (long)k);
        }
        // Buffers still tagged with a stream other than the caller's: normal for a second queue at work, but a caller that destroys
        // its streams without sp_stream_retired would strand them for good. Past kForeignMax of them: one device-wide wait, after
        // which every tag is void (nothing on the device can touch any pooled buffer any more).
        size_t foreign = 0;
        for (const auto& e : g_pool) foreign += (!e.busy && e.owner != nullptr && e.owner != st) ? 1 : 0;
        if (foreign > kForeignMax) {
            (void)hipDeviceSynchronize();
            for (auto& e : g_pool)
                if (!e.busy) e.owner = nullptr;
        }
        PoolEntry* best = nullptr;
        for (auto& e : g_pool)  // best fit among the idle buffers of this device, not more than 4x oversized
            if (!e.busy && e.device == dev && e.bytes >= bytes && e.bytes <= 4 * bytes && (!best || e.bytes < best->bytes) &&
                ((e.owner != nullptr && st != nullptr && e.owner == st) || entry_idle(e)))
                best = &e;
        if (best) { best->busy = true; best->owner = nullptr; *ptr = best->p; }
    }
    for (void* q : surplus) (void)hipFree(q);
    if (*ptr) return hipSuccess;
    void* p = nullptr;
    const hipError_t err = hipMalloc(&p, bytes);
    if (err != hipSuccess) return err;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pool.push_back(PoolEntry{p, bytes, dev, true, {}, nullptr});
    *ptr = p;
    return hipSuccess;
}

namespace {
void release_impl(void* ptr, std::vector<hipEvent_t>&& pending, hipStream_t owner = nullptr) {
    void* drop = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        size_t idle = 0;
        for (auto& e : g_pool) {
            if (e.p == ptr) { e.busy = false; e.pending = std::move(pending); e.owner = owner; }
            idle += e.busy ? 0 : 1;
        }
        if (idle > kKeep) {  // drop the smallest buffer that is known to be idle (hipFree of a buffer in use would wait for it).
            // No hipEventQuery here: a release may run in a destructor while another stream is being captured, and on ROCm 7.2
            // hipEventQuery invalidates a global-mode capture (hipEventCreate / hipEventRecord on another stream do not;
            // scratch/dbg_capture.py). Entries with pending events are settled by the next scratch_acquire.
            size_t k = g_pool.size();
            for (size_t i = 0; i < g_pool.size(); ++i)
                if (!g_pool[i].busy && g_pool[i].pending.empty() && g_pool[i].owner == nullptr && (k == g_pool.size() || g_pool[i].bytes < g_pool[k].bytes)) k = i;
            if (k != g_pool.size()) {
                drop = g_pool[k].p;
                g_pool.erase(g_pool.begin() + (long)k);
            }
        }
        // hipFree is a device-wide wait and, like hipEventQuery, invalidates a capture running on another stream: a release
        // (destructor) only queues the buffer; the next acquire — a call that allocates and synchronises anyway — frees it
        if (drop) g_to_free.push_back(drop);
    }
}
}  // namespace

void scratch_release(void* ptr) { release_impl(ptr, {}); }

void scratch_release_after(void* ptr, const StreamSet& streams) {
    if (!streams.overflow && streams.n == 1 && streams.s[0] != nullptr) {  // one stream: its order protects the buffer, no event
        release_impl(ptr, {}, streams.s[0]);
        return;
    }
    std::vector<hipEvent_t> pending;
    bool ok = !streams.overflow;
    for (int i = 0; ok && i < streams.n; ++i) {
        hipEvent_t ev = nullptr;
        ok = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        if (ok && hipEventRecord(ev, streams.s[i]) != hipSuccess) { (void)hipEventDestroy(ev); ok = false; }
        if (ok) pending.push_back(ev);
    }
    if (!ok) {  // too many streams to track, or an event could not be recorded: the device-wide wait instead
        (void)hipGetLastError();
        for (hipEvent_t ev : pending) (void)hipEventDestroy(ev);
        pending.clear();
        (void)hipDeviceSynchronize();
    }
    release_impl(ptr, std::move(pending));
}
}  // namespace sp

extern "C" void sp_stream_retired(void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (st == nullptr) return;
    bool any = false;
    {
        std::lock_guard<std::mutex> lock(sp::g_pool_mutex);
        for (const auto& e : sp::g_pool) any = any || e.owner == st;
    }
    if (!any) return;
    (void)hipStreamSynchronize(st);  // (still alive: the caller retires it BEFORE hipStreamDestroy)
    std::lock_guard<std::mutex> lock(sp::g_pool_mutex);
    for (auto& e : sp::g_pool)
        if (e.owner == st) e.owner = nullptr;
}
extern "C" int sp_abi_version(void) { return SP_ABI_VERSION; }
extern "C" const char* sp_last_error(void) { return g_last_error.c_str(); }

extern "C" int sp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int sp_set_device(int device) {
    const hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        sp_set_error(hipGetErrorString(e));
        return SP_ERR_HIP;
    }
    return SP_OK;
}

extern "C" void sp_se3_exp_host(const float* twist6, float* T_out16) {
    const sp::Rigid r = sp::se3_exp(twist6);
    sp::store_rigid_colmajor(r, T_out16);
}
extern "C" void sp_rigid_mul_host(const float* A16, const float* B16, float* out16) {
    const sp::Rigid r = sp::rigid_mul(sp::load_rigid_colmajor(A16), sp::load_rigid_colmajor(B16));
    sp::store_rigid_colmajor(r, out16);
}
extern "C" int sp_ldlt6_solve_host(const float* H36_rowmajor, const float* rhs6, float* x6) {
    sp::LdltScratch w;
    return sp::ldlt6_solve(H36_rowmajor, rhs6, x6, w) ? SP_OK : SP_ERR_RUNTIME;
}
extern "C" void sp_dogleg_step_host(const float* H36_rowmajor, const float* g6, float trust_region_radius, float* p_out6,
                                    float* step_norm_out, float* predicted_reduction_out) {
    sp::LdltScratch w;
    const sp::DoglegStep6 r = sp::dogleg_step6(H36_rowmajor, g6, trust_region_radius, w);
    for (int i = 0; i < 6; ++i) p_out6[i] = r.p[i];
    if (step_norm_out) *step_norm_out = r.step_norm;
    if (predicted_reduction_out) *predicted_reduction_out = r.predicted_reduction;
}

// Multi-GPU exchange of the GICP loop inside the C ABI (SURVEY.md 8b "sp_allreduce44", 8e): one process per GPU, source
// sharded, target replicated, ONE collective per iteration — an all-reduce (sum) of the rank's 128-byte fan-in row
// (registration.hip: fanin_reduce) over xGMI — plus the all-gather that shares the target covariances of a pre-loop
// sharded by query. RCCL is what moves the bytes; it is bound at run time (dlopen of librccl.so.1) so that the library
// loads, and every single-GPU entry point works, on a machine without RCCL: only sp_comm_* then report SP_ERR_RUNTIME.
// A 128-byte all-reduce is latency-bound (no bandwidth tuning applies: 7 x 153 GB/s links move it in nanoseconds; the
// cost is RCCL's launch + ring/tree hops), which is why the row was shrunk from 32 KB to 128 B inside the kernel and the
// whole alignment is captured into one hipGraph by the callers that can (sp_gicp_align_sharded only enqueues).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "sp_common.h"
#include "sp_xchg.h"

void sp_set_error(const char* msg);

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a process that already holds an RCCL (PyTorch bundles one under the same SONAME) gets that one
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) return;
        auto sym = [&](const char* n) { return dlsym(r.so, n); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather && r.GetErrorString;
    });
    return r;
}

int need_rccl() {
    if (rccl().ok) return SP_OK;
    sp_set_error("[sp_comm] RCCL (librccl.so.1) could not be loaded");
    return SP_ERR_RUNTIME;
}
int nccl_status(ncclResult_t r) {
    if (r == ncclSuccess) return SP_OK;
    sp_set_error(rccl().GetErrorString(r));
    return SP_ERR_HIP;
}

}  // namespace

struct sp_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

static_assert(SP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "sp_comm id is RCCL's ncclUniqueId");

extern "C" int sp_comm_unique_id(void* id_out) {
    if (!id_out) return SP_ERR_INVALID_ARGUMENT;
    if (const int rc = need_rccl(); rc != SP_OK) return rc;
    ncclUniqueId id;
    const int rc = nccl_status(rccl().GetUniqueId(&id));
    if (rc == SP_OK) std::memcpy(id_out, id.internal, SP_COMM_ID_BYTES);
    return rc;
}

extern "C" int sp_comm_create(const void* id, int rank, int world, sp_comm** out) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return SP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (const int rc = need_rccl(); rc != SP_OK) return rc;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, SP_COMM_ID_BYTES);
    sp_comm* c = new sp_comm();
    c->rank = rank;
    c->world = world;
    const int rc = nccl_status(rccl().CommInitRank(&c->comm, world, uid, rank));  // collective over the `world` callers
    if (rc != SP_OK) { delete c; return rc; }
    *out = c;
    return SP_OK;
}

extern "C" void sp_comm_destroy(sp_comm* c) {
    if (!c) return;
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
}
extern "C" int sp_comm_rank(const sp_comm* c) { return c ? c->rank : 0; }
extern "C" int sp_comm_world(const sp_comm* c) { return c ? c->world : 1; }

extern "C" int sp_allreduce_f32(sp_comm* c, float* buf, size_t n_floats, void* stream) {
    if (!c || (!buf && n_floats)) return SP_ERR_INVALID_ARGUMENT;
    if (n_floats == 0) return SP_OK;
    return nccl_status(rccl().AllReduce(buf, buf, n_floats, ncclFloat32, ncclSum, c->comm, sp::as_stream(stream)));
}

extern "C" int sp_allreduce_rows(sp_comm* c, void* workspace, int k, void* stream) {
    size_t n = 0;
    float* row = sp_gicp_align_row(workspace, k, &n);
    if (!c || !row) return SP_ERR_INVALID_ARGUMENT;
    return sp_allreduce_f32(c, row, n, stream);
}

extern "C" int sp_allgather(sp_comm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
    if (!c || ((!send || !recv) && bytes_per_rank)) return SP_ERR_INVALID_ARGUMENT;
    if (bytes_per_rank == 0) return SP_OK;
    return nccl_status(rccl().AllGather(send, recv, bytes_per_rank, ncclUint8, c->comm, sp::as_stream(stream)));
}

// Registration::align's Gauss-Newton loop with the source sharded over the ranks of `comm`: per iteration
// ONE launch (prologue + search / certified reuse + linearise of this rank's shard + in-kernel fan-in to one row) and ONE
// all-reduce of that row; launch k + 1's prologue then solves the same 6x6 system from the same row on every rank, so all
// ranks hold the identical pose without a broadcast. Only enqueues: a caller may capture it into a hipGraph.
extern "C" int sp_gicp_align_sharded(const sp_gicp_target* target, const sp_gicp_source* source, float* transT_device,
                                     const sp_factor_params* params, const sp_gn_params* gn, int max_iterations,
                                     sp_comm* comm, int32_t* nn_idx_out, float* nn_d2_out, sp_linearized* lin_out,
                                     float* delta_out8, uint32_t* iterations_out, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!comm) return SP_ERR_INVALID_ARGUMENT;
    if (max_iterations <= 0)
        return sp_gicp_align_fused(target, source, transT_device, params, gn, max_iterations, nn_idx_out, nn_d2_out, lin_out,
                                   delta_out8, iterations_out, workspace, workspace_bytes, stream);
    // (a rank whose shard is empty still takes part in every collective: its launch writes a zero row)
    for (int k = 0; k < max_iterations; ++k) {
        int rc = sp_gicp_align_step(target, source, transT_device, params, gn, k, 2, nn_idx_out, nn_d2_out, lin_out, workspace,
                                    workspace_bytes, stream);
        if (rc == SP_OK) rc = sp_allreduce_rows(comm, workspace, k, stream);
        if (rc != SP_OK) return rc;
    }
    return sp_gicp_align_finish(source, transT_device, gn, max_iterations - 1, 2, lin_out, delta_out8, iterations_out,
                                workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------ direct exchange (sp_xchg.h)
extern "C" void sp_xchg_destroy(sp_xchg* x) {
    if (!x) return;
    (void)hipDeviceSynchronize();  // no kernel of this process may still be storing into a peer's buffer
    for (int r = 0; r < x->world && r < kXchgMaxWorld; ++r)
        if (x->connected && r != x->rank && x->peers_host[r]) (void)hipIpcCloseMemHandle(x->peers_host[r]);
    if (x->peers_dev) (void)hipFree(x->peers_dev);
    if (x->epoch_dev) (void)hipFree(x->epoch_dev);
    if (x->local) (void)hipFree(x->local);
    delete x;
}

extern "C" int sp_xchg_create(int rank, int world, sp_xchg** out) {
    if (!out || world < 1 || world > kXchgMaxWorld || rank < 0 || rank >= world) return SP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    sp_xchg* x = new sp_xchg();
    x->rank = rank;
    x->world = world;
    const size_t bytes = (size_t)kXchgSlots * world * kXchgRow * sizeof(unsigned long long);
    // uncached (or at least fine-grained): written by other agents (and other processes), polled here — nothing of it may sit
    // in this GPU's L2. No silent fall-back to ordinary memory: polls could then be served stale lines and every alignment would
    // run into its time limit; the caller falls back to the collective instead (bench.py --exchange auto).
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&x->local), bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(reinterpret_cast<void**>(&x->local), bytes, hipDeviceMallocFinegrained);
    }
    if (e == hipSuccess) e = hipMemset(x->local, 0, bytes);  // tag 0 is never a sequence number
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x->peers_dev), kXchgMaxWorld * sizeof(void*));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x->epoch_dev), sizeof(unsigned));
    const unsigned one = 1u;
    if (e == hipSuccess) e = hipMemcpy(x->epoch_dev, &one, sizeof one, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&x->handle, x->local);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        sp_set_error(hipGetErrorString(e));
        sp_xchg_destroy(x);
        return SP_ERR_HIP;
    }
    *out = x;
    return SP_OK;
}

static_assert(sizeof(hipIpcMemHandle_t) == SP_XCHG_HANDLE_BYTES, "sp_xchg handle is a hipIpcMemHandle_t");

extern "C" int sp_xchg_handle(const sp_xchg* x, void* handle_out) {
    if (!x || !handle_out) return SP_ERR_INVALID_ARGUMENT;
    std::memcpy(handle_out, &x->handle, SP_XCHG_HANDLE_BYTES);
    return SP_OK;
}

extern "C" int sp_xchg_connect(sp_xchg* x, const void* handles_all) {
    if (!x || !handles_all || x->connected) return SP_ERR_INVALID_ARGUMENT;
    const char* h = static_cast<const char*>(handles_all);
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank) { x->peers_host[r] = x->local; continue; }
        hipIpcMemHandle_t hd;
        std::memcpy(&hd, h + (size_t)r * SP_XCHG_HANDLE_BYTES, SP_XCHG_HANDLE_BYTES);
        const hipError_t e = hipIpcOpenMemHandle(&x->peers_host[r], hd, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            sp_set_error(hipGetErrorString(e));
            for (int q = 0; q < r; ++q)
                if (q != x->rank && x->peers_host[q]) { (void)hipIpcCloseMemHandle(x->peers_host[q]); x->peers_host[q] = nullptr; }
            return SP_ERR_HIP;
        }
    }
    if (hipMemcpy(x->peers_dev, x->peers_host, x->world * sizeof(void*), hipMemcpyHostToDevice) != hipSuccess) return SP_ERR_HIP;
    x->connected = true;
    return SP_OK;
}

extern "C" int sp_xchg_set_timeout_ms(sp_xchg* x, unsigned ms) {
    if (!x || ms == 0) return SP_ERR_INVALID_ARGUMENT;
    x->timeout_ms = ms;
    return SP_OK;
}
extern "C" int sp_xchg_rank(const sp_xchg* x) { return x ? x->rank : 0; }
extern "C" int sp_xchg_world(const sp_xchg* x) { return x ? x->world : 1; }

// qg_comm.hip — the ONE exchange step of the multi-GPU path, owned by the library: the gather of packed C bands to the root
// rank over RCCL (xGMI inside a node).  One process per GPU; rows of C are independent, B is replicated, so
// this gather is the only collective.  The reference has no counterpart (it is a single-threaded header, QuBLAS.h has no
// communication of any kind); the interface is include/qgemul.h "one process per GPU".
//
// RCCL is bound at the first qgemul_comm_* call with dlopen / dlsym, not at link time:
//   * a process that already holds an RCCL (PyTorch ships its own librccl.so next to libtorch_hip.so) must keep using THAT
//     one — two RCCL instances in one process would each bring their own topology / IPC state — so an already loaded
//     library is taken first (RTLD_NOLOAD), and only a process without one loads /opt/rocm's librccl.so.1;
//   * single-GPU users never pay for mapping a 570 MB library.
// qgemul_comm_info() reports ncclGetVersion() and ncclCommCount() of the bound library, which is how bench.py shows that the
// gather ran on RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>

#include "../../include/qgemul.h"

namespace {

thread_local int g_last_rccl = 0;

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};

RcclApi& rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* loaded[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : loaded)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // the process's own RCCL, if it has one
        const char* fresh[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : fresh)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!api.handle) return;
        bool all = true;
        auto sym = [&](auto& fn, const char* name) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, name));
            all = all && fn != nullptr;
        };
        sym(api.GetVersion, "ncclGetVersion");
        sym(api.GetUniqueId, "ncclGetUniqueId");
        sym(api.CommInitRank, "ncclCommInitRank");
        sym(api.CommDestroy, "ncclCommDestroy");
        sym(api.CommCount, "ncclCommCount");
        sym(api.CommUserRank, "ncclCommUserRank");
        sym(api.GroupStart, "ncclGroupStart");
        sym(api.GroupEnd, "ncclGroupEnd");
        sym(api.Send, "ncclSend");
        sym(api.Recv, "ncclRecv");
        sym(api.AllReduce, "ncclAllReduce");
        api.ok = all;
    });
    return api;
}

#define QG_RCCL(expr)                          \
    do {                                       \
        ncclResult_t r_ = (expr);              \
        if (r_ != ncclSuccess) {               \
            g_last_rccl = (int)r_;             \
            return QG_ERCCL;                   \
        }                                      \
    } while (0)
#define QG_HIPC(expr)                          \
    do {                                       \
        if ((expr) != hipSuccess) return QG_EHIP; \
    } while (0)

struct DevScope {
    int prev = -1;
    bool changed = false;
    explicit DevScope(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DevScope() { if (changed) hipSetDevice(prev); }
};

} // namespace

struct qgemul_comm {
    qgemul_ctx* ctx;
    int device;
    ncclComm_t comm;
    int nranks, rank;
    hipStream_t stream;     // the gathers' own stream: a band travels while the next GEMM runs on the context's stream
    hipEvent_t produced;    // context stream -> comm stream: the band to send is complete
    hipEvent_t sent[QG_COMM_SLOTS];   // comm stream -> context stream: the buffers of the gathers issued under slot s may be reused
    double* scratch;        // 16 bytes on the device (barrier / max)
};

extern "C" {

// (declared in qg_api.hip's translation unit as well: the context's device and stream)
void* qgemul_ctx_stream(qgemul_ctx* c);
int qgemul_ctx_device(const qgemul_ctx* c);

int qgemul_last_rccl_error(void) { return g_last_rccl; }

int qgemul_comm_unique_id(void* id_out)
{
    if (!id_out) return QG_EINVAL;
    RcclApi& a = rccl();
    if (!a.ok) { g_last_rccl = -1; return QG_ERCCL; }   // no usable librccl in this process / on this machine
    ncclUniqueId id;
    QG_RCCL(a.GetUniqueId(&id));
    static_assert(sizeof id == QG_COMM_ID_BYTES, "qgemul.h: QG_COMM_ID_BYTES");
    memcpy(id_out, &id, sizeof id);
    return QG_OK;
}

int qgemul_comm_create(qgemul_ctx* c, int nranks, int rank, const void* unique_id, qgemul_comm** out)
{
    if (!c || !unique_id || !out || nranks < 1 || rank < 0 || rank >= nranks) return QG_EINVAL;
    RcclApi& a = rccl();
    if (!a.ok) { g_last_rccl = -1; return QG_ERCCL; }
    qgemul_comm* m = new (std::nothrow) qgemul_comm;
    if (!m) return QG_EINVAL;
    memset(m, 0, sizeof *m);
    m->ctx = c;
    m->device = qgemul_ctx_device(c);
    m->nranks = nranks;
    m->rank = rank;
    DevScope scope(m->device);   // ncclCommInitRank binds the communicator to the CURRENT device
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    const ncclResult_t r = a.CommInitRank(&m->comm, nranks, id, rank);
    if (r != ncclSuccess) { g_last_rccl = (int)r; delete m; return QG_ERCCL; }
    bool ok = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&m->produced, hipEventDisableTiming) == hipSuccess &&
              hipMalloc((void**)&m->scratch, 16) == hipSuccess;
    for (int i = 0; ok && i < QG_COMM_SLOTS; ++i) {
        ok = hipEventCreateWithFlags(&m->sent[i], hipEventDisableTiming) == hipSuccess;
        if (ok) ok = hipEventRecord(m->sent[i], m->stream) == hipSuccess;   // (a slot nobody has used yet is "done")
    }
    if (!ok) {
        qgemul_comm_destroy(m);
        return QG_EHIP;
    }
    *out = m;
    return QG_OK;
}

void qgemul_comm_destroy(qgemul_comm* m)
{
    if (!m) return;
    DevScope scope(m->device);
    if (m->stream) hipStreamSynchronize(m->stream);
    if (m->comm) rccl().CommDestroy(m->comm);
    if (m->scratch) hipFree(m->scratch);
    if (m->produced) hipEventDestroy(m->produced);
    for (int i = 0; i < QG_COMM_SLOTS; ++i)
        if (m->sent[i]) hipEventDestroy(m->sent[i]);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
}

int qgemul_comm_info(const qgemul_comm* m, int* nranks, int* rank, int* rccl_version)
{
    if (!m) return QG_EINVAL;
    RcclApi& a = rccl();
    if (nranks) QG_RCCL(a.CommCount(m->comm, nranks));       // (asked of the communicator, not remembered from the caller)
    if (rank) QG_RCCL(a.CommUserRank(m->comm, rank));
    if (rccl_version) QG_RCCL(a.GetVersion(rccl_version));
    return QG_OK;
}

int qgemul_gather_packed_c(qgemul_comm* m, const void* send, size_t send_bytes, void* const* recv, const size_t* recv_bytes, int root, int slot)
{
    if (!m || root < 0 || root >= m->nranks || (send_bytes && !send) || slot < 0 || slot >= QG_COMM_SLOTS) return QG_EINVAL;
    if (m->rank == root && (!recv || !recv_bytes)) return QG_EINVAL;
    RcclApi& a = rccl();
    DevScope scope(m->device);
    hipStream_t cs = (hipStream_t)qgemul_ctx_stream(m->ctx);
    // the band was produced on the context's stream; it travels on the communicator's own stream
    QG_HIPC(hipEventRecord(m->produced, cs));
    QG_HIPC(hipStreamWaitEvent(m->stream, m->produced, 0));
    if (m->rank == root) {
        if (recv[root] && recv[root] != send && send_bytes) {
            if (recv_bytes[root] != send_bytes) return QG_EINVAL;
            QG_HIPC(hipMemcpyAsync(recv[root], send, send_bytes, hipMemcpyDeviceToDevice, m->stream));
        }
        QG_RCCL(a.GroupStart());
        for (int r = 0; r < m->nranks; ++r)
            if (r != root && recv_bytes[r]) {
                if (!recv[r]) { a.GroupEnd(); return QG_EINVAL; }
                const ncclResult_t e = a.Recv(recv[r], recv_bytes[r], ncclChar, r, m->comm, m->stream);
                if (e != ncclSuccess) { a.GroupEnd(); g_last_rccl = (int)e; return QG_ERCCL; }
            }
        QG_RCCL(a.GroupEnd());
    } else if (send_bytes) {
        QG_RCCL(a.Send(send, send_bytes, ncclChar, root, m->comm, m->stream));
    }
    QG_HIPC(hipEventRecord(m->sent[slot], m->stream));
    return QG_OK;
}

int qgemul_comm_fence(qgemul_comm* m, int slot)
{
    if (!m || slot < -1 || slot >= QG_COMM_SLOTS) return QG_EINVAL;
    DevScope scope(m->device);
    hipStream_t cs = (hipStream_t)qgemul_ctx_stream(m->ctx);
    for (int i = 0; i < QG_COMM_SLOTS; ++i)
        if (slot < 0 || slot == i) QG_HIPC(hipStreamWaitEvent(cs, m->sent[i], 0));
    return QG_OK;
}

int qgemul_comm_sync(qgemul_comm* m)
{
    if (!m) return QG_EINVAL;
    DevScope scope(m->device);
    QG_HIPC(hipStreamSynchronize(m->stream));
    return QG_OK;
}

int qgemul_comm_max_f64(qgemul_comm* m, double* inout)
{
    if (!m || !inout) return QG_EINVAL;
    RcclApi& a = rccl();
    DevScope scope(m->device);
    QG_HIPC(hipMemcpyAsync(m->scratch, inout, sizeof(double), hipMemcpyHostToDevice, m->stream));
    QG_RCCL(a.AllReduce(m->scratch, m->scratch, 1, ncclDouble, ncclMax, m->comm, m->stream));
    QG_HIPC(hipMemcpyAsync(inout, m->scratch, sizeof(double), hipMemcpyDeviceToHost, m->stream));
    QG_HIPC(hipStreamSynchronize(m->stream));
    return QG_OK;
}

int qgemul_comm_barrier(qgemul_comm* m)
{
    if (!m) return QG_EINVAL;
    DevScope scope(m->device);
    QG_HIPC(hipStreamSynchronize((hipStream_t)qgemul_ctx_stream(m->ctx)));   // this rank's GEMMs are done ...
    double one = 1.0;
    return qgemul_comm_max_f64(m, &one);                                     // ... and so are every other rank's (and the gathers before it)
}

} // extern "C"

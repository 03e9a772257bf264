// qr_comm.hip -- RCCL glue for the C-level multi-GPU TSQR entry point (qr_thin_mgpu in qr_host.c).
//
// The library does not link librccl: the single-GPU drop-in path must not pay for loading a collective library it never
// uses.  The first multi-GPU call dlopen()s librccl.so and resolves the four entry points it needs.  One communicator per
// device, created together by ncclCommInitAll from the calling thread; afterwards every device is driven by its own host
// thread (one thread per GPU) and the only collective of the algorithm is ONE ncclAllGather of the n x n R
// factors (RCCL has no user-defined reduction, so the "all-reduce of R" is all-gather + redundant stacked QR).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include "qr_device.h"

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    int state = 0;      // 0 = not tried, 1 = loaded, -1 = unavailable
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.state) return g_rccl.state > 0 ? 0 : QRD_E_NORCCL;
    // (On hosts whose driver only supports dmabuf IPC -- this pool -- RCCL's intra-node transport needs HSA_ENABLE_IPC_MODE_LEGACY=0 in
    // the environment BEFORE the HSA runtime starts, i.e. before the process's first GPU call: the launcher's job (bench.py,
    // INTEGRATION.md section 5), never this library's -- a setenv from here came too late for every caller that had created a plan and
    // raced with getenv in other threads.  A failing communicator creation with the variable unset gets a hint, below.)
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) { g_rccl.state = -1; return QRD_E_NORCCL; }
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(dlsym(g_rccl.handle, "ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.handle, "ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(g_rccl.handle, "ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.handle, "ncclGetErrorString"));
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(g_rccl.handle, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(g_rccl.handle, "ncclCommInitRank"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(g_rccl.handle, "ncclCommCount"));
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetUniqueId || !g_rccl.CommInitRank) {
        g_rccl.state = -1;
        return QRD_E_NORCCL;
    }
    g_rccl.state = 1;
    return 0;
}
}   // namespace

// ---- optional roctx ranges (tracing): MI355XQR_ROCTX=1 wraps every outer step's panel / look-ahead / wide update in
// named ranges for `rocprofv3 --marker-trace`; librocprofiler-sdk-roctx (or libroctx64) is dlopen()ed on first use, never linked
namespace {
struct RoctxApi { std::atomic<int> state{0}; int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };
RoctxApi g_roctx;
std::mutex g_roctx_mutex;
bool roctx_ready()
{
    const int st = g_roctx.state.load(std::memory_order_acquire);
    if (st) return st > 0;                                  // the usual case (off): one atomic load per range
    std::lock_guard<std::mutex> lock(g_roctx_mutex);
    if (g_roctx.state.load()) return g_roctx.state.load() > 0;
    const char* e = getenv("MI355XQR_ROCTX");
    if (!e || atoi(e) == 0) { g_roctx.state.store(-1, std::memory_order_release); return false; }
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (h) {
        g_roctx.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        g_roctx.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    }
    g_roctx.state.store((g_roctx.push && g_roctx.pop) ? 1 : -1, std::memory_order_release);
    return g_roctx.state.load() > 0;
}
}   // namespace

extern "C" {

void qrd_range_push(const char* name) { if (roctx_ready()) g_roctx.push(name); }
void qrd_range_pop(void) { if (roctx_ready()) g_roctx.pop(); }

// comms: array of n opaque communicator handles, one per entry of devs (all on this node)
int qrd_comm_init_all(void** comms, int n, const int* devs)
{
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclComm_t) == sizeof(void*), "communicator handles travel as void*");
    const ncclResult_t r = g_rccl.CommInitAll(reinterpret_cast<ncclComm_t*>(comms), n, devs);
    return r == ncclSuccess ? 0 : QRD_E_RCCL - (int) r;
}

// one process (or thread) per GPU: rank 0 makes the id, the caller carries its QRD_UNIQUE_ID_BYTES to every rank by its own means
// (MPI, a torch.distributed store, a file), then every rank calls qrd_comm_init_rank with its current device set
int qrd_comm_unique_id(void* id)
{
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == QRD_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    const ncclResult_t r = g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id));
    return r == ncclSuccess ? 0 : QRD_E_RCCL - (int) r;
}

int qrd_comm_init_rank(void** comm, int nranks, const void* id, int rank)
{
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    const ncclResult_t r = g_rccl.CommInitRank(reinterpret_cast<ncclComm_t*>(comm), nranks, uid, rank);
    return r == ncclSuccess ? 0 : QRD_E_RCCL - (int) r;
}

// ranks the communicator really spans, as RCCL reports it
int qrd_comm_count(void* comm, int* n)
{
    if (g_rccl.state <= 0 || !g_rccl.CommCount) return QRD_E_NORCCL;
    const ncclResult_t r = g_rccl.CommCount((ncclComm_t) comm, n);
    return r == ncclSuccess ? 0 : QRD_E_RCCL - (int) r;
}

const char* qrd_rccl_error_string(int r)
{
    // unhandled system / internal errors of communicator creation are what the dmabuf-only hosts produce when the variable is missing
    if ((r == (int) ncclUnhandledCudaError || r == (int) ncclSystemError || r == (int) ncclInternalError) && !getenv("HSA_ENABLE_IPC_MODE_LEGACY"))
        return "RCCL call failed (if this is hipIpcGetMemHandle: invalid argument, export HSA_ENABLE_IPC_MODE_LEGACY=0 before the process "
               "touches the GPU: see INTEGRATION.md, multi-GPU)";
    if (g_rccl.state > 0 && g_rccl.GetErrorString) return g_rccl.GetErrorString((ncclResult_t) r);
    return "RCCL call failed";
}

int qrd_comm_destroy(void* comm)
{
    if (!comm || g_rccl.state <= 0) return 0;
    return g_rccl.CommDestroy((ncclComm_t) comm) == ncclSuccess ? 0 : QRD_E_RCCL;
}

// every rank contributes `count` doubles at `send`; `recv` receives world * count doubles in rank order.  Stream-ordered.
int qrd_allgather_f64(void* comm, void* stream, const double* send, double* recv, size_t count)
{
    if (g_rccl.state <= 0) return QRD_E_NORCCL;
    const ncclResult_t r = g_rccl.AllGather(send, recv, count, ncclDouble, (ncclComm_t) comm, (hipStream_t) stream);
    return r == ncclSuccess ? 0 : QRD_E_RCCL - (int) r;
}

}   // extern "C"

// RCCL behind the C-ABI (gpmi_comm_*): the collectives of the row-block partitioned path -- broadcast of a factored
// diagonal block, all-gather of a panel column, the small all-reduces -- issued on the CALLER'S stream, straight into
// librccl, no process-group layer in between.  The library is not linked: it is opened at run time (dlopen) so that
// libgpmi355x.so loads on hosts without it, and so that a process that already carries a copy of RCCL (PyTorch ships
// one next to its HIP runtime) binds THAT copy -- two HIP runtimes or two RCCLs in one process do not share a device.
// panel all-gather / L_kk broadcast over xGMI; north_star: "a thin C-ABI ... RCCL
// broadcast/all-gather".  Every entry point returns the usual status (GPMI_ERR_RUNTIME with gpmi_last_error()).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "gpmi_ctx.h"

using namespace gpmi;

namespace {

struct RcclApi {
    void* handle = nullptr;
    std::string path;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};

std::mutex g_mu;
RcclApi g_api;

template <class F> bool sym(void* h, const char* name, F& out) {
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

// opens librccl (explicit path, or: a copy this process already carries, then the system's) and resolves the entry points
int load_rccl(const char* path) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_api.handle) return GPMI_OK;
    void* h = nullptr;
    std::string tried;
    if (path && *path) {
        h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        tried = path;
    } else {
        const char* env = getenv("GPMI_RCCL_LIB");
        const char* already[] = {"librccl.so", "librccl.so.1"};
        const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (env && *env) { h = dlopen(env, RTLD_NOW | RTLD_GLOBAL); tried = env; }
        for (const char* n : already) if (!h) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) tried = std::string(n) + " (already in the process)"; }
        for (const char* n : fresh) if (!h) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); tried = n; }
    }
    if (!h) {
        g_err = "gpmi_comm: cannot open librccl (" + tried + "): " + (dlerror() ? dlerror() : "?");
        return GPMI_ERR_RUNTIME;
    }
    RcclApi a;
    a.handle = h; a.path = tried;
    const bool ok = sym(h, "ncclGetUniqueId", a.GetUniqueId) && sym(h, "ncclCommInitRank", a.CommInitRank) &&
                    sym(h, "ncclCommDestroy", a.CommDestroy) && sym(h, "ncclBroadcast", a.Broadcast) &&
                    sym(h, "ncclAllGather", a.AllGather) && sym(h, "ncclAllReduce", a.AllReduce) &&
                    sym(h, "ncclGetErrorString", a.GetErrorString);
    (void)sym(h, "ncclCommAbort", a.CommAbort);
    (void)sym(h, "ncclGetVersion", a.GetVersion);
    if (!ok) {
        g_err = "gpmi_comm: " + tried + " lacks an RCCL entry point";
        return GPMI_ERR_RUNTIME;
    }
    g_api = a;
    return GPMI_OK;
}

int fail_rccl(ncclResult_t r, const char* what) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (ncclResult %d)", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "?", (int)r);
    g_err = buf;
    return GPMI_ERR_RUNTIME;
}

#define RCCL_TRY(expr)                                          \
    do {                                                        \
        ncclResult_t _r = (expr);                               \
        if (_r != ncclSuccess) return fail_rccl(_r, #expr);     \
    } while (0)

}  // namespace

struct gpmi_comm {
    ncclComm_t c = nullptr;
    int rank = 0, size = 1, device = 0;
};

extern "C" {

int gpmi_comm_load(const char* librccl_path) { return load_rccl(librccl_path); }

int gpmi_comm_library(char* out, int64_t cap, int* version) {
    if (!out || cap < 2) return fail_arg("gpmi_comm_library: null argument");
    const int rc = load_rccl(nullptr);
    if (rc) return rc;
    snprintf(out, (size_t)cap, "%s", g_api.path.c_str());
    if (version) { *version = 0; if (g_api.GetVersion) (void)g_api.GetVersion(version); }
    return GPMI_OK;
}

int gpmi_comm_unique_id(char* id128) {
    if (!id128) return fail_arg("gpmi_comm_unique_id: null argument");
    const int rc = load_rccl(nullptr);
    if (rc) return rc;
    ncclUniqueId id;
    RCCL_TRY(g_api.GetUniqueId(&id));
    static_assert(sizeof id.internal == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, id.internal, 128);
    return GPMI_OK;
}

int gpmi_comm_create(const char* id128, int rank, int size, int device, gpmi_comm** out) {
    if (!id128 || !out) return fail_arg("gpmi_comm_create: null argument");
    if (size < 1 || rank < 0 || rank >= size) return fail_arg("gpmi_comm_create: rank must be in [0, size)");
    const int rc = load_rccl(nullptr);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    gpmi_comm* c = new gpmi_comm;
    c->rank = rank; c->size = size; c->device = device;
    const ncclResult_t r = g_api.CommInitRank(&c->c, size, id, rank);
    if (r != ncclSuccess) { delete c; return fail_rccl(r, "ncclCommInitRank"); }
    *out = c;
    return GPMI_OK;
}

int gpmi_comm_destroy(gpmi_comm* c) {
    if (!c) return GPMI_OK;
    ncclResult_t r = ncclSuccess;
    if (c->c) r = g_api.CommDestroy(c->c);
    delete c;
    if (r != ncclSuccess) return fail_rccl(r, "ncclCommDestroy");
    return GPMI_OK;
}

// every rank's buf (nbytes) <- root's, on `stream`
int gpmi_comm_broadcast(gpmi_comm* c, void* stream, void* buf_dev, int64_t nbytes, int root) {
    if (!c || !buf_dev) return fail_arg("gpmi_comm_broadcast: null argument");
    if (nbytes < 0 || root < 0 || root >= c->size) return fail_arg("gpmi_comm_broadcast: bad size or root");
    if (nbytes == 0) return GPMI_OK;
    RCCL_TRY(g_api.Broadcast(buf_dev, buf_dev, (size_t)nbytes, ncclChar, root, c->c, (hipStream_t)stream));
    return GPMI_OK;
}

// recv (size * nbytes_per_rank) <- every rank's send (nbytes_per_rank), in rank order, on `stream`
int gpmi_comm_all_gather(gpmi_comm* c, void* stream, const void* send_dev, void* recv_dev, int64_t nbytes_per_rank) {
    if (!c || !send_dev || !recv_dev) return fail_arg("gpmi_comm_all_gather: null argument");
    if (nbytes_per_rank < 0) return fail_arg("gpmi_comm_all_gather: negative size");
    if (nbytes_per_rank == 0) return GPMI_OK;
    RCCL_TRY(g_api.AllGather(send_dev, recv_dev, (size_t)nbytes_per_rank, ncclChar, c->c, (hipStream_t)stream));
    return GPMI_OK;
}

// in place; dtype 0 float64, 1 int64; op 0 sum, 1 min, 2 max
int gpmi_comm_all_reduce(gpmi_comm* c, void* stream, void* buf_dev, int64_t count, int dtype, int op) {
    if (!c || !buf_dev) return fail_arg("gpmi_comm_all_reduce: null argument");
    if (count < 0 || dtype < 0 || dtype > 1 || op < 0 || op > 2) return fail_arg("gpmi_comm_all_reduce: bad count, dtype or op");
    if (count == 0) return GPMI_OK;
    const ncclDataType_t dt = dtype == 0 ? ncclFloat64 : ncclInt64;
    const ncclRedOp_t ro = op == 0 ? ncclSum : op == 1 ? ncclMin : ncclMax;
    RCCL_TRY(g_api.AllReduce(buf_dev, buf_dev, (size_t)count, dt, ro, c->c, (hipStream_t)stream));
    return GPMI_OK;
}

}  // extern "C"

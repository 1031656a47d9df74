// RCCL side of the C ABI (include/dffw.h, "multi-GPU"): the all-gather that collects the per-rank depth maps of a
// batch sharded over the GPUs of one node.  Replaces the gather half of nn.DataParallel in the reference
// (Depth_Estimation_Test/test.py:32): there the outputs of all replicas are copied to device 0 by torch's comm layer;
// here every rank (one process per GPU, or one thread / one communicator per GPU in a single process) contributes its
// (b, H, W) fp32 maps to ONE ncclAllGather over xGMI, enqueued on the compute stream right behind the last head kernel.
// librccl is bound with dlopen at the first call, so libdffw.so carries no load-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "../../include/dffw.h"

int dffw_fail(int code, const char *fmt, ...);

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    const char *why = "";
};

Rccl g_rccl;
std::once_flag g_once;

void bind() {
    const char *names[] = {getenv("DFFW_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) {
        g_rccl.why = "librccl.so not found (set DFFW_RCCL_LIB)";
        return;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(g_rccl.lib, name);
        if (!p) ok = false;
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        g_rccl.why = "librccl.so lacks an expected nccl* symbol";
        dlclose(g_rccl.lib);
        g_rccl.lib = nullptr;
    }
}

const Rccl *rccl() {
    std::call_once(g_once, bind);
    return g_rccl.lib ? &g_rccl : nullptr;
}

int nccl_fail(const Rccl *r, const char *what, ncclResult_t rc) {
    return dffw_fail(DFFW_EHIP, "%s -> %s", what, r->GetErrorString ? r->GetErrorString(rc) : "rccl error");
}

}  // namespace

struct dffw_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
};

static_assert(DFFW_COMM_ID_BYTES == sizeof(ncclUniqueId), "dffw.h's id size must be RCCL's");

extern "C" {

int dffw_comm_unique_id(char id[DFFW_COMM_ID_BYTES]) {
    if (!id) return dffw_fail(DFFW_EINVAL, "id is null");
    const Rccl *r = rccl();
    if (!r) return dffw_fail(DFFW_EHIP, "RCCL unavailable: %s", g_rccl.why);
    ncclUniqueId u;
    const ncclResult_t rc = r->GetUniqueId(&u);
    if (rc != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId", rc);
    memcpy(id, &u, sizeof u);
    return DFFW_OK;
}

int dffw_comm_init_rank(int device, int nranks, int rank, const char id[DFFW_COMM_ID_BYTES], dffw_comm **out) {
    if (!out || !id) return dffw_fail(DFFW_EINVAL, "null argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return dffw_fail(DFFW_EINVAL, "bad rank %d of %d", rank, nranks);
    const Rccl *r = rccl();
    if (!r) return dffw_fail(DFFW_EHIP, "RCCL unavailable: %s", g_rccl.why);
    const hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return dffw_fail(DFFW_EHIP, "hipSetDevice(%d) -> %s", device, hipGetErrorString(he));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    dffw_comm *c = new dffw_comm;
    c->rank = rank;
    c->nranks = nranks;
    c->device = device;
    const ncclResult_t rc = r->CommInitRank(&c->comm, nranks, u, rank);
    if (rc != ncclSuccess) {
        delete c;
        return nccl_fail(r, "ncclCommInitRank", rc);
    }
    *out = c;
    return DFFW_OK;
}

int dffw_comm_init_all(int ndev, const int *devices, dffw_comm **out) {
    if (!out || ndev < 1) return dffw_fail(DFFW_EINVAL, "bad argument");
    for (int i = 0; i < ndev; ++i) out[i] = nullptr;
    const Rccl *r = rccl();
    if (!r) return dffw_fail(DFFW_EHIP, "RCCL unavailable: %s", g_rccl.why);
    ncclComm_t *cs = new ncclComm_t[ndev];
    const ncclResult_t rc = r->CommInitAll(cs, ndev, devices);
    if (rc != ncclSuccess) {
        delete[] cs;
        return nccl_fail(r, "ncclCommInitAll", rc);
    }
    for (int i = 0; i < ndev; ++i) {
        out[i] = new dffw_comm;
        out[i]->comm = cs[i];
        out[i]->rank = i;
        out[i]->nranks = ndev;
        out[i]->device = devices ? devices[i] : i;
    }
    delete[] cs;
    return DFFW_OK;
}

void dffw_comm_destroy(dffw_comm *c) {
    if (!c) return;
    const Rccl *r = rccl();
    if (r && c->comm) {
        (void)hipSetDevice(c->device);
        (void)r->CommDestroy(c->comm);
    }
    delete c;
}

int dffw_comm_rank(const dffw_comm *c) { return c ? c->rank : dffw_fail(DFFW_EINVAL, "null comm"); }
int dffw_comm_size(const dffw_comm *c) { return c ? c->nranks : dffw_fail(DFFW_EINVAL, "null comm"); }

int dffw_allgather(dffw_comm *c, const float *send, float *recv, int64_t count, void *hip_stream) {
    if (!c || !send || !recv) return dffw_fail(DFFW_EINVAL, "null argument");
    if (count < 0) return dffw_fail(DFFW_EINVAL, "negative count");
    const Rccl *r = rccl();
    if (!r) return dffw_fail(DFFW_EHIP, "RCCL unavailable: %s", g_rccl.why);
    const hipError_t he = hipSetDevice(c->device);
    if (he != hipSuccess) return dffw_fail(DFFW_EHIP, "hipSetDevice(%d) -> %s", c->device, hipGetErrorString(he));
    const ncclResult_t rc = r->AllGather(send, recv, (size_t)count, ncclFloat, c->comm, (hipStream_t)hip_stream);
    if (rc != ncclSuccess) return nccl_fail(r, "ncclAllGather", rc);
    return DFFW_OK;
}

int dffw_comm_group_start(void) {
    const Rccl *r = rccl();
    if (!r) return dffw_fail(DFFW_EHIP, "RCCL unavailable: %s", g_rccl.why);
    const ncclResult_t rc = r->GroupStart();
    return rc == ncclSuccess ? DFFW_OK : nccl_fail(r, "ncclGroupStart", rc);
}

int dffw_comm_group_end(void) {
    const Rccl *r = rccl();
    if (!r) return dffw_fail(DFFW_EHIP, "RCCL unavailable: %s", g_rccl.why);
    const ncclResult_t rc = r->GroupEnd();
    return rc == ncclSuccess ? DFFW_OK : nccl_fail(r, "ncclGroupEnd", rc);
}

}  // extern "C"

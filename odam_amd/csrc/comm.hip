// comm.hip -- include/odam_comm.h: the detections all-gather on RCCL.  The library is resolved with dlopen / dlsym: first the
// librccl the process has mapped already (PyTorch ships its own: two RCCL instances in one process would each bootstrap their own
// transport), then the system one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/odam_comm.h"
#include "odam_err.h"

namespace {
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[200] = "";
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    void* h = nullptr;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // what the process has mapped already
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { std::snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl not found: %s", dlerror()); return; }
    auto sym = [&](const char* n) { return dlsym(h, n); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.GroupStart && g_rccl.GroupEnd;
    if (!g_rccl.ok) std::snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl lacks an entry point");
}
int need_rccl(const char* who) {
    std::call_once(g_once, load_rccl);
    if (g_rccl.ok) return 0;
    std::snprintf(g_odam_err, sizeof(g_odam_err), "%s: %s", who, g_rccl.why);
    return 2;
}
int rccl_fail(const char* what, ncclResult_t r) {
    std::snprintf(g_odam_err, sizeof(g_odam_err), "%s: RCCL error %d (%s)", what, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return 2;
}
}  // namespace

struct odam_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0;
};

extern "C" int odam_comm_unique_id(unsigned char* id) {
    if (!id) return odam_fail(1, "odam_comm_unique_id: null pointer");
    if (int rc = need_rccl("odam_comm_unique_id")) return rc;
    ncclUniqueId u;
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    const ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

extern "C" int odam_comm_create(const unsigned char* id, int world, int rank, odam_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return odam_fail(1, "odam_comm_create: bad argument");
    if (int rc = need_rccl("odam_comm_create")) return rc;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    odam_comm* c = new odam_comm();
    c->world = world; c->rank = rank;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return rccl_fail("ncclCommInitRank", r); }
    *out = c;
    return 0;
}

extern "C" int odam_comm_destroy(odam_comm* c) {
    if (!c) return 0;
    if (c->comm && g_rccl.ok) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}

extern "C" int odam_comm_info(odam_comm* c, int* world, int* rank) {
    if (!c) return odam_fail(1, "odam_comm_info: null handle");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return 0;
}

extern "C" int odam_allgather_detections(odam_comm* c, const float* local, const int* n_local, int F_pad, float* all, int* n_all, void* stream) {
    if (!c || F_pad < 0 || (F_pad && (!local || !n_local || !all || !n_all))) return odam_fail(1, "odam_allgather_detections: bad argument");
    if (F_pad == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    ncclResult_t r = g_rccl.GroupStart();
    if (r != ncclSuccess) return rccl_fail("ncclGroupStart", r);
    const ncclResult_t r1 = g_rccl.AllGather(local, all, (size_t)F_pad * 30 * 15, ncclFloat32, c->comm, st);
    const ncclResult_t r2 = g_rccl.AllGather(n_local, n_all, (size_t)F_pad, ncclInt32, c->comm, st);
    r = g_rccl.GroupEnd();
    if (r1 != ncclSuccess) return rccl_fail("ncclAllGather (blocks)", r1);
    if (r2 != ncclSuccess) return rccl_fail("ncclAllGather (counts)", r2);
    if (r != ncclSuccess) return rccl_fail("ncclGroupEnd", r);
    return 0;
}

extern "C" int odam_allgather_f32(odam_comm* c, const float* local, long long n, float* all, void* stream) {
    if (!c || n < 0 || (n && (!local || !all))) return odam_fail(1, "odam_allgather_f32: bad argument");
    if (n == 0) return 0;
    const ncclResult_t r = g_rccl.AllGather(local, all, (size_t)n, ncclFloat32, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
    return 0;
}

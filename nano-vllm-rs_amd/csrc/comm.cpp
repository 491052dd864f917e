#include "comm.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace nvr {

namespace {
struct Api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Api g_api;

int load_api() {
    if (g_api.lib) return NVR_OK;
    // the ROCm 7.2 copy first, by path: a process that imported torch already holds torch's bundled (ROCm 7.0)
    // librccl under the same soname, and a bare-soname dlopen would return that one
    const char *names[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    void *lib = nullptr;
    for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) return fail(NVR_ERR_RCCL, "cannot dlopen librccl: %s", dlerror());
#define NVR_SYM(field, name)                                                             \
    *(void **)(&g_api.field) = dlsym(lib, name);                                         \
    if (!g_api.field) return fail(NVR_ERR_RCCL, "librccl lacks symbol %s", name);
    NVR_SYM(GetUniqueId, "ncclGetUniqueId")
    NVR_SYM(CommInitRank, "ncclCommInitRank")
    NVR_SYM(CommDestroy, "ncclCommDestroy")
    NVR_SYM(AllReduce, "ncclAllReduce")
    NVR_SYM(AllGather, "ncclAllGather")
    NVR_SYM(GetErrorString, "ncclGetErrorString")
#undef NVR_SYM
    g_api.lib = lib;
    return NVR_OK;
}
#define NVR_NCCL(expr)                                                                   \
    do {                                                                                 \
        ncclResult_t _r = (expr);                                                        \
        if (_r != ncclSuccess) return fail(NVR_ERR_RCCL, "%s failed: %s", #expr, g_api.GetErrorString(_r)); \
    } while (0)
}  // namespace

int Comm::unique_id(uint8_t out[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    int rc = load_api(); if (rc) return rc;
    ncclUniqueId id;
    NVR_NCCL(g_api.GetUniqueId(&id));
    std::memcpy(out, &id, 128);
    return NVR_OK;
}

int Comm::init(const uint8_t idb[128], int nr, int rk) {
    int rc = load_api(); if (rc) return rc;
    ncclUniqueId id; std::memcpy(&id, idb, 128);
    ncclComm_t c = nullptr;
    NVR_NCCL(g_api.CommInitRank(&c, nr, id, rk));
    comm = c; nranks = nr; rank = rk;
    if (const char *e = std::getenv("NVR_TP_FORCE_COMM")) force = (e[0] == '1');
    return NVR_OK;
}

int Comm::all_reduce_sum_f16(void *buf, size_t count, hipStream_t s) {
    NVR_NCCL(g_api.AllReduce(buf, buf, count, ncclFloat16, ncclSum, (ncclComm_t)comm, s));
    return NVR_OK;
}
int Comm::all_gather_bytes(const void *send, void *recv, size_t bytes, hipStream_t s) {
    NVR_NCCL(g_api.AllGather(send, recv, bytes, ncclInt8, (ncclComm_t)comm, s));
    return NVR_OK;
}
void Comm::destroy() {
    if (comm && g_api.CommDestroy) g_api.CommDestroy((ncclComm_t)comm);
    comm = nullptr;
}

}  // namespace nvr

#include "comm.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstdlib>
#include <chrono>
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

// ---- in-process group (comm.h) ---------------------------------------------------------------------------------------
int LocalGroup::rendezvous(int rank, const void *ptr) {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return fail(NVR_ERR_RCCL, "local group: a rank timed out earlier");
    ptrs[(size_t)rank] = ptr;
    const uint64_t gen = generation;
    if (++arrived == nranks) { arrived = 0; ++generation; cv.notify_all(); return NVR_OK; }
    if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen || broken; })) {
        broken = true; cv.notify_all();
        return fail(NVR_ERR_RCCL, "local group: rank %d waited 120 s for its peers", rank);
    }
    return broken ? fail(NVR_ERR_RCCL, "local group: a peer timed out") : NVR_OK;
}
int Comm::init_local(LocalGroup *g, int rk) {
    if (!g || rk < 0 || rk >= g->nranks) return fail(NVR_ERR_INVALID_ARG, "init_local: rank %d of %d", rk, g ? g->nranks : 0);
    local = g; nranks = g->nranks; rank = rk;
    return NVR_OK;
}
int local_sum_f16(const void *const *ptrs, int n, void *out, size_t count, hipStream_t s);     // comm_local.hip

int Comm::all_reduce_sum_f16(void *buf, size_t count, hipStream_t s) {
    if (local) {
        if (local_tmp_bytes < count * 2) {
            if (local_tmp) (void)hipFree(local_tmp);
            NVR_HIP_CHECK(hipMalloc(&local_tmp, count * 2)); local_tmp_bytes = count * 2;
        }
        NVR_HIP_CHECK(hipStreamSynchronize(s));                         // my partial sums are complete
        if (int rc = local->rendezvous(rank, buf)) return rc;           // everyone's are, and their addresses are known
        std::vector<const void *> peers;
        { std::lock_guard<std::mutex> lk(local->m); peers = local->ptrs; }
        if (int rc = local_sum_f16(peers.data(), nranks, local_tmp, count, s)) return rc;
        NVR_HIP_CHECK(hipStreamSynchronize(s));
        if (int rc = local->rendezvous(rank, buf)) return rc;           // every rank has read every input
        NVR_HIP_CHECK(hipMemcpyAsync(buf, local_tmp, count * 2, hipMemcpyDeviceToDevice, s));
        return NVR_OK;
    }
    NVR_NCCL(g_api.AllReduce(buf, buf, count, ncclFloat16, ncclSum, (ncclComm_t)comm, s));
    return NVR_OK;
}
int Comm::all_gather_bytes(const void *send, void *recv, size_t bytes, hipStream_t s) {
    if (local) {
        NVR_HIP_CHECK(hipStreamSynchronize(s));
        if (int rc = local->rendezvous(rank, send)) return rc;
        std::vector<const void *> peers;
        { std::lock_guard<std::mutex> lk(local->m); peers = local->ptrs; }
        for (int r = 0; r < nranks; ++r)
            NVR_HIP_CHECK(hipMemcpyAsync((char *)recv + (size_t)r * bytes, peers[(size_t)r], bytes, hipMemcpyDeviceToDevice, s));
        NVR_HIP_CHECK(hipStreamSynchronize(s));
        return local->rendezvous(rank, send);                           // nobody reuses its send buffer before all copies are done
    }
    NVR_NCCL(g_api.AllGather(send, recv, bytes, ncclInt8, (ncclComm_t)comm, s));
    return NVR_OK;
}
void Comm::destroy() {
    if (comm && g_api.CommDestroy) g_api.CommDestroy((ncclComm_t)comm);
    comm = nullptr;
    if (local_tmp) { (void)hipFree(local_tmp); local_tmp = nullptr; local_tmp_bytes = 0; }
    local = nullptr;
}

}  // namespace nvr

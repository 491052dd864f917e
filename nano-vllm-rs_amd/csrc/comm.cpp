#include "comm.h"
#include <dlfcn.h>
#include <algorithm>
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
    if (g->use_p2p && g->nranks > 1) {
        if (int rc = p2p_alloc(g->nranks, rk)) return rc;
        std::lock_guard<std::mutex> lk(g->m);
        g->arenas[rk] = arena;
        ++g->registered;
    }
    return NVR_OK;
}
// in-process group: the arenas of all ranks are known once every rank has attached (before the first step)
static int local_p2p_ready(Comm &c) {
    if (c.p2p_ready || !c.local || !c.local->use_p2p || c.nranks < 2) return NVR_OK;
    std::lock_guard<std::mutex> lk(c.local->m);
    if (c.local->registered < c.nranks) return fail(NVR_ERR_RCCL, "local group: %d of %d ranks attached", c.local->registered, c.nranks);
    return c.p2p_attach_ptrs(c.local->arenas);
}
int Comm::prepare() { return local_p2p_ready(*this); }

// ---- peer-to-peer arenas (kernels/comm_p2p.hip) ----------------------------------------------------------------------
static size_t p2p_flags_offset() { return 2 * 8 * Comm::kP2PSlotBytes; }
static size_t p2p_gslots_offset() { return p2p_flags_offset() + 1024; }                       // flags: 2*8*4 words = 256 B
static size_t p2p_gflags_offset() { return p2p_gslots_offset() + 2 * 8 * (size_t)P2P_GATHER_BYTES; }
static size_t p2p_arena_size() { return p2p_gflags_offset() + 1024; }

int Comm::p2p_alloc(int nr, int rk) {
    if (arena) return NVR_OK;
    if (nr < 2 || nr > 8 || rk < 0 || rk >= nr) return fail(NVR_ERR_INVALID_ARG, "p2p_alloc: rank %d of %d", rk, nr);
    nranks = nr; rank = rk;
    arena_bytes = p2p_arena_size();
    // fine-grained device memory: peers' system-scope stores and my polls of them must not sit in a non-coherent cache
    if (hipExtMallocWithFlags(&arena, arena_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        NVR_HIP_CHECK(hipMalloc(&arena, arena_bytes));
    }
    NVR_HIP_CHECK(hipMemset(arena, 0, arena_bytes));
    // the plain all-reduce's output buffer is allocated here, not on first use: an allocation synchronises the device, and with
    // the in-process group a peer's collective may already be spinning on it, waiting for this rank
    NVR_HIP_CHECK(hipMalloc(&p2p_tmp, kP2PSlotBytes)); p2p_tmp_bytes = kP2PSlotBytes;
    NVR_HIP_CHECK(hipMalloc((void **)&p2p_words, 64));
    const unsigned int init[4] = {1u, 0u, 0u, 0u};                       // epoch 1 (flags start at 0), done 0, err 0
    NVR_HIP_CHECK(hipMemcpy(p2p_words, init, sizeof init, hipMemcpyHostToDevice));
    NVR_HIP_CHECK(hipDeviceSynchronize());
    return NVR_OK;
}
int Comm::p2p_export(uint8_t handle[64]) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!arena) return fail(NVR_ERR_INVALID_ARG, "p2p_export: no arena (p2p_alloc first)");
    hipIpcMemHandle_t hnd;
    NVR_HIP_CHECK(hipIpcGetMemHandle(&hnd, arena));
    std::memcpy(handle, &hnd, 64);
    return NVR_OK;
}
int Comm::p2p_attach_ipc(const uint8_t *handles, const int *devices) {
    if (!arena) return fail(NVR_ERR_INVALID_ARG, "p2p_attach: no arena (p2p_alloc first)");
    int me = 0;
    NVR_HIP_CHECK(hipGetDevice(&me));
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) { peer_arena[r] = arena; continue; }
        if (devices && devices[r] != me) {
            int can = 0;
            NVR_HIP_CHECK(hipDeviceCanAccessPeer(&can, me, devices[r]));
            if (!can) return fail(NVR_ERR_RCCL, "p2p_attach: device %d cannot access device %d", me, devices[r]);
            hipError_t e = hipDeviceEnablePeerAccess(devices[r], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(NVR_ERR_RCCL, "hipDeviceEnablePeerAccess(%d): %s", devices[r], hipGetErrorString(e));
            (void)hipGetLastError();
        }
        hipIpcMemHandle_t hnd; std::memcpy(&hnd, handles + (size_t)r * 64, 64);
        hipError_t e = hipIpcOpenMemHandle(&peer_arena[r], hnd, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(NVR_ERR_RCCL, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e)); }
        peer_opened[r] = true;
    }
    p2p_ready = true;
    return NVR_OK;
}
int Comm::p2p_attach_ptrs(void *const *arenas) {
    if (!arena) return fail(NVR_ERR_INVALID_ARG, "p2p_attach: no arena (p2p_alloc first)");
    for (int r = 0; r < nranks; ++r) peer_arena[r] = r == rank ? arena : arenas[r];
    p2p_ready = true;
    return NVR_OK;
}
static void p2p_fill(const Comm &c, P2PArgs &a) {
    for (int r = 0; r < c.nranks; ++r) {
        a.peer_slots[r] = (p2p_half *)c.peer_arena[r];
        a.peer_flags[r] = (unsigned int *)((char *)c.peer_arena[r] + p2p_flags_offset());
        a.peer_gslots[r] = (char *)c.peer_arena[r] + p2p_gslots_offset();
        a.peer_gflags[r] = (unsigned int *)((char *)c.peer_arena[r] + p2p_gflags_offset());
    }
    a.gslots = (char *)c.arena + p2p_gslots_offset(); a.gflags = (unsigned int *)((char *)c.arena + p2p_gflags_offset());
    a.slots = (p2p_half *)c.arena; a.flags = (unsigned int *)((char *)c.arena + p2p_flags_offset());
    a.nranks = c.nranks; a.rank = c.rank; a.slot_bytes = Comm::kP2PSlotBytes;
    a.epoch = c.p2p_words; a.done = c.p2p_words + 1; a.err = c.p2p_words + 2;
    // a peer that never arrives: far longer than any host-side stall of a live peer (graph instantiation, first-launch code
    // loading), so that a slow rank is waited for and only a dead one sets the error word
    a.timeout_ticks = (unsigned long long)c.timeout_ms * 100000ull;      // wall clock at 100 MHz
    a.fenced = c.p2p_fenced ? 1 : 0;
}
int Comm::all_reduce_add_rmsnorm(const void *in, void *h, const void *wn, float eps, int rows, int Hd, void *out, hipStream_t s) {
    if (!p2p_usable((size_t)rows * Hd)) return fail(NVR_ERR_INVALID_ARG, "all_reduce_add_rmsnorm: peer arenas not attached or message too large");
    P2PArgs a{};
    p2p_fill(*this, a);
    a.in = (const p2p_half *)in; a.count = (size_t)rows * Hd; a.Hd = Hd;
    a.h = (p2p_half *)h; a.wn = (const p2p_half *)wn; a.eps = eps; a.out = (p2p_half *)out;
    return bf16 ? kb::p2p_allreduce_launch(a, rows, s) : k::p2p_allreduce_launch(a, rows, s);
}
int Comm::p2p_check_error(hipStream_t s) {
    if (!p2p_ready) return NVR_OK;
    unsigned int e = 0;
    NVR_HIP_CHECK(hipMemcpyAsync(&e, p2p_words + 2, 4, hipMemcpyDeviceToHost, s));
    NVR_HIP_CHECK(hipStreamSynchronize(s));
    if (e) {
        NVR_HIP_CHECK(hipMemsetAsync(p2p_words + 2, 0, 4, s));
        return fail(NVR_ERR_RCCL, "peer-to-peer all-reduce: a peer did not arrive (epoch %u)", e);
    }
    return NVR_OK;
}

int Comm::all_reduce_sum_f16(void *buf, size_t count, hipStream_t s) {
    if (p2p_usable(count) && count % 4 == 0) {
        // plain one-shot all-reduce: rows of <= 4096 elements into a private buffer (the push workgroups read buf while the
        // reduce workgroups write), then copied back
        int Hd = 4096;
        while (count % (size_t)Hd) Hd /= 2;
        if (Hd >= 4) {
            if (p2p_tmp_bytes < count * 2) {
                if (p2p_tmp) (void)hipFree(p2p_tmp);
                NVR_HIP_CHECK(hipMalloc(&p2p_tmp, count * 2)); p2p_tmp_bytes = count * 2;
            }
            P2PArgs a{};
            p2p_fill(*this, a);
            a.in = (const p2p_half *)buf; a.count = count; a.Hd = Hd; a.out = (p2p_half *)p2p_tmp;
            if (int rc = bf16 ? kb::p2p_allreduce_launch(a, (int)(count / (size_t)Hd), s) : k::p2p_allreduce_launch(a, (int)(count / (size_t)Hd), s)) return rc;
            NVR_HIP_CHECK(hipMemcpyAsync(buf, p2p_tmp, count * 2, hipMemcpyDeviceToDevice, s));
            return NVR_OK;
        }
    }
    if (p2p_ready && !local && !comm && count % 4 == 0) {
        // no other backend: a large message goes through the arenas slot by slot
        const size_t chunk = kP2PSlotBytes / 2;
        for (size_t off = 0; off < count; off += chunk)
            if (int rc = all_reduce_sum_f16((char *)buf + off * 2, std::min(chunk, count - off), s)) return rc;
        return NVR_OK;
    }
    if (local) {
        if (local_tmp_bytes < count * 2) {
            if (local_tmp) (void)hipFree(local_tmp);
            NVR_HIP_CHECK(hipMalloc(&local_tmp, count * 2)); local_tmp_bytes = count * 2;
        }
        NVR_HIP_CHECK(hipStreamSynchronize(s));                         // my partial sums are complete
        if (int rc = local->rendezvous(rank, buf)) return rc;           // everyone's are, and their addresses are known
        std::vector<const void *> peers;
        { std::lock_guard<std::mutex> lk(local->m); peers = local->ptrs; }
        if (int rc = bf16 ? kb::local_sum_16(peers.data(), nranks, local_tmp, count, s) : k::local_sum_16(peers.data(), nranks, local_tmp, count, s)) return rc;
        NVR_HIP_CHECK(hipStreamSynchronize(s));
        if (int rc = local->rendezvous(rank, buf)) return rc;           // every rank has read every input
        NVR_HIP_CHECK(hipMemcpyAsync(buf, local_tmp, count * 2, hipMemcpyDeviceToDevice, s));
        return NVR_OK;
    }
    if (!comm || !g_api.AllReduce)
        return fail(NVR_ERR_RCCL, "all-reduce of %zu fp16 values: no RCCL communicator, and the peer-to-peer arenas take multiples of 4 values only", count);
    NVR_NCCL(g_api.AllReduce(buf, buf, count, bf16 ? ncclBfloat16 : ncclFloat16, ncclSum, (ncclComm_t)comm, s));
    return NVR_OK;
}
int Comm::all_gather_bytes(const void *send, void *recv, size_t bytes, hipStream_t s) {
    if (p2p_ready && bytes % 4 == 0 && bytes <= (size_t)P2P_GATHER_BYTES) {
        P2PArgs a{};
        p2p_fill(*this, a);
        return k::p2p_allgather_launch(a, send, recv, bytes, s);          // bytes: one build serves both types
    }
    if (p2p_ready && bytes % 8 == 0 && (!comm || bytes <= kP2PSlotBytes)) {
        // a record of up to one slot (the f32 partial sums of a float32 rank's decode step: 128 KB at 32 rows x 1024) goes through the all-reduce slots in
        // their all-gather form — one launch, capturable — whatever other backend exists; with no other backend (peer-to-peer arenas only) so does a large
        // record (vocabulary-shard logits of stochastic sampling, greedy batches over 512 rows), one slot-sized piece per launch
        const size_t total = bytes / 2, chunk = kP2PSlotBytes / 2;                  // in fp16-sized units
        for (size_t off = 0; off < total; off += chunk) {
            const size_t cnt = std::min(chunk, total - off);
            int Hd = 4096;
            while (cnt % (size_t)Hd) Hd /= 2;                                        // >= 4: bytes % 8 == 0
            P2PArgs a{};
            p2p_fill(*this, a);
            a.in = (const p2p_half *)send + off; a.count = cnt; a.Hd = Hd;
            a.out = (p2p_half *)recv + off; a.gather_stride = total;
            if (int rc = k::p2p_allreduce_launch(a, (int)(cnt / (size_t)Hd), s)) return rc;     // gather form: elements are moved, never summed
        }
        return NVR_OK;
    }
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
    if (!comm || !g_api.AllGather)
        return fail(NVR_ERR_RCCL, "all-gather of %zu bytes per rank: no RCCL communicator (the peer-to-peer arenas take multiples of 8 bytes)", bytes);
    NVR_NCCL(g_api.AllGather(send, recv, bytes, ncclInt8, (ncclComm_t)comm, s));
    return NVR_OK;
}
int Comm::p2p_reset() {
    if (!arena) return NVR_OK;
    NVR_HIP_CHECK(hipDeviceSynchronize());                               // nothing of mine is still polling or pushing
    NVR_HIP_CHECK(hipMemset((char *)arena + p2p_flags_offset(), 0, 1024));
    NVR_HIP_CHECK(hipMemset((char *)arena + p2p_gflags_offset(), 0, 1024));
    const unsigned int init[4] = {1u, 0u, 0u, 0u};
    NVR_HIP_CHECK(hipMemcpy(p2p_words, init, sizeof init, hipMemcpyHostToDevice));
    NVR_HIP_CHECK(hipDeviceSynchronize());
    return NVR_OK;
}
void Comm::drop_rccl() {
    if (comm && g_api.CommDestroy) g_api.CommDestroy((ncclComm_t)comm);
    comm = nullptr;
}
void Comm::destroy() {
    if (comm && g_api.CommDestroy) g_api.CommDestroy((ncclComm_t)comm);
    comm = nullptr;
    if (local_tmp) { (void)hipFree(local_tmp); local_tmp = nullptr; local_tmp_bytes = 0; }
    local = nullptr;
    for (int r = 0; r < 8; ++r) if (peer_opened[r]) { (void)hipIpcCloseMemHandle(peer_arena[r]); peer_opened[r] = false; }
    if (arena) { (void)hipFree(arena); arena = nullptr; }
    if (p2p_words) { (void)hipFree(p2p_words); p2p_words = nullptr; }
    if (p2p_tmp) { (void)hipFree(p2p_tmp); p2p_tmp = nullptr; p2p_tmp_bytes = 0; }
    p2p_ready = false;
}

}  // namespace nvr

// comm.h — tensor-parallel exchange over RCCL/xGMI, one process per GPU.
// Fills the reference's three TODO communication sites: row-parallel all-reduce
// (src/layers/linear.rs:236-238), embedding all-reduce (src/layers/embed_head.rs:130-139, skipped:
// the embedding table is replicated) and the vocab-shard logits gather (embed_head.rs:321-336).
// librccl is dlopen'ed on first use so that single-GPU runs never load it.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <condition_variable>
#include <mutex>
#include <vector>

namespace nvr {

// In-process stand-in for the communicator (tests / bring-up on a one-GPU box): the N ranks are N runners of ONE process on
// ONE device, each driven by its own host thread.  A collective is a host rendezvous (every rank's stream drained, buffer
// pointers exchanged), a device kernel / copies over the peers' buffers, and a second rendezvous before anyone overwrites
// its input.  Same results as the RCCL path for two ranks (one fp16 rounding of a + b); no performance claim.
struct LocalGroup {
    explicit LocalGroup(int n) : nranks(n), ptrs((size_t)n, nullptr) {}
    int nranks;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;                       // a rank timed out: every later rendezvous fails at once
    std::vector<const void *> ptrs;
    int rendezvous(int rank, const void *ptr); // publishes ptr, returns when all ranks arrived (NVR_ERR_RCCL after 120 s)
    // peer-to-peer mode (default): every rank registers its arena when it attaches; the collectives then run as the product's
    // one-shot kernels (kernels/comm_p2p.hip) on each rank's own stream, with no host rendezvous — the same code path the
    // multi-process ranks take after exchanging hipIpc handles.  use_p2p = false keeps the host-rendezvous collectives.
    bool use_p2p = true;
    void *arenas[8] = {};
    int registered = 0;
};

// One-shot all-reduce over peer-mapped arenas (kernels/comm_p2p.hip).  Arena layout: slots[2][8][slot_bytes] then
// flags[2][8][P2P_PUSH_SPLIT] (uint32 epochs); the private words (epoch, done, err) live in ordinary device memory.
constexpr int P2P_PUSH_SPLIT = 4;
constexpr int P2P_GATHER_BYTES = 4096;                           // all-gather record per rank (gslots[2][8][4096], gflags[2][8])
typedef unsigned short p2p_half;                                 // a 16-bit element (fp16 or bfloat16: the kernel build decides, Comm::bf16 picks the build)
struct P2PArgs {
    p2p_half *peer_slots[8]; unsigned int *peer_flags[8];       // every rank's arena as mapped into THIS process ([rank] = my own)
    p2p_half *slots; unsigned int *flags;                        // my own arena (what my reduce workgroups read)
    char *peer_gslots[8]; unsigned int *peer_gflags[8]; char *gslots; unsigned int *gflags;   // the all-gather part of the arenas
    const p2p_half *in; size_t count;                            // my partial sums [rows][Hd]
    int nranks, rank, Hd;
    size_t slot_bytes;
    unsigned int *epoch, *done, *err;                            // private device words
    unsigned long long timeout_ticks;                            // of the 100 MHz wall clock (s_memrealtime)
    size_t gather_stride;                                        // all-gather form (h == nullptr): out[r * gather_stride + i] = rank r's in[i]; 0 = all-reduce
    p2p_half *h; const p2p_half *wn; float eps;                  // fused residual + RMSNorm (h == nullptr: plain all-reduce into out)
    p2p_half *out;
    int fenced;                                                  // r04's release / acquire fences around the hand-offs (comm_p2p.hip header); 0: fence-free (r05)
};
// kernels/comm_p2p.hip and comm_local.hip exist twice like the other kernels (device_utils.h): sums rounded to fp16 (k) or bfloat16 (kb)
#define NVR_COMM_DECLS \
    int p2p_allreduce_launch(const P2PArgs &a, int rows, hipStream_t s); \
    int p2p_allgather_launch(const P2PArgs &a, const void *send, void *recv, size_t bytes, hipStream_t s); \
    int local_sum_16(const void *const *ptrs, int n, void *out, size_t count, hipStream_t s);
namespace k { NVR_COMM_DECLS }
namespace kb { NVR_COMM_DECLS }
#undef NVR_COMM_DECLS

struct Comm {
    bool bf16 = false;                                           // the 16-bit type of the payloads (Config.dtype = "bfloat16"); set by the runner before any collective
    // ---- peer-to-peer arenas
    static constexpr size_t kP2PSlotBytes = 1u << 20;            // payload per (parity, source): decode-sized messages only
    void *arena = nullptr; size_t arena_bytes = 0;               // mine (fine-grained HBM when the allocator offers it)
    void *peer_arena[8] = {};                                    // every rank's arena in this process' address space
    bool peer_opened[8] = {};                                    // mapped with hipIpcOpenMemHandle (closed in destroy)
    unsigned int *p2p_words = nullptr;                           // epoch, done, err
    void *p2p_tmp = nullptr; size_t p2p_tmp_bytes = 0;           // output of the plain (unfused) all-reduce
    bool p2p_ready = false;
    bool p2p_fenced = false;                                     // Env::p2p_fenced / nvr_runner_p2p_set_fenced: the fenced form of the one-shot collectives
    int p2p_alloc(int nranks_, int rank_);                       // allocate + zero my arena
    int p2p_export(uint8_t handle[64]);                          // hipIpcGetMemHandle of my arena
    int p2p_attach_ipc(const uint8_t *handles /* [nranks][64] */, const int *devices /* [nranks] HIP ordinals */);
    int p2p_attach_ptrs(void *const *arenas);                    // in-process group: plain pointers
    bool p2p_usable(size_t count) const { return p2p_ready && count * 2 <= kP2PSlotBytes; }
    // h <- fp16(h + fp16(sum_ranks in)), out = rmsnorm(h)*wn — one launch; rows*Hd*2 <= kP2PSlotBytes
    int all_reduce_add_rmsnorm(const void *in, void *h, const void *wn, float eps, int rows, int Hd, void *out, hipStream_t s);
    int prepare();                                               // in-process group: attach the peers' arenas once all ranks registered
    int p2p_check_error(hipStream_t s);                          // NVR_ERR_RCCL if a collective since the last check timed out
    int timeout_ms = 20000;                                      // how long a collective waits for a peer (Env::p2p_timeout_ms)
    // After a timed-out collective the ranks' epoch words differ: every rank calls p2p_reset (epoch, flags and error word back to
    // their initial values; the device is drained first), the caller's control plane barriers, and the group is usable again.
    int p2p_reset();
    void drop_rccl();                                            // forget the RCCL communicator (the group agreed not to use it)

    void *lib = nullptr;
    void *comm = nullptr;
    int nranks = 1, rank = 0;
    LocalGroup *local = nullptr;
    void *local_tmp = nullptr; size_t local_tmp_bytes = 0;
    int init_local(LocalGroup *g, int rank);

    static int unique_id(uint8_t out[128]);
    int init(const uint8_t id[128], int nranks, int rank);
    bool force = false;     // NVR_TP_FORCE_COMM=1: enqueue the collectives even with one rank (exercises RCCL on a 1-GPU box)
    bool active() const { return (local != nullptr && nranks > 1) || (comm != nullptr && (nranks > 1 || force)) || (p2p_ready && nranks > 1); }
    int all_reduce_sum_f16(void *buf, size_t count, hipStream_t s);
    int all_gather_bytes(const void *send, void *recv, size_t bytes_per_rank, hipStream_t s);
    void destroy();
};

}  // namespace nvr

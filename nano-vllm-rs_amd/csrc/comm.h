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
};

struct Comm {
    void *lib = nullptr;
    void *comm = nullptr;
    int nranks = 1, rank = 0;
    LocalGroup *local = nullptr;
    void *local_tmp = nullptr; size_t local_tmp_bytes = 0;
    int init_local(LocalGroup *g, int rank);

    static int unique_id(uint8_t out[128]);
    int init(const uint8_t id[128], int nranks, int rank);
    bool force = false;     // NVR_TP_FORCE_COMM=1: enqueue the collectives even with one rank (exercises RCCL on a 1-GPU box)
    bool active() const { return (local != nullptr && nranks > 1) || (comm != nullptr && (nranks > 1 || force)); }
    int all_reduce_sum_f16(void *buf, size_t count, hipStream_t s);
    int all_gather_bytes(const void *send, void *recv, size_t bytes_per_rank, hipStream_t s);
    void destroy();
};

}  // namespace nvr

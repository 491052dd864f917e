// comm.h — tensor-parallel exchange over RCCL/xGMI, one process per GPU.
// Fills the reference's three TODO communication sites: row-parallel all-reduce
// (src/layers/linear.rs:236-238), embedding all-reduce (src/layers/embed_head.rs:130-139, skipped:
// the embedding table is replicated) and the vocab-shard logits gather (embed_head.rs:321-336).
// librccl is dlopen'ed on first use so that single-GPU runs never load it.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nvr {

struct Comm {
    void *lib = nullptr;
    void *comm = nullptr;
    int nranks = 1, rank = 0;

    static int unique_id(uint8_t out[128]);
    int init(const uint8_t id[128], int nranks, int rank);
    bool force = false;     // NVR_TP_FORCE_COMM=1: enqueue the collectives even with one rank (exercises RCCL on a 1-GPU box)
    bool active() const { return comm != nullptr && (nranks > 1 || force); }
    int all_reduce_sum_f16(void *buf, size_t count, hipStream_t s);
    int all_gather_bytes(const void *send, void *recv, size_t bytes_per_rank, hipStream_t s);
    void destroy();
};

}  // namespace nvr

// block_manager.h — paged KV-cache allocator (mirrors BlockManager, reference
// src/engine/block_manager.rs:69-361): ref-counted fixed-size blocks, FIFO free list, xxh64 chained
// prefix-hash index, per-sequence block tables.
//
// Observable behaviour (block ids handed out, table contents, cached-token counts, stats) is
// identical to the reference; the containers are not: the free list is an intrusive doubly linked
// list over block ids (O(1) take-front / push-back / remove-by-id where the reference does
// VecDeque::retain, :130), "used" is a flag per block, and block token content is kept in one
// flat arena sized num_blocks*block_size (the pool is sized for 288 GB of HBM: ~8.8 k blocks of
// 256 tokens for Qwen3-0.6B, 100 k+ at smaller block sizes).
#pragma once
#include <cstdint>
#include <unordered_map>
#include <vector>
#include "sequence.h"

namespace nvr {

class BlockManager {
public:
    BlockManager(size_t num_blocks, size_t block_size);

    static uint64_t compute_hash(const int64_t *tokens, size_t n, bool has_prefix, uint64_t prefix);  // :109
    bool can_allocate(const nvr_seq &s) const { return free_count_ >= s.num_blocks(); }             // :152
    int allocate(nvr_seq &s);                                                                         // :157
    int deallocate(nvr_seq &s);                                                                       // :240
    bool can_append(const nvr_seq &s) const {                                                         // :255
        return (s.len() % block_size_ == 1) ? free_count_ > 0 : true;
    }
    int may_append(nvr_seq &s);                                                                       // :265
    void get_stats(nvr_bm_stats *out) const;                                                          // :307
    bool get_block(size_t id, nvr_block_info *out) const;                                             // :318
    size_t block_size() const { return block_size_; }
    size_t num_blocks() const { return num_blocks_; }
    size_t free_list(int32_t *out, size_t cap) const;

private:
    struct Block {                 // Block, :12-24
        uint32_t ref_count = 0;
        bool has_hash = false;
        bool used = false;
        uint64_t hash = 0;
        uint32_t num_tokens = 0;   // token_ids.len()
        int32_t prev = -1, next = -1;   // free-list links
    };
    int allocate_block(int32_t id);                                  // :126
    int deallocate_block(int32_t id);                                // :137
    int allocate_new_block(bool has_hash, uint64_t hash, const int64_t *tok, size_t n, int32_t *out);  // :222
    void list_remove(int32_t id);
    void list_push_back(int32_t id);
    int64_t *tokens_of(int32_t id) { return arena_.data() + (size_t)id * block_size_; }
    const int64_t *tokens_of(int32_t id) const { return arena_.data() + (size_t)id * block_size_; }

    size_t num_blocks_, block_size_;
    std::vector<Block> blocks_;
    std::vector<int64_t> arena_;
    std::unordered_map<uint64_t, int32_t> hash_to_block_;
    int32_t head_ = -1, tail_ = -1;
    size_t free_count_ = 0, used_count_ = 0;
};

}  // namespace nvr

struct nvr_block_manager { nvr::BlockManager impl; nvr_block_manager(size_t n, size_t b) : impl(n, b) {} };

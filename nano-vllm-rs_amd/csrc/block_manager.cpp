// block_manager.cpp — see block_manager.h.  Line cites are into the reference's
// src/engine/block_manager.rs.
#include "block_manager.h"
#include <cstring>
#include "common.h"
#include "xxh64.h"

namespace nvr {

BlockManager::BlockManager(size_t num_blocks, size_t block_size)
    : num_blocks_(num_blocks), block_size_(block_size), blocks_(num_blocks), arena_(num_blocks * block_size) {
    for (size_t i = 0; i < num_blocks; ++i) list_push_back((int32_t)i);   // free ids 0..n, :96
    hash_to_block_.reserve(num_blocks * 2);
}

uint64_t BlockManager::compute_hash(const int64_t *tokens, size_t n, bool has_prefix, uint64_t prefix) {
    return block_hash(tokens, n, has_prefix, prefix);
}

void BlockManager::list_remove(int32_t id) {
    Block &b = blocks_[id];
    if (b.prev >= 0) blocks_[b.prev].next = b.next; else head_ = b.next;
    if (b.next >= 0) blocks_[b.next].prev = b.prev; else tail_ = b.prev;
    b.prev = b.next = -1;
    --free_count_;
}
void BlockManager::list_push_back(int32_t id) {
    Block &b = blocks_[id];
    b.prev = tail_; b.next = -1;
    if (tail_ >= 0) blocks_[tail_].next = id; else head_ = id;
    tail_ = id;
    ++free_count_;
}

int BlockManager::allocate_block(int32_t id) {                       // :126-134
    Block &b = blocks_[id];
    if (b.ref_count != 0) return fail(NVR_ERR_INVARIANT, "Block %d is not free", id);
    b.ref_count = 1; b.has_hash = false; b.hash = 0; b.num_tokens = 0;   // Block::reset :43-47
    list_remove(id);
    b.used = true; ++used_count_;
    return NVR_OK;
}

int BlockManager::deallocate_block(int32_t id) {                     // :137-149
    Block &b = blocks_[id];
    if (b.ref_count != 0) return fail(NVR_ERR_INVARIANT, "Block %d still has references", id);
    if (b.used) { b.used = false; --used_count_; }
    list_push_back(id);
    if (b.has_hash) {
        auto it = hash_to_block_.find(b.hash);
        if (it != hash_to_block_.end() && it->second == id) hash_to_block_.erase(it);
    }
    return NVR_OK;
}

int BlockManager::allocate_new_block(bool has_hash, uint64_t hash, const int64_t *tok, size_t n, int32_t *out) {  // :222-237
    if (head_ < 0) return fail(NVR_ERR_NO_FREE_BLOCKS, "No free blocks available");
    int32_t id = head_;
    int rc = allocate_block(id);
    if (rc) return rc;
    Block &b = blocks_[id];
    if (n) std::memcpy(tokens_of(id), tok, n * sizeof(int64_t));
    b.num_tokens = (uint32_t)n;
    if (has_hash) { b.has_hash = true; b.hash = hash; hash_to_block_[hash] = id; }
    *out = id;
    return NVR_OK;
}

int BlockManager::allocate(nvr_seq &s) {                             // :157-219
    if (!s.block_table.empty()) return fail(NVR_ERR_ALREADY_ALLOCATED, "Sequence already has allocated blocks");
    if (!can_allocate(s)) return fail(NVR_ERR_NO_FREE_BLOCKS, "Not enough free blocks to allocate sequence");
    bool has_prefix = false, cache_miss = false;
    uint64_t prefix = 0;
    const size_t nb = s.num_blocks();
    s.block_table.reserve(nb + 1);
    for (size_t bi = 0; bi < nb; ++bi) {
        const int64_t *tok; size_t n = s.block_tokens(bi, &tok);
        bool full = (n == block_size_);                              // only full blocks are hashed :173-177
        uint64_t h = full ? block_hash(tok, n, has_prefix, prefix) : 0;
        int32_t id = -1;
        bool hit = false;
        if (full && !cache_miss) {
            auto it = hash_to_block_.find(h);
            if (it != hash_to_block_.end()) {
                const Block &e = blocks_[it->second];
                // hash-collision guard: stored tokens must equal the block's tokens :184
                if (e.num_tokens == n && std::memcmp(tokens_of(it->second), tok, n * sizeof(int64_t)) == 0) {
                    hit = true; id = it->second;
                }
            }
        }
        if (hit) {
            s.num_cached_tokens += block_size_;                      // :187
            if (blocks_[id].used) blocks_[id].ref_count += 1;        // :189-192
            else { int rc = allocate_block(id); if (rc) return rc; } // :193-197 (unreachable, SURVEY A-4)
        } else {
            cache_miss = true;                                       // :200,205,210
            int rc = allocate_new_block(full, h, tok, n, &id);
            if (rc) return rc;
        }
        s.block_table.push_back(id);
        has_prefix = full; prefix = h;                               // prefix_hash = current_hash :215
    }
    return NVR_OK;
}

int BlockManager::deallocate(nvr_seq &s) {                           // :240-252
    for (size_t i = s.block_table.size(); i-- > 0;) {
        int32_t id = s.block_table[i];
        Block &b = blocks_[id];
        if (b.ref_count == 0) return fail(NVR_ERR_INVARIANT, "Cannot remove reference from block with zero refs");
        if (--b.ref_count == 0) { int rc = deallocate_block(id); if (rc) return rc; }
    }
    s.num_cached_tokens = 0;
    s.block_table.clear();
    return NVR_OK;
}

int BlockManager::may_append(nvr_seq &s) {                           // :265-304
    if (s.block_table.empty()) return fail(NVR_ERR_NOT_ALLOCATED, "Sequence has no allocated blocks");
    const size_t last_idx = s.block_table.size() - 1;
    const int32_t last_id = s.block_table[last_idx];
    const size_t r = s.len() % block_size_;
    if (r == 1) {
        if (blocks_[last_id].has_hash) {                             // previous block full and hashed :276
            if (head_ < 0) return fail(NVR_ERR_NO_FREE_BLOCKS, "No free blocks for append");
            int32_t id = head_;
            int rc = allocate_block(id);
            if (rc) return rc;
            s.block_table.push_back(id);
        }
    } else if (r == 0) {
        Block &lb = blocks_[last_id];
        if (!lb.has_hash) {                                          // block just completed :287-300
            const int64_t *tok; size_t n = s.block_tokens(s.num_blocks() - 1, &tok);
            bool has_prefix = false; uint64_t prefix = 0;
            if (s.block_table.size() > 1) {
                const Block &pb = blocks_[s.block_table[last_idx - 1]];
                has_prefix = pb.has_hash; prefix = pb.hash;
            }
            uint64_t h = block_hash(tok, n, has_prefix, prefix);
            std::memcpy(tokens_of(last_id), tok, n * sizeof(int64_t));
            lb.num_tokens = (uint32_t)n; lb.has_hash = true; lb.hash = h;
            hash_to_block_[h] = last_id;
        }
    }
    return NVR_OK;
}

void BlockManager::get_stats(nvr_bm_stats *o) const {
    o->total_blocks = num_blocks_; o->free_blocks = free_count_; o->used_blocks = used_count_;
    o->cached_blocks = hash_to_block_.size(); o->block_size = block_size_;
}
bool BlockManager::get_block(size_t id, nvr_block_info *o) const {
    if (id >= num_blocks_) return false;
    const Block &b = blocks_[id];
    o->block_id = id; o->ref_count = b.ref_count; o->has_hash = b.has_hash; o->hash = b.hash; o->num_tokens = b.num_tokens;
    return true;
}
size_t BlockManager::free_list(int32_t *out, size_t cap) const {
    size_t n = 0;
    for (int32_t i = head_; i >= 0 && n < cap; i = blocks_[i].next) out[n++] = i;
    return free_count_;
}

}  // namespace nvr

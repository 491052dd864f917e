// sequence.h — request state (mirrors Sequence, reference src/engine/sequence.rs:50-237).
// Unlike the reference the scheduler never clones a Sequence (scheduler.rs:164,215): queues hold
// pointers to one live object, so a step costs O(1) per sequence instead of O(tokens).
#pragma once
#include <atomic>
#include <cstdint>
#include <vector>
#include "../../include/nvr.h"

struct nvr_seq {
    uint64_t seq_id = 0;
    int32_t status = NVR_SEQ_WAITING;
    std::vector<int64_t> token_ids;
    int64_t last_token = 0;
    size_t num_tokens = 0;
    size_t num_prompt_tokens = 0;
    size_t num_cached_tokens = 0;
    std::vector<int32_t> block_table;
    nvr_sampling_params sampling{};
    size_t block_size = 256;
    bool owned_by_scheduler = false;
    bool in_running = false;               // member of the scheduler's running queue (kept by every queue operation: O(1) membership in postprocess)
    // chunked prefill (extension A-23; the reference schedules whole sequences, scheduler.rs:135-138): tokens whose K/V are in
    // the cache, and the token range [chunk_start, chunk_start + chunk_len) the sequence contributes to the step it is in
    size_t num_computed_tokens = 0, chunk_start = 0, chunk_len = 0;
    bool chunk_is_partial() const { return chunk_start + chunk_len < num_tokens; }

    size_t len() const { return num_tokens; }                                        // :104
    size_t num_completion_tokens() const { return num_tokens - num_prompt_tokens; }  // :135
    size_t num_blocks() const { return (num_tokens + block_size - 1) / block_size; } // :157
    size_t last_block_num_tokens() const {                                           // :167
        size_t r = num_tokens % block_size;
        return (r == 0 && num_tokens > 0) ? block_size : r;
    }
    // number of tokens in block `idx` and pointer to its first token (get_block_tokens :177)
    size_t block_tokens(size_t idx, const int64_t **p) const {
        size_t start = idx * block_size;
        if (start >= num_tokens) { *p = nullptr; return 0; }
        size_t end = (idx + 1) * block_size;
        if (end > num_tokens) end = num_tokens;
        *p = token_ids.data() + start;
        return end - start;
    }
    void append_token(int64_t t) { token_ids.push_back(t); last_token = t; ++num_tokens; }  // :150
    bool should_stop(bool has_eos, int64_t eos) const {                                     // :189
        if (num_completion_tokens() >= sampling.max_tokens) return true;
        if (!sampling.ignore_eos && has_eos && last_token == eos) return true;
        return false;
    }
    void preempt() { status = NVR_SEQ_PREEMPTED; block_table.clear(); num_cached_tokens = 0; num_computed_tokens = 0; chunk_start = chunk_len = 0; }  // :213
};

namespace nvr {
extern std::atomic<uint64_t> g_sequence_counter;  // SEQUENCE_COUNTER, sequence.rs:12
}

// scheduler.cpp — see scheduler.h.  Line cites are into the reference's src/engine/scheduler.rs.
#include "scheduler.h"
#include <algorithm>
#include "common.h"

namespace nvr {

Scheduler::Scheduler(const nvr_config &cfg)
    : max_num_seqs_(cfg.max_num_seqs), max_num_batched_tokens_(cfg.max_num_batched_tokens),
      has_eos_(cfg.has_eos != 0), eos_(cfg.eos_token_id), chunked_(cfg.enable_chunked_prefill != 0),
      bm_(new nvr_block_manager(cfg.num_kvcache_blocks >= 0 ? (size_t)cfg.num_kvcache_blocks : 1000,  // :71-74
                                cfg.kvcache_block_size)) {}

Scheduler::~Scheduler() {
    for (auto *s : waiting_) delete s;
    for (auto *s : running_) delete s;
    for (auto *s : finished_) delete s;
}

void Scheduler::add_sequence(nvr_seq *s) {                           // :93-98
    s->status = NVR_SEQ_WAITING;
    s->owned_by_scheduler = true;
    waiting_.push_back(s);
    stats_.total_sequences += 1;
    update_stats();
}

int Scheduler::schedule(std::vector<nvr_seq *> &out, bool *is_prefill) {   // :103-116
    out.clear();
    int rc = NVR_OK;
    if (try_schedule_prefill(out, &rc)) {
        stats_.prefill_batches += 1;
        double n = (double)stats_.prefill_batches;                   // :283-288
        stats_.avg_prefill_batch_size = (stats_.avg_prefill_batch_size * (n - 1.0) + (double)out.size()) / n;
        *is_prefill = true;
        return NVR_OK;
    }
    if (rc) return rc;
    rc = try_schedule_decode(out);
    if (rc) return rc;
    stats_.decode_batches += 1;
    double n = (double)stats_.decode_batches;                        // :291-296
    stats_.avg_decode_batch_size = (stats_.avg_decode_batch_size * (n - 1.0) + (double)out.size()) / n;
    *is_prefill = false;
    return NVR_OK;
}

// Extension A-23 — intra-sequence chunked prefill on top of :119-168 (twin of oracle/engine_oracle.py::_try_schedule_prefill_chunked):
// the head of the waiting queue is scheduled for min(remaining, budget left) tokens instead of being held back until its whole
// remainder fits the token budget.  Its blocks are allocated for the whole prompt when its first chunk is scheduled
// (can_allocate / allocate as :141-149); a sequence whose prompt is not finished stays at the FRONT of the waiting queue with its
// blocks and closes the batch; its last chunk moves it to running like :152-165.
bool Scheduler::try_schedule_prefill_chunked(std::vector<nvr_seq *> &out, int *rc) {
    if (waiting_.empty()) return false;
    size_t num_seqs = 0, num_batched_tokens = 0;
    BlockManager &bm = bm_->impl;
    std::vector<nvr_seq *> done;
    while (!waiting_.empty()) {
        nvr_seq *s = waiting_.front();
        if (num_seqs >= max_num_seqs_) break;
        if (num_batched_tokens >= max_num_batched_tokens_) break;
        const size_t budget_left = max_num_batched_tokens_ - num_batched_tokens;
        if (s->block_table.empty()) {                                // first chunk: blocks for the whole prompt
            if (!bm.can_allocate(*s)) break;
            int r = bm.allocate(*s);
            if (r) { *rc = r; return false; }
            s->num_computed_tokens = 0;
        }
        const size_t remaining = s->len() - s->num_computed_tokens;
        const size_t chunk = std::min(remaining, budget_left);
        s->chunk_start = s->num_computed_tokens; s->chunk_len = chunk;
        num_seqs += 1;
        num_batched_tokens += chunk;
        out.push_back(s);
        if (chunk < remaining) break;                                // budget exhausted inside this prompt
        waiting_.pop_front();
        s->status = NVR_SEQ_RUNNING;
        done.push_back(s);
    }
    if (out.empty()) return false;
    for (nvr_seq *s : done) { running_.push_back(s); s->in_running = true; }
    return true;
}

bool Scheduler::try_schedule_prefill(std::vector<nvr_seq *> &out, int *rc) {   // :119-168
    if (chunked_) return try_schedule_prefill_chunked(out, rc);
    if (waiting_.empty()) return false;
    size_t num_seqs = 0, num_batched_tokens = 0;
    BlockManager &bm = bm_->impl;
    while (!waiting_.empty()) {
        nvr_seq *s = waiting_.front();
        if (num_seqs >= max_num_seqs_) break;                        // :131
        size_t seq_tokens = s->len() - s->num_cached_tokens;         // :135
        if (num_batched_tokens + seq_tokens > max_num_batched_tokens_) break;
        if (!bm.can_allocate(*s)) break;                             // :141
        waiting_.pop_front();
        int r = bm.allocate(*s);                                     // :149
        if (r) { waiting_.push_front(s); *rc = r; return false; }
        num_seqs += 1;
        num_batched_tokens += seq_tokens;
        s->status = NVR_SEQ_RUNNING;
        s->chunk_start = 0; s->chunk_len = s->len();
        out.push_back(s);
    }
    if (out.empty()) return false;
    for (nvr_seq *s : out) { running_.push_back(s); s->in_running = true; }   // :163-165
    return true;
}

int Scheduler::try_schedule_decode(std::vector<nvr_seq *> &out) {    // :171-223 (intent per SURVEY A-16)
    BlockManager &bm = bm_->impl;
    size_t num_seqs = 0;
    std::vector<nvr_seq *> reschedule;
    // (an error return leaves the popped sequences outside the queue, as the reference's early `?` returns do: their membership flag follows)
    auto lost = [&](nvr_seq *cur, int rc) { cur->in_running = false; for (nvr_seq *q : out) q->in_running = false; for (nvr_seq *q : reschedule) q->in_running = false; return rc; };
    while (!running_.empty()) {
        nvr_seq *s = running_.front(); running_.pop_front();
        if (num_seqs >= max_num_seqs_) { reschedule.push_back(s); continue; }   // :179-182
        bool self_preempted = false;
        while (!bm.can_append(*s)) {                                 // :185-198
            int rc;
            if (!running_.empty()) { nvr_seq *v = running_.back(); running_.pop_back(); rc = preempt_sequence(v); }
            else if (!out.empty()) { nvr_seq *v = out.back(); out.pop_back(); rc = preempt_sequence(v); }
            else { rc = preempt_sequence(s); self_preempted = true; }
            if (rc) return lost(s, rc);
            if (self_preempted) break;
        }
        if (!self_preempted && bm.can_append(*s)) {                  // :201-205
            num_seqs += 1;
            int rc = bm.may_append(*s);
            if (rc) return lost(s, rc);
            s->chunk_start = s->len() - 1; s->chunk_len = 1;
            out.push_back(s);
        }
    }
    // running = scheduled ++ rescheduled (:208-216)
    for (size_t i = reschedule.size(); i-- > 0;) running_.push_front(reschedule[i]);
    for (size_t i = out.size(); i-- > 0;) running_.push_front(out[i]);
    if (out.empty()) return fail(NVR_ERR_NOTHING_TO_SCHEDULE, "No sequences could be scheduled for decode");
    return NVR_OK;
}

int Scheduler::preempt_sequence(nvr_seq *s) {                        // :226-231
    s->status = NVR_SEQ_PREEMPTED; s->in_running = false;             // (every caller has taken it out of running_)
    int rc = bm_->impl.deallocate(*s);
    s->num_computed_tokens = 0; s->chunk_start = s->chunk_len = 0;  // recompute-style preemption: nothing of it is in the cache any more
    waiting_.push_front(s);
    stats_.preemptions += 1;
    return rc;
}

int Scheduler::postprocess(nvr_seq *const *seqs, const int64_t *token_ids, size_t n) {   // :234-257
    // (the reference's length check, :235-237, is enforced at the ABI: one n for both arrays)
    for (size_t i = 0; i < n; ++i) {
        nvr_seq *s = seqs[i];
        // A-23: a prefill chunk that does not finish its prompt yields no token.  Only a sequence THIS scheduler stamped with a
        // partial chunk can be in that state (chunked prefill on, not yet moved to running: status WAITING or PREEMPTED); every other sequence
        // gets its token appended exactly as the reference does (:240-242), whatever its chunk fields hold.
        if (chunked_ && s->status != NVR_SEQ_RUNNING && s->status != NVR_SEQ_FINISHED && s->chunk_len > 0 && s->chunk_start + s->chunk_len < s->len()) {
            s->num_computed_tokens = s->chunk_start + s->chunk_len;
            continue;
        }
        s->num_computed_tokens = s->len();
        s->append_token(token_ids[i]);
        if (s->should_stop(has_eos_, eos_)) {
            s->status = NVR_SEQ_FINISHED;
            int rc = bm_->impl.deallocate(*s);
            if (rc) return rc;
            // remove_from_running :260-262.  Membership is a flag kept with every queue operation (the reference scans the queue by seq_id for
            // every sequence of every step, :260-274: O(n^2) per step, 0.1 ms at 512 sequences); batches finish front to back
            if (s->in_running) {
                if (!running_.empty() && running_.front() == s) running_.pop_front();
                else { auto it = std::find(running_.begin(), running_.end(), s); if (it != running_.end()) running_.erase(it); }
                s->in_running = false;
            }
            finished_.push_back(s);
            stats_.finished_sequences += 1;
        } else if (!s->in_running) {
            running_.push_back(s); s->in_running = true;             // update_running_sequence fallback :272-273
        }
    }
    update_stats();
    return NVR_OK;
}

void Scheduler::abort_batch(nvr_seq *const *seqs, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        nvr_seq *s = seqs[i];
        auto it = running_.end();
        if (s->in_running) { it = std::find(running_.begin(), running_.end(), s); if (it != running_.end()) running_.erase(it); s->in_running = false; }
        it = std::find(waiting_.begin(), waiting_.end(), s);          // a partially prefilled prompt lives at the front of waiting
        if (it != waiting_.end()) waiting_.erase(it);
        if (!s->block_table.empty()) (void)bm_->impl.deallocate(*s);
        s->status = NVR_SEQ_FINISHED;
        finished_.push_back(s);
        stats_.finished_sequences += 1;
    }
    update_stats();
}

void Scheduler::preempt_all() {                                      // :314-319
    std::vector<nvr_seq *> seqs(running_.begin(), running_.end());
    running_.clear();
    for (nvr_seq *s : seqs) preempt_sequence(s);
    for (nvr_seq *s : waiting_)                                       // A-23: a partially prefilled prompt gives its blocks back too
        if (!s->block_table.empty()) { (void)bm_->impl.deallocate(*s); s->num_computed_tokens = 0; }
}

double Scheduler::memory_pressure() const {                          // :322-329
    nvr_bm_stats st; bm_->impl.get_stats(&st);
    return st.total_blocks == 0 ? 0.0 : 1.0 - (double)st.free_blocks / (double)st.total_blocks;
}

size_t Scheduler::take_finished(nvr_seq **out, size_t cap) {
    size_t n = std::min(cap, finished_.size());
    for (size_t i = 0; i < n; ++i) { out[i] = finished_[i]; out[i]->owned_by_scheduler = false; }
    finished_.erase(finished_.begin(), finished_.begin() + n);
    return n;
}

nvr_seq *Scheduler::take_finished_id(uint64_t seq_id) {
    for (size_t i = 0; i < finished_.size(); ++i)
        if (finished_[i]->seq_id == seq_id) {
            nvr_seq *s = finished_[i];
            finished_.erase(finished_.begin() + (long)i);
            s->owned_by_scheduler = false;
            return s;
        }
    return nullptr;
}

}  // namespace nvr

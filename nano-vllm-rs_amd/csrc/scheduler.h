// scheduler.h — prefill-first continuous batching with recompute-style preemption (mirrors
// Scheduler, reference src/engine/scheduler.rs:13-365).  Sequences are owned by the scheduler
// from add_sequence until they are taken out as finished; the waiting/running queues hold
// pointers (the reference clones whole Sequences in and out of its queues every step,
// scheduler.rs:164,215, and scans them by seq_id, :260-274).
#pragma once
#include <deque>
#include <memory>
#include <vector>
#include "block_manager.h"

namespace nvr {

class Scheduler {
public:
    explicit Scheduler(const nvr_config &cfg);
    ~Scheduler();

    bool is_finished() const { return waiting_.empty() && running_.empty(); }          // :88
    void add_sequence(nvr_seq *s);                                                      // :93
    int schedule(std::vector<nvr_seq *> &out, bool *is_prefill);                        // :103
    int postprocess(nvr_seq *const *seqs, const int64_t *token_ids, size_t n);          // :234
    void preempt_all();                                                                 // :314
    // a batch whose model step failed: its sequences leave the queues with their blocks returned (status FINISHED,
    // parked with the finished ones) so that the next schedule() does not run into the same failure again
    void abort_batch(nvr_seq *const *seqs, size_t n);
    const nvr_sched_stats &stats() const { return stats_; }
    void restore_stats(const nvr_sched_stats &st) { stats_ = st; }                      // a cancelled launch-ahead step (engine.cpp)
    // the next schedule() is a decode step over exactly `seqs`, in this order (nothing waiting, everything running fits)
    bool next_is_decode_of(nvr_seq *const *seqs, size_t n) const {
        if (!waiting_.empty() || running_.size() != n || n > max_num_seqs_) return false;
        for (size_t i = 0; i < n; ++i) if (running_[i] != seqs[i]) return false;
        return true;
    }
    BlockManager &block_manager() { return bm_->impl; }
    const BlockManager &block_manager() const { return bm_->impl; }
    nvr_block_manager *block_manager_handle() { return bm_.get(); }
    size_t max_num_seqs() const { return max_num_seqs_; }
    // a handle the scheduler still schedules (finished sequences may already be with — and destroyed by — the caller: never dereferenced here)
    bool is_live(const nvr_seq *s) const {
        for (const nvr_seq *r : running_) if (r == s) return true;
        for (const nvr_seq *w : waiting_) if (w == s) return true;
        return false;
    }
    size_t waiting_len() const { return waiting_.size(); }
    size_t running_len() const { return running_.size(); }
    double memory_pressure() const;                                                     // :322
    size_t take_finished(nvr_seq **out, size_t cap);
    nvr_seq *take_finished_id(uint64_t seq_id);      // one finished sequence by id (nullptr: not finished / unknown)
    bool has_eos() const { return has_eos_; }
    int64_t eos() const { return eos_; }

private:
    bool try_schedule_prefill(std::vector<nvr_seq *> &out, int *rc);                    // :119
    bool try_schedule_prefill_chunked(std::vector<nvr_seq *> &out, int *rc);            // extension A-23
    int try_schedule_decode(std::vector<nvr_seq *> &out);                               // :171
    int preempt_sequence(nvr_seq *s);                                                   // :226
    void update_stats() { stats_.waiting_sequences = waiting_.size(); stats_.running_sequences = running_.size(); }

    size_t max_num_seqs_, max_num_batched_tokens_;
    bool has_eos_; int64_t eos_;
    bool chunked_ = false;                       // nvr_config.enable_chunked_prefill
    std::unique_ptr<nvr_block_manager> bm_;
    std::deque<nvr_seq *> waiting_, running_;
    std::vector<nvr_seq *> finished_;
    nvr_sched_stats stats_{};
};

}  // namespace nvr

struct nvr_scheduler { nvr::Scheduler impl; explicit nvr_scheduler(const nvr_config &c) : impl(c) {} };

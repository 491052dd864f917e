// engine.h — the hot loop of LLMEngine (reference src/engine/llm_engine.rs:155-197):
// schedule -> execute_model -> sample_tokens -> postprocess, and the callers around it:
// generate / generate_stream with the placeholder char tokenizer (:70-128,:200-230), stats / health (:312-357).
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "model_runner.h"
#include "scheduler.h"
#include "tokenizer.h"

struct nvr_engine {
    nvr_config cfg;
    std::unique_ptr<nvr_scheduler> scheduler;
    std::unique_ptr<nvr_model_runner> runner;
    std::vector<nvr_seq *> batch;
    std::vector<uint64_t> last_ids;
    std::vector<int64_t> last_tokens;
    bool is_running = true;                              // llm_engine.rs:37,353: cleared by shutdown()
    struct HostTrace { bool on = false; double acc[4] = {0, 0, 0, 0}; long n = 0; double pacc[4] = {0, 0, 0, 0}; long pn = 0; ~HostTrace(); } trace;
    // host time of the integer side of every step ("BlockManager / scheduler: host-side, reported as us/step"): microseconds spent
    // inside Scheduler::schedule and Scheduler::postprocess (block manager calls included), and the steps they belong to — always counted
    // (four clock reads per step), read by nvr_engine_host_times
    double host_schedule_us = 0, host_postprocess_us = 0; uint64_t host_steps = 0;
    int step(nvr_step_info *info);
    // launch-ahead of greedy decode steps (nvr_config.async_decode): the step enqueued behind the one being reported
    struct Ahead { bool pending = false; std::vector<nvr_seq *> batch; int parity = 0; nvr_sched_stats stats_before{}; } ahead;
    int step_async(nvr_step_info *info);
    bool can_launch_ahead(const std::vector<nvr_seq *> &cur) const;
    void cancel_ahead();                                 // a request arrived (or the engine shuts down) while a step is in flight
    uint64_t ahead_launched = 0;                         // decode steps enqueued ahead of their predecessor's tokens
    uint64_t ahead_declined = 0;                         // steps whose successor could not be enqueued ahead (they then ran synchronously)
    void abort_last_batch();                             // control-plane abort of the batch the engine scheduled last (tensor-parallel ranks)

    // SequenceOutput storage of the last generate / generate_stream call (sequence.rs:30-47)
    struct SeqOut { uint64_t seq_id = 0; std::string text; std::vector<int64_t> tokens; size_t nprompt = 0; int32_t status = 0; };
    std::vector<SeqOut> gen_store;
    std::vector<nvr_sequence_output> gen_view;
    int add_ids(const int64_t *prompt, size_t n, const nvr_sampling_params *sp, uint64_t *id_out);
    int generate(const std::vector<std::vector<int64_t>> &prompts, const nvr_sampling_params *sp, nvr_stream_fn fn, void *user);
};


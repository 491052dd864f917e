// engine.h — the hot loop of LLMEngine (reference src/engine/llm_engine.rs:155-197):
// schedule -> execute_model -> sample_tokens -> postprocess.  Everything else in that file (tokio
// streaming, builder, health, char tokenizer) is outside the hot path.
#pragma once
#include <memory>
#include <vector>
#include "model_runner.h"
#include "scheduler.h"

struct nvr_engine {
    nvr_config cfg;
    std::unique_ptr<nvr_scheduler> scheduler;
    std::unique_ptr<nvr_model_runner> runner;
    std::vector<nvr_seq *> batch;
    std::vector<uint64_t> last_ids;
    std::vector<int64_t> last_tokens;
    bool is_running = true;                              // llm_engine.rs:37,353: cleared by shutdown()
    int step(nvr_step_info *info);
};

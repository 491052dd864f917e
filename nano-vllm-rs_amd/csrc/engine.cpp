#include "engine.h"
#include "common.h"

int nvr_engine::step(nvr_step_info *info) {                          // LLMEngine::step, llm_engine.rs:155-197
    bool is_prefill = false;
    int rc = scheduler->impl.schedule(batch, &is_prefill);           // :160-166
    if (rc) return rc;
    rc = runner->execute(batch.data(), batch.size(), is_prefill);    // :176-179
    if (rc) return rc;
    last_tokens.resize(batch.size());
    rc = runner->sample(batch.data(), batch.size(), last_tokens.data());   // :182-185
    if (rc) return rc;
    last_ids.resize(batch.size());
    for (size_t i = 0; i < batch.size(); ++i) last_ids[i] = batch[i]->seq_id;
    const uint64_t ntok = (uint64_t)runner->last_tokens;                 // rows fed through the model (a prefill skips cached prefixes)
    const uint64_t fin_before = scheduler->impl.stats().finished_sequences;
    rc = scheduler->impl.postprocess(batch.data(), last_tokens.data(), batch.size());   // :188-189
    if (rc) return rc;
    if (info) {
        info->is_prefill = is_prefill; info->num_seqs = batch.size(); info->num_tokens = ntok;
        info->num_finished = scheduler->impl.stats().finished_sequences - fin_before;
    }
    return NVR_OK;
}

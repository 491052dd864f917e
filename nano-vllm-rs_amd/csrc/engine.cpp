#include "engine.h"
#include "common.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

// NVR_TRACE_HOST=1 (read once when the engine is created): average host time per step spent in schedule / execute (input prep +
// upload + launch) / sample (arg-max merge launch + D2H + wait for the GPU) / postprocess, printed when the engine is destroyed
nvr_engine::HostTrace::~HostTrace() {
    if (on && pn) std::fprintf(stderr, "[nvr host trace] %ld prefill steps: schedule %.1f us, execute %.1f us, sample(+GPU wait) %.1f us, postprocess %.1f us\n",
                               pn, pacc[0] / pn, pacc[1] / pn, pacc[2] / pn, pacc[3] / pn);
    if (on && n) std::fprintf(stderr, "[nvr host trace] %ld decode steps: schedule %.2f us, execute %.2f us, sample(+GPU wait) %.2f us, postprocess %.2f us\n",
                              n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n);
}
namespace {
inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// a scheduler call with its host time added to acc
template <class F> inline int timed(double &acc, F &&f) { const double t = now_us(); const int rc = f(); acc += now_us() - t; return rc; }
}

// ---- launch-ahead (nvr_config.async_decode) -------------------------------------------------------------------------------
// A decode step's batch, positions, slots and block tables follow from the sequence LENGTHS; only its input ids are the
// previous step's sampled tokens, and those can stay on the device.  So when the step after the current one is provably the
// decode step the reference's scheduler would build whatever the tokens turn out to be — nothing waiting, the running queue
// is the current batch, no sequence can stop on this token (max_tokens not reached, EOS ignored or not configured), no block
// boundary (no new block, no block hash, which needs the token values: may_append is then a no-op) — it is scheduled with
// placeholder tokens and enqueued BEFORE the host waits for the current step's tokens; the placeholders are patched when they
// arrive.  The GPU never idles between steps (the host gap was 35-40 us of a 1.53 ms step, profiles/r01 step_gap).
bool nvr_engine::can_launch_ahead(const std::vector<nvr_seq *> &cur) const {
    if (!cfg.async_decode || cur.empty() || !runner->ahead_ok(cur.size())) return false;
    const nvr::Scheduler &sc = scheduler->impl;
    if (!sc.next_is_decode_of(cur.data(), cur.size())) return false;
    const size_t bs = cfg.kvcache_block_size;
    for (const nvr_seq *s : cur) {
        if (s->chunk_is_partial() || s->sampling.temperature != 0.0f) return false;
        if (s->num_completion_tokens() + 1 >= s->sampling.max_tokens) return false;          // this token could be the last
        if (!s->sampling.ignore_eos && sc.has_eos()) return false;                           // ... or an EOS
        const size_t L = s->len() + 1;                                                       // length the next step sees
        if (L % bs == 0 || L % bs == 1 || (int64_t)L > runner->max_pos) return false;        // may_append would hash / allocate
    }
    return true;
}

void nvr_engine::cancel_ahead() {
    if (!ahead.pending) return;
    // the step in flight is abandoned: its K/V rows are the ones a rescheduled step writes again, its tokens are never read;
    // on the host side only the scheduler's counters moved (may_append was a no-op by construction)
    scheduler->impl.restore_stats(ahead.stats_before);
    ahead.pending = false;
}

int nvr_engine::step_async(nvr_step_info *info) {
    bool is_prefill = false;
    int parity;
    if (ahead.pending) {                                                 // this step was enqueued during the previous call
        batch = ahead.batch; parity = ahead.parity; ahead.pending = false;
    } else {
        int rc = timed(host_schedule_us, [&] { return scheduler->impl.schedule(batch, &is_prefill); });
        if (rc) return rc;
        rc = runner->execute(batch.data(), batch.size(), is_prefill);
        if (rc) { scheduler->impl.abort_batch(batch.data(), batch.size()); return rc; }
        bool greedy = runner->ahead_ok(batch.size()) && runner->lm_parts_of_last_step() > 0;
        for (const nvr_seq *s : batch) greedy = greedy && s->sampling.temperature == 0.0f;
        if (!greedy) {                                                   // stochastic rows: the ordinary synchronous tail
            last_tokens.resize(batch.size());
            rc = runner->sample(batch.data(), batch.size(), last_tokens.data());
            if (rc) { scheduler->impl.abort_batch(batch.data(), batch.size()); return rc; }
            parity = -1;
        } else {
            parity = 0;
            rc = runner->sample_launch(batch.data(), batch.size(), parity);
            if (rc) { scheduler->impl.abort_batch(batch.data(), batch.size()); return rc; }
        }
    }
    const uint64_t ntok = is_prefill ? (uint64_t)runner->last_tokens : (uint64_t)batch.size();
    const uint64_t fin_before = scheduler->impl.stats().finished_sequences;
    last_tokens.resize(batch.size());
    ++host_steps;
    if (parity >= 0 && can_launch_ahead(batch)) {
        // schedule the next step on placeholder tokens and enqueue it, THEN wait for this step's tokens.  Whatever happens to the
        // step launched ahead, the CURRENT step ends like a synchronous one: its tokens are collected and patched in.  If the
        // step behind it could not be enqueued (graph cache full, capture failure, ...) the speculative schedule is rolled back
        // (cancel_ahead semantics: may_append was a no-op by construction, only the counters moved) and the next call takes the
        // synchronous path, which reports a persistent failure itself and aborts ITS batch.
        std::vector<int64_t> placeholder(batch.size(), -1);
        int rc = timed(host_postprocess_us, [&] { return scheduler->impl.postprocess(batch.data(), placeholder.data(), batch.size()); });
        if (rc) return rc;                                               // (cannot happen: nobody can stop on this token)
        ahead.stats_before = scheduler->impl.stats();
        bool pf = false;
        int arc = timed(host_schedule_us, [&] { return scheduler->impl.schedule(ahead.batch, &pf); });
        const bool scheduled = arc == NVR_OK;
        if (scheduled && (pf || ahead.batch != batch)) arc = nvr::fail(NVR_ERR_INVARIANT, "launch-ahead: the scheduler built another batch than predicted");
        if (!arc) {
            ahead.parity = parity ^ 1;
            arc = runner->execute_decode_ahead(ahead.batch.data(), ahead.batch.size(), ahead.parity);
            if (!arc) arc = runner->sample_launch(ahead.batch.data(), ahead.batch.size(), ahead.parity);
        }
        std::string ahead_err; int ahead_status = 0;
        if (arc) { ahead_err = nvr::last_error_slot(); ahead_status = nvr::last_status_slot(); }
        rc = runner->sample_wait(batch.size(), parity, last_tokens.data());
        if (rc) {                                                        // the device never delivered: this batch is lost
            if (scheduled) scheduler->impl.restore_stats(ahead.stats_before);
            scheduler->impl.abort_batch(batch.data(), batch.size());
            return rc;
        }
        for (size_t i = 0; i < batch.size(); ++i) {                      // the placeholders become the sampled tokens
            nvr_seq *s = batch[i];
            s->token_ids[s->num_tokens - 1] = last_tokens[i]; s->last_token = last_tokens[i];
        }
        if (!arc) { ahead.pending = true; ++ahead_launched; }
        else {
            if (scheduled) scheduler->impl.restore_stats(ahead.stats_before);
            if (arc == NVR_ERR_INVARIANT) { nvr::last_error_slot() = ahead_err; nvr::last_status_slot() = ahead_status; return arc; }
            ++ahead_declined;                                            // (diagnostic counter; the step itself succeeded)
        }
    } else {
        if (parity >= 0) {
            int rc = runner->sample_wait(batch.size(), parity, last_tokens.data());
            if (rc) { scheduler->impl.abort_batch(batch.data(), batch.size()); return rc; }   // as the synchronous step and the launch-ahead branch
        }
        for (size_t i = 0; i < batch.size(); ++i) if (batch[i]->chunk_is_partial()) last_tokens[i] = -1;
        int rc = timed(host_postprocess_us, [&] { return scheduler->impl.postprocess(batch.data(), last_tokens.data(), batch.size()); });
        if (rc) return rc;
    }
    // the logits accessors (nvr_runner_copy_logits, the borrowed pointer) refer to the step being returned, not to the one launched behind it
    if (parity >= 0) runner->present_step(parity, batch.size());
    last_ids.resize(batch.size());
    for (size_t i = 0; i < batch.size(); ++i) last_ids[i] = batch[i]->seq_id;
    if (info) {
        info->is_prefill = is_prefill; info->num_seqs = batch.size(); info->num_tokens = ntok;
        info->num_finished = (ahead.pending ? ahead.stats_before.finished_sequences : scheduler->impl.stats().finished_sequences) - fin_before;
    }
    return NVR_OK;
}

// A batch that must not go on (a tensor-parallel peer reported a failed collective: every rank drops the same batch): the step
// launched ahead is cancelled and the sequences of the last scheduled batch leave the engine with their blocks returned.
void nvr_engine::abort_last_batch() {
    cancel_ahead();
    std::vector<nvr_seq *> live;
    // by membership, not through the handles: sequences that finished in that step may have been taken (and destroyed) by the caller —
    // step -> take_finished -> learn of the peer's failure -> abort is the natural order of an external control plane
    for (nvr_seq *s : batch) if (scheduler->impl.is_live(s)) live.push_back(s);
    if (!live.empty()) scheduler->impl.abort_batch(live.data(), live.size());
    batch.clear();
}

int nvr_engine::step(nvr_step_info *info) {                          // LLMEngine::step, llm_engine.rs:155-197
    if (cfg.async_decode) return step_async(info);
    bool is_prefill = false;
    const double t0 = trace.on ? now_us() : 0;
    int rc = timed(host_schedule_us, [&] { return scheduler->impl.schedule(batch, &is_prefill); });   // :160-166
    if (rc) return rc;
    ++host_steps;
    const double t1 = trace.on ? now_us() : 0;
    // A model step that fails after schedule() has allocated blocks and moved the batch to running must not wedge the engine
    // (every later step would schedule the same sequences into the same failure): the batch is aborted — blocks returned,
    // sequences parked as finished — and the error is reported once.
    rc = runner->execute(batch.data(), batch.size(), is_prefill);    // :176-179
    if (rc) { scheduler->impl.abort_batch(batch.data(), batch.size()); return rc; }
    const double t2 = trace.on ? now_us() : 0;
    last_tokens.resize(batch.size());
    rc = runner->sample(batch.data(), batch.size(), last_tokens.data());   // :182-185
    if (rc) { scheduler->impl.abort_batch(batch.data(), batch.size()); return rc; }
    const double t3 = trace.on ? now_us() : 0;
    for (size_t i = 0; i < batch.size(); ++i)                            // A-23: a prompt this step did not finish has no token yet
        if (batch[i]->chunk_is_partial()) last_tokens[i] = -1;
    last_ids.resize(batch.size());
    for (size_t i = 0; i < batch.size(); ++i) last_ids[i] = batch[i]->seq_id;
    const uint64_t ntok = (uint64_t)runner->last_tokens;                 // rows fed through the model (a prefill skips cached prefixes)
    const uint64_t fin_before = scheduler->impl.stats().finished_sequences;
    rc = timed(host_postprocess_us, [&] { return scheduler->impl.postprocess(batch.data(), last_tokens.data(), batch.size()); });   // :188-189
    if (rc) return rc;
    if (trace.on) {
        const double t4 = now_us();
        double *a = is_prefill ? trace.pacc : trace.acc;
        a[0] += t1 - t0; a[1] += t2 - t1; a[2] += t3 - t2; a[3] += t4 - t3;
        ++(is_prefill ? trace.pn : trace.n);
    }
    if (info) {
        info->is_prefill = is_prefill; info->num_seqs = batch.size(); info->num_tokens = ntok;
        info->num_finished = scheduler->impl.stats().finished_sequences - fin_before;
    }
    return NVR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Callers around the hot loop: placeholder tokenizer, generate, generate_stream.

int nvr_engine::add_ids(const int64_t *prompt, size_t n, const nvr_sampling_params *sp, uint64_t *id_out) {
    if (sp) { int rc = nvr_sampling_params_validate(sp); if (rc) return rc; }
    if (n == 0) return nvr::fail(NVR_ERR_INVALID_ARG, "empty prompt");
    for (size_t i = 0; i < n; ++i)                                        // a request the model cannot embed is refused here,
        if ((uint64_t)prompt[i] >= (uint64_t)runner->V)                   // not when its batch reaches execute_model
            return nvr::fail(NVR_ERR_INVALID_ARG, "token id %ld at position %zu is outside the vocabulary [0, %ld)", (long)prompt[i], i, (long)runner->V);
    // Admission (SURVEY A-24; the reference holds max_model_len, config.rs:27, and never checks it): a prompt the runner can
    // never execute — longer than max_model_len (RoPE table, block-table width) or than one prefill batch — is refused here,
    // and max_tokens is clamped so that the sequence's last decode step still fits max_model_len.
    if ((int64_t)n > runner->max_pos)
        return nvr::fail(NVR_ERR_INVALID_ARG, "prompt of %zu tokens exceeds max_model_len %ld", n, (long)runner->max_pos);
    if ((int64_t)n > runner->max_tokens && !cfg.enable_chunked_prefill)      // (chunked prefill, A-23, cuts such a prompt into batches)
        return nvr::fail(NVR_ERR_INVALID_ARG, "prompt of %zu tokens exceeds max_num_batched_tokens %ld", n, (long)runner->max_tokens);
    cancel_ahead();                                      // the next step must see this request (prefill-first scheduling, scheduler.rs:103-116)
    nvr_seq *s = nvr_seq_create(prompt, n, sp, cfg.kvcache_block_size);
    if (!s) return NVR_ERR_INVARIANT;
    const uint64_t room = (uint64_t)(runner->max_pos - (int64_t)n) + 1;       // decode step c feeds position n + c - 1 < max_pos
    if (s->sampling.max_tokens > room) s->sampling.max_tokens = room;
    scheduler->impl.add_sequence(s);
    if (id_out) *id_out = s->seq_id;
    return NVR_OK;
}

static void fill_output(nvr_engine::SeqOut &o, const nvr_seq &s) {
    o.seq_id = s.seq_id; o.tokens = s.token_ids; o.nprompt = s.num_prompt_tokens; o.status = s.status;
    nvr::detokenize(o.tokens.data() + o.nprompt, o.tokens.size() - o.nprompt, o.text);
}
static nvr_sequence_output view_of(const nvr_engine::SeqOut &o) {
    nvr_sequence_output v{};
    v.seq_id = o.seq_id; v.text = o.text.c_str(); v.text_len = o.text.size();
    v.token_ids = o.tokens.data(); v.num_tokens = o.tokens.size();
    v.completion_token_ids = o.tokens.data() + o.nprompt;
    v.num_prompt_tokens = o.nprompt; v.num_completion_tokens = o.tokens.size() - o.nprompt; v.status = o.status;
    return v;
}

// generate :70-97 / generate_stream :100-128 over already tokenized prompts
int nvr_engine::generate(const std::vector<std::vector<int64_t>> &prompts, const nvr_sampling_params *sp, nvr_stream_fn fn, void *user) {
    gen_store.clear(); gen_view.clear();
    if (prompts.empty()) return NVR_OK;                                           // :76-78
    // validate everything before the first sequence is queued: a bad prompt must not leave half a request behind
    if (sp) { int rc = nvr_sampling_params_validate(sp); if (rc) return rc; }
    for (size_t i = 0; i < prompts.size(); ++i) {
        if (prompts[i].empty()) return nvr::fail(NVR_ERR_INVALID_ARG, "generate: prompt %zu is empty", i);
        if ((int64_t)prompts[i].size() > runner->max_pos || ((int64_t)prompts[i].size() > runner->max_tokens && !cfg.enable_chunked_prefill))
            return nvr::fail(NVR_ERR_INVALID_ARG, "generate: prompt %zu has %zu tokens (max_model_len %ld, max_num_batched_tokens %ld)", i,
                             prompts[i].size(), (long)runner->max_pos, (long)runner->max_tokens);
        for (int64_t t : prompts[i])
            if ((uint64_t)t >= (uint64_t)runner->V)
                return nvr::fail(NVR_ERR_INVALID_ARG, "generate: prompt %zu holds token id %ld outside the vocabulary [0, %ld)", i, (long)t, (long)runner->V);
    }
    std::vector<uint64_t> ids(prompts.size());
    for (size_t i = 0; i < prompts.size(); ++i) { int rc = add_ids(prompts[i].data(), prompts[i].size(), sp, &ids[i]); if (rc) return rc; }
    SeqOut live;
    while (!scheduler->impl.is_finished()) {                                      // run_inference_loop :131-152
        int rc = step(nullptr);
        if (rc) return rc;
        if (!fn) continue;
        for (nvr_seq *s : batch) {                                                // execute_streaming_step :265-301
            fill_output(live, *s);
            const nvr_sequence_output v = view_of(live);
            if (fn(&v, user)) return NVR_OK;                                      // receiver dropped :250-253
        }
    }
    gen_store.resize(prompts.size());
    for (size_t i = 0; i < prompts.size(); ++i) {
        nvr_seq *s = scheduler->impl.take_finished_id(ids[i]);
        if (!s) return nvr::fail(NVR_ERR_INVARIANT, "generate: sequence %lu did not finish", (unsigned long)ids[i]);
        fill_output(gen_store[i], *s);
        delete s;
    }
    for (const SeqOut &o : gen_store) gen_view.push_back(view_of(o));
    return NVR_OK;
}

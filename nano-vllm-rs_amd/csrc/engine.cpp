#include "engine.h"
#include "common.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

// NVR_TRACE_HOST=1: average host time per decode step spent in schedule / execute (input prep + upload + launch) /
// sample (arg-max merge launch + D2H + wait for the GPU) / postprocess, printed when the engine is destroyed
namespace {
struct HostTrace {
    bool on = std::getenv("NVR_TRACE_HOST") != nullptr;
    double acc[4] = {0, 0, 0, 0}; long n = 0;
    double pacc[4] = {0, 0, 0, 0}; long pn = 0;
    ~HostTrace() {
        if (on && pn) std::fprintf(stderr, "[nvr host trace] %ld prefill steps: schedule %.1f us, execute %.1f us, sample(+GPU wait) %.1f us, postprocess %.1f us\n",
                                   pn, pacc[0] / pn, pacc[1] / pn, pacc[2] / pn, pacc[3] / pn);
        if (on && n) std::fprintf(stderr, "[nvr host trace] %ld decode steps: schedule %.2f us, execute %.2f us, sample(+GPU wait) %.2f us, postprocess %.2f us\n",
                                  n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n);
    }
} g_trace;
inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}

int nvr_engine::step(nvr_step_info *info) {                          // LLMEngine::step, llm_engine.rs:155-197
    bool is_prefill = false;
    const double t0 = g_trace.on ? now_us() : 0;
    int rc = scheduler->impl.schedule(batch, &is_prefill);           // :160-166
    if (rc) return rc;
    const double t1 = g_trace.on ? now_us() : 0;
    rc = runner->execute(batch.data(), batch.size(), is_prefill);    // :176-179
    if (rc) return rc;
    const double t2 = g_trace.on ? now_us() : 0;
    last_tokens.resize(batch.size());
    rc = runner->sample(batch.data(), batch.size(), last_tokens.data());   // :182-185
    if (rc) return rc;
    const double t3 = g_trace.on ? now_us() : 0;
    last_ids.resize(batch.size());
    for (size_t i = 0; i < batch.size(); ++i) last_ids[i] = batch[i]->seq_id;
    const uint64_t ntok = (uint64_t)runner->last_tokens;                 // rows fed through the model (a prefill skips cached prefixes)
    const uint64_t fin_before = scheduler->impl.stats().finished_sequences;
    rc = scheduler->impl.postprocess(batch.data(), last_tokens.data(), batch.size());   // :188-189
    if (rc) return rc;
    if (g_trace.on) {
        const double t4 = now_us();
        double *a = is_prefill ? g_trace.pacc : g_trace.acc;
        a[0] += t1 - t0; a[1] += t2 - t1; a[2] += t3 - t2; a[3] += t4 - t3;
        ++(is_prefill ? g_trace.pn : g_trace.n);
    }
    if (info) {
        info->is_prefill = is_prefill; info->num_seqs = batch.size(); info->num_tokens = ntok;
        info->num_finished = scheduler->impl.stats().finished_sequences - fin_before;
    }
    return NVR_OK;
}

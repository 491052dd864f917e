// nvr_api.cpp — the extern "C" boundary declared in include/nvr.h.  Thin: argument checks, handle
// unwrapping, status codes; no logic of its own.  Nothing throws across the ABI.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <initializer_list>
#include <new>
#include "common.h"
#include "engine.h"
#include "kernels/kernels.h"
#include "kernels/device_utils.h"

namespace k = nvr::k;
uint64_t nvr_weight_key_impl(uint64_t seed, uint64_t tid);
float nvr_weight_scale_impl(double std);

#define NVR_GUARD_BEGIN try {
#define NVR_GUARD_END(ret)                                                                  \
    } catch (const std::bad_alloc &) { nvr::fail(NVR_ERR_INVARIANT, "out of host memory"); return ret; } \
      catch (const std::exception &e) { nvr::fail(NVR_ERR_INVARIANT, "%s", e.what()); return ret; }       \
      catch (...) { nvr::fail(NVR_ERR_INVARIANT, "unknown exception"); return ret; }

extern "C" {

const char *nvr_last_error(void) { return nvr::last_error_slot().c_str(); }
int nvr_last_status(void) { return nvr::last_status_slot(); }
const char *nvr_version(void) { return "nano-vllm-rs_amd 0.1 (gfx950)"; }

// ------------------------------------------------------------------ params / config
// (nvr_sampling_params_* / nvr_config_*: config_api.cpp, host-only)
int nvr_config_runnable(const nvr_config *c);
void nvr_model_config_default(nvr_model_config *m) {                 // qwen3.rs:70-89
    std::memset(m, 0, sizeof *m);
    m->vocab_size = 151936; m->hidden_size = 4096; m->intermediate_size = 11008; m->num_hidden_layers = 32;
    m->num_attention_heads = 32; m->num_key_value_heads = 32; m->head_dim = 0; m->max_position_embeddings = 32768;
    m->rms_norm_eps = 1e-6f; m->rope_theta = 10000.0; m->tie_word_embeddings = 0; m->init_std = 0.02f; m->seed = 0;
}
void nvr_model_config_qwen3_0_6b(nvr_model_config *m) {
    nvr_model_config_default(m);
    m->hidden_size = 1024; m->intermediate_size = 3072; m->num_hidden_layers = 28; m->num_attention_heads = 16;
    m->num_key_value_heads = 8; m->head_dim = 128; m->rope_theta = 1e6; m->tie_word_embeddings = 1;
}
void nvr_model_config_qwen3_8b(nvr_model_config *m) {
    nvr_model_config_default(m);
    m->hidden_size = 4096; m->intermediate_size = 12288; m->num_hidden_layers = 36; m->num_attention_heads = 32;
    m->num_key_value_heads = 8; m->head_dim = 128; m->rope_theta = 1e6; m->tie_word_embeddings = 0;
}
int nvr_model_config_validate(const nvr_model_config *m, uint64_t tp) {   // qwen3.rs:106-124
    if (tp == 0) return nvr::fail(NVR_ERR_INVALID_ARG, "tensor parallel size must be positive");
    if (m->num_attention_heads == 0 || m->num_key_value_heads == 0) return nvr::fail(NVR_ERR_INVALID_ARG, "head counts must be positive");
    if (m->head_dim == 0 && m->hidden_size % m->num_attention_heads != 0)
        return nvr::fail(NVR_ERR_INVALID_ARG, "Hidden size must be divisible by number of attention heads");
    if (m->num_attention_heads % tp != 0) return nvr::fail(NVR_ERR_INVALID_ARG, "Number of attention heads must be divisible by tensor parallel size");
    if (m->num_key_value_heads % tp != 0) return nvr::fail(NVR_ERR_INVALID_ARG, "Number of key-value heads must be divisible by tensor parallel size");
    if (m->intermediate_size % tp != 0) return nvr::fail(NVR_ERR_INVALID_ARG, "Intermediate size must be divisible by tensor parallel size");
    if ((m->num_attention_heads / tp) % (m->num_key_value_heads / tp) != 0)
        return nvr::fail(NVR_ERR_INVALID_ARG, "attention heads must be a multiple of key-value heads");
    return NVR_OK;
}

// ------------------------------------------------------------------ Sequence
nvr_seq_t *nvr_seq_create(const int64_t *prompt, size_t n, const nvr_sampling_params *sp, size_t block_size) {
    NVR_GUARD_BEGIN
    nvr_seq *s = new nvr_seq();
    s->seq_id = nvr::g_sequence_counter.fetch_add(1, std::memory_order_relaxed);   // sequence.rs:85
    s->token_ids.assign(prompt, prompt + n);
    s->last_token = n ? prompt[n - 1] : 0;                           // :87
    s->num_tokens = s->num_prompt_tokens = n;
    if (sp) s->sampling = *sp; else nvr_sampling_params_default(&s->sampling);
    s->block_size = block_size ? block_size : 256;                   // :99 (A-1)
    return s;
    NVR_GUARD_END(nullptr)
}
void nvr_seq_destroy(nvr_seq_t *s) { if (s && !s->owned_by_scheduler) delete s; }
void nvr_seq_reset_id_counter(void) { nvr::g_sequence_counter.store(0); }
uint64_t nvr_seq_id(const nvr_seq_t *s) { return s->seq_id; }
int32_t nvr_seq_status(const nvr_seq_t *s) { return s->status; }
size_t nvr_seq_len(const nvr_seq_t *s) { return s->len(); }
size_t nvr_seq_num_prompt_tokens(const nvr_seq_t *s) { return s->num_prompt_tokens; }
size_t nvr_seq_num_completion_tokens(const nvr_seq_t *s) { return s->num_completion_tokens(); }
size_t nvr_seq_num_cached_tokens(const nvr_seq_t *s) { return s->num_cached_tokens; }
size_t nvr_seq_num_computed_tokens(const nvr_seq_t *s) { return s->num_computed_tokens; }
void nvr_seq_chunk(const nvr_seq_t *s, size_t *start, size_t *len) { if (start) *start = s->chunk_start; if (len) *len = s->chunk_len; }
int64_t nvr_seq_last_token(const nvr_seq_t *s) { return s->last_token; }
size_t nvr_seq_num_blocks(const nvr_seq_t *s) { return s->num_blocks(); }
size_t nvr_seq_last_block_num_tokens(const nvr_seq_t *s) { return s->last_block_num_tokens(); }
void nvr_seq_token_ids(const nvr_seq_t *s, const int64_t **p, size_t *len) { *p = s->token_ids.data(); *len = s->num_tokens; }
void nvr_seq_block_table(const nvr_seq_t *s, const int32_t **p, size_t *len) { *p = s->block_table.data(); *len = s->block_table.size(); }
void nvr_seq_append_token(nvr_seq_t *s, int64_t t) { s->append_token(t); }
int nvr_seq_should_stop(const nvr_seq_t *s, int has_eos, int64_t eos) { return s->should_stop(has_eos != 0, eos) ? 1 : 0; }
void nvr_seq_preempt(nvr_seq_t *s) { s->preempt(); }
void nvr_seq_finish(nvr_seq_t *s) { s->status = NVR_SEQ_FINISHED; }

// ------------------------------------------------------------------ BlockManager
nvr_block_manager_t *nvr_bm_create(size_t num_blocks, size_t block_size) {
    if (num_blocks == 0) { nvr::fail(NVR_ERR_INVARIANT, "Number of blocks must be positive"); return nullptr; }   // :92
    if (block_size == 0) { nvr::fail(NVR_ERR_INVARIANT, "Block size must be positive"); return nullptr; }         // :93
    NVR_GUARD_BEGIN
    return new nvr_block_manager(num_blocks, block_size);
    NVR_GUARD_END(nullptr)
}
void nvr_bm_destroy(nvr_block_manager_t *bm) { delete bm; }
uint64_t nvr_bm_compute_hash(const int64_t *t, size_t n, int has_prefix, uint64_t prefix) {
    return nvr::BlockManager::compute_hash(t, n, has_prefix != 0, prefix);
}
int nvr_bm_can_allocate(const nvr_block_manager_t *bm, const nvr_seq_t *s) { return bm->impl.can_allocate(*s) ? 1 : 0; }
int nvr_bm_allocate(nvr_block_manager_t *bm, nvr_seq_t *s) { NVR_GUARD_BEGIN return bm->impl.allocate(*s); NVR_GUARD_END(NVR_ERR_INVARIANT) }
int nvr_bm_deallocate(nvr_block_manager_t *bm, nvr_seq_t *s) { NVR_GUARD_BEGIN return bm->impl.deallocate(*s); NVR_GUARD_END(NVR_ERR_INVARIANT) }
int nvr_bm_can_append(const nvr_block_manager_t *bm, const nvr_seq_t *s) { return bm->impl.can_append(*s) ? 1 : 0; }
int nvr_bm_may_append(nvr_block_manager_t *bm, nvr_seq_t *s) { NVR_GUARD_BEGIN return bm->impl.may_append(*s); NVR_GUARD_END(NVR_ERR_INVARIANT) }
int nvr_bm_get_stats(const nvr_block_manager_t *bm, nvr_bm_stats *o) { bm->impl.get_stats(o); return NVR_OK; }
int nvr_bm_get_block(const nvr_block_manager_t *bm, size_t id, nvr_block_info *o) {
    return bm->impl.get_block(id, o) ? NVR_OK : nvr::fail(NVR_ERR_INVALID_ARG, "block id %zu out of range", id);
}
size_t nvr_bm_free_list(const nvr_block_manager_t *bm, int32_t *out, size_t cap) { return bm->impl.free_list(out, cap); }

// ------------------------------------------------------------------ Scheduler
nvr_scheduler_t *nvr_sched_create(const nvr_config *cfg) {
    if (cfg->kvcache_block_size == 0) { nvr::fail(NVR_ERR_INVARIANT, "Block size must be positive"); return nullptr; }
    if (cfg->num_kvcache_blocks == 0) { nvr::fail(NVR_ERR_INVARIANT, "Number of blocks must be positive"); return nullptr; }
    NVR_GUARD_BEGIN
    return new nvr_scheduler(*cfg);
    NVR_GUARD_END(nullptr)
}
void nvr_sched_destroy(nvr_scheduler_t *sc) { delete sc; }
int nvr_sched_add_sequence(nvr_scheduler_t *sc, nvr_seq_t *s) {
    if (s->owned_by_scheduler) return nvr::fail(NVR_ERR_INVALID_ARG, "sequence already belongs to a scheduler");
    NVR_GUARD_BEGIN sc->impl.add_sequence(s); return NVR_OK; NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_sched_schedule(nvr_scheduler_t *sc, nvr_seq_t **out, size_t cap, size_t *n, int *is_prefill) {
    NVR_GUARD_BEGIN
    static thread_local std::vector<nvr_seq *> tmp;
    bool pf = false;
    // checked BEFORE scheduling: schedule() allocates blocks and moves sequences to running, a batch that does not fit the
    // caller's array afterwards would be lost to it.  A batch never holds more than min(max_num_seqs, live sequences) entries.
    const size_t need = std::min(sc->impl.max_num_seqs(), sc->impl.waiting_len() + sc->impl.running_len());
    if (cap < need)
        return nvr::fail(NVR_ERR_INVALID_ARG, "schedule: output capacity %zu < min(max_num_seqs %zu, %zu live sequences)", cap,
                         sc->impl.max_num_seqs(), sc->impl.waiting_len() + sc->impl.running_len());
    int rc = sc->impl.schedule(tmp, &pf);
    if (rc) return rc;
    std::memcpy(out, tmp.data(), tmp.size() * sizeof(nvr_seq *));
    *n = tmp.size(); *is_prefill = pf ? 1 : 0;
    return NVR_OK;
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_sched_postprocess(nvr_scheduler_t *sc, nvr_seq_t *const *seqs, const int64_t *toks, size_t n) {
    NVR_GUARD_BEGIN return sc->impl.postprocess(seqs, toks, n); NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_sched_is_finished(const nvr_scheduler_t *sc) { return sc->impl.is_finished() ? 1 : 0; }
void nvr_sched_preempt_all(nvr_scheduler_t *sc) { sc->impl.preempt_all(); }
int nvr_sched_get_stats(const nvr_scheduler_t *sc, nvr_sched_stats *o) { *o = sc->impl.stats(); return NVR_OK; }
int nvr_sched_get_block_stats(const nvr_scheduler_t *sc, nvr_bm_stats *o) { sc->impl.block_manager().get_stats(o); return NVR_OK; }
void nvr_sched_queue_lengths(const nvr_scheduler_t *sc, size_t *w, size_t *r) { *w = sc->impl.waiting_len(); *r = sc->impl.running_len(); }
double nvr_sched_memory_pressure(const nvr_scheduler_t *sc) { return sc->impl.memory_pressure(); }
nvr_block_manager_t *nvr_sched_block_manager(nvr_scheduler_t *sc) { return sc->impl.block_manager_handle(); }
size_t nvr_sched_take_finished(nvr_scheduler_t *sc, nvr_seq_t **out, size_t cap) { return sc->impl.take_finished(out, cap); }

// ------------------------------------------------------------------ ModelRunner
nvr_model_runner_t *nvr_runner_create(const nvr_config *cfg, const nvr_model_config *mc) {
    NVR_GUARD_BEGIN
    if (nvr_config_validate(cfg) || nvr_config_runnable(cfg)) return nullptr;
    nvr_model_runner *r = new nvr_model_runner();
    r->cfg = *cfg; r->mc = *mc;
    if (r->init() != NVR_OK) { delete r; return nullptr; }
    return r;
    NVR_GUARD_END(nullptr)
}
void nvr_runner_destroy(nvr_model_runner_t *r) { delete r; }
int nvr_runner_execute_model(nvr_model_runner_t *r, nvr_seq_t *const *seqs, size_t n, int is_prefill, const float **logits_dev) {
    NVR_GUARD_BEGIN
    int rc = r->execute(seqs, n, is_prefill != 0);
    if (rc == NVR_OK && logits_dev) { rc = r->ensure_logits(); *logits_dev = r->logits; }
    // a tensor-parallel step whose peer never arrived holds zeros in place of the peer's sums: never hand that out as a result
    if (rc == NVR_OK && logits_dev && r->tp > 1) rc = r->comm.p2p_check_error(r->stream);
    return rc;
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_runner_sample_tokens(nvr_model_runner_t *r, nvr_seq_t *const *seqs, size_t n, int64_t *out) {
    NVR_GUARD_BEGIN return r->sample(seqs, n, out); NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_runner_load_tensor(nvr_model_runner_t *r, const char *name, int dtype, const int64_t *shape, int ndim, const void *data) {
    NVR_GUARD_BEGIN return r->load_tensor(name, dtype, shape, ndim, data); NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_runner_copy_weight(nvr_model_runner_t *r, const char *local_name, uint16_t *host_out, size_t cap, int64_t *rows, int64_t *cols) {
    NVR_GUARD_BEGIN return r->copy_weight(local_name, host_out, cap, rows, cols); NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_runner_copy_logits(nvr_model_runner_t *r, float *host_out, size_t rows) {
    if (rows > r->last_rows) return nvr::fail(NVR_ERR_LEN_MISMATCH, "copy_logits: %zu rows requested, %zu available", rows, r->last_rows);
    NVR_HIP_CHECK(hipSetDevice(r->device));
    if (int rc = r->ensure_logits()) return rc;
    if (r->tp > 1) { if (int rc = r->comm.p2p_check_error(r->stream)) return rc; }
    NVR_HIP_CHECK(hipMemcpyAsync(host_out, r->logits, rows * r->Vl * sizeof(float), hipMemcpyDeviceToHost, r->stream));
    NVR_HIP_CHECK(hipStreamSynchronize(r->stream));
    return NVR_OK;
}
uint64_t nvr_runner_num_kvcache_blocks(const nvr_model_runner_t *r) { return (uint64_t)r->num_blocks; }
int nvr_runner_kv_cache(nvr_model_runner_t *r, size_t layer, void **kd, void **vd) {
    if ((int64_t)layer >= r->L) return nvr::fail(NVR_ERR_INVALID_ARG, "layer %zu out of range", layer);
    *kd = r->k_cache(layer); *vd = r->v_cache(layer);
    return NVR_OK;
}
void *nvr_runner_stream(nvr_model_runner_t *r) { return r->stream; }
int nvr_comm_unique_id(uint8_t id_out[128]) { return nvr::Comm::unique_id(id_out); }
int nvr_runner_replay_last_decode_graph(nvr_model_runner_t *r, int n) { NVR_GUARD_BEGIN return r->replay_last_decode_graph(n); NVR_GUARD_END(NVR_ERR_INVARIANT) }
int nvr_runner_comm_selftest(nvr_model_runner_t *r) { NVR_GUARD_BEGIN return r->comm_selftest(); NVR_GUARD_END(NVR_ERR_INVARIANT) }
int nvr_runner_init_comm(nvr_model_runner_t *r, const uint8_t id[128]) {
    NVR_HIP_CHECK(hipSetDevice(r->device));
    int rc = r->comm.init(id, (int)r->tp, (int)r->rank);
    if (rc) return rc;
    // Multi-rank decode steps replay a captured hipGraph like single-rank ones (the one-shot peer-to-peer collectives are plain
    // kernel nodes; ncclAllReduce nodes capture too).  If the capture cannot be built the runner falls back to eager launches
    // (execute()); NVR_TP_GRAPH=0 forces eager.
    return r->tp > 1 ? r->comm_selftest() : NVR_OK;            // also establishes every RCCL connection up front
}
// One-shot peer-to-peer collectives (kernels/comm_p2p.hip): every rank allocates an arena and exports its hipIpc handle; the
// caller's control plane gathers the handles (and the ranks' HIP device ordinals) and hands all of them to every rank.
int nvr_runner_p2p_export(nvr_model_runner_t *r, uint8_t handle[64]) {
    NVR_GUARD_BEGIN
    NVR_HIP_CHECK(hipSetDevice(r->device));
    if (int rc = r->comm.p2p_alloc((int)r->tp, (int)r->rank)) return rc;
    return r->comm.p2p_export(handle);
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_runner_p2p_attach(nvr_model_runner_t *r, const uint8_t *handles, const int32_t *devices) {
    NVR_GUARD_BEGIN
    NVR_HIP_CHECK(hipSetDevice(r->device));
    if (!handles) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_runner_p2p_attach: handles is null");
    std::vector<int> dv;
    if (devices) dv.assign(devices, devices + r->tp);
    if (int rc = r->comm.p2p_attach_ipc(handles, devices ? dv.data() : nullptr)) return rc;
    if (!r->comm.comm && !r->comm.local) { r->comm.nranks = (int)r->tp; r->comm.rank = (int)r->rank; }
    return NVR_OK;
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_runner_p2p_disable(nvr_model_runner_t *r) { r->comm.p2p_ready = false; return NVR_OK; }
int nvr_runner_p2p_active(const nvr_model_runner_t *r) { return r->comm.p2p_ready ? 1 : 0; }
int nvr_runner_p2p_set_fenced(nvr_model_runner_t *r, int32_t on) { NVR_GUARD_BEGIN return r->set_p2p_fenced(on != 0); NVR_GUARD_END(NVR_ERR_INVARIANT) }
int nvr_runner_p2p_fenced(const nvr_model_runner_t *r) { return r->comm.p2p_fenced ? 1 : 0; }
int nvr_runner_p2p_reset(nvr_model_runner_t *r) {
    NVR_GUARD_BEGIN
    NVR_HIP_CHECK(hipSetDevice(r->device));
    if (int rc = r->comm.p2p_reset()) return rc;
    return r->rearm_tickets();
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_runner_comm_drop_rccl(nvr_model_runner_t *r) { r->comm.drop_rccl(); return NVR_OK; }
int64_t nvr_runner_last_shared_prefix_len(const nvr_model_runner_t *r) { return r->facts().shared_len; }
int nvr_runner_last_prefill_kv_source(const nvr_model_runner_t *r) { return r->facts().kv_source; }
int nvr_runner_set_tp_prefill_overlap(nvr_model_runner_t *r, int32_t mode) { r->tp_overlap = mode == 2 ? 2 : mode ? 1 : 0; return NVR_OK; }
int64_t nvr_runner_last_overlap_chunks(const nvr_model_runner_t *r) { return r->facts().overlap_chunks; }
int32_t nvr_runner_last_decode_ragged(const nvr_model_runner_t *r) { return r->facts().ragged ? 1 : 0; }
int64_t nvr_runner_last_shared_prefix_rows(const nvr_model_runner_t *r) { return r->facts().shared_rows; }

// in-process communicator (comm.h LocalGroup): N runners of one process on one device, one host thread each
struct nvr_local_group { nvr::LocalGroup g; explicit nvr_local_group(int n) : g(n) {} };
nvr_local_group_t *nvr_local_group_create(int nranks) {
    NVR_GUARD_BEGIN
    if (nranks < 1 || nranks > 8) { nvr::fail(NVR_ERR_INVALID_ARG, "nvr_local_group_create: %d ranks (1..8)", nranks); return nullptr; }
    return new nvr_local_group(nranks);
    NVR_GUARD_END(nullptr)
}
void nvr_local_group_destroy(nvr_local_group_t *g) { delete g; }
int nvr_local_group_set_p2p(nvr_local_group_t *g, int on) {
    if (!g) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_local_group_set_p2p: group is null");
    if (g->g.registered) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_local_group_set_p2p: ranks are already attached");
    g->g.use_p2p = on != 0;
    return NVR_OK;
}
int nvr_runner_init_comm_local(nvr_model_runner_t *r, nvr_local_group_t *g) {
    NVR_GUARD_BEGIN
    if (!g) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_runner_init_comm_local: group is null");
    if (g->g.nranks != (int)r->tp) return nvr::fail(NVR_ERR_INVALID_ARG, "local group of %d ranks for tensor_parallel_size %ld", g->g.nranks, (long)r->tp);
    NVR_HIP_CHECK(hipSetDevice(r->device));
    int rc = r->comm.init_local(&g->g, (int)r->rank);
    if (rc) return rc;
    r->graphs_disabled = !g->g.use_p2p;        // host-rendezvous collectives synchronise on the host: nothing to capture
    return NVR_OK;
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}

// ------------------------------------------------------------------ Engine
nvr_engine_t *nvr_engine_create(const nvr_config *cfg, const nvr_model_config *mc) {
    NVR_GUARD_BEGIN
    if (nvr_config_validate(cfg) || nvr_config_runnable(cfg)) return nullptr;
    std::unique_ptr<nvr_engine> e(new nvr_engine());
    e->cfg = *cfg;
    e->runner.reset(nvr_runner_create(cfg, mc));
    if (!e->runner) return nullptr;
    nvr_config sc = *cfg;
    sc.num_kvcache_blocks = (int64_t)e->runner->num_blocks;          // one pool size for scheduler and runner
    e->scheduler.reset(new nvr_scheduler(sc));
    e->trace.on = e->runner->env.trace_host;
    return e.release();
    NVR_GUARD_END(nullptr)
}
void nvr_engine_destroy(nvr_engine_t *e) { delete e; }
int nvr_engine_add_request(nvr_engine_t *e, const int64_t *prompt, size_t n, const nvr_sampling_params *sp, uint64_t *id_out) {
    NVR_GUARD_BEGIN
    return e->add_ids(prompt, n, sp, id_out);
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
// ---- text in, SequenceOutput out (llm_engine.rs:70-128,200-230)
int nvr_tokenize(const char *utf8, size_t nbytes, int64_t *out, size_t cap, size_t *n) {
    NVR_GUARD_BEGIN
    if (!utf8 && nbytes) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_tokenize: text is null");
    std::vector<int64_t> ids;
    int rc = nvr::tokenize(utf8, nbytes, ids);
    if (rc) return rc;
    if (n) *n = ids.size();
    if (!out) return NVR_OK;
    if (ids.size() > cap) return nvr::fail(NVR_ERR_LEN_MISMATCH, "nvr_tokenize: %zu tokens, buffer holds %zu", ids.size(), cap);
    std::memcpy(out, ids.data(), ids.size() * sizeof(int64_t));
    return NVR_OK;
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_detokenize(const int64_t *ids, size_t n, char *out, size_t cap, size_t *nbytes) {
    NVR_GUARD_BEGIN
    if (!ids && n) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_detokenize: ids is null");
    std::string t;
    nvr::detokenize(ids, n, t);
    if (nbytes) *nbytes = t.size();
    if (!out) return NVR_OK;
    if (t.size() > cap) return nvr::fail(NVR_ERR_LEN_MISMATCH, "nvr_detokenize: %zu bytes, buffer holds %zu", t.size(), cap);
    std::memcpy(out, t.data(), t.size());
    if (t.size() < cap) out[t.size()] = 0;
    return NVR_OK;
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_engine_add_prompt(nvr_engine_t *e, const char *utf8, size_t nbytes, const nvr_sampling_params *sp, uint64_t *id_out) {
    NVR_GUARD_BEGIN
    std::vector<int64_t> ids;
    int rc = nvr::tokenize(utf8, nbytes, ids);
    if (rc) return rc;
    return e->add_ids(ids.data(), ids.size(), sp, id_out);
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
static int tokenize_all(const char *const *prompts, const size_t *nbytes, size_t n, std::vector<std::vector<int64_t>> &ids) {
    if (n && (!prompts || !nbytes)) return nvr::fail(NVR_ERR_INVALID_ARG, "generate: prompts / nbytes is null");
    ids.resize(n);
    for (size_t i = 0; i < n; ++i) { int rc = nvr::tokenize(prompts[i], nbytes[i], ids[i]); if (rc) return rc; }
    return NVR_OK;
}
static void hand_out(nvr_engine_t *e, const nvr_sequence_output **outs, size_t *nout) {
    if (outs) *outs = e->gen_view.data();
    if (nout) *nout = e->gen_view.size();
}
int nvr_engine_generate(nvr_engine_t *e, const char *const *prompts, const size_t *nbytes, size_t n, const nvr_sampling_params *sp,
                        const nvr_sequence_output **outs, size_t *nout) {
    NVR_GUARD_BEGIN
    if (outs) *outs = nullptr;
    if (nout) *nout = 0;
    std::vector<std::vector<int64_t>> ids;
    int rc = tokenize_all(prompts, nbytes, n, ids);
    if (rc) return rc;
    rc = e->generate(ids, sp, nullptr, nullptr);
    if (rc) return rc;
    hand_out(e, outs, nout);
    return NVR_OK;
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_engine_generate_ids(nvr_engine_t *e, const int64_t *const *prompts, const size_t *lens, size_t n, const nvr_sampling_params *sp,
                            const nvr_sequence_output **outs, size_t *nout) {
    NVR_GUARD_BEGIN
    if (outs) *outs = nullptr;
    if (nout) *nout = 0;
    if (n && (!prompts || !lens)) return nvr::fail(NVR_ERR_INVALID_ARG, "generate: prompts / lens is null");
    std::vector<std::vector<int64_t>> ids(n);
    for (size_t i = 0; i < n; ++i) ids[i].assign(prompts[i], prompts[i] + lens[i]);
    int rc = e->generate(ids, sp, nullptr, nullptr);
    if (rc) return rc;
    hand_out(e, outs, nout);
    return NVR_OK;
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_engine_generate_stream(nvr_engine_t *e, const char *const *prompts, const size_t *nbytes, size_t n, const nvr_sampling_params *sp,
                               nvr_stream_fn fn, void *user) {
    NVR_GUARD_BEGIN
    if (!fn) return nvr::fail(NVR_ERR_INVALID_ARG, "generate_stream: callback is null");
    std::vector<std::vector<int64_t>> ids;
    int rc = tokenize_all(prompts, nbytes, n, ids);
    if (rc) return rc;
    return e->generate(ids, sp, fn, user);
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_engine_step(nvr_engine_t *e, nvr_step_info *info) { NVR_GUARD_BEGIN return e->step(info); NVR_GUARD_END(NVR_ERR_INVARIANT) }
int nvr_engine_is_finished(const nvr_engine_t *e) { return e->scheduler->impl.is_finished() ? 1 : 0; }
nvr_scheduler_t *nvr_engine_scheduler(nvr_engine_t *e) { return e->scheduler.get(); }
nvr_model_runner_t *nvr_engine_runner(nvr_engine_t *e) { return e->runner.get(); }
int nvr_engine_get_stats(nvr_engine_t *e, nvr_engine_stats *o) {                         // llm_engine.rs:312-327
    if (!o) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_engine_get_stats: out is null");
    o->scheduler = e->scheduler->impl.stats();
    if (e->ahead.pending) {                          // a step launched ahead has not been reported yet: its batch is not counted
        const nvr_sched_stats &sb = e->ahead.stats_before;
        o->scheduler.decode_batches = sb.decode_batches; o->scheduler.avg_decode_batch_size = sb.avg_decode_batch_size;
    }
    nvr_bm_stats b{}; e->scheduler->impl.block_manager().get_stats(&b);
    o->total_blocks = b.total_blocks; o->free_blocks = b.free_blocks; o->used_blocks = b.used_blocks;
    o->utilization = b.total_blocks ? (double)b.used_blocks / (double)b.total_blocks * 100.0 : 0.0;   // block_manager.rs:345-351
    o->is_running = e->is_running ? 1 : 0;
    return NVR_OK;
}
int nvr_engine_health_check(nvr_engine_t *e, nvr_health_status *o) {                     // llm_engine.rs:330-342
    if (!o) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_engine_health_check: out is null");
    nvr_engine_stats st{};
    nvr_engine_get_stats(e, &st);
    o->memory_pressure = st.utilization; o->is_healthy = st.utilization < 95.0 ? 1 : 0;
    o->active_sequences = st.scheduler.running_sequences; o->waiting_sequences = st.scheduler.waiting_sequences;
    return NVR_OK;
}
int nvr_engine_shutdown(nvr_engine_t *e) {                                               // llm_engine.rs:345-357
    e->cancel_ahead();
    e->scheduler->impl.preempt_all();
    e->is_running = false;
    return NVR_OK;
}
void nvr_engine_last_step(const nvr_engine_t *e, const uint64_t **ids, const int64_t **toks, size_t *n) {
    *ids = e->last_ids.data(); *toks = e->last_tokens.data(); *n = e->last_ids.size();
}
size_t nvr_engine_take_finished(nvr_engine_t *e, nvr_seq_t **out, size_t cap) { return e->scheduler->impl.take_finished(out, cap); }
int nvr_engine_abort_last_batch(nvr_engine_t *e) { NVR_GUARD_BEGIN e->abort_last_batch(); return NVR_OK; NVR_GUARD_END(NVR_ERR_INVARIANT) }
uint64_t nvr_engine_ahead_declined(const nvr_engine_t *e) { return e->ahead_declined; }
uint64_t nvr_engine_ahead_launched(const nvr_engine_t *e) { return e->ahead_launched; }
void nvr_engine_host_times(const nvr_engine_t *e, double *out3) { out3[0] = e->host_schedule_us; out3[1] = e->host_postprocess_us; out3[2] = (double)e->host_steps; }
size_t nvr_engine_last_batch(const nvr_engine_t *e, nvr_seq_t **out, size_t cap) {
    size_t n = 0;
    for (nvr_seq *s : e->batch) if (e->scheduler->impl.is_live(s) && n < cap) out[n++] = s;
    return n;
}

// ------------------------------------------------------------------ device utilities
int nvr_device_count(int *n) { NVR_HIP_CHECK(hipGetDeviceCount(n)); return NVR_OK; }
int nvr_device_set(int o) { NVR_HIP_CHECK(hipSetDevice(o)); return NVR_OK; }
int nvr_device_name(char *buf, size_t cap) {
    int d = 0; NVR_HIP_CHECK(hipGetDevice(&d));
    hipDeviceProp_t p; NVR_HIP_CHECK(hipGetDeviceProperties(&p, d));
    // (the marketing name comes from a driver table that minimal container images lack: the architecture, CU count and memory always identify the part)
    std::snprintf(buf, cap, "%s (%s, %d CUs, %.0f GiB)", p.name[0] ? p.name : "AMD GPU", p.gcnArchName, p.multiProcessorCount,
                  (double)p.totalGlobalMem / (double)(1ull << 30));
    return NVR_OK;
}
int nvr_device_mem_info(uint64_t *f, uint64_t *t) { size_t a = 0, b = 0; NVR_HIP_CHECK(hipMemGetInfo(&a, &b)); *f = a; *t = b; return NVR_OK; }
int nvr_device_malloc(void **p, size_t bytes) { NVR_HIP_CHECK(hipMalloc(p, bytes ? bytes : 16)); return NVR_OK; }
int nvr_device_free(void *p) { NVR_HIP_CHECK(hipFree(p)); return NVR_OK; }
int nvr_device_memset(void *p, int v, size_t bytes) { NVR_HIP_CHECK(hipMemset(p, v, bytes)); return NVR_OK; }
int nvr_memcpy_h2d(void *d, const void *s, size_t b) { NVR_HIP_CHECK(hipMemcpy(d, s, b, hipMemcpyHostToDevice)); return NVR_OK; }
int nvr_memcpy_d2h(void *d, const void *s, size_t b) { NVR_HIP_CHECK(hipMemcpy(d, s, b, hipMemcpyDeviceToHost)); return NVR_OK; }
int nvr_device_synchronize(void) { NVR_HIP_CHECK(hipDeviceSynchronize()); return NVR_OK; }
int nvr_stream_create(void **s) { hipStream_t st; NVR_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); *s = st; return NVR_OK; }
int nvr_stream_destroy(void *s) { NVR_HIP_CHECK(hipStreamDestroy((hipStream_t)s)); return NVR_OK; }
int nvr_stream_synchronize(void *s) { NVR_HIP_CHECK(hipStreamSynchronize((hipStream_t)s)); return NVR_OK; }
int nvr_event_create(void **e) { hipEvent_t ev; NVR_HIP_CHECK(hipEventCreate(&ev)); *e = ev; return NVR_OK; }
int nvr_event_destroy(void *e) { NVR_HIP_CHECK(hipEventDestroy((hipEvent_t)e)); return NVR_OK; }
int nvr_event_record(void *e, void *s) { NVR_HIP_CHECK(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); return NVR_OK; }
int nvr_stream_wait_event(void *s, void *e) { NVR_HIP_CHECK(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0)); return NVR_OK; }
int nvr_event_elapsed_ms(void *a, void *b, float *ms) {
    NVR_HIP_CHECK(hipEventSynchronize((hipEvent_t)b));
    NVR_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return NVR_OK;
}
// hipGraph capture of the stateless ops below on a caller stream (measurement harnesses: a chain of ops replayed with the
// kernel boundaries of the engine's captured decode step, without a host launch per kernel)
int nvr_graph_capture_begin(void *s) { NVR_HIP_CHECK(hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeThreadLocal)); return NVR_OK; }
int nvr_graph_capture_end(void *s, void **exec) {
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    NVR_HIP_CHECK(hipStreamEndCapture((hipStream_t)s, &g));
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    *exec = ge;
    return NVR_OK;
}
int nvr_graph_launch(void *exec, void *s) { NVR_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)s)); return NVR_OK; }
int nvr_graph_destroy(void *exec) { NVR_HIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)exec)); return NVR_OK; }

// ------------------------------------------------------------------ stateless ops
// The 16-bit type of the stateless entry points' buffers: fp16 (default) or bfloat16 — per calling thread, like nvr_last_error.
// Every kernel exists in both builds (kernels/device_utils.h); KO() picks the namespace per call.
// "float32" (r04): the ops of the reference-precision path (kernels/f32_path.hip) — the nvr_half pointers then address f32 elements; the ops that
// exist only as fused 16-bit kernels (nvr_linear_qkv_rope_store, nvr_linear_silu_mul, nvr_lm_head, the *_tiled / split-k / shared-prefix forms)
// answer NVR_ERR_UNSUPPORTED: the f32 graph runs their unfused parts.
static thread_local bool g_ops_bf16 = false, g_ops_f32 = false;
#define KO(call) (g_ops_bf16 ? nvr::kb::call : nvr::k::call)
#define FP(p) reinterpret_cast<float *>(const_cast<nvr_half *>(p))
#define NO_F32(name) if (g_ops_f32) return nvr::fail(NVR_ERR_UNSUPPORTED, name ": a fused 16-bit kernel; nvr_ops_set_dtype(\"float32\") has its unfused parts")
int nvr_ops_set_dtype(const char *dtype) {
    if (dtype && std::strcmp(dtype, "float16") == 0) { g_ops_bf16 = false; g_ops_f32 = false; return NVR_OK; }
    if (dtype && std::strcmp(dtype, "bfloat16") == 0) { g_ops_bf16 = true; g_ops_f32 = false; return NVR_OK; }
    if (dtype && std::strcmp(dtype, "float32") == 0) { g_ops_bf16 = false; g_ops_f32 = true; return nvr::kf::prepare(); }
    return nvr::fail(NVR_ERR_UNSUPPORTED, "nvr_ops_set_dtype: '%s' (float16 | bfloat16 | float32)", dtype ? dtype : "(null)");
}
const char *nvr_ops_dtype(void) { return g_ops_f32 ? "float32" : g_ops_bf16 ? "bfloat16" : "float16"; }
int nvr_embedding(const int64_t *ids, int64_t T, const nvr_half *E, int64_t Hd, nvr_half *out, void *s) {
    if (g_ops_f32) return nvr::kf::embedding(ids, T, FP(E), Hd, FP(out), (hipStream_t)s);
    return KO(embedding(ids, T, E, Hd, out, (hipStream_t)s));
}
int nvr_rmsnorm(const nvr_half *x, const nvr_half *w, float eps, int64_t T, int64_t Hd, nvr_half *out, void *s) {
    if (g_ops_f32) return nvr::kf::rmsnorm(FP(x), FP(w), eps, T, Hd, FP(out), (hipStream_t)s);
    return KO(rmsnorm(x, w, eps, T, Hd, out, (hipStream_t)s));
}
int nvr_add_rmsnorm(nvr_half *h, const nvr_half *y, const nvr_half *w, float eps, int64_t T, int64_t Hd, nvr_half *out, void *s) {
    if (g_ops_f32) return nvr::kf::add_rmsnorm(FP(h), FP(y), FP(w), eps, T, Hd, FP(out), (hipStream_t)s);
    return KO(add_rmsnorm(h, y, w, eps, T, Hd, out, (hipStream_t)s));
}
int nvr_linear(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t N, void *y, int f32, void *s) {
    if (g_ops_f32) return nvr::kf::linear(FP(x), ldx, FP(W), T, K, N, nullptr, (float *)y, (hipStream_t)s);      // (y is f32 either way)
    return KO(linear(x, ldx, W, T, K, N, y, f32 != 0, (hipStream_t)s));
}
int nvr_lm_head(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t N, float *logits, float *part_val,
                int32_t *part_idx, int32_t *nparts, void *s) {
    NO_F32("nvr_lm_head");
    if (!nparts) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_lm_head: nparts is null");
    return KO(lm_head(x, ldx, W, T, K, N, logits, part_val, part_idx, nparts, (hipStream_t)s, logits != nullptr));
}
int nvr_argmax_partials(const float *part_val, const int32_t *part_idx, int32_t nparts, int64_t T, int64_t *out_idx, float *out_val,
                        int64_t idx_offset, void *s) {
    return KO(argmax_partials(part_val, part_idx, nparts, T, out_idx, out_val, idx_offset, (hipStream_t)s));
}
int nvr_linear_splitk(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t N, int64_t S, float *slabs, void *s) {
    NO_F32("nvr_linear_splitk");
    return KO(linear_splitk(x, ldx, W, T, K, N, S, slabs, (hipStream_t)s));
}
int nvr_add_rmsnorm_slabs(nvr_half *h, const float *slabs, int64_t S, const nvr_half *w, float eps, int64_t T, int64_t Hd, nvr_half *out, void *s) {
    NO_F32("nvr_add_rmsnorm_slabs");
    return KO(add_rmsnorm_slabs(h, slabs, S, w, eps, T, Hd, out, (hipStream_t)s));
}
int nvr_linear_tiled(const nvr_half *x, int64_t ldx, const nvr_half *W, const nvr_half *Wt, int64_t T, int64_t K, int64_t N, void *y, int f32,
                     void *s) {
    NO_F32("nvr_linear_tiled");
    return KO(linear(x, ldx, W, T, K, N, y, f32 != 0, (hipStream_t)s, Wt));
}
int nvr_linear_splitk_tiled(const nvr_half *x, int64_t ldx, const nvr_half *W, const nvr_half *Wt, int64_t T, int64_t K, int64_t N, int64_t S,
                            float *slabs, void *s) {
    NO_F32("nvr_linear_splitk_tiled");
    return KO(linear_splitk(x, ldx, W, T, K, N, S, slabs, (hipStream_t)s, Wt));
}
int nvr_linear_silu_mul_tiled(const nvr_half *x, int64_t ldx, const nvr_half *W, const nvr_half *Wt, int64_t T, int64_t K, int64_t I,
                              nvr_half *out, void *s) {
    NO_F32("nvr_linear_silu_mul_tiled");
    return KO(linear_silu_mul(x, ldx, W, T, K, I, out, (hipStream_t)s, Wt));
}
int nvr_linear_qkv_rope_store_tiled(const nvr_half *x, int64_t ldx, const nvr_half *W, const nvr_half *Wt, int64_t T, int64_t K, int64_t H,
                                    int64_t KVH, int64_t D, const int64_t *pos, const int32_t *slots, const float *c, const float *sn,
                                    nvr_half *qkv, nvr_half *kc, nvr_half *vc, void *s) {
    NO_F32("nvr_linear_qkv_rope_store_tiled");
    return KO(linear_qkv_rope_store(x, ldx, W, T, K, H, KVH, D, pos, slots, c, sn, qkv, kc, vc, (hipStream_t)s, Wt));
}
int nvr_lm_head_tiled(const nvr_half *x, int64_t ldx, const nvr_half *W, const nvr_half *Wt, int64_t T, int64_t K, int64_t N, float *logits,
                      float *part_val, int32_t *part_idx, int32_t *nparts, void *s) {
    NO_F32("nvr_lm_head_tiled");
    if (!nparts) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_lm_head_tiled: nparts is null");
    return KO(lm_head(x, ldx, W, T, K, N, logits, part_val, part_idx, nparts, (hipStream_t)s, logits != nullptr, Wt));
}
int nvr_retile_weight(const nvr_half *src, nvr_half *dst, int64_t N, int64_t K, int mode, int64_t H, int64_t KVH, int64_t D, void *s) {
    NO_F32("nvr_retile_weight");
    return KO(retile_weight(src, dst, N, K, mode, H, KVH, D, (hipStream_t)s));
}
int nvr_linear_add_residual(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t N, nvr_half *h, void *s) {
    NO_F32("nvr_linear_add_residual");
    if (!k::gemm256_preferred(T, K, N, ldx))
        return nvr::fail(NVR_ERR_UNSUPPORTED, "nvr_linear_add_residual: T=%ld K=%ld N=%ld is not a shape of the 256x256 prefill GEMM", (long)T, (long)K, (long)N);
    return KO(gemm256_resid(x, ldx, W, T, K, N, h, (hipStream_t)s));
}
int nvr_decode_splitk_slices(int64_t T, int64_t K, int64_t N) { return k::decode_splitk_slices(T, K, N); }
int nvr_linear_silu_mul(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t I, nvr_half *out, void *s) {
    if (g_ops_f32) {                                                    // float32: the decode-sized form exists (gemv + SiluAndMul in one launch)
        if (!nvr::kf::linear_silu_ok(T, K, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "nvr_linear_silu_mul: float32 ops fuse decode-sized steps only (1..8 rows); nvr_linear + nvr_silu_and_mul otherwise");
        return nvr::kf::linear_silu_mul(FP(x), ldx, FP(W), T, K, I, nullptr, FP(out), (hipStream_t)s);
    }
    return KO(linear_silu_mul(x, ldx, W, T, K, I, out, (hipStream_t)s));
}
int nvr_linear_qkv_rope_store(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t H, int64_t KVH,
                              int64_t D, const int64_t *pos, const int32_t *slots, const float *c, const float *sn,
                              nvr_half *qkv, nvr_half *kc, nvr_half *vc, void *s) {
    if (g_ops_f32) {                                                    // float32: the decode-sized form exists (gemv + RoPE + KV store in one launch)
        if (!nvr::kf::linear_qkv_rope_ok(T, K, D, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "nvr_linear_qkv_rope_store: float32 ops fuse decode-sized steps only (1..8 rows); nvr_linear + nvr_rope_store_kv otherwise");
        return nvr::kf::linear_qkv_rope_store(FP(x), ldx, FP(W), T, K, H, KVH, D, nullptr, pos, slots, c, sn, FP(qkv), FP(kc), FP(vc), (hipStream_t)s);
    }
    return KO(linear_qkv_rope_store(x, ldx, W, T, K, H, KVH, D, pos, slots, c, sn, qkv, kc, vc, (hipStream_t)s));
}
int nvr_rope_store_kv(nvr_half *qkv, const int64_t *pos, const int32_t *slots, int64_t T, int64_t H, int64_t KVH, int64_t D,
                      const float *c, const float *sn, nvr_half *kc, nvr_half *vc, void *s) {
    if (g_ops_f32) return nvr::kf::rope_store_kv(FP(qkv), pos, slots, T, H, KVH, D, c, sn, FP(kc), FP(vc), nullptr, nullptr, 0.f, (hipStream_t)s);
    return KO(rope_store_kv(qkv, pos, slots, T, H, KVH, D, c, sn, kc, vc, (hipStream_t)s));
}
int nvr_qk_norm_rope_store_kv(nvr_half *qkv, const int64_t *pos, const int32_t *slots, int64_t T, int64_t H, int64_t KVH, int64_t D,
                              const float *c, const float *sn, const nvr_half *qw, const nvr_half *kw, float eps, nvr_half *kc, nvr_half *vc,
                              void *s) {
    if (!qw || !kw) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_qk_norm_rope_store_kv: norm weights are null");
    if (g_ops_f32) return nvr::kf::rope_store_kv(FP(qkv), pos, slots, T, H, KVH, D, c, sn, FP(kc), FP(vc), FP(qw), FP(kw), eps, (hipStream_t)s);
    return KO(rope_store_kv(qkv, pos, slots, T, H, KVH, D, c, sn, kc, vc, (hipStream_t)s, qw, kw, eps));
}
int nvr_rope_table(int64_t D, int64_t max_pos, double theta, float *cos_dev, float *sin_dev) {   // rotary_embedding.rs:74-119 (A-14)
    NVR_GUARD_BEGIN
    const int64_t half = D / 2;
    std::vector<float> c(max_pos * half), sn(max_pos * half);
    for (int64_t p = 0; p < max_pos; ++p)
        for (int64_t j = 0; j < half; ++j) {
            float inv = (float)(1.0 / std::pow(theta, (double)(2 * j) / (double)D));
            float ang = (float)p * inv;
            c[p * half + j] = (float)std::cos((double)ang); sn[p * half + j] = (float)std::sin((double)ang);
        }
    NVR_HIP_CHECK(hipMemcpy(cos_dev, c.data(), c.size() * 4, hipMemcpyHostToDevice));
    NVR_HIP_CHECK(hipMemcpy(sin_dev, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    return NVR_OK;
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
size_t nvr_paged_attn_workspace_bytes(int64_t B, int64_t H, int64_t D, int64_t max_ctx) { return k::attn_workspace_bytes(B, H, D, max_ctx); }
int nvr_paged_attn_decode(const nvr_half *q, int64_t ldq, const nvr_half *kc, const nvr_half *vc, const nvr_attn_meta *m,
                          int64_t H, int64_t KVH, int64_t D, int64_t bs, float scale, nvr_half *out, void *ws, void *s) {
    if (g_ops_f32) {
        nvr::kt::AttnArgsF f{};
        f.q = FP(q); f.ldq = ldq; f.k = FP(kc); f.v = FP(vc); f.ctx_lens = m->context_lens; f.block_tables = m->block_tables; f.max_blocks = m->max_blocks;
        f.block_size = (int32_t)bs; f.nq = m->batch; f.H = (int32_t)H; f.KVH = (int32_t)KVH; f.D = (int32_t)D; f.scale = scale; f.max_ctx = m->max_context_len;
        f.out = FP(out);
        return nvr::kf::attention(f, true, (hipStream_t)s);
    }
    k::AttnArgs a{};
    a.q = q; a.ldq = ldq; a.k = kc; a.v = vc; a.ctx_lens = m->context_lens; a.block_tables = m->block_tables;
    a.max_blocks = m->max_blocks; a.block_size = (int32_t)bs; a.nq = m->batch; a.H = (int32_t)H; a.KVH = (int32_t)KVH;
    a.D = (int32_t)D; a.scale = scale; a.max_ctx = m->max_context_len; a.out = out; a.workspace = ws;
    return KO(attention(a, true, (hipStream_t)s));
}
int nvr_paged_attn_decode_fused(const nvr_half *q, int64_t ldq, const nvr_half *kc, const nvr_half *vc, const nvr_attn_meta *m,
                                int64_t H, int64_t KVH, int64_t D, int64_t bs, float scale, nvr_half *out, void *ws, uint32_t *tickets, void *s) {
    NO_F32("nvr_paged_attn_decode_fused");
    if (!tickets) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_paged_attn_decode_fused: tickets is null");
    k::AttnArgs a{};
    a.q = q; a.ldq = ldq; a.k = kc; a.v = vc; a.ctx_lens = m->context_lens; a.block_tables = m->block_tables;
    a.max_blocks = m->max_blocks; a.block_size = (int32_t)bs; a.nq = m->batch; a.H = (int32_t)H; a.KVH = (int32_t)KVH;
    a.D = (int32_t)D; a.scale = scale; a.max_ctx = m->max_context_len; a.out = out; a.workspace = ws; a.tickets = tickets;
    return KO(attention(a, true, (hipStream_t)s));
}
int nvr_paged_attn_decode_shared(const nvr_half *q, int64_t ldq, const nvr_half *kc, const nvr_half *vc, const nvr_attn_meta *m,
                                 int64_t H, int64_t KVH, int64_t D, int64_t bs, float scale, int64_t shared_len, const int32_t *rows,
                                 const int32_t *kv0, const int32_t *count, nvr_half *out, void *ws, void *s) {
    NO_F32("nvr_paged_attn_decode_shared");
    k::AttnArgs a{};
    a.q = q; a.ldq = ldq; a.k = kc; a.v = vc; a.ctx_lens = m->context_lens; a.block_tables = m->block_tables;
    a.max_blocks = m->max_blocks; a.block_size = (int32_t)bs; a.nq = m->batch; a.H = (int32_t)H; a.KVH = (int32_t)KVH;
    a.D = (int32_t)D; a.scale = scale; a.max_ctx = m->max_context_len; a.out = out; a.workspace = ws;
    a.shared_len = (int32_t)shared_len; a.shared_rows = rows; a.shared_kv0 = kv0; a.shared_count = count;
    return KO(attention(a, true, (hipStream_t)s));
}
// host-side tile list for the flash kernels from cu_seqlens_q (+ context lens for the paged variant)
static int run_prefill_attn(const nvr_half *q, int64_t ldq, const nvr_half *kk, const nvr_half *v, int64_t ldkv, const nvr_attn_meta *m,
                            bool paged, int64_t bs, int64_t T, int64_t H, int64_t KVH, int64_t D, float scale, nvr_half *out, hipStream_t st) {
    std::vector<int32_t> cu(m->batch + 1), ctxl(m->batch);
    NVR_HIP_CHECK(hipMemcpy(cu.data(), m->cu_seqlens_q, cu.size() * 4, hipMemcpyDeviceToHost));
    if (cu[m->batch] != T) return nvr::fail(NVR_ERR_LEN_MISMATCH, "cu_seqlens_q ends at %d but T=%ld", cu[m->batch], (long)T);
    if (paged) NVR_HIP_CHECK(hipMemcpy(ctxl.data(), m->context_lens, ctxl.size() * 4, hipMemcpyDeviceToHost));
    int rc;
    if (!g_ops_f32 && k::flash_prefill_ok((int)D, (int)H, (int)KVH)) {
        const int qb = k::flash_tile_positions((int)H, (int)KVH);
        std::vector<k::FlashTile> tiles;
        for (int b = 0; b < m->batch; ++b) {
            const int nq = cu[b + 1] - cu[b], p0 = paged ? ctxl[b] - nq : 0;      // query i sits at position p0 + i
            if (p0 < 0) return nvr::fail(NVR_ERR_INVALID_ARG, "sequence %d: %d queries but context %d", b, nq, ctxl[b]);
            for (int q0 = 0; q0 < nq; q0 += qb)
                tiles.push_back(k::FlashTile{cu[b] + q0, std::min(qb, nq - q0), p0 + q0, paged ? b : cu[b]});
        }
        k::FlashTile *d = nullptr;
        NVR_HIP_CHECK(hipMalloc((void **)&d, tiles.size() * sizeof(k::FlashTile) + 16));
        NVR_HIP_CHECK(hipMemcpy(d, tiles.data(), tiles.size() * sizeof(k::FlashTile), hipMemcpyHostToDevice));
        k::FlashArgs f{};
        f.q = q; f.ldq = ldq; f.k = kk; f.v = v; f.ldkv = ldkv; f.block_tables = m->block_tables; f.max_blocks = m->max_blocks;
        f.block_size = (int32_t)bs; f.tiles = d; f.ntiles = (int32_t)tiles.size(); f.H = (int32_t)H; f.KVH = (int32_t)KVH; f.D = (int32_t)D;
        f.scale = scale; f.out = out;
        rc = KO(flash_prefill(f, paged, st));
        hipStreamSynchronize(st);
        hipFree(d);
    } else {
        std::vector<int32_t> ctx(T), ref(T);
        int maxc = 1;
        for (int b = 0; b < m->batch; ++b)
            for (int t = cu[b]; t < cu[b + 1]; ++t) {
                ctx[t] = (paged ? ctxl[b] - (cu[b + 1] - cu[b]) : 0) + (t - cu[b]) + 1; ref[t] = paged ? b : cu[b];
                maxc = std::max(maxc, ctx[t]);
            }
        int32_t *d = nullptr;
        NVR_HIP_CHECK(hipMalloc((void **)&d, (size_t)(2 * T + 4) * 4));
        NVR_HIP_CHECK(hipMemcpy(d, ctx.data(), T * 4, hipMemcpyHostToDevice));
        NVR_HIP_CHECK(hipMemcpy(d + T, ref.data(), T * 4, hipMemcpyHostToDevice));
        if (g_ops_f32) {
            nvr::kt::AttnArgsF f{};
            f.q = FP(q); f.ldq = ldq; f.k = FP(kk); f.v = FP(v); f.ldkv = ldkv; f.ctx_lens = d; f.nq = (int32_t)T;
            if (paged) { f.seq_of_q = d + T; f.block_tables = m->block_tables; f.max_blocks = m->max_blocks; f.block_size = (int32_t)bs; }
            else f.kv_base = d + T;
            f.H = (int32_t)H; f.KVH = (int32_t)KVH; f.D = (int32_t)D; f.scale = scale; f.max_ctx = maxc; f.out = FP(out);
            rc = nvr::kf::attention(f, paged, st);
            hipStreamSynchronize(st);
            hipFree(d);
            return rc;
        }
        k::AttnArgs a{};
        a.q = q; a.ldq = ldq; a.k = kk; a.v = v; a.ldkv = ldkv; a.ctx_lens = d; a.nq = (int32_t)T;
        if (paged) { a.seq_of_q = d + T; a.block_tables = m->block_tables; a.max_blocks = m->max_blocks; a.block_size = (int32_t)bs; }
        else a.kv_base = d + T;
        a.H = (int32_t)H; a.KVH = (int32_t)KVH; a.D = (int32_t)D; a.scale = scale; a.max_ctx = maxc; a.out = out;
        rc = KO(attention(a, paged, st));
        hipStreamSynchronize(st);
        hipFree(d);
    }
    return rc;
}
int nvr_attn_prefill_varlen(const nvr_half *q, const nvr_half *kk, const nvr_half *v, int64_t ld, const nvr_attn_meta *m,
                            int64_t T, int64_t H, int64_t KVH, int64_t D, float scale, nvr_half *out, void *s) {
    NVR_GUARD_BEGIN
    return run_prefill_attn(q, ld, kk, v, ld, m, false, 0, T, H, KVH, D, scale, out, (hipStream_t)s);
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_attn_prefill_paged(const nvr_half *q, int64_t ldq, const nvr_half *kc, const nvr_half *vc, const nvr_attn_meta *m, int64_t T,
                           int64_t H, int64_t KVH, int64_t D, int64_t bs, float scale, nvr_half *out, void *s) {
    NVR_GUARD_BEGIN
    return run_prefill_attn(q, ldq, kc, vc, 0, m, true, bs, T, H, KVH, D, scale, out, (hipStream_t)s);
    NVR_GUARD_END(NVR_ERR_INVARIANT)
}
int nvr_activation_type_from_str(const char *name, int32_t *type_out) {                       // ActivationType::from_str, activation.rs:169-182
    if (!name || !type_out) return nvr::fail(NVR_ERR_INVALID_ARG, "nvr_activation_type_from_str: null argument");
    std::string s(name);
    for (char &c : s) c = (char)std::tolower((unsigned char)c);
    if (s == "silu" || s == "swish") *type_out = NVR_ACT_SILU;
    else if (s == "gelu") *type_out = NVR_ACT_GELU;
    else if (s == "relu") *type_out = NVR_ACT_RELU;
    else if (s == "silu_and_mul" || s == "siluandmul") *type_out = NVR_ACT_SILU_AND_MUL;
    else if (s == "gelu_and_mul" || s == "geluandmul") *type_out = NVR_ACT_GELU_AND_MUL;
    else return nvr::fail(NVR_ERR_INVALID_ARG, "Unknown activation function: %s", name);
    return NVR_OK;
}
int nvr_activation(int32_t type, const nvr_half *x, int64_t T, int64_t cols, nvr_half *out, void *s) {   // Activation::forward, activation.rs:147-159
    if (g_ops_f32) return nvr::kf::activation(type, FP(x), T, cols, FP(out), (hipStream_t)s);
    return KO(activation(type, x, T, cols, out, (hipStream_t)s));
}
int nvr_silu_and_mul(const nvr_half *x, int64_t T, int64_t I, nvr_half *out, void *s) {
    if (g_ops_f32) return nvr::kf::silu_and_mul(FP(x), T, I, FP(out), (hipStream_t)s);
    return KO(silu_and_mul(x, T, I, out, (hipStream_t)s));
}
int nvr_add_bias(nvr_half *y, const nvr_half *b, int64_t T, int64_t N, void *s) {
    NO_F32("nvr_add_bias");                                              // (the f32 linear takes its bias as an argument inside the runner)
    return KO(add_bias(y, b, T, N, (hipStream_t)s));
}
int nvr_select_last_tokens(const nvr_half *h, const int32_t *cu, int64_t B, int64_t Hd, nvr_half *out, void *s) {
    if (g_ops_f32) return nvr::kf::select_last_tokens(FP(h), cu, B, Hd, FP(out), (hipStream_t)s);
    return KO(select_last_tokens(h, cu, B, Hd, out, (hipStream_t)s));
}
int nvr_argmax(const float *logits, int64_t B, int64_t V, int64_t *out, void *s) { return k::argmax(logits, B, V, out, nullptr, 0, (hipStream_t)s); }
size_t nvr_sample_workspace_bytes(int64_t B, int64_t V) { return k::sample_workspace_bytes(B, V); }
int nvr_sample(const float *logits, int64_t B, int64_t V, const float *temp, const int64_t *top_k, const float *top_p,
               const uint64_t *keys, int64_t *out, void *ws, void *s) {
    return KO(sample(logits, B, V, temp, top_k, top_p, keys, out, ws, (hipStream_t)s));
}
uint64_t nvr_sample_key(uint64_t seed, uint64_t seq_id, uint64_t step) {
    return nvr::splitmix64(nvr_weight_key_impl(seed, seq_id) + step * 0xA24BAED4963EE407ULL);
}
uint64_t nvr_weight_key(uint64_t seed, uint64_t tid) { return nvr_weight_key_impl(seed, tid); }
float nvr_weight_scale(double std) { return nvr_weight_scale_impl(std); }
int nvr_fill_weight(nvr_half *dst, int64_t rows, int64_t cols, int64_t ld, int64_t gcols, int64_t row0, int64_t col0,
                    uint64_t key, float scale, void *s) {
    if (g_ops_f32) return nvr::kf::fill_weight(FP(dst), rows, cols, ld, gcols, row0, col0, key, scale, (hipStream_t)s);   // unrounded values
    return KO(fill_weight(dst, rows, cols, ld, gcols, row0, col0, key, scale, (hipStream_t)s));
}
int nvr_fill_const(nvr_half *dst, int64_t n, float v, void *s) {
    if (g_ops_f32) return nvr::kf::fill_const(FP(dst), n, v, (hipStream_t)s);
    return KO(fill_const(dst, n, v, (hipStream_t)s));
}

}  // extern "C"

// model_runner.h — step execution on one MI355X (mirrors ModelRunner, reference
// src/engine/model_runner.rs:19-464, and the Qwen3 graph it drives, src/models/qwen3.rs:208-505).
//
// Owns: fp16 weights (synthetic), the KV pool [L][K|V][NB, bs, KVH/tp, D] in one HBM
// allocation (model_runner.rs:364-396), persistent activation workspaces, pinned step-input staging
// and their fixed device twins (so a decode step is graph-replayable), and the hipGraph cache keyed
// by (batch size, context bucket) that replaces the reference's stub (model_runner.rs:303-361).
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <vector>
#include "sequence.h"
#include "comm.h"
#include "common.h"
#include "kernels/kernels.h"

struct nvr_model_runner {
    nvr::Env env;                          // NVR_* switches, read once in init()
    nvr_config cfg{};
    nvr_model_config mc{};
    // derived, per rank (qwen3.rs:158-159, linear.rs:300-304)
    int64_t tp = 1, rank = 0;
    bool bf16 = false;                     // Config.dtype == "bfloat16": the nvr::kb build of the kernels; 16-bit words are bfloat16 everywhere
    bool f32 = false; int64_t em = 1;      // Config.dtype == "float32": the f32 path (kernels/f32_path.hip); the uint16_t* buffers then hold em = 2 words per element
    int64_t Hd = 0, H = 0, KVH = 0, D = 0, I = 0, V = 0, Vl = 0, vocab_start = 0, L = 0, QKV = 0;
    int64_t block_size = 256, num_blocks = 0, max_tokens = 0, max_seqs = 0, max_blocks_per_seq = 0, max_pos = 0;
    float scale = 1.f;
    int device = 0;
    hipStream_t stream = nullptr;
    // tensor-parallel prefill: the row-parallel exchange of token chunk i runs on comm_stream under the GEMM of chunk i+1 (row_parallel_norm)
    hipStream_t comm_stream = nullptr;
    static constexpr int kMaxChunks = 8;
    hipEvent_t ev_gemm[kMaxChunks] = {}, ev_reduced[kMaxChunks] = {};
    int64_t tp_overlap_chunks = 0;         // chunks of the last overlapped exchange (0: none ran; diagnostics / tests)
    int tp_overlap = 1;                    // nvr_runner_set_tp_prefill_overlap: 1 (default) chunks of one GEMM; 2 two micro-batches per step; 0: serial on one stream
    int mb_route[4] = {0, 0, 0, 0};         // tp_overlap = 2: kernel of the step's qkv / o / gate_up / down GEMMs (0: 256-row tiles, 1: 128-row tiles)
    int64_t mb_rows = 0, mb_tiles = 0;     // tp_overlap = 2: rows / flash tiles of the first micro-batch of the current prefill step (0: not split)
    int forward_prefill_two(int64_t T, int64_t B);

    struct Layer { uint16_t *qkv, *o, *gate_up, *down, *ln1, *ln2;
                   uint16_t *qkv_t, *o_t, *gate_up_t, *down_t;      // *_t: tiled copies for the decode kernels (retile_weight), or null
                   uint16_t *q_norm, *k_norm;                       // mc.qk_norm: [D] each (ones until loaded), else null
                   uint16_t *qkv_b, *o_b, *gate_up_b, *down_b; };   // mc.use_bias: [QKV] / [Hd] (rank 0, else null) / [2I] / [Hd] (rank 0), else null
    std::vector<Layer> layers;
    uint16_t *embed = nullptr, *lm_head = nullptr, *norm = nullptr;
    uint16_t *lm_head_t = nullptr;         // tiled copy of the LM head (decode-sized steps)
    bool tiled_weights = true, tiled_dirty = true;   // env.tiled_weights = false keeps the row-major parameters only
    int retile_all();                      // (re)build the tiled copies from the row-major parameters (init, after load_tensor)
    float *cos_t = nullptr, *sin_t = nullptr;
    uint16_t *kv_pool = nullptr;
    size_t kv_layer_elems = 0;             // elements of one K (or V) cache of one layer

    // activations
    uint16_t *h = nullptr, *n = nullptr, *qkv = nullptr, *attn = nullptr, *proj = nullptr, *gu = nullptr, *act = nullptr,
             *nlast = nullptr;
    float *logits = nullptr; void *attn_ws = nullptr; size_t attn_ws_bytes = 0;
    unsigned int *attn_tickets = nullptr;  // [max_seqs * KVH] arrival counters of the split-KV decode attention (fused merge; zero between launches)
    float *slabs = nullptr;                // split-k partials of o_proj / down_proj: [S][T][Hd] f32, S * T <= 4 * slab_rows
    // step inputs: one pinned host arena mirrored by one device arena
    char *in_host = nullptr, *in_dev = nullptr; size_t in_bytes = 0;
    int64_t *d_ids = nullptr, *d_pos = nullptr; int32_t *d_slots = nullptr, *d_cu = nullptr, *d_ctx = nullptr,
            *d_kvbase = nullptr, *d_bt = nullptr;
    size_t off_ids = 0, off_pos = 0, off_slots = 0, off_cu = 0, off_ctx = 0, off_kvbase = 0, off_bt = 0, off_tiles = 0;
    int64_t n_tiles = 0;                   // flash prefill tiles of the current step
    // decode steps use one compact region (ids|pos|slots|ctx|block tables) uploaded with a single memcpy (K19)
    size_t off_dec = 0, dec_bytes = 0, dof_ids = 0, dof_pos = 0, dof_slots = 0, dof_ctx = 0, dof_bt = 0;
    size_t dof_skv0 = 0, dof_srows = 0, dof_scount = 0;  // shared-prefix group of the step: kv0 per row, member rows, member count
    int64_t *dd_ids = nullptr, *dd_pos = nullptr; int32_t *dd_slots = nullptr, *dd_ctx = nullptr, *dd_bt = nullptr;
    // sampling
    int64_t *d_tok = nullptr, *h_tok = nullptr; float *d_maxval = nullptr;
    float *d_temp = nullptr; int64_t *d_topk = nullptr; float *d_topp = nullptr; uint64_t *d_keys = nullptr;
    char *samp_host = nullptr; void *sample_ws = nullptr;
    bool allow_missing_comm = false;                    // NVR_TP_NO_COMM=1: one rank's compute without its collectives (profiling only)
    // TP exchange buffers for the greedy (max, idx) merge
    float *d_gather_val = nullptr; int64_t *d_gather_idx = nullptr;
    float *d_gather_logits = nullptr, *d_full_logits = nullptr; void *sample_ws_full = nullptr;   // stochastic sampling under TP (lazy)
    nvr::k::TpArgmaxRec *d_rec = nullptr, *d_gather_rec = nullptr;     // greedy launch-ahead under TP: this rank's records, every rank's

    int num_cus = 256;                     // compute units of the device
    std::map<uint64_t, hipGraphExec_t> graphs;
    size_t last_rows = 0; bool last_prefill = false; int64_t last_tokens = 0;
    int64_t decode_shared_len = 0;                       // the last decode step: tokens its sharing group holds in the same leading blocks
    bool decode_ragged = false;                          // the decode step's contexts are ragged enough for the work-balanced attention launch (ragged_batch)
    bool ragged_batch(size_t nseq, int64_t sum_ctx, int64_t max_ctx);   // (also sets decode_shares)
    int32_t decode_shares = 0;                           // ... and the number of shares its launch is asked for
    int64_t decode_shared_rows = 0;                      // ... and how many of the step's sequences belong to that group (== batch: all)
    nvr::Comm comm;
    bool graphs_disabled = false;   // set when capture with RCCL nodes fails: fall back to eager launches
    int comm_selftest();
    int comm_selftest_calls = 0;           // (Env::selftest_inject)
    int rearm_tickets();                   // zero the split-KV attention's arrival counters (failure paths)
    int set_p2p_fenced(bool on);           // switch the one-shot collectives' protocol (captured decode graphs hold the old one: dropped)

    ~nvr_model_runner();
    int init();
    int execute(nvr_seq *const *seqs, size_t nseq, bool is_prefill);
    // a full (un-sharded) checkpoint tensor by its HF / reference name -> this rank's slice of the packed
    // device weights (dtype 0 = f16, 1 = bf16, 2 = f32); copy_weight reads a local tensor back (tests)
    int load_tensor(const char *name, int dtype, const int64_t *shape, int ndim, const void *data);
    int copy_weight(const char *local_name, uint16_t *host_out, size_t cap, int64_t *rows, int64_t *cols);
    void *last_decode_graph = nullptr;   // hipGraphExec_t of the last decode step
    int replay_last_decode_graph(int n); // diagnostic: launch chain without the host gap (nvr_runner_replay_last_decode_graph)
    // Launch-ahead of greedy decode steps (nvr_config.async_decode, engine.cpp): the sampled tokens of step k go straight into
    // step k+1's device-side input ids, and step k+1 is enqueued before step k's tokens have reached the host.
    int32_t lm_parts_of_last_step() const { return lm_parts; }
    // single rank, or tensor-parallel ranks whose collectives are the stream-ordered peer-to-peer kernels (no host rendezvous): every
    // rank takes the same decisions from the same scheduler state and merges the same gathered (max, arg-max) records on the device
    int last_prefill_kv_source() const { return !last_prefill ? -1 : prefill_paged ? 2 : (prefill_kv_cache && n_tiles > 0) ? 1 : 0; }
    // what the diagnostic accessors report about "the last step" (nvr_runner_last_*): of the step the engine has just returned — kept per step in
    // flight like the LM head's input rows (present_step) — or, for a caller that drives execute_model itself, of the step executed last
    struct StepFacts { int kv_source = -1; int64_t shared_len = 0, shared_rows = 0, overlap_chunks = 0; bool ragged = false; };
    StepFacts facts_now() const {
        StepFacts f; f.kv_source = last_prefill_kv_source(); f.overlap_chunks = tp_overlap_chunks;
        f.shared_len = last_prefill ? 0 : decode_shared_len; f.shared_rows = (last_prefill || decode_shared_len == 0) ? 0 : decode_shared_rows;
        f.ragged = !last_prefill && decode_ragged;
        return f;
    }
    StepFacts facts_kept[2], facts_shown; bool facts_shown_valid = false;
    StepFacts facts() const { return facts_shown_valid ? facts_shown : facts_now(); }
    bool ahead_capable() const {
        return lm_fused && h_tok_dev != nullptr && ahead_tok[0] != nullptr && ((tp == 1 && !comm.active()) || (tp > 1 && comm.p2p_ready));
    }
    bool ahead_ok(size_t nseq) const { return ahead_capable() && (tp == 1 || nseq * sizeof(nvr::k::TpArgmaxRec) <= (size_t)nvr::P2P_GATHER_BYTES); }
    int sample_launch(nvr_seq *const *seqs, size_t nseq, int parity);        // greedy rows: arg-max merge -> pinned ahead_tok[parity] and dd_ids
    int sample_wait(size_t nseq, int parity, int64_t *out);                  // spins on the pinned buffer (no stream synchronisation)
    int execute_decode_ahead(nvr_seq *const *seqs, size_t nseq, int parity); // decode step whose ids are already on the device
    int64_t *ahead_tok[2] = {nullptr, nullptr}, *ahead_tok_dev[2] = {nullptr, nullptr};
    char *ahead_host[2] = {nullptr, nullptr};            // pinned twins of the decode input region (one per step in flight)
    uint16_t *lm_snap[2] = {nullptr, nullptr};           // [max_seqs][hidden]: the LM head's input rows of the step sampled with this parity
    void present_step(int parity, size_t rows);          // the logits accessors now refer to THAT step, whatever has been launched behind it
    int ensure_logits();                                 // materialise the last step's f32 logits if it skipped their stores
    int sample(nvr_seq *const *seqs, size_t nseq, int64_t *out);
    uint16_t *k_cache(size_t l) { return kv_pool + (2 * l) * kv_layer_elems * em; }
    uint16_t *v_cache(size_t l) { return kv_pool + (2 * l + 1) * kv_layer_elems * em; }

private:
    int forward(int64_t T, int64_t B, bool is_prefill, int64_t max_ctx);
    int gen_weights();
    int gen_weights_f32();
    int forward_f32(int64_t T, int64_t B, bool is_prefill, int64_t max_ctx);
    int row_parallel_norm_f32(int64_t T, const float *wn);
    float *f32_h2 = nullptr;                             // float32, one rank: second residual-stream buffer [8][hidden] of the decode-sized steps whose add + norm
                                                         // rides on the consumer GEMV (the workgroups of that launch still read the first while one of them writes)
    float *f32_gather = nullptr; int64_t f32_gather_rows = 0;   // float32 tensor-parallel ranks: every rank's partial sums [tp][f32_gather_rows][hidden], a piece of the step's rows at a time
    int row_parallel_norm(const uint16_t *x, int64_t K, const uint16_t *W, const uint16_t *Wt, int64_t T, const uint16_t *wn, const uint16_t *bias = nullptr);
    int64_t *h_tok_dev = nullptr;                        // device-visible address of the pinned token buffer h_tok
    int64_t slab_rows = 64;           // rows the split-k slab buffers hold (row_parallel_norm)
    // sharing group of a decode batch (0 = none / too small: plain paged attention); fills kv0[nseq], rows[nseq], *count of the arena
    int64_t shared_prefix_plan(nvr_seq *const *seqs, size_t nseq, int32_t *kv0, int32_t *rows, int32_t *count, int64_t *members) const;
    mutable std::vector<int32_t> plan_keys, plan_cnt;    // scratch of the plan (first-block id -> count)
    bool prefill_kv_cache = false;                  // whole-prompt prefill whose K/V the flash kernel reads from the (contiguous) cache rows
    bool prefill_paged = false;                          // this prefill step skips cached prefixes (K/V via block tables)
    bool lazy_logits = true, want_logits = true, logits_valid = true; const uint16_t *lm_input = nullptr;
    bool lm_fused = true; int32_t lm_parts = 0;          // lm_head arg-max partials of the last step (0: none)
    float *d_lm_pval = nullptr; int32_t *d_lm_pidx = nullptr;
};

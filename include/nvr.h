/*
 * nvr.h — C ABI of the MI355X-native drop-in for nano-vllm-rs's paged-attention
 * prefill/decode hot path and block-based KV cache.
 *
 * The reference (ssvgopal/nano-vllm-rs @ 2025-07-18) is pure Rust with no FFI of its own;
 * its seam is the Rust public API re-exported at src/lib.rs:91-94, src/engine/mod.rs:14-17
 * and src/layers/mod.rs:16-22.  Every entry point below names the reference item it
 * replaces (file:line under the reference tree).  INTEGRATION.md shows the Rust-side
 * `extern "C"` block + safe wrappers that keep the LLMEngine / ModelRunner / BlockManager
 * surface.
 *
 * Conventions:
 *  - plain pointers and sizes only; no C++/torch types; hipStream_t travels as void*.
 *  - every fallible call returns an int status: NVR_OK (0) or a negative nvr_status;
 *    nvr_last_error() returns a thread-local message carrying the reference's error text.
 *    The reference's assert!/panic sites map to NVR_ERR_INVARIANT; nothing unwinds across
 *    the ABI.
 *  - handles are not internally synchronised (the reference wraps them in a Mutex,
 *    src/engine/llm_engine.rs:25-28): one thread at a time per handle.
 *  - the library owns all device memory it allocates (weights, KV pool, workspaces);
 *    pointers handed out are borrowed and stay valid until the documented next call.
 *  - the HIP path is the only compute path: if no gfx950 device is usable, device calls
 *    fail with NVR_ERR_HIP.  There is no CPU fallback.
 */
#ifndef NVR_H
#define NVR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVR_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ status ---- */
typedef enum nvr_status {
    NVR_OK = 0,
    NVR_ERR_NO_FREE_BLOCKS = -1,     /* block_manager.rs:163,225,280 */
    NVR_ERR_ALREADY_ALLOCATED = -2,  /* block_manager.rs:159 */
    NVR_ERR_NOT_ALLOCATED = -3,      /* block_manager.rs:267 */
    NVR_ERR_LEN_MISMATCH = -4,       /* scheduler.rs:236 */
    NVR_ERR_NOTHING_TO_SCHEDULE = -5,/* scheduler.rs:219 */
    NVR_ERR_INVARIANT = -6,          /* assert! sites: block_manager.rs:62,92-93,127,138 ... */
    NVR_ERR_INVALID_ARG = -7,        /* config.rs:83-119, sampling_params.rs:91-119 */
    NVR_ERR_HIP = -8,
    NVR_ERR_RCCL = -9,
    NVR_ERR_UNSUPPORTED = -10
} nvr_status;

NVR_API const char *nvr_last_error(void);
NVR_API int nvr_last_status(void);          /* status code that came with nvr_last_error (constructors return NULL: the code is here) */
NVR_API const char *nvr_version(void);

/* ------------------------------------------------------------- plain structs ---- */
/* SamplingParams, src/engine/sampling_params.rs:10-28 (Option<T> -> has_* flag) */
typedef struct nvr_sampling_params {
    float temperature;            /* default 1.0; 0.0 == greedy (:86-88) */
    uint64_t max_tokens;          /* default 64 */
    int32_t ignore_eos;
    int32_t has_top_p;  float top_p;
    int32_t has_top_k;  uint64_t top_k;
    int32_t has_repetition_penalty; float repetition_penalty;
} nvr_sampling_params;
NVR_API void nvr_sampling_params_default(nvr_sampling_params *sp);   /* sampling_params.rs:30-41 */
NVR_API int nvr_sampling_params_validate(const nvr_sampling_params *sp); /* :91-119 */

/* Config, src/config.rs:16-52 (model_path omitted: synthetic weights; loader is §8f) */
typedef struct nvr_config {
    uint64_t max_num_batched_tokens;   /* 32768 */
    uint64_t max_num_seqs;             /* 512 */
    uint64_t max_model_len;            /* 4096 */
    float gpu_memory_utilization;      /* 0.9 */
    uint64_t tensor_parallel_size;     /* 1 (1..=8) */
    int32_t enforce_eager;             /* 0: decode steps replay a captured hipGraph */
    int32_t has_eos; int64_t eos_token_id;
    uint64_t kvcache_block_size;       /* 256 */
    int64_t num_kvcache_blocks;        /* -1 == None: scheduler default 1000 (scheduler.rs:71-74);
                                          -2 == size the pool from free HBM * gpu_memory_utilization */
    /* extensions that the reference hard-codes (model_runner.rs:75,405): */
    uint64_t tensor_parallel_rank;     /* 0 */
    int32_t device_ordinal;            /* HIP device index of this rank */
    uint64_t sample_seed;              /* A-20 counter-RNG seed */
    int32_t skip_block_size_check;     /* reference unit tests use block sizes 4/16 that
                                          Config::validate would reject (config.rs:94) */
    uint32_t decode_chain;             /* reserved (0).  r01-r04 selected alternative launch chains of the decode step's GEMMs here (4: norms in
                                          the GEMM prologues + last-arriver residual adds, 5: the MLP pair as one persistent launch); both were
                                          measured a wash / a loss against the default six launches and left the library in r05 (sources and
                                          measurements: scratch/r04_decode_chain/, profiles/r02_decode_chain_ablation.txt, r04_mlp_engine.txt) */
    int32_t recompute_cached_prefix;   /* 0 (default): a prefill step computes only the tokens after a sequence's
                                          cached prefix (num_cached_tokens, block_manager.rs:187) and attends to the
                                          prefix through the block table (K8, attention.rs:211-222) — SURVEY §8f row 2;
                                          1: the reference's prepare_prefill_inputs (model_runner.rs:176-182), which
                                          recomputes every token from position 0 */
    int32_t enable_chunked_prefill;    /* extension A-23 (0 = the reference's whole-sequence prefill batches, scheduler.rs:135-138): the head
                                          of the waiting queue is scheduled for min(remaining, token budget left) tokens; later chunks
                                          attend to the earlier ones through the block table (K8); a step samples a token only for
                                          sequences whose prompt it finished (others report token -1) */
    int32_t async_decode;              /* 1: LLMEngine::step launches the NEXT greedy decode step before this step's tokens have reached
                                          the host (their ids go device to device; positions / slots / block tables follow from the
                                          lengths), whenever that is provably the step the reference would schedule: nothing waiting,
                                          no sequence able to stop, no block boundary.  Same batches, tokens and statistics as 0 (a
                                          request added in between cancels the step launched ahead).  The logits accessors of the engine's
                                          runner (nvr_runner_copy_logits) refer to the step nvr_engine_step has just RETURNED, as
                                          ModelRunner::execute_model does (model_runner.rs:105-128), never to the step launched behind it
                                          (r06: the LM head's input rows are kept per step in flight and the logits are produced from them on
                                          demand, bit-identical to the step's own).  Default 1 (r05; 0 in r02-r04). */
    int32_t shared_prefix_min_seqs;    /* decode batches of at least this many sequences that ALL begin with the same cache blocks
                                          (prefix-cache hits, block_manager.rs:181-197; BASELINE configs[4]) attend to those blocks in one
                                          MFMA pass for the whole batch (nvr_paged_attn_decode_shared) instead of once per sequence; same
                                          results within the attention tolerance.  0 = default (32), < 0 = never */
    char device[16];                   /* config.rs:48, validated like :108-111 plus "hip": "hip" (default) | "cuda" (the reference's
                                          default, taken as "the GPU") | "cpu" | "metal"; a runner exists only for "hip" / "cuda" —
                                          there is no CPU path in this library (NVR_ERR_UNSUPPORTED) */
    char dtype[16];                    /* config.rs:51,:113-116: "float16" (default) | "bfloat16" | "float32".  A runner exists for the two
                                          16-bit types: every kernel and collective of the library is built twice (fp16 and bf16 storage,
                                          f32 accumulation, MFMA at the same rate) and the runner picks one build at creation — weights,
                                          activations, KV cache and the residual stream are all of that type, logits are f32;
                                          nvr_runner_load_tensor converts any of f16 / bf16 / f32 to it (bf16 checkpoints into a bf16
                                          runner bit for bit), nvr_runner_copy_weight returns its raw 16-bit elements.  "float32" (r04):
                                          the REFERENCE-PRECISION path (kernels/f32_path.hip) — weights (the generator's values
                                          unrounded, or a checkpoint's values exactly), activations and KV cache as f32, plain FMA
                                          kernels, one GPU (tensor_parallel_size > 1: NVR_ERR_UNSUPPORTED); the
                                          arithmetic of the reference's own CPU path: its logits to ~1e-5.  nvr_runner_copy_weight then
                                          returns f32 values (two 16-bit words per element).  The stateless op entry points
                                          (nvr_linear, nvr_paged_attn_*, ...) take their 16-bit type from nvr_ops_set_dtype */
} nvr_config;
NVR_API void nvr_config_default(nvr_config *cfg);                    /* config.rs:54-71 */
NVR_API int nvr_config_validate(const nvr_config *cfg);              /* config.rs:83-119 */

/* Qwen3Config, src/models/qwen3.rs:26-125 (+ head_dim override, SURVEY A-17) */
typedef struct nvr_model_config {
    uint64_t vocab_size, hidden_size, intermediate_size, num_hidden_layers;
    uint64_t num_attention_heads, num_key_value_heads;
    uint64_t head_dim;                 /* 0 => hidden_size / num_attention_heads (:101-103) */
    uint64_t max_position_embeddings;
    float rms_norm_eps;
    double rope_theta;
    int32_t tie_word_embeddings;
    float init_std;                    /* synthetic weights: N(0, std^2)-like, SURVEY §8d */
    uint64_t seed;
    int32_t qk_norm;                   /* extension (0 = the reference graph): RMSNorm over head_dim on every q and k head before
                                        * RoPE, weights "layers.N.self_attn.q_norm.weight" / "k_norm.weight" (real Qwen3 checkpoints;
                                        * SURVEY §8f row 1, DESIGN A-27) */
    int32_t use_bias;                  /* Qwen3Config::use_bias (qwen3.rs:54-55, default false :82): a bias on qkv_proj :167, o_proj :178,
                                        * gate_up_proj :276 and down_proj :287 (row-parallel ones on rank 0 only, linear.rs:206); tensors
                                        * "...q_proj.bias" etc.  A-30: y = 16bit(16bit(x·Wᵀ) + b), candle_nn::Linear's two tensor ops.  Such a
                                        * model runs the plain GEMMs + one bias launch each (the fused epilogues carry no bias) */
} nvr_model_config;
NVR_API void nvr_model_config_default(nvr_model_config *mc);         /* qwen3.rs:70-89 */
NVR_API void nvr_model_config_qwen3_0_6b(nvr_model_config *mc);
NVR_API void nvr_model_config_qwen3_8b(nvr_model_config *mc);
NVR_API int nvr_model_config_validate(const nvr_model_config *mc, uint64_t tp); /* :106-124 */

/* ------------------------------------------------------------------ Sequence ---- */
/* Sequence, src/engine/sequence.rs:50-237.  status values = SequenceStatus :15-27 */
typedef struct nvr_seq nvr_seq_t;
enum { NVR_SEQ_WAITING = 0, NVR_SEQ_RUNNING = 1, NVR_SEQ_FINISHED = 2, NVR_SEQ_PREEMPTED = 3, NVR_SEQ_ERROR = 4 };

NVR_API nvr_seq_t *nvr_seq_create(const int64_t *prompt, size_t n, const nvr_sampling_params *sp,
                                  size_t block_size);                 /* Sequence::new :84-101 (A-1) */
NVR_API void nvr_seq_destroy(nvr_seq_t *s);   /* only for sequences never handed to a scheduler/engine */
NVR_API void nvr_seq_reset_id_counter(void);                          /* SEQUENCE_COUNTER :12 (tests) */
NVR_API uint64_t nvr_seq_id(const nvr_seq_t *s);
NVR_API int32_t nvr_seq_status(const nvr_seq_t *s);
NVR_API size_t nvr_seq_len(const nvr_seq_t *s);                       /* :104 */
NVR_API size_t nvr_seq_num_prompt_tokens(const nvr_seq_t *s);
NVR_API size_t nvr_seq_num_completion_tokens(const nvr_seq_t *s);     /* :135 */
NVR_API size_t nvr_seq_num_cached_tokens(const nvr_seq_t *s);
/* chunked prefill (A-23): tokens already in the cache, and the token range of the step the sequence was last scheduled into */
NVR_API size_t nvr_seq_num_computed_tokens(const nvr_seq_t *s);
NVR_API void nvr_seq_chunk(const nvr_seq_t *s, size_t *start, size_t *len);
NVR_API int64_t nvr_seq_last_token(const nvr_seq_t *s);
NVR_API size_t nvr_seq_num_blocks(const nvr_seq_t *s);                /* :157 */
NVR_API size_t nvr_seq_last_block_num_tokens(const nvr_seq_t *s);     /* :167 */
NVR_API void nvr_seq_token_ids(const nvr_seq_t *s, const int64_t **ptr, size_t *len);   /* borrowed */
NVR_API void nvr_seq_block_table(const nvr_seq_t *s, const int32_t **ptr, size_t *len); /* borrowed */
NVR_API void nvr_seq_append_token(nvr_seq_t *s, int64_t token);       /* :150 */
NVR_API int nvr_seq_should_stop(const nvr_seq_t *s, int has_eos, int64_t eos); /* :189 */
NVR_API void nvr_seq_preempt(nvr_seq_t *s);                           /* :213 */
NVR_API void nvr_seq_finish(nvr_seq_t *s);                            /* :208 */

/* -------------------------------------------------------------- BlockManager ---- */
/* BlockManager, src/engine/block_manager.rs:69-361 */
typedef struct nvr_block_manager nvr_block_manager_t;
typedef struct nvr_bm_stats {          /* BlockManagerStats :325-332 */
    uint64_t total_blocks, free_blocks, used_blocks, cached_blocks, block_size;
} nvr_bm_stats;
typedef struct nvr_block_info {        /* Block :12-24 */
    uint64_t block_id, ref_count; int32_t has_hash; uint64_t hash; uint64_t num_tokens;
} nvr_block_info;

NVR_API nvr_block_manager_t *nvr_bm_create(size_t num_blocks, size_t block_size);      /* :91 */
NVR_API void nvr_bm_destroy(nvr_block_manager_t *bm);
NVR_API uint64_t nvr_bm_compute_hash(const int64_t *tokens, size_t n, int has_prefix, uint64_t prefix); /* :109 */
NVR_API int nvr_bm_can_allocate(const nvr_block_manager_t *bm, const nvr_seq_t *s);    /* :152 -> 0/1 */
NVR_API int nvr_bm_allocate(nvr_block_manager_t *bm, nvr_seq_t *s);                    /* :157 */
NVR_API int nvr_bm_deallocate(nvr_block_manager_t *bm, nvr_seq_t *s);                  /* :240 */
NVR_API int nvr_bm_can_append(const nvr_block_manager_t *bm, const nvr_seq_t *s);      /* :255 -> 0/1 */
NVR_API int nvr_bm_may_append(nvr_block_manager_t *bm, nvr_seq_t *s);                  /* :265 */
NVR_API int nvr_bm_get_stats(const nvr_block_manager_t *bm, nvr_bm_stats *out);        /* :307 */
NVR_API int nvr_bm_get_block(const nvr_block_manager_t *bm, size_t block_id, nvr_block_info *out); /* :318 */
NVR_API size_t nvr_bm_free_list(const nvr_block_manager_t *bm, int32_t *out, size_t cap); /* free_block_ids order (A-2) */

/* ----------------------------------------------------------------- Scheduler ---- */
/* Scheduler, src/engine/scheduler.rs:13-365 */
typedef struct nvr_scheduler nvr_scheduler_t;
typedef struct nvr_sched_stats {       /* SchedulerStats :38-66 */
    uint64_t total_sequences, waiting_sequences, running_sequences, finished_sequences;
    uint64_t preemptions, prefill_batches, decode_batches;
    double avg_prefill_batch_size, avg_decode_batch_size;
} nvr_sched_stats;

NVR_API nvr_scheduler_t *nvr_sched_create(const nvr_config *cfg);                      /* :70 */
NVR_API void nvr_sched_destroy(nvr_scheduler_t *sc);
NVR_API int nvr_sched_add_sequence(nvr_scheduler_t *sc, nvr_seq_t *s);  /* :93; ownership moves to the scheduler */
/* :103 — writes up to cap borrowed sequence handles (valid until they finish); *is_prefill 0/1.  cap must be at least
 * min(max_num_seqs, waiting + running sequences) — the largest batch this call can build; it is checked BEFORE anything is
 * scheduled (NVR_ERR_INVALID_ARG, nothing moved) */
NVR_API int nvr_sched_schedule(nvr_scheduler_t *sc, nvr_seq_t **out, size_t cap, size_t *n, int *is_prefill);
NVR_API int nvr_sched_postprocess(nvr_scheduler_t *sc, nvr_seq_t *const *seqs, const int64_t *token_ids, size_t n); /* :234 */
NVR_API int nvr_sched_is_finished(const nvr_scheduler_t *sc);                          /* :88 */
NVR_API void nvr_sched_preempt_all(nvr_scheduler_t *sc);                               /* :314 */
NVR_API int nvr_sched_get_stats(const nvr_scheduler_t *sc, nvr_sched_stats *out);      /* :299 */
NVR_API int nvr_sched_get_block_stats(const nvr_scheduler_t *sc, nvr_bm_stats *out);   /* :304 */
NVR_API void nvr_sched_queue_lengths(const nvr_scheduler_t *sc, size_t *waiting, size_t *running); /* :309 */
NVR_API double nvr_sched_memory_pressure(const nvr_scheduler_t *sc);                   /* :322 */
NVR_API nvr_block_manager_t *nvr_sched_block_manager(nvr_scheduler_t *sc);             /* borrowed */
/* finished sequences stay owned by the scheduler until taken (SequenceOutput, sequence.rs:30-47) */
NVR_API size_t nvr_sched_take_finished(nvr_scheduler_t *sc, nvr_seq_t **out, size_t cap); /* caller destroys */

/* --------------------------------------------------------------- ModelRunner ---- */
/* ModelRunner, src/engine/model_runner.rs:19-464 */
typedef struct nvr_model_runner nvr_model_runner_t;

NVR_API nvr_model_runner_t *nvr_runner_create(const nvr_config *cfg, const nvr_model_config *mc); /* :67 */
NVR_API void nvr_runner_destroy(nvr_model_runner_t *r);
/* :105 — logits_dev: borrowed device pointer to f32 [n, vocab/tp], valid until the next execute */
NVR_API int nvr_runner_execute_model(nvr_model_runner_t *r, nvr_seq_t *const *seqs, size_t n, int is_prefill,
                                     const float **logits_dev);
/* :131 — samples from the logits of the last execute_model; writes n token ids (host) */
NVR_API int nvr_runner_sample_tokens(nvr_model_runner_t *r, nvr_seq_t *const *seqs, size_t n, int64_t *out_ids);
NVR_API int nvr_runner_copy_logits(nvr_model_runner_t *r, float *host_out, size_t rows); /* D2H, tests */
/* Weight loading (Qwen3Model::load_weights qwen3.rs:518-570, ModelLoader utils/loader.rs:43-198): one
 * FULL (un-sharded) checkpoint tensor by its name — "embed_tokens.weight", "norm.weight", "lm_head.weight",
 * "layers.N.input_layernorm.weight", "layers.N.post_attention_layernorm.weight", "layers.N.self_attn.{q,k,v,o,qkv}_proj.weight",
 * "layers.N.mlp.{gate,up,gate_up,down}_proj.weight", with or without the HF "model." prefix — is converted to fp16
 * (dtype: 0 = f16, 1 = bf16, 2 = f32; row-major, shape[ndim]) and this rank's slice is written into the packed device
 * parameter by the reference's shard rules (ColumnParallelLinear::load_weight linear.rs:154-171 narrows dim 0,
 * RowParallelLinear::load_weight :249-267 dim 1, VocabParallelEmbedding embed_head.rs:142-161).  A wrong shape is
 * NVR_ERR_LEN_MISMATCH ("Partition weight shape mismatch"), a name outside the reference graph NVR_ERR_UNSUPPORTED.
 * nvr_runner_copy_weight reads a LOCAL packed tensor back ("embed", "lm_head", "norm", "layers.N.{qkv,o,gate_up,down,ln1,ln2}");
 * host_out may be NULL to query the shape. */
NVR_API int nvr_runner_load_tensor(nvr_model_runner_t *r, const char *name, int dtype, const int64_t *shape, int ndim,
                                   const void *host_data);
NVR_API int nvr_runner_copy_weight(nvr_model_runner_t *r, const char *local_name, uint16_t *host_out, size_t cap_elems,
                                   int64_t *rows, int64_t *cols);
NVR_API uint64_t nvr_runner_num_kvcache_blocks(const nvr_model_runner_t *r);
/* KV pool of layer l (borrowed device pointers, fp16 [NB, bs, KVH/tp, D], model_runner.rs:364-396) */
NVR_API int nvr_runner_kv_cache(nvr_model_runner_t *r, size_t layer, void **k_dev, void **v_dev);
NVR_API void *nvr_runner_stream(nvr_model_runner_t *r);
/* diagnostic: re-launch the hipGraph of the last decode step n times back to back (same inputs, no host round trip; the caller
 * synchronises): the GPU's launch chain without the host gap between engine steps (execute_with_cuda_graph, model_runner.rs:303-326) */
NVR_API int nvr_runner_replay_last_decode_graph(nvr_model_runner_t *r, int n);
/* tensor-parallel wiring (the reference's TODO all-reduce/gather sites, linear.rs:236-238,
 * embed_head.rs:130-139,321-336): rank 0 creates the 128-byte RCCL id, the host broadcasts it. */
NVR_API int nvr_comm_unique_id(uint8_t id_out[128]);
NVR_API int nvr_runner_init_comm(nvr_model_runner_t *r, const uint8_t id[128]);
/* collective self-check (all ranks call it): all-reduce of ones == tensor_parallel_size, all-gather of ranks */
NVR_API int nvr_runner_comm_selftest(nvr_model_runner_t *r);
/* In-process stand-in for the communicator (tests / bring-up on a one-GPU box; no performance claim): the ranks are N runners
 * of ONE process on ONE device, each driven by its own host thread; a collective is a host rendezvous plus a device sum /
 * copies over the peers' buffers.  Same call sites and results as the RCCL path (linear.rs:236-238, embed_head.rs:321-336);
 * decode steps run eagerly.  The group outlives its runners' use of it.
 * Ranks that share a device wait for each other INSIDE kernels (the one-shot collectives), so while the ranks are stepping no thread
 * of the process may make a call that waits for the whole device — hipDeviceSynchronize, hipFree / hipMalloc, a default-stream copy,
 * destroying another engine (a garbage collector doing so counts): it would wait for a peer's kernel that waits for the caller's own
 * next launch, until NVR_P2P_TIMEOUT_MS ends the wait with NVR_ERR_RCCL.  One process per GPU (the deployment) has no such coupling. */
typedef struct nvr_local_group nvr_local_group_t;
NVR_API nvr_local_group_t *nvr_local_group_create(int nranks);
NVR_API void nvr_local_group_destroy(nvr_local_group_t *g);
NVR_API int nvr_runner_init_comm_local(nvr_model_runner_t *r, nvr_local_group_t *g);
/* 1 (default): the group's collectives are the product's one-shot peer-to-peer kernels over the ranks' arenas (the code path of
 * real multi-GPU ranks, minus the hipIpc mapping); 0: host-rendezvous collectives.  Call before any rank attaches. */
NVR_API int nvr_local_group_set_p2p(nvr_local_group_t *g, int on);
/* One-shot all-reduce (+ residual + RMSNorm) and small all-gather over peer-mapped HBM (xGMI), kernels/comm_p2p.hip: the
 * exchange sites the reference leaves as TODOs (RowParallelLinear::forward linear.rs:236-238; ParallelLMHead::gather_logits
 * embed_head.rs:321-336), as plain kernels that capture into the decode hipGraph.  Each rank exports the hipIpc handle of its
 * arena; the caller's control plane gathers all handles (rank order) and the ranks' HIP device ordinals and attaches them on
 * every rank.  Messages larger than a slot (1 MiB, prefill) keep using the RCCL communicator when one exists. */
NVR_API int nvr_runner_p2p_export(nvr_model_runner_t *r, uint8_t handle[64]);
NVR_API int nvr_runner_p2p_attach(nvr_model_runner_t *r, const uint8_t *handles /* [tp][64] */, const int32_t *devices /* [tp] or NULL */);
NVR_API int nvr_runner_p2p_disable(nvr_model_runner_t *r);
NVR_API int nvr_runner_p2p_active(const nvr_model_runner_t *r);
/* Protocol of the one-shot collectives (kernels/comm_p2p.hip).  0 (default): FENCE-FREE — every payload byte is a system-scope write-through
 * store and a system-scope load, ordered by the writers' s_waitcnt vmcnt(0) + barrier in front of the flag store; 1: FENCED — r04's release /
 * acquire fences around every hand-off (the conservative form: slower by ~11 % of a two-rank step measured on one device).  The fence-free form
 * has been measured on ranks sharing ONE GPU only; nvr_runner_comm_selftest exercises whichever is set with the largest decode message, back to back
 * on both slot parities, and a control plane falls back fence-free -> fenced -> RCCL (bench.py does; NVR_P2P_FENCED=1 starts fenced).  Every rank
 * of a group must use the same setting; switching drops the captured decode graphs (they hold the old protocol). */
NVR_API int nvr_runner_p2p_set_fenced(nvr_model_runner_t *r, int32_t on);
NVR_API int nvr_runner_p2p_fenced(const nvr_model_runner_t *r);
/* A collective whose peer never arrived (bounded wait, NVR_P2P_TIMEOUT_MS, default 20 s) fails its step with NVR_ERR_RCCL on the
 * waiting rank (the batch is aborted, linear.rs:236-238 has no error path of its own).  The ranks' epochs then differ: the caller's
 * control plane tells EVERY rank to abort that batch (nvr_engine_abort_last_batch), to call nvr_runner_p2p_reset (epoch words and
 * arrival flags back to their initial values; drains the device first) and then barriers; the group is usable again. */
NVR_API int nvr_runner_p2p_reset(nvr_model_runner_t *r);
/* forget this rank's RCCL communicator: the group agreed (control plane) that RCCL is not used because a peer failed to build its
 * own — otherwise the ranks would pick different backends for messages larger than an arena slot */
NVR_API int nvr_runner_comm_drop_rccl(nvr_model_runner_t *r);
/* tokens of the last decode step that went through the shared-prefix attention pass (nvr_config.shared_prefix_min_seqs); 0 = plain */
NVR_API int64_t nvr_runner_last_shared_prefix_len(const nvr_model_runner_t *r);
/* Where the last PREFILL step's attention read K / V (attention.rs:177-222): 0 = the step's qkv buffer (FlashAttention::forward_varlen on
 * the fresh projections), 1 = the cache rows themselves, contiguous (whole prompts whose blocks are consecutive: the qkv GEMM then writes
 * K / V once, into the caches), 2 = the caches through the block tables (cached prefixes, prompt chunks); -1 = the last step was a decode */
NVR_API int nvr_runner_last_prefill_kv_source(const nvr_model_runner_t *r);
NVR_API int64_t nvr_runner_last_shared_prefix_rows(const nvr_model_runner_t *r);   /* sequences of that step inside the sharing group */
/* 1 when the last decode step's contexts were RAGGED enough (their sum < batch x longest / 1.3, up to 1024 (sequence, kv head) pairs, >= 6 x 64 keys per CU) for the
 * work-balanced attention launch — two or three workgroups per CU walking equal shares of all pairs' keys instead of workgroups per pair sized by the longest
 * context (Attention::flash_attention_decode, attention.rs:225-235: same result within one f32 merge); 0 otherwise, and for prefill steps. */
NVR_API int32_t nvr_runner_last_decode_ragged(const nvr_model_runner_t *r);
/* Tensor-parallel PREFILL steps (row-parallel o_proj / down_proj, linear.rs:228-239 with its all-reduce :236-238): on = 1 (default) cuts a step of
 * >= 512 rows into up to 4 token chunks and runs the all-reduce of chunk i on a second HIP stream under the GEMM of chunk i + 1 (events between
 * the streams; residual add + RMSNorm of a chunk behind its reduce); on = 0 keeps GEMM -> all-reduce -> add + norm in a row on one stream.  Same
 * bits either way (chunks are whole tiles of the GEMM kernel the step is routed to: 256 or 128 rows).  on = 2: the step runs as TWO micro-batches of whole sequences, one behind the
 * other through every layer, and each exchange runs on the second stream under the OTHER micro-batch's compute segment (a third of a layer
 * instead of one GEMM: what a step needs whose exchanges outweigh its compute, e.g. configs[3] at tp 8); steps it does not fit (one group of
 * sequences, GQA groups outside the MFMA attention kernel, q/k norm or bias graphs, chunked prefill) take the chunks of mode 1.  Same bits again.
 * nvr_runner_last_overlap_chunks: chunks (mode 1) or micro-batches (mode 2) of the last such step (0: none). */
NVR_API int nvr_runner_set_tp_prefill_overlap(nvr_model_runner_t *r, int32_t on);
NVR_API int64_t nvr_runner_last_overlap_chunks(const nvr_model_runner_t *r);

/* -------------------------------------------------------------------- Engine ---- */
/* LLMEngine::step loop, src/engine/llm_engine.rs:155-197 (driver of the hot path only) */
typedef struct nvr_engine nvr_engine_t;
typedef struct nvr_step_info {
    int32_t is_prefill; uint64_t num_seqs; uint64_t num_tokens; uint64_t num_finished;
} nvr_step_info;

NVR_API nvr_engine_t *nvr_engine_create(const nvr_config *cfg, const nvr_model_config *mc); /* :34-63 */
NVR_API void nvr_engine_destroy(nvr_engine_t *e);
NVR_API int nvr_engine_add_request(nvr_engine_t *e, const int64_t *prompt, size_t n,
                                   const nvr_sampling_params *sp, uint64_t *seq_id_out);   /* :200-217 */
NVR_API int nvr_engine_step(nvr_engine_t *e, nvr_step_info *info);                         /* :155 */
NVR_API int nvr_engine_is_finished(const nvr_engine_t *e);
NVR_API nvr_scheduler_t *nvr_engine_scheduler(nvr_engine_t *e);     /* borrowed */
/* LLMEngine::get_stats / health_check / shutdown, llm_engine.rs:312-357 (EngineStats, MemoryStats, HealthStatus :360-398):
 * scheduler counters + block-pool occupancy; utilization = used / total * 100 (block_manager.rs:345-351); healthy while the
 * pool is < 95 % used; shutdown preempts every running sequence (scheduler.preempt_all) and clears is_running */
typedef struct nvr_engine_stats {
    nvr_sched_stats scheduler;
    uint64_t total_blocks, free_blocks, used_blocks; double utilization;     /* MemoryStats */
    int32_t is_running;
} nvr_engine_stats;
typedef struct nvr_health_status {
    int32_t is_healthy; double memory_pressure; uint64_t active_sequences, waiting_sequences;
} nvr_health_status;
NVR_API int nvr_engine_get_stats(nvr_engine_t *e, nvr_engine_stats *out);
NVR_API int nvr_engine_health_check(nvr_engine_t *e, nvr_health_status *out);
NVR_API int nvr_engine_shutdown(nvr_engine_t *e);
NVR_API nvr_model_runner_t *nvr_engine_runner(nvr_engine_t *e);     /* borrowed */
/* ids + tokens sampled by the last step (borrowed until the next step) */
NVR_API void nvr_engine_last_step(const nvr_engine_t *e, const uint64_t **seq_ids, const int64_t **tokens, size_t *n);
NVR_API size_t nvr_engine_take_finished(nvr_engine_t *e, nvr_seq_t **out, size_t cap);   /* caller destroys */
/* Control-plane abort (tensor-parallel ranks after a peer reported NVR_ERR_RCCL; the reference's step has no error path,
 * llm_engine.rs:155-197): the sequences of the batch this engine scheduled last that are still alive leave the engine as finished,
 * their blocks returned — what a failed model step does on the rank that saw the failure.  Safe before or after
 * taking (and destroying) that step's finished sequences: the batch is matched against the scheduler's queues by handle, never dereferenced. */
NVR_API int nvr_engine_abort_last_batch(nvr_engine_t *e);
/* nvr_config.async_decode: decode steps whose successor could not be enqueued ahead and therefore ran synchronously (diagnostic) */
NVR_API uint64_t nvr_engine_ahead_declined(const nvr_engine_t *e);
NVR_API uint64_t nvr_engine_ahead_launched(const nvr_engine_t *e);   /* decode steps that were enqueued ahead */
/* Host time of the integer side of the steps so far: out3[0] = microseconds inside Scheduler::schedule (scheduler.rs:103-223, block
 * manager calls included), out3[1] = inside Scheduler::postprocess (:234-257), out3[2] = steps counted.  Always on (four clock reads per step). */
NVR_API void nvr_engine_host_times(const nvr_engine_t *e, double *out3);
/* sequences of the last step's batch (borrowed handles; finished ones are excluded) */
NVR_API size_t nvr_engine_last_batch(const nvr_engine_t *e, nvr_seq_t **out, size_t cap);

/* Text in, SequenceOutput out.  LLMEngine::tokenize (llm_engine.rs:220-230) is the reference's
 * placeholder tokenizer: one token id per Unicode scalar value of the prompt, first 100 characters only (its `tokenizers`
 * dependency is unused, Cargo.toml:21); nvr_tokenize restates it for UTF-8 input (malformed UTF-8, which a Rust String
 * cannot hold, is NVR_ERR_INVALID_ARG) and nvr_detokenize is its inverse (ids that are not scalar values — surrogates,
 * negatives, > 0x10FFFF — become U+FFFD).  out may be NULL to query the length. */
#define NVR_TOKENIZE_MAX_CHARS 100
NVR_API int nvr_tokenize(const char *utf8, size_t nbytes, int64_t *out, size_t cap, size_t *n);
NVR_API int nvr_detokenize(const int64_t *ids, size_t n, char *out, size_t cap, size_t *nbytes);

/* SequenceOutput, sequence.rs:30-47.  Pointers are borrowed from the engine: valid until the next generate /
 * generate_stream call on it (inside a stream callback: until the callback returns). */
typedef struct nvr_sequence_output {
    uint64_t seq_id;
    const char *text; size_t text_len;            /* detokenized completion, UTF-8 (also NUL-terminated) */
    const int64_t *token_ids; size_t num_tokens;  /* prompt + completion */
    const int64_t *completion_token_ids;          /* = token_ids + num_prompt_tokens */
    size_t num_prompt_tokens, num_completion_tokens;
    int32_t status;                               /* NVR_SEQ_* */
} nvr_sequence_output;
/* create_sequences for one prompt (llm_engine.rs:200-217): tokenize + Sequence::new + scheduler.add_sequence */
NVR_API int nvr_engine_add_prompt(nvr_engine_t *e, const char *utf8, size_t nbytes, const nvr_sampling_params *sp,
                                  uint64_t *seq_id_out);
/* LLMEngine::generate (llm_engine.rs:70-97): tokenize the prompts, queue them with one SamplingParams, step until the
 * scheduler is finished (run_inference_loop :131-152), return one output per prompt in prompt order (the reference leaves
 * the collection as a "real implementation would" note, :188-196; this is that implementation).  n == 0 -> no outputs.
 * Sequences queued earlier through add_request / add_prompt are stepped too and stay with take_finished. */
NVR_API int nvr_engine_generate(nvr_engine_t *e, const char *const *prompts, const size_t *nbytes, size_t n,
                                const nvr_sampling_params *sp, const nvr_sequence_output **outs, size_t *nout);
/* the same over token-id prompts (the placeholder tokenizer cannot express ids that are not scalar values) */
NVR_API int nvr_engine_generate_ids(nvr_engine_t *e, const int64_t *const *prompts, const size_t *lens, size_t n,
                                    const nvr_sampling_params *sp, const nvr_sequence_output **outs, size_t *nout);
/* LLMEngine::generate_stream (llm_engine.rs:100-128, run_streaming_inference :233-262): after every step the callback
 * receives one cumulative output per sequence of the step's batch (status RUNNING, or FINISHED on its last one) — the
 * items the reference sends through its mpsc channel; a non-zero return is the dropped receiver (:250-253): the loop
 * stops, NVR_OK is returned and unfinished sequences stay queued (step / shutdown deal with them, as in the reference).
 * The call is synchronous on the caller's thread (the Rust shim runs it inside its spawned task). */
typedef int (*nvr_stream_fn)(const nvr_sequence_output *out, void *user);
NVR_API int nvr_engine_generate_stream(nvr_engine_t *e, const char *const *prompts, const size_t *nbytes, size_t n,
                                       const nvr_sampling_params *sp, nvr_stream_fn fn, void *user);

/* --------------------------------------------------------- device utilities ---- */
NVR_API int nvr_device_count(int *n);
NVR_API int nvr_device_set(int ordinal);
NVR_API int nvr_device_name(char *buf, size_t cap);
NVR_API int nvr_device_mem_info(uint64_t *free_bytes, uint64_t *total_bytes);
NVR_API int nvr_device_malloc(void **ptr, size_t bytes);
NVR_API int nvr_device_free(void *ptr);
NVR_API int nvr_device_memset(void *ptr, int value, size_t bytes);
NVR_API int nvr_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
NVR_API int nvr_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
NVR_API int nvr_device_synchronize(void);
NVR_API int nvr_stream_create(void **stream);
NVR_API int nvr_stream_destroy(void *stream);
NVR_API int nvr_stream_synchronize(void *stream);
NVR_API int nvr_event_create(void **ev);
NVR_API int nvr_event_destroy(void *ev);
NVR_API int nvr_event_record(void *ev, void *stream);
NVR_API int nvr_stream_wait_event(void *stream, void *ev);   /* also forks / joins streams inside a graph capture */
NVR_API int nvr_event_elapsed_ms(void *start, void *stop, float *ms);   /* synchronises on stop */
/* hipGraph capture of stateless-op calls on a caller-created stream (measurement harnesses; the engine captures its own decode
 * steps, execute_with_cuda_graph model_runner.rs:303-326): ops enqueued between begin and end become one replayable graph */
NVR_API int nvr_graph_capture_begin(void *stream);
NVR_API int nvr_graph_capture_end(void *stream, void **graph_exec);
NVR_API int nvr_graph_launch(void *graph_exec, void *stream);
NVR_API int nvr_graph_destroy(void *graph_exec);

/* ------------------------------------------------ stateless op entry points ---- */
/* One per kernel-shaped op site of the hot path.  All tensors are
 * device pointers; activations/weights/caches are 16-bit (nvr_half = uint16_t bits: fp16 by
 * default, bfloat16 after nvr_ops_set_dtype("bfloat16")) unless noted; math is f32 inside an
 * op.  `stream` is a hipStream_t (NULL = default stream). */
typedef uint16_t nvr_half;
/* Config.dtype (config.rs:51,113-116) for the stateless entry points below: "float16" | "bfloat16" select the fp16 / bf16 build of the
 * kernels for the CALLING THREAD (thread-local, like nvr_last_error); "float32" (r04) selects the ops of the reference-precision path — the
 * nvr_half pointers of nvr_embedding, nvr_rmsnorm, nvr_add_rmsnorm, nvr_linear, nvr_rope_store_kv, nvr_qk_norm_rope_store_kv, nvr_silu_and_mul,
 * nvr_select_last_tokens, nvr_attn_prefill_varlen / _paged, nvr_paged_attn_decode, nvr_fill_weight (unrounded values) and nvr_fill_const then
 * address f32 elements; nvr_linear_qkv_rope_store and nvr_linear_silu_mul exist there for decode-sized steps (1..8 rows: r05, the bits of their parts;
 * NVR_ERR_UNSUPPORTED for more rows); the ops that exist only as fused 16-bit kernels answer NVR_ERR_UNSUPPORTED; anything else: NVR_ERR_UNSUPPORTED.
 * Runners and engines take their type from nvr_config.dtype and are not affected. */
NVR_API int nvr_ops_set_dtype(const char *dtype);
NVR_API const char *nvr_ops_dtype(void);

/* Attention metadata of one step = Context, src/utils/context.rs:11-35 (explicit, not global) */
typedef struct nvr_attn_meta {
    int32_t is_prefill;
    const int32_t *cu_seqlens_q;   /* [B+1] device; prefill */
    const int32_t *cu_seqlens_k;   /* [B+1] device; prefill (== q today, model_runner.rs:233-234) */
    int32_t max_seqlen_q, max_seqlen_k;
    const int32_t *slot_mapping;   /* [T] device; slot = block*bs + offset (A-6), <0 skips */
    const int32_t *context_lens;   /* [B] device; decode */
    const int32_t *block_tables;   /* [B, max_blocks] device, -1 padded; decode / prefix prefill */
    int32_t max_blocks;
    int32_t batch;                 /* B */
    int32_t max_context_len;       /* host-side max of context_lens (decode grid sizing) */
} nvr_attn_meta;

/* K1 embedding gather, embed_head.rs:77-97 */
NVR_API int nvr_embedding(const int64_t *ids, int64_t T, const nvr_half *E, int64_t Hd, nvr_half *out, void *stream);
/* K2 RMSNorm, layernorm.rs:58-75 */
NVR_API int nvr_rmsnorm(const nvr_half *x, const nvr_half *w, float eps, int64_t T, int64_t Hd, nvr_half *out, void *stream);
/* K11+K2 fused: h = fp16(h + y) in place, out = rmsnorm(h)*w  (layernorm.rs:170-176) */
NVR_API int nvr_add_rmsnorm(nvr_half *h, const nvr_half *y, const nvr_half *w, float eps, int64_t T, int64_t Hd,
                            nvr_half *out, void *stream);
/* K3/K10/K12/K14/K16 y = x·Wᵀ; x [T,K] (row stride ldx), W [N,K], y [T,N] fp16 or f32 */
NVR_API int nvr_linear(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t N,
                       void *y, int y_is_f32, void *stream);
/* K16: logits[T,N] f32 = x·Wᵀ (ParallelLMHead::compute_logits, embed_head.rs:292-306) plus the greedy arg-max of every row
 * folded into the epilogue (Sampler::sample greedy branch, sampler.rs:126-151): part_val/part_idx [*nparts][T] hold
 * per-workgroup (max, lowest index) pairs, *nparts <= NVR_LM_HEAD_MAX_PARTS (size the buffers for that); nvr_argmax_partials
 * merges them (ties -> lowest index), adds idx_offset, out_val nullable.  T <= 32 (K multiple of 256 and <= 2048): the
 * weight-streaming decode kernel; T > 32 (K multiple of 64, N <= 128·NVR_LM_HEAD_MAX_PARTS): 128x128 MFMA tiles, one partial
 * per 128 columns; N a multiple of 16; other shapes NVR_ERR_UNSUPPORTED (use nvr_linear + nvr_argmax).
 * logits == NULL: only the partials are produced (a greedy batch never reads its 4·T·N logit bytes). */
#define NVR_LM_HEAD_MAX_PARTS 2048
NVR_API int nvr_lm_head(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t N, float *logits,
                        float *part_val, int32_t *part_idx, int32_t *nparts, void *stream);
NVR_API int nvr_argmax_partials(const float *part_val, const int32_t *part_idx, int32_t nparts, int64_t T, int64_t *out_idx,
                                float *out_val, int64_t idx_offset, void *stream);
/* split-k form of the narrow row-parallel GEMMs (o_proj / down_proj, linear.rs:228-239) for T <= 64:
 * slabs[z][T][N] f32 partial sums over k-slice z, consumed by nvr_add_rmsnorm_slabs
 * (h <- fp16(h + fp16(sum_z slabs[z])), out = rmsnorm(h)·w; layernorm.rs:170-176) */
NVR_API int nvr_linear_splitk(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t N, int64_t S,
                              float *slabs, void *stream);
NVR_API int nvr_add_rmsnorm_slabs(nvr_half *h, const float *slabs, int64_t S, const nvr_half *w, float eps, int64_t T,
                                  int64_t Hd, nvr_half *out, void *stream);
/* Tiled copy of a row-major weight W[N][K] for the weight-streaming decode kernels: dst[N/16][K/32][16 rows][32 k], so that one
 * 16x32 MFMA operand tile is 1 KiB contiguous (the runner keeps such copies of qkv / o / gate_up / down / LM head next to the
 * row-major parameters and reads them in decode-sized steps: same values, same summation order, same bits).  mode 0: tile t =
 * rows 16t..16t+15; mode 1: the qkv row order of the RoPE epilogue (rotation partners of a q / k head in one tile; needs H, KVH, D). */
NVR_API int nvr_retile_weight(const nvr_half *src, nvr_half *dst, int64_t N, int64_t K, int mode, int64_t H, int64_t KVH, int64_t D,
                              void *stream);
/* The same ops reading a tiled copy: W stays the row-major weight (taken when the shape routes to a tile GEMM, T > 64), Wt is its
 * nvr_retile_weight copy (mode 1 for qkv, mode 0 otherwise) and is what the weight-streaming decode kernels read.  Results are
 * bit-identical to the row-major entry points (tests/test_kernels_gpu.py::test_tiled_entry_points_match_row_major). */
NVR_API int nvr_linear_tiled(const nvr_half *x, int64_t ldx, const nvr_half *W, const nvr_half *Wt, int64_t T, int64_t K, int64_t N,
                             void *y, int y_is_f32, void *stream);
NVR_API int nvr_linear_splitk_tiled(const nvr_half *x, int64_t ldx, const nvr_half *W, const nvr_half *Wt, int64_t T, int64_t K,
                                    int64_t N, int64_t S, float *slabs, void *stream);
NVR_API int nvr_linear_silu_mul_tiled(const nvr_half *x, int64_t ldx, const nvr_half *W, const nvr_half *Wt, int64_t T, int64_t K,
                                      int64_t I, nvr_half *out, void *stream);
NVR_API int nvr_linear_qkv_rope_store_tiled(const nvr_half *x, int64_t ldx, const nvr_half *W, const nvr_half *Wt, int64_t T,
                                            int64_t K, int64_t H, int64_t KVH, int64_t D, const int64_t *positions,
                                            const int32_t *slot_mapping, const float *cos_t, const float *sin_t,
                                            nvr_half *qkv, nvr_half *k_cache, nvr_half *v_cache, void *stream);
NVR_API int nvr_lm_head_tiled(const nvr_half *x, int64_t ldx, const nvr_half *W, const nvr_half *Wt, int64_t T, int64_t K, int64_t N,
                              float *logits, float *part_val, int32_t *part_idx, int32_t *nparts, void *stream);
/* k-slices S of nvr_linear_splitk for the N = hidden GEMMs of a decode-sized step (o_proj, down_proj): enough to reach ~256 workgroups */
NVR_API int nvr_decode_splitk_slices(int64_t T, int64_t K, int64_t N);
/* h[T,N] <- fp16(h + fp16(x · Wᵀ)) with the residual add (qwen3.rs:382,389) in the epilogue of the
 * 256x256 MFMA GEMM; only for shapes that kernel takes (T >= 256, N % 256 == 0, K % 64 == 0 and preferred by the routing of nvr_linear:
 * NVR_ERR_UNSUPPORTED otherwise).  Bit-identical to nvr_linear into a scratch tensor followed by the add of nvr_add_rmsnorm. */
NVR_API int nvr_linear_add_residual(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t N, nvr_half *h,
                                    void *stream);
/* K12+K13 fused: out[T,I] = SiluAndMul(x · W_gate_upᵀ), W [2I,K] gate rows then up rows
 * (MergedColumnParallelLinear::forward linear.rs:437-439 + SiluAndMul::forward activation.rs:46-63) */
NVR_API int nvr_linear_silu_mul(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K, int64_t I,
                                nvr_half *out, void *stream);
/* K3..K6 fused: qkv = x · W_qkvᵀ (QKVParallelLinear linear.rs:354-356), RoPE on the q and k heads
 * (rotary_embedding.rs:145-158), k,v rows stored at slot_mapping (attention.rs:150-174) */
NVR_API int nvr_linear_qkv_rope_store(const nvr_half *x, int64_t ldx, const nvr_half *W, int64_t T, int64_t K,
                                      int64_t H, int64_t KVH, int64_t D, const int64_t *positions,
                                      const int32_t *slot_mapping, const float *cos_t, const float *sin_t,
                                      nvr_half *qkv, nvr_half *k_cache, nvr_half *v_cache, void *stream);
/* K5+K6 RoPE (rotary_embedding.rs:23-48) on the q and k heads of a packed qkv buffer
 * [T, (H+2KVH)*D] in place, then store k,v rows to the caches at slot_mapping (attention.rs:150-174) */
NVR_API int nvr_rope_store_kv(nvr_half *qkv, const int64_t *positions, const int32_t *slot_mapping, int64_t T,
                              int64_t H, int64_t KVH, int64_t D, const float *cos_t, const float *sin_t,
                              nvr_half *k_cache, nvr_half *v_cache, void *stream);
/* the same with RMSNorm::forward_simple (layernorm.rs:58-75) over head_dim on every q and k head in front of the rotation
 * (nvr_model_config.qk_norm; q_norm_w / k_norm_w [D]) */
NVR_API int nvr_qk_norm_rope_store_kv(nvr_half *qkv, const int64_t *positions, const int32_t *slot_mapping, int64_t T,
                                      int64_t H, int64_t KVH, int64_t D, const float *cos_t, const float *sin_t,
                                      const nvr_half *q_norm_w, const nvr_half *k_norm_w, float eps,
                                      nvr_half *k_cache, nvr_half *v_cache, void *stream);
NVR_API int nvr_rope_table(int64_t D, int64_t max_pos, double theta, float *cos_dev, float *sin_dev); /* rotary_embedding.rs:74-119 */
/* K9 decode paged attention, attention.rs:225-235,264-318 (A-8).  q rows have stride ldq
 * elements ([B, H, D] inside the packed qkv buffer).  workspace: nvr_paged_attn_workspace_bytes. */
NVR_API size_t nvr_paged_attn_workspace_bytes(int64_t B, int64_t H, int64_t D, int64_t max_context_len);
NVR_API int nvr_paged_attn_decode(const nvr_half *q, int64_t ldq, const nvr_half *k_cache, const nvr_half *v_cache,
                                  const nvr_attn_meta *meta, int64_t H, int64_t KVH, int64_t D, int64_t block_size,
                                  float scale, nvr_half *out, void *workspace, void *stream);
/* The same in ONE launch when the context is cut into split-KV partitions (batches with fewer than ~192 (sequence, kv head) pairs: small
 * batches, tensor-parallel ranks holding 1-4 kv heads): the last partition workgroup of a (sequence, kv head) to finish merges the pair's
 * partials, instead of a second (merge) launch; bit-identical to nvr_paged_attn_decode.  tickets: [B * KVH] device words, ZERO before the
 * first call and owned by these calls afterwards (the kernel re-arms them; one call at a time per array). */
NVR_API int nvr_paged_attn_decode_fused(const nvr_half *q, int64_t ldq, const nvr_half *k_cache, const nvr_half *v_cache,
                                        const nvr_attn_meta *meta, int64_t H, int64_t KVH, int64_t D, int64_t block_size,
                                        float scale, nvr_half *out, void *workspace, uint32_t *tickets, void *stream);
/* The same when EVERY sequence of the batch holds its first shared_len tokens (a multiple of block_size) in the cache blocks that
 * block-table row 0 starts with (prefix-cache hits of BlockManager::allocate, block_manager.rs:181-197; BASELINE configs[4]): the
 * shared keys go through one MFMA pass for the whole batch, the remainder per sequence, merged as split-KV partials.  Same
 * semantics as nvr_paged_attn_decode (A-8); workspace sized by nvr_paged_attn_workspace_bytes.
 * rows / kv0 / count (device arrays, all three or all NULL): only the sequences rows[0 .. *count) share the prefix (the block table
 * of rows[0] names the shared blocks); kv0[b] = shared_len for those, 0 for every other sequence, which is attended to in full by
 * the per-sequence kernel. */
NVR_API int nvr_paged_attn_decode_shared(const nvr_half *q, int64_t ldq, const nvr_half *k_cache, const nvr_half *v_cache,
                                         const nvr_attn_meta *meta, int64_t H, int64_t KVH, int64_t D, int64_t block_size,
                                         float scale, int64_t shared_len, const int32_t *rows, const int32_t *kv0, const int32_t *count,
                                         nvr_half *out, void *workspace, void *stream);
/* K7 varlen causal prefill attention, attention.rs:177-208 (q,k,v inside packed qkv, stride ld) */
NVR_API int nvr_attn_prefill_varlen(const nvr_half *q, const nvr_half *k, const nvr_half *v, int64_t ld,
                                    const nvr_attn_meta *meta, int64_t T, int64_t H, int64_t KVH, int64_t D,
                                    float scale, nvr_half *out, void *stream);
/* K8 prefix-cached prefill attention, attention.rs:211-222,264-318: the nq_b queries of sequence b (cu_seqlens_q) sit at
 * positions context_lens[b]-nq_b .. context_lens[b]-1 and read K/V (cached prefix + the new tokens, already stored)
 * through the block table; causal inside the new tokens */
NVR_API int nvr_attn_prefill_paged(const nvr_half *q, int64_t ldq, const nvr_half *k_cache, const nvr_half *v_cache,
                                   const nvr_attn_meta *meta, int64_t T, int64_t H, int64_t KVH, int64_t D, int64_t block_size,
                                   float scale, nvr_half *out, void *stream);
/* K13 SiluAndMul, activation.rs:46-63: [T,2I] -> [T,I] */
NVR_API int nvr_silu_and_mul(const nvr_half *x, int64_t T, int64_t I, nvr_half *out, void *stream);
/* The rest of src/layers/activation.rs (r06; not on the Qwen3 path, which uses SiluAndMul only): ActivationType (:111-117), its FromStr (:169-182) and
 * Activation::forward (:147-159).  nvr_activation: [T, cols] -> [T, cols] for NVR_ACT_SILU (:12-15), NVR_ACT_GELU (:20-22: candle's tanh form) and
 * NVR_ACT_RELU (:25-27); [T, cols] -> [T, cols / 2] = act(x[:, :cols/2]) * x[:, cols/2:] for NVR_ACT_SILU_AND_MUL (:46-63) and NVR_ACT_GELU_AND_MUL
 * (:74-100) — an odd cols is NVR_ERR_INVALID_ARG with the reference's message (:50-52, :88-90); output columns % 8 == 0.  f32 inside, one rounding to
 * the ops' 16-bit type at the store; under nvr_ops_set_dtype("float32") the pointers address f32 elements.  nvr_activation_type_from_str: "silu" |
 * "swish" | "gelu" | "relu" | "silu_and_mul" | "siluandmul" | "gelu_and_mul" | "geluandmul", case-insensitive; anything else NVR_ERR_INVALID_ARG
 * ("Unknown activation function: ..."). */
typedef enum nvr_activation_type { NVR_ACT_SILU = 0, NVR_ACT_GELU = 1, NVR_ACT_RELU = 2, NVR_ACT_SILU_AND_MUL = 3, NVR_ACT_GELU_AND_MUL = 4 } nvr_activation_type;
NVR_API int nvr_activation_type_from_str(const char *name, int32_t *type_out);
NVR_API int nvr_activation(int32_t type, const nvr_half *x, int64_t T, int64_t cols, nvr_half *out, void *stream);
/* The bias of a Linear with Qwen3Config::use_bias (A-30): y[T,N] <- 16bit(y + b[N]) in place, N % 8 == 0 — candle_nn::Linear::forward is
 * matmul, then broadcast_add, each rounding to the tensor dtype; linear.rs:124-139 (column-parallel: the local slice of b),
 * :206 / :228-239 (row-parallel: rank 0 only, before the all-reduce) */
NVR_API int nvr_add_bias(nvr_half *y, const nvr_half *b, int64_t T, int64_t N, void *stream);
/* K15 last-token select, embed_head.rs:272-289 */
NVR_API int nvr_select_last_tokens(const nvr_half *h, const int32_t *cu_seqlens_q, int64_t B, int64_t Hd,
                                   nvr_half *out, void *stream);
/* K17 greedy argmax over f32 logits [B,V] (lowest index wins, A-12) */
NVR_API int nvr_argmax(const float *logits, int64_t B, int64_t V, int64_t *out_ids, void *stream);
/* K18 temperature / top-k / top-p / Gumbel-max, sampler.rs:71-218 (A-18..A-20).  Per-row
 * params are device arrays; top_k[i]==0 and top_p[i]<0 mean disabled; keys[i] = counter-RNG key. */
NVR_API size_t nvr_sample_workspace_bytes(int64_t B, int64_t V);
NVR_API int nvr_sample(const float *logits, int64_t B, int64_t V, const float *temperature, const int64_t *top_k,
                       const float *top_p, const uint64_t *keys, int64_t *out_ids, void *workspace, void *stream);
NVR_API uint64_t nvr_sample_key(uint64_t seed, uint64_t seq_id, uint64_t step);
/* synthetic weights: dst[i*ld+j] = fp16(value(key, (row0+i)*global_cols + col0+j)) */
NVR_API uint64_t nvr_weight_key(uint64_t seed, uint64_t tensor_id);
NVR_API float nvr_weight_scale(double std);
NVR_API int nvr_fill_weight(nvr_half *dst, int64_t rows, int64_t cols, int64_t ld, int64_t global_cols,
                            int64_t row0, int64_t col0, uint64_t key, float scale, void *stream);
NVR_API int nvr_fill_const(nvr_half *dst, int64_t n, float value, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NVR_H */

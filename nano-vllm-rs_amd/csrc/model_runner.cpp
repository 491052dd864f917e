// model_runner.cpp — see model_runner.h.  Line cites: reference src/engine/model_runner.rs unless
// another file is named.
#include "model_runner.h"
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"
#include "kernels/kernels.h"
#include "kernels/device_utils.h"

namespace k = nvr::k;
// dtype dispatch: the kernels exist twice (fp16 in nvr::k, bfloat16 in nvr::kb: kernels/device_utils.h); `bf16` is the runner's
// Config.dtype == "bfloat16" (reference src/config.rs:51,113-116).  Argument structs and constants are shared (nvr::kt via nvr::k).
#define KD(call) (bf16 ? nvr::kb::call : nvr::k::call)

// tensor ids of the synthetic weight generator (twin of oracle/model_oracle.py)
enum { TID_QKV = 0, TID_O = 1, TID_GATE_UP = 2, TID_DOWN = 3, TID_BIAS = 4 };   // TID_BIAS + id: the bias of that projection (use_bias)
static const uint64_t TID_EMBED = 1ull << 20, TID_LM_HEAD = (1ull << 20) + 1;

uint64_t nvr_weight_key_impl(uint64_t seed, uint64_t tid) {
    return nvr::splitmix64(nvr::splitmix64(seed) + tid * 0xD1B54A32D192ED03ULL);
}
float nvr_weight_scale_impl(double std) {
    return (float)(std / std::sqrt(4.0 * (65536.0 * 65536.0 - 1.0) / 12.0));
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

template <class T>
static int dmalloc(T **p, size_t count) {
    NVR_HIP_CHECK(hipMalloc((void **)p, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
    return NVR_OK;
}

nvr_model_runner::~nvr_model_runner() {
    if (stream) hipStreamSynchronize(stream);
    for (auto &g : graphs) hipGraphExecDestroy(g.second);
    comm.destroy();
    if (attn_tickets) hipFree(attn_tickets);
    for (auto &l : layers) { hipFree(l.qkv); hipFree(l.o); hipFree(l.gate_up); hipFree(l.down); hipFree(l.ln1); hipFree(l.ln2);
                             void *ts[] = {l.qkv_t, l.o_t, l.gate_up_t, l.down_t, l.q_norm, l.k_norm, l.qkv_b, l.o_b, l.gate_up_b, l.down_b}; for (void *t : ts) if (t) hipFree(t); }
    if (lm_head_t) hipFree(lm_head_t);
    void *ptrs[] = {embed, mc.tie_word_embeddings ? nullptr : lm_head, norm, cos_t, sin_t, kv_pool, h, n, qkv, attn,
                    proj, gu, act, nlast, logits, attn_ws, slabs, in_dev, d_tok, d_maxval, d_temp, d_topk, d_topp, d_keys,
                    sample_ws, d_gather_val, d_gather_idx, d_gather_logits, d_full_logits, sample_ws_full, d_lm_pval, d_lm_pidx, d_rec, d_gather_rec, f32_gather, f32_h2};
    for (void *p : ptrs) if (p) hipFree(p);
    if (in_host) hipHostFree(in_host);
    if (h_tok) hipHostFree(h_tok);
    for (int i = 0; i < 2; ++i) { if (ahead_tok[i]) hipHostFree(ahead_tok[i]); if (ahead_host[i]) hipHostFree(ahead_host[i]); if (lm_snap[i]) hipFree(lm_snap[i]); }
    if (samp_host) hipHostFree(samp_host);
    for (int i = 0; i < kMaxChunks; ++i) { if (ev_gemm[i]) hipEventDestroy(ev_gemm[i]); if (ev_reduced[i]) hipEventDestroy(ev_reduced[i]); }
    if (comm_stream) hipStreamDestroy(comm_stream);
    if (stream) hipStreamDestroy(stream);
}

int nvr_model_runner::init() {                                       // ModelRunner::new, :67-102
    env = nvr::Env::read();                                          // the only place the runner looks at the environment
    tp = (int64_t)cfg.tensor_parallel_size; rank = (int64_t)cfg.tensor_parallel_rank;
    bf16 = std::strcmp(cfg.dtype, "bfloat16") == 0;                               // config.rs:51 (fp16 otherwise)
    f32 = std::strcmp(cfg.dtype, "float32") == 0;                                 // the reference-precision path (kernels/f32_path.hip): 4-byte storage,
    em = f32 ? 2 : 1;                                                             // every "16-bit" buffer below holds em x 2 bytes per element
    if (f32) { tiled_weights = false; lm_fused = false; RC(nvr::kf::prepare()); }   // (decode steps of the f32 path replay a captured graph like the 16-bit ones)
    comm.bf16 = bf16;                                                              // the collectives round their sums to the same 16-bit type
    if (tp < 1 || rank >= tp) return nvr::fail(NVR_ERR_INVALID_ARG, "bad tensor parallel rank %ld of %ld", (long)rank, (long)tp);
    RC(nvr_model_config_validate(&mc, (uint64_t)tp));
    Hd = mc.hidden_size; L = mc.num_hidden_layers; V = mc.vocab_size;
    D = mc.head_dim ? mc.head_dim : mc.hidden_size / mc.num_attention_heads;     // qwen3.rs:101-103 (+A-17)
    H = mc.num_attention_heads / tp; KVH = mc.num_key_value_heads / tp;           // qwen3.rs:158-159
    I = mc.intermediate_size / tp;
    QKV = (H + 2 * KVH) * D;
    Vl = V / tp; vocab_start = rank * Vl; if (rank == tp - 1) Vl = V - vocab_start;   // embed_head.rs:57-59
    scale = 1.0f / std::sqrt((float)D);                                           // attention.rs:45
    block_size = cfg.kvcache_block_size;
    max_tokens = cfg.max_num_batched_tokens; max_seqs = cfg.max_num_seqs;
    max_pos = std::min<int64_t>(mc.max_position_embeddings, std::max<int64_t>(cfg.max_model_len, 1));
    max_blocks_per_seq = (max_pos + block_size - 1) / block_size + 1;
    if (f32) {
        // the f32 attention kernel keeps a query's scores in LDS (kernels/f32_path.hip): decode steps are launched for the 256-token context
        // bucket, so refuse here what would otherwise fail at step time, inside a stream capture
        const int64_t bucket = (max_pos + 255) / 256 * 256;
        if ((bucket + 5 * D + 8) * 4 > 160 * 1024)
            return nvr::fail(NVR_ERR_UNSUPPORTED, "dtype float32: max_model_len %ld (context bucket %ld) does not fit the attention kernel's score buffer "
                             "(%ld bytes of LDS, 163840 available): at most %ld tokens at head_dim %ld", (long)max_pos, (long)bucket,
                             (long)((bucket + 5 * D + 8) * 4), (long)((160 * 1024 / 4 - 5 * D - 8) / 256 * 256), (long)D);
    }

    device = cfg.device_ordinal;
    NVR_HIP_CHECK(hipSetDevice(device));
    NVR_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (tp > 1) {
        NVR_HIP_CHECK(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
        for (int i = 0; i < kMaxChunks; ++i) {
            NVR_HIP_CHECK(hipEventCreateWithFlags(&ev_gemm[i], hipEventDisableTiming));
            NVR_HIP_CHECK(hipEventCreateWithFlags(&ev_reduced[i], hipEventDisableTiming));
        }
    }

    RC(gen_weights());

    // RoPE tables (rotary_embedding.rs:74-119, A-14): computed on the host in f64 from the f32 angle,
    // so the oracle (same libm, same box) holds bit-identical tables.
    {
        const int64_t half = D / 2;
        std::vector<float> c(max_pos * half), sn(max_pos * half);
        for (int64_t p = 0; p < max_pos; ++p)
            for (int64_t j = 0; j < half; ++j) {
                float inv = (float)(1.0 / std::pow(mc.rope_theta, (double)(2 * j) / (double)D));
                float ang = (float)p * inv;
                c[p * half + j] = (float)std::cos((double)ang);
                sn[p * half + j] = (float)std::sin((double)ang);
            }
        RC(dmalloc(&cos_t, c.size())); RC(dmalloc(&sin_t, sn.size()));
        NVR_HIP_CHECK(hipMemcpy(cos_t, c.data(), c.size() * 4, hipMemcpyHostToDevice));
        NVR_HIP_CHECK(hipMemcpy(sin_t, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    }

    // activations (persistent: graph-replayable, no allocation on the step path)
    RC(dmalloc(&h, em * max_tokens * Hd)); RC(dmalloc(&n, em * max_tokens * Hd)); RC(dmalloc(&qkv, em * max_tokens * QKV));
    RC(dmalloc(&attn, em * max_tokens * H * D)); RC(dmalloc(&proj, em * max_tokens * Hd)); RC(dmalloc(&gu, em * max_tokens * 2 * I));
    RC(dmalloc(&act, em * max_tokens * I)); RC(dmalloc(&nlast, em * max_seqs * Hd)); RC(dmalloc(&logits, max_seqs * Vl));
    // float32 tensor-parallel ranks gather every rank's partial sums before they add them (allocated here, not on first use: an allocation
    // synchronises the device, and with the in-process group a peer's collective may already be spinning on it)
    if (f32 && tp == 1 && env.f32_fused_norm) NVR_HIP_CHECK(hipMalloc((void **)&f32_h2, (size_t)8 * (size_t)Hd * sizeof(float)));
    f32_gather_rows = std::min<int64_t>(max_tokens, std::max<int64_t>(max_seqs, 2048));
    if (f32 && tp > 1) NVR_HIP_CHECK(hipMalloc((void **)&f32_gather, (size_t)tp * (size_t)f32_gather_rows * (size_t)Hd * sizeof(float)));
    // (rows of a STEP, prefill steps included: sized by max_seqs alone, a prefill of 257..1024 tokens on an engine of few sequences lost the split-k route
    //  and fell to 9-12 row blocks of the streaming kernel — Qwen3-0.6B, 8 sequences: 257 tokens 2.38 ms against 1.49 ms for 256; scratch/prefill_scan.py)
    slab_rows = std::max<int64_t>(64, std::min<int64_t>(1024, std::max<int64_t>({max_seqs, max_tokens, 256})));
    RC(dmalloc(&slabs, 4 * slab_rows * Hd));
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) num_cus = v; }
    allow_missing_comm = env.tp_no_comm;                                               // compute-only profiling of one rank
    if (!env.tp_graph && tp > 1) graphs_disabled = true;
    comm.force = env.tp_force_comm; comm.timeout_ms = env.p2p_timeout_ms; comm.p2p_fenced = env.p2p_fenced;
    RC(KD(linear_stream_prepare()));
    RC(KD(gemm_tiled_prepare()));
    lazy_logits = env.lazy_logits && !f32;
    {   // arg-max partials [parts][rows]: <= LM_HEAD_MAX_PARTS x 32 rows (lm_head_kernel), or one per 128 vocabulary columns x all rows
        const size_t pe = std::max<size_t>((size_t)k::LM_HEAD_MAX_PARTS * 32, (size_t)((Vl + 127) / 128) * (size_t)max_seqs);
        RC(dmalloc(&d_lm_pval, pe)); RC(dmalloc(&d_lm_pidx, pe));
    }
    // the hipGraph decode path launches attention with max_ctx = the 256-token context bucket, which can exceed max_pos:
    // the split-KV workspace is sized for the largest bucket (launch_attn checks the bytes it is given)
    attn_ws_bytes = KD(attn_workspace_bytes(max_seqs, H, D, (max_pos + 255) / 256 * 256));
    NVR_HIP_CHECK(hipMalloc(&attn_ws, attn_ws_bytes));
    if (env.attn_fused_merge && !f32) {
        NVR_HIP_CHECK(hipMalloc((void **)&attn_tickets, (size_t)(max_seqs * KVH) * sizeof(unsigned int)));
        NVR_HIP_CHECK(hipMemset(attn_tickets, 0, (size_t)(max_seqs * KVH) * sizeof(unsigned int)));
    }
    // step-input arena
    auto carve = [&](size_t &off, size_t bytes) { off = in_bytes; in_bytes += (bytes + 255) / 256 * 256; };
    // prefill region (the start of the arena): capacity for the largest step; a step lays its arrays out back to back for its own
    // token count (execute()), so these offsets are only the initial layout
    carve(off_ids, max_tokens * 8); carve(off_pos, max_tokens * 8); carve(off_slots, max_tokens * 4);
    carve(off_cu, (max_seqs + 1) * 4); carve(off_ctx, std::max(max_tokens, max_seqs) * 4);
    carve(off_kvbase, max_tokens * 4); carve(off_bt, 16);
    carve(off_tiles, (size_t)(max_tokens / 16 + max_seqs + 1) * sizeof(k::FlashTile));
    {
        size_t o = 0;
        auto sub = [&](size_t &f, size_t bytes) { f = o; o += (bytes + 15) / 16 * 16; };
        sub(dof_ids, max_seqs * 8); sub(dof_pos, max_seqs * 8); sub(dof_slots, max_seqs * 4); sub(dof_ctx, max_seqs * 4);
        sub(dof_skv0, max_seqs * 4); sub(dof_srows, max_seqs * 4); sub(dof_scount, 16);
        sub(dof_bt, max_seqs * max_blocks_per_seq * 4);
        dec_bytes = o; carve(off_dec, dec_bytes);
    }
    NVR_HIP_CHECK(hipHostMalloc((void **)&in_host, in_bytes, hipHostMallocDefault));
    NVR_HIP_CHECK(hipMalloc((void **)&in_dev, in_bytes));
    d_ids = (int64_t *)(in_dev + off_ids); d_pos = (int64_t *)(in_dev + off_pos); d_slots = (int32_t *)(in_dev + off_slots);
    d_cu = (int32_t *)(in_dev + off_cu); d_ctx = (int32_t *)(in_dev + off_ctx); d_kvbase = (int32_t *)(in_dev + off_kvbase);
    d_bt = (int32_t *)(in_dev + off_bt);
    dd_ids = (int64_t *)(in_dev + off_dec + dof_ids); dd_pos = (int64_t *)(in_dev + off_dec + dof_pos);
    dd_slots = (int32_t *)(in_dev + off_dec + dof_slots); dd_ctx = (int32_t *)(in_dev + off_dec + dof_ctx);
    dd_bt = (int32_t *)(in_dev + off_dec + dof_bt);

    RC(dmalloc(&d_tok, max_seqs)); RC(dmalloc(&d_maxval, max_seqs));
    NVR_HIP_CHECK(hipHostMalloc((void **)&h_tok, max_seqs * 8, hipHostMallocDefault));
    if (hipHostGetDevicePointer((void **)&h_tok_dev, h_tok, 0) != hipSuccess) { h_tok_dev = nullptr; (void)hipGetLastError(); }
    if (cfg.async_decode && h_tok_dev) {                                 // launch-ahead: one token buffer and one input twin per step in flight
        for (int i = 0; i < 2; ++i) {
            NVR_HIP_CHECK(hipHostMalloc((void **)&ahead_tok[i], (max_seqs + 1) * 8, hipHostMallocDefault));   // + the collectives' error word
            NVR_HIP_CHECK(hipHostGetDevicePointer((void **)&ahead_tok_dev[i], ahead_tok[i], 0));
            NVR_HIP_CHECK(hipHostMalloc((void **)&ahead_host[i], dec_bytes, hipHostMallocDefault));
            RC(dmalloc(&lm_snap[i], em * max_seqs * Hd));                   // the LM head's input rows of the step with this parity (present_step)
        }
    }
    RC(dmalloc(&d_temp, max_seqs)); RC(dmalloc(&d_topk, max_seqs)); RC(dmalloc(&d_topp, max_seqs)); RC(dmalloc(&d_keys, max_seqs));
    NVR_HIP_CHECK(hipHostMalloc((void **)&samp_host, max_seqs * 24, hipHostMallocDefault));
    NVR_HIP_CHECK(hipMalloc(&sample_ws, KD(sample_workspace_bytes(max_seqs, Vl))));
    RC(dmalloc(&d_gather_val, tp * max_seqs)); RC(dmalloc(&d_gather_idx, tp * max_seqs));   // (max, argmax) pairs of every rank
    if (tp > 1) { RC(dmalloc(&d_rec, max_seqs)); RC(dmalloc(&d_gather_rec, tp * max_seqs)); }

    // KV pool (create_kv_cache, :364-396): [NB, bs, KVH/tp, D] per layer per K/V, one allocation.
    const size_t block_elems = (size_t)block_size * KVH * D;
    if (cfg.num_kvcache_blocks >= 0) num_blocks = cfg.num_kvcache_blocks;
    else if (cfg.num_kvcache_blocks == -1) num_blocks = 1000;                     // :85 unwrap_or(1000)
    else {
        // size from free HBM (the reference has the config field, config.rs:30, but no sizing code)
        size_t free_b = 0, total_b = 0;
        NVR_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        double budget = (double)free_b - (1.0 - cfg.gpu_memory_utilization) * (double)total_b;
        int64_t nb = (int64_t)(budget / (double)(block_elems * 2 * em * 2 * L));
        if (nb < 1) return nvr::fail(NVR_ERR_HIP, "not enough free HBM for one KV block");
        num_blocks = nb;
    }
    kv_layer_elems = (size_t)num_blocks * block_elems;
    NVR_HIP_CHECK(hipMalloc((void **)&kv_pool, kv_layer_elems * 2 * L * sizeof(uint16_t) * em));
    NVR_HIP_CHECK(hipMemsetAsync(kv_pool, 0, kv_layer_elems * 2 * L * sizeof(uint16_t) * em, stream));   // Tensor::zeros :379-389
    NVR_HIP_CHECK(hipStreamSynchronize(stream));
    return NVR_OK;
}

// Tiled copies of the GEMM weights for the decode kernels (kernels/elementwise.hip: retile_weight): +1x the weight bytes
// (1.2 GB at Qwen3-0.6B of 288 GB), rebuilt whenever a parameter changed.  Shapes the tiled reader does not cover keep null.
int nvr_model_runner::retile_all() {
    tiled_dirty = false;
    if (!tiled_weights) return NVR_OK;
    const bool gemm_ok = Hd % 32 == 0 && (H * D) % 32 == 0 && I % 32 == 0 && QKV % 16 == 0 && Hd % 16 == 0 && I % 16 == 0 && D % 16 == 0;
    for (auto &w : layers) {
        if (!gemm_ok || !w.qkv_t) continue;
        RC(KD(retile_weight(w.qkv, w.qkv_t, QKV, Hd, (mc.qk_norm || mc.use_bias) ? 0 : 1, H, KVH, D, stream)));   // qk_norm / use_bias: plain GEMM, rows in place
        RC(KD(retile_weight(w.o, w.o_t, Hd, H * D, 0, 0, 0, 0, stream)));
        RC(KD(retile_weight(w.gate_up, w.gate_up_t, 2 * I, Hd, 0, 0, 0, 0, stream)));
        RC(KD(retile_weight(w.down, w.down_t, Hd, I, 0, 0, 0, 0, stream)));
    }
    if (lm_head_t) RC(KD(retile_weight(lm_head, lm_head_t, Vl, Hd, 0, 0, 0, 0, stream)));
    NVR_HIP_CHECK(hipStreamSynchronize(stream));
    return NVR_OK;
}

// Config.dtype = "float32": the same tensors as 4-byte values — the generator's values unrounded, as the f32 oracle holds them
int nvr_model_runner::gen_weights_f32() {
    const float sc = nvr_weight_scale_impl(mc.init_std);
    const int64_t Hg = mc.num_attention_heads, KVHg = mc.num_key_value_heads, Ig = mc.intermediate_size;
    namespace kf = nvr::kf;
    auto F = [](uint16_t *p) { return reinterpret_cast<float *>(p); };
    layers.resize(L);
    for (int64_t l = 0; l < L; ++l) {
        Layer &w = layers[l];
        std::memset(&w, 0, sizeof w);
        auto key = [&](uint64_t tid) { return nvr_weight_key_impl(mc.seed, (uint64_t)l * 8 + tid); };
        RC(dmalloc(&w.qkv, 2 * QKV * Hd)); RC(dmalloc(&w.o, 2 * Hd * H * D)); RC(dmalloc(&w.gate_up, 2 * 2 * I * Hd)); RC(dmalloc(&w.down, 2 * Hd * I));
        RC(dmalloc(&w.ln1, 2 * Hd)); RC(dmalloc(&w.ln2, 2 * Hd));
        // this rank's slices, by the reference's shard rules (gen_weights below: the same rows / columns of the same global tensors)
        RC(kf::fill_weight(F(w.qkv), H * D, Hd, Hd, Hd, rank * H * D, 0, key(TID_QKV), sc, stream));
        RC(kf::fill_weight(F(w.qkv) + H * D * Hd, KVH * D, Hd, Hd, Hd, Hg * D + rank * KVH * D, 0, key(TID_QKV), sc, stream));
        RC(kf::fill_weight(F(w.qkv) + (H + KVH) * D * Hd, KVH * D, Hd, Hd, Hd, (Hg + KVHg) * D + rank * KVH * D, 0, key(TID_QKV), sc, stream));
        RC(kf::fill_weight(F(w.o), Hd, H * D, H * D, Hg * D, 0, rank * H * D, key(TID_O), sc, stream));
        RC(kf::fill_weight(F(w.gate_up), I, Hd, Hd, Hd, rank * I, 0, key(TID_GATE_UP), sc, stream));
        RC(kf::fill_weight(F(w.gate_up) + I * Hd, I, Hd, Hd, Hd, Ig + rank * I, 0, key(TID_GATE_UP), sc, stream));
        RC(kf::fill_weight(F(w.down), Hd, I, I, Ig, 0, rank * I, key(TID_DOWN), sc, stream));
        RC(kf::fill_const(F(w.ln1), Hd, 1.0f, stream)); RC(kf::fill_const(F(w.ln2), Hd, 1.0f, stream));
        if (mc.use_bias) {
            RC(dmalloc(&w.qkv_b, 2 * QKV)); RC(dmalloc(&w.gate_up_b, 2 * 2 * I));
            RC(kf::fill_weight(F(w.qkv_b), H * D, 1, 1, 1, rank * H * D, 0, key(TID_BIAS + TID_QKV), sc, stream));
            RC(kf::fill_weight(F(w.qkv_b) + H * D, KVH * D, 1, 1, 1, Hg * D + rank * KVH * D, 0, key(TID_BIAS + TID_QKV), sc, stream));
            RC(kf::fill_weight(F(w.qkv_b) + (H + KVH) * D, KVH * D, 1, 1, 1, (Hg + KVHg) * D + rank * KVH * D, 0, key(TID_BIAS + TID_QKV), sc, stream));
            RC(kf::fill_weight(F(w.gate_up_b), I, 1, 1, 1, rank * I, 0, key(TID_BIAS + TID_GATE_UP), sc, stream));
            RC(kf::fill_weight(F(w.gate_up_b) + I, I, 1, 1, 1, Ig + rank * I, 0, key(TID_BIAS + TID_GATE_UP), sc, stream));
            if (rank == 0) {                                              // row-parallel biases live on rank 0 (linear.rs:206)
                RC(dmalloc(&w.o_b, 2 * Hd)); RC(dmalloc(&w.down_b, 2 * Hd));
                RC(kf::fill_weight(F(w.o_b), Hd, 1, 1, 1, 0, 0, key(TID_BIAS + TID_O), sc, stream));
                RC(kf::fill_weight(F(w.down_b), Hd, 1, 1, 1, 0, 0, key(TID_BIAS + TID_DOWN), sc, stream));
            }
        }
        if (mc.qk_norm) {
            RC(dmalloc(&w.q_norm, 2 * D)); RC(dmalloc(&w.k_norm, 2 * D));
            RC(kf::fill_const(F(w.q_norm), D, 1.0f, stream)); RC(kf::fill_const(F(w.k_norm), D, 1.0f, stream));
        }
    }
    RC(dmalloc(&embed, 2 * V * Hd));
    RC(kf::fill_weight(F(embed), V, Hd, Hd, Hd, 0, 0, nvr_weight_key_impl(mc.seed, TID_EMBED), sc, stream));
    if (mc.tie_word_embeddings) lm_head = embed + vocab_start * Hd * 2;             // this rank's vocabulary rows (two 16-bit words per f32 element)
    else {
        RC(dmalloc(&lm_head, 2 * Vl * Hd));
        RC(kf::fill_weight(F(lm_head), Vl, Hd, Hd, Hd, vocab_start, 0, nvr_weight_key_impl(mc.seed, TID_LM_HEAD), sc, stream));
    }
    RC(dmalloc(&norm, 2 * Hd)); RC(kf::fill_const(F(norm), Hd, 1.0f, stream));
    NVR_HIP_CHECK(hipStreamSynchronize(stream));
    tiled_dirty = false;
    return NVR_OK;
}

// RowParallelLinear's exchange on the float32 path (linear.rs:236-238) + residual + RMSNorm: every rank's f32 partial sums [T, hidden] are
// gathered (the communicator's all-gather: the one-shot arenas for decode-sized rows — bytes are moved, never summed there —, RCCL or the
// in-process rendezvous otherwise), then summed in RANK ORDER in f32, added to the residual stream and normalised by one kernel: every rank holds
// the same bits, equal to the f32 oracle's tensor-parallel sum.  One rank: the plain add + RMSNorm.
int nvr_model_runner::row_parallel_norm_f32(int64_t T, const float *wn) {
    namespace kf = nvr::kf;
    float *fh = reinterpret_cast<float *>(h), *fn = reinterpret_cast<float *>(n), *fp = reinterpret_cast<float *>(proj);
    if (!comm.active()) return kf::add_rmsnorm(fh, fp, wn, mc.rms_norm_eps, T, Hd, fn, stream);
    if (!f32_gather) return nvr::fail(NVR_ERR_INVARIANT, "float32 tensor-parallel rank without its gather buffer");
    // in pieces of at most f32_gather_rows rows (the buffer holds tp x that many rows: sized for a decode batch and a few thousand prefill rows, not for
    // tp x max_num_batched_tokens x hidden x 4 bytes — 4.3 GB per rank on Qwen3-8B at tp 8); rows are independent: same bits
    for (int64_t r0 = 0; r0 < T; r0 += f32_gather_rows) {
        const int64_t nr = std::min(f32_gather_rows, T - r0);
        RC(comm.all_gather_bytes(fp + r0 * Hd, f32_gather, (size_t)(nr * Hd) * sizeof(float), stream));
        RC(kf::sum_ranks_add_rmsnorm(fh + r0 * Hd, f32_gather, (int)tp, nr * Hd, wn, mc.rms_norm_eps, nr, Hd, fn + r0 * Hd, stream));
    }
    return NVR_OK;
}

// the f32 graph (Qwen3Model::forward, qwen3.rs:487-505; layer wiring :372-392): one launch per op, eager
int nvr_model_runner::forward_f32(int64_t T, int64_t B, bool is_prefill, int64_t max_ctx) {
    namespace kf = nvr::kf;
    hipStream_t st = stream;
    auto F = [](const uint16_t *p) { return reinterpret_cast<float *>(const_cast<uint16_t *>(p)); };
    float *fh = F(h), *fn = F(n), *fq = F(qkv), *fa = F(attn), *fp = F(proj), *fg = F(gu), *fact = F(act), *fnl = F(nlast);
    const int64_t *ids = is_prefill ? d_ids : dd_ids, *pos = is_prefill ? d_pos : dd_pos;
    const int32_t *slots = is_prefill ? d_slots : dd_slots, *ctx = is_prefill ? d_ctx : dd_ctx;
    RC(kf::embedding(ids, T, F(embed), Hd, fh, st));
    RC(kf::rmsnorm(fh, F(L > 0 ? layers[0].ln1 : norm), mc.rms_norm_eps, T, Hd, fn, st));
    // Decode-sized steps on one rank (r05): the residual add + RMSNorm in front of a consumer GEMV is done by that launch's workgroups themselves
    // (kf::add_norm_*: the bits of add_rmsnorm + consumer) — the residual stream then alternates between h and f32_h2, because the workgroups of
    // the launch still read the old one while one of them writes the new one.  `owed`: the add of `proj` + the norm of the next layer, left to its qkv launch.
    const bool ride = f32_h2 && !comm.active() && kf::fused_norm_ok(T, Hd);
    float *hc = fh, *hn = f32_h2;
    bool owed = false;
    for (int64_t l = 0; l < L; ++l) {
        const Layer &w = layers[l];
        if (owed) {
            RC(kf::add_norm_linear_qkv_rope_store(hc, fp, F(w.ln1), mc.rms_norm_eps, hn, F(w.qkv), T, Hd, H, KVH, D, w.qkv_b ? F(w.qkv_b) : nullptr, pos, slots,
                                                  cos_t, sin_t, fq, F(k_cache(l)), F(v_cache(l)), st));
            std::swap(hc, hn); owed = false;
        } else if (!w.q_norm && !w.k_norm && kf::linear_qkv_rope_ok(T, Hd, D, Hd)) {               // decode-sized: K3..K6 in one launch
            RC(kf::linear_qkv_rope_store(fn, Hd, F(w.qkv), T, Hd, H, KVH, D, w.qkv_b ? F(w.qkv_b) : nullptr, pos, slots, cos_t, sin_t, fq,
                                         F(k_cache(l)), F(v_cache(l)), st));
        } else {
            RC(kf::linear(fn, Hd, F(w.qkv), T, Hd, QKV, w.qkv_b ? F(w.qkv_b) : nullptr, fq, st));
            RC(kf::rope_store_kv(fq, pos, slots, T, H, KVH, D, cos_t, sin_t, F(k_cache(l)), F(v_cache(l)), w.q_norm ? F(w.q_norm) : nullptr,
                                 w.k_norm ? F(w.k_norm) : nullptr, mc.rms_norm_eps, st));
        }
        nvr::kt::AttnArgsF a{};
        a.q = fq; a.ldq = QKV; a.ctx_lens = ctx; a.nq = (int32_t)T; a.H = (int32_t)H; a.KVH = (int32_t)KVH; a.D = (int32_t)D; a.scale = scale;
        a.max_ctx = (int32_t)max_ctx; a.out = fa;
        const bool paged = !is_prefill || prefill_paged;                    // decode, or a prefill step that skips cached prefixes / continues a chunked prompt
        if (!paged) { a.k = fq + H * D; a.v = fq + (H + KVH) * D; a.ldkv = QKV; a.kv_base = d_kvbase; }          // flash_attention_varlen, attention.rs:177-208
        else {                                                              // ..._with_cache :211-222 / flash_attention_decode :225-235: every key through the block table
            a.k = F(k_cache(l)); a.v = F(v_cache(l)); a.block_tables = dd_bt; a.max_blocks = (int32_t)max_blocks_per_seq; a.block_size = (int32_t)block_size;
            if (is_prefill) a.seq_of_q = d_kvbase;
        }
        RC(kf::attention(a, paged, st));
        RC(kf::linear(fa, H * D, F(w.o), T, H * D, Hd, w.o_b ? F(w.o_b) : nullptr, fp, st));
        if (ride) {                                                                                  // residual :382, norm :385, K12 + K13: one launch
            RC(kf::add_norm_linear_silu_mul(hc, fp, F(w.ln2), mc.rms_norm_eps, hn, F(w.gate_up), T, Hd, I, w.gate_up_b ? F(w.gate_up_b) : nullptr, fact, st));
            std::swap(hc, hn);
        } else {
        RC(row_parallel_norm_f32(T, F(w.ln2)));                                                      // (exchange,) residual :382, norm :385
        if (kf::linear_silu_ok(T, Hd, Hd)) {                                                         // decode-sized: K12 + K13 in one launch
            RC(kf::linear_silu_mul(fn, Hd, F(w.gate_up), T, Hd, I, w.gate_up_b ? F(w.gate_up_b) : nullptr, fact, st));
        } else {
            RC(kf::linear(fn, Hd, F(w.gate_up), T, Hd, 2 * I, w.gate_up_b ? F(w.gate_up_b) : nullptr, fg, st));
            RC(kf::silu_and_mul(fg, T, I, fact, st));
        }
        }
        RC(kf::linear(fact, I, F(w.down), T, I, Hd, w.down_b ? F(w.down_b) : nullptr, fp, st));
        if (ride && l + 1 < L && !layers[l + 1].q_norm && !layers[l + 1].k_norm) owed = true;        // residual :389 + next norm :378: on the next layer's qkv launch
        else if (ride) RC(kf::add_rmsnorm(hc, fp, F(l + 1 < L ? layers[l + 1].ln1 : norm), mc.rms_norm_eps, T, Hd, fn, st));
        else RC(row_parallel_norm_f32(T, F(l + 1 < L ? layers[l + 1].ln1 : norm)));                   // (exchange,) residual :389, next norm :378 / :501
    }
    const float *hl = fn;
    if (is_prefill) { RC(kf::select_last_tokens(fn, d_cu, B, Hd, fnl, st)); hl = fnl; }
    return kf::linear(hl, Hd, F(lm_head), B, Hd, Vl, nullptr, logits, st);
}

int nvr_model_runner::gen_weights() {
    if (f32) return gen_weights_f32();
    const float sc = nvr_weight_scale_impl(mc.init_std);
    tiled_weights = env.tiled_weights;
    if (Hd % 32 || (H * D) % 32 || I % 32 || QKV % 16 || D % 16) tiled_weights = false;
    const int64_t Hg = mc.num_attention_heads, KVHg = mc.num_key_value_heads, Ig = mc.intermediate_size;
    layers.resize(L);
    for (int64_t l = 0; l < L; ++l) {
        Layer &w = layers[l];
        auto key = [&](uint64_t tid) { return nvr_weight_key_impl(mc.seed, (uint64_t)l * 8 + tid); };
        RC(dmalloc(&w.qkv, QKV * Hd)); RC(dmalloc(&w.o, Hd * H * D)); RC(dmalloc(&w.gate_up, 2 * I * Hd));
        RC(dmalloc(&w.down, Hd * I)); RC(dmalloc(&w.ln1, Hd)); RC(dmalloc(&w.ln2, Hd));
        w.qkv_t = w.o_t = w.gate_up_t = w.down_t = w.q_norm = w.k_norm = nullptr;
        w.qkv_b = w.o_b = w.gate_up_b = w.down_b = nullptr;
        if (mc.use_bias) {
            // Qwen3Config::use_bias: one value per output feature, generated like a one-column weight indexed by the GLOBAL output row (shards
            // take their slices); the row-parallel projections hold theirs on rank 0 only (linear.rs:206)
            RC(dmalloc(&w.qkv_b, QKV)); RC(dmalloc(&w.gate_up_b, 2 * I));
            RC(KD(fill_weight(w.qkv_b, H * D, 1, 1, 1, rank * H * D, 0, key(TID_BIAS + TID_QKV), sc, stream)));
            RC(KD(fill_weight(w.qkv_b + H * D, KVH * D, 1, 1, 1, Hg * D + rank * KVH * D, 0, key(TID_BIAS + TID_QKV), sc, stream)));
            RC(KD(fill_weight(w.qkv_b + (H + KVH) * D, KVH * D, 1, 1, 1, (Hg + KVHg) * D + rank * KVH * D, 0, key(TID_BIAS + TID_QKV), sc, stream)));
            RC(KD(fill_weight(w.gate_up_b, I, 1, 1, 1, rank * I, 0, key(TID_BIAS + TID_GATE_UP), sc, stream)));
            RC(KD(fill_weight(w.gate_up_b + I, I, 1, 1, 1, Ig + rank * I, 0, key(TID_BIAS + TID_GATE_UP), sc, stream)));
            if (rank == 0) {
                RC(dmalloc(&w.o_b, Hd)); RC(dmalloc(&w.down_b, Hd));
                RC(KD(fill_weight(w.o_b, Hd, 1, 1, 1, 0, 0, key(TID_BIAS + TID_O), sc, stream)));
                RC(KD(fill_weight(w.down_b, Hd, 1, 1, 1, 0, 0, key(TID_BIAS + TID_DOWN), sc, stream)));
            }
        }
        if (mc.qk_norm) {
            RC(dmalloc(&w.q_norm, D)); RC(dmalloc(&w.k_norm, D));
            RC(KD(fill_const(w.q_norm, D, 1.0f, stream))); RC(KD(fill_const(w.k_norm, D, 1.0f, stream)));
        }
        if (tiled_weights) {
            RC(dmalloc(&w.qkv_t, QKV * Hd)); RC(dmalloc(&w.o_t, Hd * H * D)); RC(dmalloc(&w.gate_up_t, 2 * I * Hd)); RC(dmalloc(&w.down_t, Hd * I));
        }
        // QKVParallelLinear, linear.rs:300-340: global rows [q heads | k heads | v heads], per-rank head slices
        RC(KD(fill_weight(w.qkv, H * D, Hd, Hd, Hd, rank * H * D, 0, key(TID_QKV), sc, stream)));
        RC(KD(fill_weight(w.qkv + H * D * Hd, KVH * D, Hd, Hd, Hd, Hg * D + rank * KVH * D, 0, key(TID_QKV), sc, stream)));
        RC(KD(fill_weight(w.qkv + (H + KVH) * D * Hd, KVH * D, Hd, Hd, Hd, (Hg + KVHg) * D + rank * KVH * D, 0, key(TID_QKV), sc, stream)));
        // RowParallelLinear o_proj, linear.rs:180-268: global [Hd, H*D], input columns sharded
        RC(KD(fill_weight(w.o, Hd, H * D, H * D, Hg * D, 0, rank * H * D, key(TID_O), sc, stream)));
        // MergedColumnParallelLinear, linear.rs:378-454: global rows [gate | up], each sharded
        RC(KD(fill_weight(w.gate_up, I, Hd, Hd, Hd, rank * I, 0, key(TID_GATE_UP), sc, stream)));
        RC(KD(fill_weight(w.gate_up + I * Hd, I, Hd, Hd, Hd, Ig + rank * I, 0, key(TID_GATE_UP), sc, stream)));
        RC(KD(fill_weight(w.down, Hd, I, I, Ig, 0, rank * I, key(TID_DOWN), sc, stream)));
        RC(KD(fill_const(w.ln1, Hd, 1.0f, stream))); RC(KD(fill_const(w.ln2, Hd, 1.0f, stream)));   // layernorm.rs:29
    }
    // embedding replicated (SURVEY §8e skips C2); LM head vocab-sharded, tied when tie_word_embeddings (qwen3.rs:461-473)
    RC(dmalloc(&embed, V * Hd));
    RC(KD(fill_weight(embed, V, Hd, Hd, Hd, 0, 0, nvr_weight_key_impl(mc.seed, TID_EMBED), sc, stream)));
    if (mc.tie_word_embeddings) lm_head = embed + vocab_start * Hd;
    else {
        RC(dmalloc(&lm_head, Vl * Hd));
        RC(KD(fill_weight(lm_head, Vl, Hd, Hd, Hd, vocab_start, 0, nvr_weight_key_impl(mc.seed, TID_LM_HEAD), sc, stream)));
    }
    RC(dmalloc(&norm, Hd)); RC(KD(fill_const(norm, Hd, 1.0f, stream)));
    if (tiled_weights && Vl % 16 == 0) RC(dmalloc(&lm_head_t, Vl * Hd));
    NVR_HIP_CHECK(hipStreamSynchronize(stream));
    return retile_all();
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight loading.  The reference's loaders only narrow and shape-check (ColumnParallelLinear::load_weight
// linear.rs:154-171: narrow(0, rank*out_per_partition); RowParallelLinear::load_weight :249-267: narrow(1,
// rank*in_per_partition); VocabParallelEmbedding embed_head.rs:142-161; Qwen3Model::load_weights qwen3.rs:518-570 names
// "embed_tokens.weight", "layers.N.input_layernorm.weight", ..., "norm.weight", "lm_head.weight") and never fill the
// packed tensors; here the same shard rules write this rank's slice of qkv_proj ([q | k | v] rows), gate_up_proj
// ([gate | up] rows), o_proj / down_proj (input columns) and the vocabulary shard of the LM head.
namespace {
inline uint16_t f32_to_f16_bits(float f) { _Float16 h = (_Float16)f; uint16_t b; std::memcpy(&b, &h, 2); return b; }
inline float f16_bits_to_f32(uint16_t b) { _Float16 h; std::memcpy(&h, &b, 2); return (float)h; }
inline float bf16_bits_to_f32(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; std::memcpy(&f, &u, 4); return f; }
inline uint16_t f32_to_bf16_bits(float f) {                                       // round to nearest even; NaN stays NaN
    uint32_t u; std::memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
bool match_layer(const char *name, const char *suffix, int64_t *layer) {          // "layers.<l>.<suffix>"
    if (std::strncmp(name, "layers.", 7) != 0) return false;
    char *end = nullptr;
    const long l = std::strtol(name + 7, &end, 10);
    if (end == name + 7 || *end != '.') return false;
    if (std::strcmp(end + 1, suffix) != 0) return false;
    *layer = l;
    return true;
}
}  // namespace

int nvr_model_runner::load_tensor(const char *name_in, int dtype, const int64_t *shape, int ndim, const void *data) {
    NVR_HIP_CHECK(hipSetDevice(device));
    if (!name_in || !shape || !data || ndim < 1 || ndim > 2) return nvr::fail(NVR_ERR_INVALID_ARG, "load_tensor: bad arguments");
    if (dtype < 0 || dtype > 2) return nvr::fail(NVR_ERR_INVALID_ARG, "load_tensor: dtype %d (0 = f16, 1 = bf16, 2 = f32)", dtype);
    tiled_dirty = true;                                                   // the tiled copies are rebuilt before the next step
    const char *name = std::strncmp(name_in, "model.", 6) == 0 ? name_in + 6 : name_in;
    const int64_t R = shape[0], Cc = ndim == 2 ? shape[1] : 1;
    const int64_t Hg = mc.num_attention_heads, KVHg = mc.num_key_value_heads, Ig = mc.intermediate_size;
    auto want = [&](int64_t r, int64_t c) -> int {
        if (R != r || Cc != c || (ndim == 1) != (c == 1 && ndim == 1))
            return nvr::fail(NVR_ERR_LEN_MISMATCH, "Partition weight shape mismatch: %s expected [%ld, %ld], got [%ld, %ld]", name_in,
                             (long)r, (long)c, (long)R, (long)Cc);
        return NVR_OK;
    };
    // rows [r0, r0+nr) x columns [c0, c0+nc) of the source -> dst (row stride dst_ld)
    auto put = [&](uint16_t *dst, int64_t dst_off, int64_t dst_ld, int64_t r0, int64_t nr, int64_t c0, int64_t nc) -> int {
        if (f32) {                                                        // the f32 path keeps the checkpoint's values as f32 (exact for every source type)
            std::vector<float> sf((size_t)(nr * nc));
            for (int64_t r = 0; r < nr; ++r)
                for (int64_t c = 0; c < nc; ++c) {
                    const size_t si = (size_t)((r0 + r) * Cc + c0 + c);
                    sf[(size_t)(r * nc + c)] = dtype == 0 ? f16_bits_to_f32(((const uint16_t *)data)[si])
                                             : dtype == 1 ? bf16_bits_to_f32(((const uint16_t *)data)[si]) : ((const float *)data)[si];
                }
            NVR_HIP_CHECK(hipStreamSynchronize(stream));
            NVR_HIP_CHECK(hipMemcpy2D(reinterpret_cast<float *>(dst) + dst_off, (size_t)dst_ld * 4, sf.data(), (size_t)nc * 4, (size_t)nc * 4, (size_t)nr, hipMemcpyHostToDevice));
            return NVR_OK;
        }
        dst += dst_off;
        std::vector<uint16_t> st((size_t)(nr * nc));
        for (int64_t r = 0; r < nr; ++r)
            for (int64_t c = 0; c < nc; ++c) {
                const size_t si = (size_t)((r0 + r) * Cc + c0 + c);
                uint16_t b;                                               // the checkpoint value in the runner's 16-bit type
                if (dtype == (bf16 ? 1 : 0)) b = ((const uint16_t *)data)[si];
                else {
                    const float f = dtype == 0 ? f16_bits_to_f32(((const uint16_t *)data)[si])
                                  : dtype == 1 ? bf16_bits_to_f32(((const uint16_t *)data)[si]) : ((const float *)data)[si];
                    b = bf16 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f);
                }
                st[(size_t)(r * nc + c)] = b;
            }
        NVR_HIP_CHECK(hipStreamSynchronize(stream));
        NVR_HIP_CHECK(hipMemcpy2D(dst, (size_t)dst_ld * 2, st.data(), (size_t)nc * 2, (size_t)nc * 2, (size_t)nr, hipMemcpyHostToDevice));
        return NVR_OK;
    };
    int64_t l = -1;
    if (!std::strcmp(name, "embed_tokens.weight")) { RC(want(V, Hd)); return put(embed, 0, Hd, 0, V, 0, Hd); }           // replicated (§8e)
    if (!std::strcmp(name, "norm.weight")) { if (ndim != 1) return want(-1, -1); RC(want(Hd, 1)); return put(norm, 0, 1, 0, Hd, 0, 1); }
    if (!std::strcmp(name, "lm_head.weight")) {
        RC(want(V, Hd));
        if (mc.tie_word_embeddings) return nvr::fail(NVR_ERR_UNSUPPORTED, "lm_head.weight: tie_word_embeddings is set (qwen3.rs:461-473), the head is the embedding");
        return put(lm_head, 0, Hd, vocab_start, Vl, 0, Hd);                                                              // embed_head.rs:57-59
    }
    auto layer_ok = [&]() -> int { return (l < 0 || l >= L) ? nvr::fail(NVR_ERR_INVALID_ARG, "%s: layer out of range (0..%ld)", name_in, (long)L - 1) : NVR_OK; };
    if (match_layer(name, "input_layernorm.weight", &l)) { RC(layer_ok()); RC(want(Hd, 1)); return put(layers[l].ln1, 0, 1, 0, Hd, 0, 1); }
    if (match_layer(name, "post_attention_layernorm.weight", &l)) { RC(layer_ok()); RC(want(Hd, 1)); return put(layers[l].ln2, 0, 1, 0, Hd, 0, 1); }
    // head_dim norms of the real checkpoints (replicated on every rank): part of the graph only with mc.qk_norm (A-27)
    if (mc.qk_norm && match_layer(name, "self_attn.q_norm.weight", &l)) { RC(layer_ok()); RC(want(D, 1)); return put(layers[l].q_norm, 0, 1, 0, D, 0, 1); }
    if (mc.qk_norm && match_layer(name, "self_attn.k_norm.weight", &l)) { RC(layer_ok()); RC(want(D, 1)); return put(layers[l].k_norm, 0, 1, 0, D, 0, 1); }
    // QKVParallelLinear: local rows [q heads | k heads | v heads] (linear.rs:300-340), each a rank slice of its projection
    if (match_layer(name, "self_attn.q_proj.weight", &l)) { RC(layer_ok()); RC(want(Hg * D, Hd)); return put(layers[l].qkv, 0, Hd, rank * H * D, H * D, 0, Hd); }
    if (match_layer(name, "self_attn.k_proj.weight", &l)) { RC(layer_ok()); RC(want(KVHg * D, Hd)); return put(layers[l].qkv, H * D * Hd, Hd, rank * KVH * D, KVH * D, 0, Hd); }
    if (match_layer(name, "self_attn.v_proj.weight", &l)) { RC(layer_ok()); RC(want(KVHg * D, Hd)); return put(layers[l].qkv, (H + KVH) * D * Hd, Hd, rank * KVH * D, KVH * D, 0, Hd); }
    if (match_layer(name, "self_attn.qkv_proj.weight", &l)) {                                                          // packed, global [q | k | v]
        RC(layer_ok()); RC(want((Hg + 2 * KVHg) * D, Hd));
        RC(put(layers[l].qkv, 0, Hd, rank * H * D, H * D, 0, Hd));
        RC(put(layers[l].qkv, H * D * Hd, Hd, Hg * D + rank * KVH * D, KVH * D, 0, Hd));
        return put(layers[l].qkv, (H + KVH) * D * Hd, Hd, (Hg + KVHg) * D + rank * KVH * D, KVH * D, 0, Hd);
    }
    if (match_layer(name, "self_attn.o_proj.weight", &l)) { RC(layer_ok()); RC(want(Hd, Hg * D)); return put(layers[l].o, 0, H * D, 0, Hd, rank * H * D, H * D); }   // :249-267
    if (mc.use_bias) {                                                    // Qwen3Config::use_bias: output-feature slices like the weight rows; row-parallel biases live on rank 0 (linear.rs:206)
        if (match_layer(name, "self_attn.q_proj.bias", &l)) { RC(layer_ok()); RC(want(Hg * D, 1)); return put(layers[l].qkv_b, 0, 1, rank * H * D, H * D, 0, 1); }
        if (match_layer(name, "self_attn.k_proj.bias", &l)) { RC(layer_ok()); RC(want(KVHg * D, 1)); return put(layers[l].qkv_b, H * D, 1, rank * KVH * D, KVH * D, 0, 1); }
        if (match_layer(name, "self_attn.v_proj.bias", &l)) { RC(layer_ok()); RC(want(KVHg * D, 1)); return put(layers[l].qkv_b, (H + KVH) * D, 1, rank * KVH * D, KVH * D, 0, 1); }
        if (match_layer(name, "self_attn.o_proj.bias", &l)) { RC(layer_ok()); RC(want(Hd, 1)); return rank == 0 ? put(layers[l].o_b, 0, 1, 0, Hd, 0, 1) : NVR_OK; }
        if (match_layer(name, "mlp.gate_proj.bias", &l)) { RC(layer_ok()); RC(want(Ig, 1)); return put(layers[l].gate_up_b, 0, 1, rank * I, I, 0, 1); }
        if (match_layer(name, "mlp.up_proj.bias", &l)) { RC(layer_ok()); RC(want(Ig, 1)); return put(layers[l].gate_up_b, I, 1, rank * I, I, 0, 1); }
        if (match_layer(name, "mlp.down_proj.bias", &l)) { RC(layer_ok()); RC(want(Hd, 1)); return rank == 0 ? put(layers[l].down_b, 0, 1, 0, Hd, 0, 1) : NVR_OK; }
    }
    // MergedColumnParallelLinear: local rows [gate | up] (linear.rs:378-454)
    if (match_layer(name, "mlp.gate_proj.weight", &l)) { RC(layer_ok()); RC(want(Ig, Hd)); return put(layers[l].gate_up, 0, Hd, rank * I, I, 0, Hd); }
    if (match_layer(name, "mlp.up_proj.weight", &l)) { RC(layer_ok()); RC(want(Ig, Hd)); return put(layers[l].gate_up, I * Hd, Hd, rank * I, I, 0, Hd); }
    if (match_layer(name, "mlp.gate_up_proj.weight", &l)) {
        RC(layer_ok()); RC(want(2 * Ig, Hd));
        RC(put(layers[l].gate_up, 0, Hd, rank * I, I, 0, Hd));
        return put(layers[l].gate_up, I * Hd, Hd, Ig + rank * I, I, 0, Hd);
    }
    if (match_layer(name, "mlp.down_proj.weight", &l)) { RC(layer_ok()); RC(want(Hd, Ig)); return put(layers[l].down, 0, I, 0, Hd, rank * I, I); }
    return nvr::fail(NVR_ERR_UNSUPPORTED, "load_tensor: no parameter named '%s' in this graph (biases need nvr_model_config.use_bias, q_norm / k_norm nvr_model_config.qk_norm, SURVEY A-17)", name_in);
}

int nvr_model_runner::copy_weight(const char *ln, uint16_t *out, size_t cap, int64_t *rows, int64_t *cols) {
    NVR_HIP_CHECK(hipSetDevice(device));
    const uint16_t *src = nullptr; int64_t r = 0, c = 0, l = -1;
    if (!std::strcmp(ln, "embed")) { src = embed; r = V; c = Hd; }
    else if (!std::strcmp(ln, "lm_head")) { src = lm_head; r = Vl; c = Hd; }
    else if (!std::strcmp(ln, "norm")) { src = norm; r = Hd; c = 1; }
    else if (match_layer(ln, "qkv", &l) && l >= 0 && l < L) { src = layers[l].qkv; r = QKV; c = Hd; }
    else if (match_layer(ln, "o", &l) && l >= 0 && l < L) { src = layers[l].o; r = Hd; c = H * D; }
    else if (match_layer(ln, "gate_up", &l) && l >= 0 && l < L) { src = layers[l].gate_up; r = 2 * I; c = Hd; }
    else if (match_layer(ln, "down", &l) && l >= 0 && l < L) { src = layers[l].down; r = Hd; c = I; }
    else if (match_layer(ln, "ln1", &l) && l >= 0 && l < L) { src = layers[l].ln1; r = Hd; c = 1; }
    else if (match_layer(ln, "ln2", &l) && l >= 0 && l < L) { src = layers[l].ln2; r = Hd; c = 1; }
    else if (mc.qk_norm && match_layer(ln, "q_norm", &l) && l >= 0 && l < L) { src = layers[l].q_norm; r = D; c = 1; }
    else if (mc.qk_norm && match_layer(ln, "k_norm", &l) && l >= 0 && l < L) { src = layers[l].k_norm; r = D; c = 1; }
    else if (mc.use_bias && match_layer(ln, "qkv_b", &l) && l >= 0 && l < L) { src = layers[l].qkv_b; r = QKV; c = 1; }
    else if (mc.use_bias && match_layer(ln, "gate_up_b", &l) && l >= 0 && l < L) { src = layers[l].gate_up_b; r = 2 * I; c = 1; }
    else if (mc.use_bias && rank == 0 && match_layer(ln, "o_b", &l) && l >= 0 && l < L) { src = layers[l].o_b; r = Hd; c = 1; }
    else if (mc.use_bias && rank == 0 && match_layer(ln, "down_b", &l) && l >= 0 && l < L) { src = layers[l].down_b; r = Hd; c = 1; }
    else return nvr::fail(NVR_ERR_INVALID_ARG, "copy_weight: unknown local tensor '%s'", ln);
    if (rows) *rows = r;
    if (cols) *cols = c;
    if (!out) return NVR_OK;
    if ((size_t)(r * c * em) > cap) return nvr::fail(NVR_ERR_LEN_MISMATCH, "copy_weight: %s has %ld elements (%ld 16-bit words), buffer holds %zu words", ln, (long)(r * c), (long)(r * c * em), cap);
    NVR_HIP_CHECK(hipStreamSynchronize(stream));
    NVR_HIP_CHECK(hipMemcpy(out, src, (size_t)(r * c) * 2 * em, hipMemcpyDeviceToHost));    // (float32 runner: f32 values, two words per element)
    return NVR_OK;
}

// RowParallelLinear::forward (o_proj / down_proj, linear.rs:228-239) + the residual add and the next RMSNorm
// (qwen3.rs:382-389): decode-sized
// steps on one GPU split k over S workgroups per output tile so that the N = hidden GEMMs reach all 256 CUs; the f32 partial
// slabs are summed, added to the residual and normalised by the following add_rmsnorm_slabs launch.  Otherwise the plain kernel
// writes fp16 `proj` (+ all-reduce when tensor parallel) and add_rmsnorm follows.
int nvr_model_runner::row_parallel_norm(const uint16_t *x, int64_t K, const uint16_t *W, const uint16_t *Wt, int64_t T, const uint16_t *wn,
                                        const uint16_t *bias) {
    if (mc.use_bias) {
        // Qwen3Config::use_bias (A-30): candle's Linear is matmul, then broadcast_add — two roundings —, and the row-parallel bias is added on
        // rank 0 BEFORE the all-reduce (linear.rs:206, :228-239): the GEMM keeps its plain epilogue and the bias is a launch of its own
        RC(KD(linear(x, K, W, T, K, Hd, proj, false, stream, Wt)));
        if (bias) RC(KD(add_bias(proj, bias, T, Hd, stream)));
        if (comm.active() && T <= 64 && comm.p2p_usable((size_t)(T * Hd)))
            return comm.all_reduce_add_rmsnorm(proj, h, wn, mc.rms_norm_eps, (int)T, (int)Hd, n, stream);
        if (comm.active()) RC(comm.all_reduce_sum_f16(proj, (size_t)(T * Hd), stream));
        return KD(add_rmsnorm(h, proj, wn, mc.rms_norm_eps, T, Hd, n, stream));
    }
    int64_t S = 1;
    if (!comm.active() && tp == 1 && T <= 64 && T <= KD(stream_row_limit()) && Hd <= 2048) {
        S = KD(decode_splitk_slices(T, K, Hd));
    } else if (!comm.active() && tp == 1 && T > 32 && T <= 64 && Hd <= 8192 && KD(splitk_prefers_tiles(T, K, Hd)) && 8 * T <= 4 * slab_rows &&
               KD(gemm_tiled_splitk_ok(T, K, Hd, 4, K))) {
        // 33..64 rows over large weights (Qwen3-8B bs 64): one 64-token tile per 128 weight rows, k split until ~256 workgroups stream the
        // weights ONCE (the streaming split-k kernel below walks them once per 32-row block: o + down 92 -> 27 us per layer)
        S = (((Hd + 127) / 128) * 8 <= 320 && KD(gemm_tiled_splitk_ok(T, K, Hd, 8, K))) ? 8 : 4;
    } else if (!comm.active() && tp == 1 && T <= 32 && Hd <= 8192 && Hd % 64 == 0 && Hd * K * 2 >= (24ll << 20) && K % 128 == 0) {
        S = 4;                                       // large weights: 64-column workgroups x 4 k-slices (linear_splitk)
    } else if (!comm.active() && tp == 1 && T > KD(stream_row_limit()) && T <= slab_rows && Hd <= 8192 && ((Hd + 127) / 128) * ((T + 127) / 128) <= 64 &&
               KD(gemm_tiled_splitk_ok(T, K, Hd, 4, K))) {
        S = 4;                                       // 97..1024 rows, few 128x128 tiles: k-split of the tiled kernel (gemm_tiled_splitk;
                                                     // bs = 256 / 512 decode 4.04 -> 3.67 / 6.18 -> 5.33 ms in r01; with the 32- / 64-token
                                                     // tiles of r02 it also wins from 97 rows on: same boundary as prefer_stream, linear.hip)
    }
    if (S > 1) {
        RC(KD(linear_splitk(x, K, W, T, K, Hd, S, slabs, stream, Wt)));
        return KD(add_rmsnorm_slabs(h, slabs, S, wn, mc.rms_norm_eps, T, Hd, n, stream));
    }
    if (!comm.active() && tp == 1 && KD(gemm256_preferred(T, K, Hd, K))) {
        // prefill-sized steps on one rank: the residual add rides in the 256x256 GEMM's epilogue (h <- fp16(h + fp16(x W^T)), the same
        // rounding points), and the norm reads one tensor instead of h and the projection (r02: 41.6 -> ~21 us per norm at 32 x 1024)
        RC(KD(gemm256_resid(x, K, W, T, K, Hd, h, stream)));
        return KD(rmsnorm(h, wn, mc.rms_norm_eps, T, Hd, n, stream));
    }
    // (the step's GEMM kernel by linear.hip's routing: 256-row tiles when its cost rule prefers them, else the 128-row kernel; the chunks call THAT kernel)
    const bool big = KD(gemm256_preferred(T, K, Hd, K));
    if (comm.active() && comm_stream && tp_overlap && ((big && T >= 1024) || (!big && T >= 512 && T * Hd > 384 * 1024 && KD(gemm_tiled_ok(T, K, Hd, K))))) {   // (T * N <= 384 Ki: linear() streams the weights instead)
        // Prefill on tensor-parallel ranks (row g): the rows are cut into chunks (multiples of the 256-row GEMM tile, so every tile and every
        // row is computed exactly as in one piece: bit-identical), and the all-reduce of chunk i runs on the communication stream while the
        // compute stream works on the GEMM of chunk i+1; the residual add + RMSNorm of a chunk follows its reduce.  Host order
        // G0 G1 R0 N0 G2 R1 N1 ...: a backend that blocks the host inside the reduce (in-process ranks) still finds the next GEMM queued.
        // configs[3] at tp 8: 72 all-reduces of 268 MB per 32 768-token step are the larger half of the step: serial 160 ms,
        // overlapped ~max(comm, compute) + one chunk.
        const int64_t gran = big ? 256 : 128;                         // rows of a GEMM tile: chunks are whole tiles
        int64_t C = std::min<int64_t>(4, T / (2 * gran));
        const int64_t rows = ((T + C - 1) / C + gran - 1) / gran * gran;
        C = (T + rows - 1) / rows;
        if (C > 1 && T - (C - 1) * rows < gran) --C;                  // (the last chunk takes a short remainder along: every chunk is a launch of the same kernel)
        tp_overlap_chunks = C;
        auto reduce_and_norm = [&](int64_t i) -> int {
            const int64_t r0 = i * rows, nr = i + 1 == C ? T - r0 : rows;
            NVR_HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_gemm[i], 0));
            RC(comm.all_reduce_sum_f16(proj + r0 * Hd, (size_t)(nr * Hd), comm_stream));
            NVR_HIP_CHECK(hipEventRecord(ev_reduced[i], comm_stream));
            NVR_HIP_CHECK(hipStreamWaitEvent(stream, ev_reduced[i], 0));
            return KD(add_rmsnorm(h + r0 * Hd, proj + r0 * Hd, wn, mc.rms_norm_eps, nr, Hd, n + r0 * Hd, stream));
        };
        for (int64_t i = 0; i < C; ++i) {
            const int64_t r0 = i * rows, nr = i + 1 == C ? T - r0 : rows;
            if (big) RC(KD(gemm256(x + r0 * K, K, W, nr, K, Hd, proj + r0 * Hd, stream)));   // (the kernel linear() takes for the whole step: same tiles, same bits)
            else RC(KD(gemm_tiled(x + r0 * K, K, W, nr, K, Hd, proj + r0 * Hd, stream)));
            NVR_HIP_CHECK(hipEventRecord(ev_gemm[i], stream));
            if (i >= 1) RC(reduce_and_norm(i - 1));
        }
        return reduce_and_norm(C - 1);
    }
    RC(KD(linear(x, K, W, T, K, Hd, proj, false, stream, Wt)));
    // linear.rs:236-238 (all-reduce) + qwen3.rs:382-389 (residual, norm): one launch over the peer-mapped arenas when the
    // message fits a slot (decode-sized steps), else the communicator's all-reduce followed by add+RMSNorm
    // (T <= 64: the fused kernel reduces a row's squares exactly like the decode-sized add+RMSNorm kernel, so fused and unfused
    // steps agree bit for bit; larger steps take the plain one-shot all-reduce and the row kernels of their size)
    if (comm.active() && T <= 64 && comm.p2p_usable((size_t)(T * Hd)))
        return comm.all_reduce_add_rmsnorm(proj, h, wn, mc.rms_norm_eps, (int)T, (int)Hd, n, stream);
    if (comm.active()) RC(comm.all_reduce_sum_f16(proj, (size_t)(T * Hd), stream));
    return KD(add_rmsnorm(h, proj, wn, mc.rms_norm_eps, T, Hd, n, stream));
}

// Decode batches in which many sequences start with the same cache blocks (prefix-cache hits of BlockManager::allocate,
// block_manager.rs:181-197: one system prompt in front of the requests, BASELINE configs[4]).  The GROUP is the set of sequences
// whose first block is the batch's most common first block; its shared length is the run of full blocks all members have in common.  The
// attention launch then sends those keys through one MFMA pass for the whole group (kernels/flash_prefill.hip, SHARED) instead of
// once per sequence; sequences outside the group (kv0 = 0) are attended to in full by the row kernel.  Groups under
// nvr_config.shared_prefix_min_seqs sequences (default 32) and block sizes the kernel does not take keep the plain kernel.
int64_t nvr_model_runner::shared_prefix_plan(nvr_seq *const *seqs, size_t nseq, int32_t *kv0, int32_t *rows, int32_t *count,
                                             int64_t *members) const {
    *members = 0;
    if (f32) return 0;
    const int64_t min_seqs = cfg.shared_prefix_min_seqs == 0 ? 32 : cfg.shared_prefix_min_seqs;
    if (min_seqs < 0 || (int64_t)nseq < min_seqs || nseq < 2) return 0;
    if (block_size < 64 || (block_size & (block_size - 1)) || !KD(flash_prefill_ok((int)D, (int)H, (int)KVH))) return 0;
    auto full_blocks = [&](const nvr_seq &s) {                            // full blocks below the token of this step
        return std::min<size_t>(s.block_table.size(), (size_t)(((int64_t)s.len() - 1) / block_size));
    };
    // the most common first block among the sequences that have a full one (a small open-addressing count table)
    int32_t cand = -1;
    {
        size_t cap = 16;
        while (cap < 2 * nseq) cap <<= 1;
        plan_keys.assign(cap, -1); plan_cnt.assign(cap, 0);
        int32_t best = 0;
        for (size_t b = 0; b < nseq; ++b) {
            if (full_blocks(*seqs[b]) == 0) continue;
            const int32_t f = seqs[b]->block_table[0];
            size_t i = ((uint32_t)f * 2654435761u) & (cap - 1);
            while (plan_keys[i] != -1 && plan_keys[i] != f) i = (i + 1) & (cap - 1);
            plan_keys[i] = f;
            if (++plan_cnt[i] > best) { best = plan_cnt[i]; cand = f; }
        }
    }
    if (cand < 0) return 0;
    size_t common = SIZE_MAX; int64_t n = 0; const nvr_seq *first = nullptr;
    for (size_t b = 0; b < nseq; ++b) {
        const nvr_seq &s = *seqs[b];
        const size_t fb = full_blocks(s);
        if (fb == 0 || s.block_table[0] != cand) continue;
        if (!first) { first = &s; common = fb; }
        else {
            size_t j = 0;
            const size_t lim = std::min(common, fb);
            while (j < lim && s.block_table[j] == first->block_table[j]) ++j;
            common = j;
        }
        ++n;
    }
    if (n < min_seqs || n < 2 || common == 0 || common == SIZE_MAX) return 0;
    const int64_t S = (int64_t)common * block_size;
    int64_t m = 0;
    for (size_t b = 0; b < nseq; ++b) {
        const nvr_seq &s = *seqs[b];
        const bool in = full_blocks(s) >= common && s.block_table[0] == cand;
        kv0[b] = in ? (int32_t)S : 0;
        if (in) rows[m++] = (int32_t)b;
    }
    for (size_t b = (size_t)m; b < nseq; ++b) rows[b] = rows[0];
    *count = (int32_t)m; *members = m;
    return S;
}

// a captured decode step is a function of (batch size, context bucket, shared-prefix length, whole batch or a group of it, logits wanted)
static inline uint64_t graph_key(bool want_logits, size_t nseq, int64_t bucket, int64_t shared_len, bool group, bool ragged) {
    return ((uint64_t)want_logits << 63) | ((uint64_t)group << 62) | ((uint64_t)ragged << 61) | ((uint64_t)nseq << 44) | ((uint64_t)(shared_len / 64) << 24) | (uint64_t)(bucket / 256);
}
// A decode batch whose contexts are ragged: the attention launch sizes every (sequence, kv head) pair's partitions by the LONGEST context of the step (its bucket, under
// a captured graph): the short sequences' workgroups idle (or finish early) while the long ones stream — the work-balanced launch (attn_share_kernel) cuts the keys
// evenly instead.  Ragged = the contexts sum to less than 1 / 1.3 of batch x bound, and there are two 64-key units of work for every CU.  (float32 runners,
// shared-prefix steps and batches of > 1024 pairs — 4+ workgroups per CU, which the dispatcher balances: 200 ragged sequences 3.96 per-pair against 4.12 — keep their launches.)
bool nvr_model_runner::ragged_batch(size_t nseq, int64_t sum_ctx, int64_t max_ctx) {
    if (f32 || nseq == 0 || decode_shared_len > 0) return false;
    const int64_t pairs = (int64_t)nseq * KVH, units = sum_ctx * KVH / 64, cus = num_cus;
    // shares: two per CU (three from 2 pairs per CU on), but none of less than four units (a tensor-parallel rank with one kv head: 32 sequences of 256..8192
    // keys are 1176 units — 512 shares of 2.3 units lost 2-5 % to the per-pair launch)
    decode_shares = (int32_t)std::max<int64_t>(64, std::min<int64_t>(cus * (pairs <= 2 * cus ? 2 : 3), units / 4));
    // ... and not below six units per CU in all: the step is then launch-bound and the per-pair launch's shorter ramp wins (12 sequences of 20..1500 keys: 1.08
    // against 1.24 ms per step; 8 of 64..8192 keys — 2040 units — 1.65 -> 1.33 the other way)
    return pairs <= 1024 && max_ctx * (int64_t)nseq * 100 >= sum_ctx * 130 && units >= 6 * cus;
}

// Prefill on tensor-parallel ranks as TWO micro-batches of whole sequences (row g, nvr_runner_set_tp_prefill_overlap(r, 2)): rows [0, mb_rows)
// and [mb_rows, T) go through every layer one behind the other on the compute stream, and each row-parallel exchange (o_proj, down_proj:
// linear.rs:236-238) runs on the communication stream under the OTHER micro-batch's compute segment:
//     compute:  attn(A) attn(B) | mlp(A)      mlp(B)      | post(A) attn'(A) post(B) attn'(B) | ...
//     comm:            AR(oA)    AR(oB)  AR(dA)      AR(dB) ...
// (attn = [input norm,] qkv + RoPE + store, flash attention, o_proj; mlp = residual + post-attention norm, gate_up + SiLU, down_proj; post =
// residual + the next layer's input norm).  With chunks of one GEMM (mode 1) an exchange hides only behind that GEMM; here it hides behind a
// third of a layer, which is what a TP-8 prefill of configs[3] needs (its exchanges are the larger half of the step).
// Every op is a row-wise kernel, a 256-row-tile GEMM (each part is big enough to take the same kernel as the whole step) or attention
// over whole sequences, so the bits are those of the serial form.  Host order keeps one segment queued ahead of every reduce (a backend
// that blocks the host inside the reduce still finds work queued).
int nvr_model_runner::forward_prefill_two(int64_t T, int64_t B) {
    hipStream_t st = stream;
    const int64_t r0s[2] = {0, mb_rows}, nrs[2] = {mb_rows, T - mb_rows};
    const int64_t t0s[2] = {0, mb_tiles}, nts[2] = {mb_tiles, n_tiles - mb_tiles};
    RC(KD(embedding(d_ids, T, embed, Hd, h, st)));
    auto attn_part = [&](int64_t l, int m) -> int {
        const Layer &w = layers[l];
        const int64_t r0 = r0s[m], nr = nrs[m];
        if (l == 0) RC(KD(rmsnorm(h + r0 * Hd, w.ln1, mc.rms_norm_eps, nr, Hd, n + r0 * Hd, st)));
        if (mb_route[0] == 0) RC(KD(gemm256_qkv_rope_store(n + r0 * Hd, Hd, w.qkv, nr, Hd, H, KVH, D, d_pos + r0, d_slots + r0, cos_t, sin_t, qkv + r0 * QKV,
                                                           k_cache(l), v_cache(l), st, prefill_paged || prefill_kv_cache)));
        else RC(KD(gemm_tiled_qkv_rope_store(n + r0 * Hd, Hd, w.qkv, nr, Hd, H, KVH, D, d_pos + r0, d_slots + r0, cos_t, sin_t, qkv + r0 * QKV, k_cache(l), v_cache(l), st)));
        k::FlashArgs f{};
        f.q = qkv; f.ldq = QKV;
        if (prefill_paged) {
            f.k = k_cache(l); f.v = v_cache(l); f.block_tables = dd_bt; f.max_blocks = (int32_t)max_blocks_per_seq; f.block_size = (int32_t)block_size;
        } else if (prefill_kv_cache) { f.k = k_cache(l); f.v = v_cache(l); f.ldkv = KVH * D; }
        else { f.k = qkv + H * D; f.v = qkv + (H + KVH) * D; f.ldkv = QKV; }
        f.tiles = (const k::FlashTile *)(in_dev + off_tiles) + t0s[m]; f.ntiles = (int32_t)nts[m];
        f.H = (int32_t)H; f.KVH = (int32_t)KVH; f.D = (int32_t)D; f.scale = scale; f.out = attn;
        RC(KD(flash_prefill(f, prefill_paged, st)));
        return mb_route[1] == 0 ? KD(gemm256(attn + r0 * H * D, H * D, w.o, nr, H * D, Hd, proj + r0 * Hd, st))
                                : KD(gemm_tiled(attn + r0 * H * D, H * D, w.o, nr, H * D, Hd, proj + r0 * Hd, st));
    };
    auto mlp_part = [&](int64_t l, int m) -> int {
        const Layer &w = layers[l];
        const int64_t r0 = r0s[m], nr = nrs[m];
        RC(KD(add_rmsnorm(h + r0 * Hd, proj + r0 * Hd, w.ln2, mc.rms_norm_eps, nr, Hd, n + r0 * Hd, st)));
        if (mb_route[2] == 0) RC(KD(gemm256_silu_mul(n + r0 * Hd, Hd, w.gate_up, nr, Hd, I, act + r0 * I, st)));
        else RC(KD(gemm_tiled_silu_mul(n + r0 * Hd, Hd, w.gate_up, nr, Hd, I, act + r0 * I, st)));
        return mb_route[3] == 0 ? KD(gemm256(act + r0 * I, I, w.down, nr, I, Hd, proj + r0 * Hd, st))
                                : KD(gemm_tiled(act + r0 * I, I, w.down, nr, I, Hd, proj + r0 * Hd, st));
    };
    auto post_part = [&](int64_t l, int m) -> int {
        const int64_t r0 = r0s[m], nr = nrs[m];
        return KD(add_rmsnorm(h + r0 * Hd, proj + r0 * Hd, l + 1 < L ? layers[l + 1].ln1 : norm, mc.rms_norm_eps, nr, Hd, n + r0 * Hd, st));
    };
    // exchange of micro-batch m's projection rows on the communication stream, bracketed by events (slot e = 2 * m + which)
    auto exchange = [&](int m, int e) -> int {
        NVR_HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_gemm[e], 0));
        RC(comm.all_reduce_sum_f16(proj + r0s[m] * Hd, (size_t)(nrs[m] * Hd), comm_stream));
        NVR_HIP_CHECK(hipEventRecord(ev_reduced[e], comm_stream));
        return NVR_OK;
    };
    tp_overlap_chunks = 2;
    for (int64_t l = 0; l < L; ++l) {
        // (post(l-1) of a micro-batch runs in front of its attn(l): the order below is per micro-batch A, B inside every stage)
        for (int m = 0; m < 2; ++m) {
            if (l > 0) { NVR_HIP_CHECK(hipStreamWaitEvent(st, ev_reduced[2 * m + 1], 0)); RC(post_part(l - 1, m)); }
            RC(attn_part(l, m));
            NVR_HIP_CHECK(hipEventRecord(ev_gemm[2 * m], st));
            if (m == 1) { RC(exchange(0, 0)); }                       // AR(o, A) goes out once attn(B) is queued behind it
        }
        RC(exchange(1, 2));
        for (int m = 0; m < 2; ++m) {
            NVR_HIP_CHECK(hipStreamWaitEvent(st, ev_reduced[2 * m], 0));
            RC(mlp_part(l, m));
            NVR_HIP_CHECK(hipEventRecord(ev_gemm[2 * m + 1], st));
            if (m == 1) { RC(exchange(0, 1)); }
        }
        RC(exchange(1, 3));
    }
    for (int m = 0; m < 2; ++m) { NVR_HIP_CHECK(hipStreamWaitEvent(st, ev_reduced[2 * m + 1], 0)); RC(post_part(L - 1, m)); }
    RC(KD(select_last_tokens(n, d_cu, B, Hd, nlast, st)));                                       // embed_head.rs:272-289
    if (lm_parts > 0) {
        int32_t np = 0;
        RC(KD(lm_head(nlast, Hd, lm_head, B, Hd, Vl, logits, d_lm_pval, d_lm_pidx, &np, st, want_logits, (tiled_weights && B <= 32) ? lm_head_t : nullptr)));
        if (np != lm_parts) return nvr::fail(NVR_ERR_INVARIANT, "lm_head produced %d partials, planned %d", np, lm_parts);
    } else {
        RC(KD(linear(nlast, Hd, lm_head, B, Hd, Vl, logits, true, st)));
    }
    return NVR_OK;
}

// Qwen3Model::forward, src/models/qwen3.rs:487-505; layer wiring :372-392; attention :208-240; MLP :305-314.
int nvr_model_runner::forward(int64_t T, int64_t B, bool is_prefill, int64_t max_ctx) {
    hipStream_t st = stream;
    const int64_t *ids = is_prefill ? d_ids : dd_ids, *pos = is_prefill ? d_pos : dd_pos;
    const int32_t *slots = is_prefill ? d_slots : dd_slots, *ctx = is_prefill ? d_ctx : dd_ctx;
    const int32_t *bt = dd_bt;
    if (f32) return forward_f32(T, B, is_prefill, max_ctx);
    if (is_prefill && tp_overlap == 2 && mb_rows > 0) return forward_prefill_two(T, B);
    const bool tl = tiled_weights && T <= 64;                            // decode-sized steps stream the tiled weight copies
    const bool embed_norm = L > 0 && KD(embedding_rmsnorm_ok(T, Hd));                    // decode-sized: K1 + the first norm in one launch
    if (embed_norm) RC(KD(embedding_rmsnorm(ids, T, embed, layers[0].ln1, mc.rms_norm_eps, Hd, h, n, st)));
    else RC(KD(embedding(ids, T, embed, Hd, h, st)));
    for (int64_t l = 0; l < L; ++l) {
        const Layer &w = layers[l];
        if (l == 0 && !embed_norm) RC(KD(rmsnorm(h, w.ln1, mc.rms_norm_eps, T, Hd, n, st)));   // later layers: see down_proj
        if (mc.qk_norm || mc.use_bias) {         // A-27: the head norms sit between the projection and RoPE: plain GEMM, then one
            RC(KD(linear(n, Hd, w.qkv, T, Hd, QKV, qkv, false, st, tl ? w.qkv_t : nullptr)));      // norm + RoPE + KV-store launch
            if (mc.use_bias) RC(KD(add_bias(qkv, w.qkv_b, T, QKV, st)));                            // A-30: the bias precedes the norms and RoPE (qwen3.rs:208-222)
            RC(KD(rope_store_kv(qkv, pos, slots, T, H, KVH, D, cos_t, sin_t, k_cache(l), v_cache(l), st, w.q_norm, w.k_norm, mc.rms_norm_eps)));
        } else {
            // qkv GEMM with the RoPE + KV-store epilogue (K3..K6 in one launch)
            RC(KD(linear_qkv_rope_store(n, Hd, w.qkv, T, Hd, H, KVH, D, pos, slots, cos_t, sin_t, qkv, k_cache(l), v_cache(l), st, tl ? w.qkv_t : nullptr,
                                        is_prefill && n_tiles > 0 && (prefill_paged || prefill_kv_cache))));
        }
        k::AttnArgs a{};
        a.q = qkv; a.ldq = QKV; a.ctx_lens = ctx; a.nq = (int32_t)T; a.H = (int32_t)H; a.KVH = (int32_t)KVH; a.D = (int32_t)D;
        a.scale = scale; a.max_ctx = (int32_t)max_ctx; a.out = attn;
        if (is_prefill && n_tiles > 0) {                             // flash_attention_varlen, attention.rs:177-208 (MFMA)
            k::FlashArgs f{};
            f.q = qkv; f.ldq = QKV;
            if (prefill_paged) {                                     // cached prefixes: K/V through the block tables (K8)
                f.k = k_cache(l); f.v = v_cache(l); f.block_tables = dd_bt; f.max_blocks = (int32_t)max_blocks_per_seq;
                f.block_size = (int32_t)block_size;
            } else if (prefill_kv_cache) {                          // every sequence's blocks are consecutive: its K/V rows are contiguous IN the cache
                f.k = k_cache(l); f.v = v_cache(l); f.ldkv = KVH * D;
            } else { f.k = qkv + H * D; f.v = qkv + (H + KVH) * D; f.ldkv = QKV; }
            f.tiles = (const k::FlashTile *)(in_dev + off_tiles); f.ntiles = (int32_t)n_tiles;
            f.H = (int32_t)H; f.KVH = (int32_t)KVH; f.D = (int32_t)D; f.scale = scale; f.out = attn;
            RC(KD(flash_prefill(f, prefill_paged, st)));
        } else if (is_prefill) {                                     // head shapes outside the MFMA kernel: row kernel
            a.k = qkv + H * D; a.v = qkv + (H + KVH) * D; a.ldkv = QKV; a.kv_base = d_kvbase; a.workspace = nullptr;
            RC(KD(attention(a, false, st)));
        } else {                                                     // flash_attention_decode, attention.rs:225-235
            a.k = k_cache(l); a.v = v_cache(l); a.block_tables = bt; a.max_blocks = (int32_t)max_blocks_per_seq;
            a.block_size = (int32_t)block_size; a.workspace = attn_ws; a.workspace_bytes = attn_ws_bytes; a.tickets = attn_tickets;
            a.shared_len = (int32_t)decode_shared_len;
            a.balance_hint = decode_ragged ? decode_shares : 0;
            if (decode_shared_len > 0 && decode_shared_rows < T) {     // a group inside the batch: per-row kv0, member rows, member count
                a.shared_kv0 = (const int32_t *)(in_dev + off_dec + dof_skv0); a.shared_rows = (const int32_t *)(in_dev + off_dec + dof_srows);
                a.shared_count = (const int32_t *)(in_dev + off_dec + dof_scount);
            }
            RC(KD(attention(a, true, st)));
        }
        RC(row_parallel_norm(attn, H * D, w.o, tl ? w.o_t : nullptr, T, w.ln2, w.o_b));  // o_proj, residual :382, norm :385
        if (mc.use_bias) {                       // A-30: gate_up_proj, its bias, then SiluAndMul on the biased halves (qwen3.rs:305-314)
            RC(KD(linear(n, Hd, w.gate_up, T, Hd, 2 * I, gu, false, st, tl ? w.gate_up_t : nullptr)));
            RC(KD(add_bias(gu, w.gate_up_b, T, 2 * I, st)));
            RC(KD(silu_and_mul(gu, T, I, act, st)));
            RC(row_parallel_norm(act, I, w.down, tl ? w.down_t : nullptr, T, l + 1 < L ? layers[l + 1].ln1 : norm, w.down_b));
            continue;
        }
        RC(KD(linear_silu_mul(n, Hd, w.gate_up, T, Hd, I, act, st, tl ? w.gate_up_t : nullptr)));   // K12 + K13 in one launch
        // down_proj, residual :389 and the NEXT layer's input norm :378 (or the final norm :501)
        RC(row_parallel_norm(act, I, w.down, tl ? w.down_t : nullptr, T, l + 1 < L ? layers[l + 1].ln1 : norm));
    }
    if (L == 0) RC(KD(rmsnorm(h, norm, mc.rms_norm_eps, T, Hd, n, st)));                  // final norm :501
    const uint16_t *hl = n;
    if (is_prefill) { RC(KD(select_last_tokens(n, d_cu, B, Hd, nlast, st))); hl = nlast; }      // embed_head.rs:272-289
    if (lm_parts > 0) {                                                                        // f32 logits (A-21) + arg-max partials
        int32_t np = 0;
        RC(KD(lm_head(hl, Hd, lm_head, B, Hd, Vl, logits, d_lm_pval, d_lm_pidx, &np, st, want_logits, (tiled_weights && B <= 32) ? lm_head_t : nullptr)));
        if (np != lm_parts) return nvr::fail(NVR_ERR_INVARIANT, "lm_head produced %d partials, planned %d", np, lm_parts);
    } else {
        RC(KD(linear(hl, Hd, lm_head, B, Hd, Vl, logits, true, st)));
    }
    return NVR_OK;
}

// execute_model :105-128 with prepare_*_inputs :172-210 and create_*_context :222-300
int nvr_model_runner::execute(nvr_seq *const *seqs, size_t nseq, bool is_prefill) {
    NVR_HIP_CHECK(hipSetDevice(device));
    if (tiled_dirty) RC(retile_all());
    if (tp > 1) RC(comm.prepare());
    if (tp > 1 && !comm.active() && !allow_missing_comm)
        return nvr::fail(NVR_ERR_RCCL, "execute_model: tensor_parallel_size %ld but no communicator is attached (nvr_runner_init_comm / "
                         "nvr_local_group_attach); partial sums would be returned as results", (long)tp);
    if (nseq == 0) return nvr::fail(NVR_ERR_INVALID_ARG, "execute_model: empty batch");
    facts_shown_valid = false;                                           // the accessors follow this step until the engine presents another
    if ((int64_t)nseq > max_seqs) return nvr::fail(NVR_ERR_INVALID_ARG, "execute_model: %zu sequences > max_num_seqs %ld", nseq, (long)max_seqs);
    char *hd = in_host + off_dec;
    int64_t *ids = (int64_t *)(hd + dof_ids), *pos = (int64_t *)(hd + dof_pos);
    int32_t *slots = (int32_t *)(hd + dof_slots), *ctx = (int32_t *)(hd + dof_ctx);
    int32_t *cu = nullptr, *kvb = nullptr, *bt = (int32_t *)(hd + dof_bt);
    size_t prefill_bytes = 0;                                            // bytes of the prefill region this step uploads (one copy)
    int64_t T = 0, max_ctx = 0, sum_ctx = 0;
    const int64_t bs = block_size;
    NVR_HIP_CHECK(hipStreamSynchronize(stream));   // staging arena is reused: previous uploads must have landed
    if (is_prefill) {
        // slot(pos) = table[pos/bs]*bs + pos%bs (A-6).  Tokens: the reference feeds every token from position 0
        // (model_runner.rs:176-182, A-7); unless recompute_cached_prefix is set, a sequence's cached prefix
        // (num_cached_tokens: whole blocks another live sequence already holds, block_manager.rs:181-187) is skipped —
        // its K/V rows are in the cache (written by an earlier step, or by the owner's rows of THIS step's qkv launch,
        // which precedes the attention launch of the layer) and the new tokens attend to them through the block table
        // (K8, attention.rs:211-222).  At least the last token is always computed (its logits are the step's output).
        // With enable_chunked_prefill (A-23) a sequence contributes the token range [chunk_start, chunk_start + chunk_len) the
        // scheduler gave it; earlier tokens are reached through the block table exactly like a cached prefix.
        const bool flash_ok = !f32 && KD(flash_prefill_ok((int)D, (int)H, (int)KVH));   // (the f32 path: row attention over the step's own K / V rows)
        const bool chunked = cfg.enable_chunked_prefill != 0;
        auto range_of = [&](const nvr_seq &sq, int64_t *lo, int64_t *hi) {        // rows fed through the model for this sequence
            const int64_t len = (int64_t)sq.len();
            int64_t a0 = 0, b0 = len;
            if (chunked && sq.chunk_len > 0) { a0 = (int64_t)sq.chunk_start; b0 = a0 + (int64_t)sq.chunk_len; }
            const bool paged_ok = flash_ok || f32;                                // (the 16-bit row kernel has no paged prefill form; the f32 path's has)
            int64_t c = (cfg.recompute_cached_prefix || !paged_ok) ? 0 : std::min<int64_t>((int64_t)sq.num_cached_tokens, b0 - 1);
            if (!paged_ok) a0 = 0;
            *lo = std::max<int64_t>(a0, std::max<int64_t>(c, 0)); *hi = b0;
        };
        if (chunked && !flash_ok && !f32)
            for (size_t b = 0; b < nseq; ++b)
                if (seqs[b]->chunk_start > 0) return nvr::fail(NVR_ERR_UNSUPPORTED, "chunked prefill needs the paged flash kernel (head_dim 64/128)");
        prefill_paged = false;
        mb_rows = mb_tiles = 0;
        tp_overlap_chunks = 0;
        int64_t total = 0;
        for (size_t b = 0; b < nseq; ++b) {
            int64_t lo, hi; range_of(*seqs[b], &lo, &hi);
            prefill_paged |= lo > 0;
            total += hi - lo;
        }
        if (total > max_tokens) return nvr::fail(NVR_ERR_INVALID_ARG, "prefill of %ld tokens exceeds max_num_batched_tokens %ld", (long)total, (long)max_tokens);
        // Whole prompts: the flash kernel reads K / V from the caches, never from the step's qkv buffer, so the qkv GEMM writes them once
        // (gemm256 skips the k / v columns of the qkv buffer: -9 % on that GEMM).  If every sequence's blocks are consecutive (blocks come off
        // the free list in order until it has been recycled) token p of a sequence sits at cache row table[0] * bs + p and the kernel runs in
        // its contiguous form (-1.7 ms of a 32 x 1024 prefill); otherwise it walks the block tables like a prefill behind a cached prefix
        // (-0.5 ms).  q/k-norm models keep the qkv buffer (their norm + RoPE launch writes both anyway).
        prefill_kv_cache = flash_ok && !prefill_paged && !mc.qk_norm && !mc.use_bias;
        for (size_t b = 0; b < nseq && prefill_kv_cache; ++b) {
            const nvr_seq &sq = *seqs[b];
            const size_t nb = std::min(sq.block_table.size(), (size_t)(((int64_t)sq.len() + bs - 1) / bs));
            for (size_t j = 1; j < nb; ++j)
                if (sq.block_table[j] != sq.block_table[j - 1] + 1) { prefill_kv_cache = false; break; }
        }
        if (flash_ok && !prefill_paged && !mc.qk_norm && !mc.use_bias && !prefill_kv_cache) prefill_paged = true;
        const int qb = flash_ok ? KD(flash_tile_positions((int)H, (int)KVH)) : 1;
        {   // the step's arrays back to back at the start of the arena (sized by THIS step's token count): one upload instead of seven
            size_t o = 0;
            auto sub = [&](size_t &f, size_t bytes) { f = o; o += (bytes + 63) / 64 * 64; };
            sub(off_ids, (size_t)total * 8); sub(off_pos, (size_t)total * 8); sub(off_slots, (size_t)total * 4); sub(off_ctx, (size_t)total * 4);
            sub(off_kvbase, (size_t)total * 4); sub(off_cu, (nseq + 1) * 4);
            // (tiles exist only for the flash kernel; the row-kernel path — GQA group 8, head_dim outside 64 / 128 — reserves none: with
            //  qb = 1 a tile per token would not fit what init() carved, max_tokens / 16 tiles)
            sub(off_tiles, flash_ok ? (size_t)(total / qb + (int64_t)nseq + 1) * sizeof(k::FlashTile) : 0);
            prefill_bytes = o;
            if (prefill_bytes > off_dec) return nvr::fail(NVR_ERR_INVARIANT, "prefill input region: %zu bytes needed, %zu carved", prefill_bytes, off_dec);
            d_ids = (int64_t *)(in_dev + off_ids); d_pos = (int64_t *)(in_dev + off_pos); d_slots = (int32_t *)(in_dev + off_slots);
            d_ctx = (int32_t *)(in_dev + off_ctx); d_kvbase = (int32_t *)(in_dev + off_kvbase); d_cu = (int32_t *)(in_dev + off_cu);
            ids = (int64_t *)(in_host + off_ids); pos = (int64_t *)(in_host + off_pos); slots = (int32_t *)(in_host + off_slots);
            ctx = (int32_t *)(in_host + off_ctx); kvb = (int32_t *)(in_host + off_kvbase); cu = (int32_t *)(in_host + off_cu);
        }
        cu[0] = 0;
        k::FlashTile *tl = (k::FlashTile *)(in_host + off_tiles);
        n_tiles = 0;
        for (size_t b = 0; b < nseq; ++b) {
            const nvr_seq &s = *seqs[b];
            int64_t c0, end; range_of(s, &c0, &end);
            if (!prefill_paged) c0 = 0;
            if (end > max_pos) return nvr::fail(NVR_ERR_INVALID_ARG, "sequence of %ld tokens exceeds max_model_len %ld", (long)end, (long)max_pos);
            if ((int64_t)s.block_table.size() * bs < end) return nvr::fail(NVR_ERR_NOT_ALLOCATED, "Sequence has no allocated blocks");
            {   // rows [c0, end) of this sequence, array by array (tight loops the compiler vectorises; one division per block, not per token)
                const int64_t n = end - c0, *tok = s.token_ids.data() + c0;
                uint64_t worst = 0;
                for (int64_t i = 0; i < n; ++i) { const uint64_t t = (uint64_t)tok[i]; worst = t > worst ? t : worst; ids[T + i] = (int64_t)t; }
                if (worst >= (uint64_t)V) {                                   // candle's index_select rejects these (embed_head.rs:80)
                    int64_t p = c0; while ((uint64_t)s.token_ids[p] < (uint64_t)V) ++p;
                    return nvr::fail(NVR_ERR_INVALID_ARG, "token id %ld at position %ld is outside the vocabulary [0, %ld)", (long)s.token_ids[p], (long)p, (long)V);
                }
                for (int64_t i = 0; i < n; ++i) pos[T + i] = c0 + i;
                for (int64_t i = 0; i < n; ++i) ctx[T + i] = (int32_t)(c0 + i + 1);
                const int32_t cub = cu[b];
                if (f32 && prefill_paged) { for (int64_t i = 0; i < n; ++i) kvb[T + i] = (int32_t)b; }   // f32 paged prefill: the row's block table (seq_of_q)
                else for (int64_t i = 0; i < n; ++i) kvb[T + i] = cub;
                for (int64_t p = c0; p < end;) {                              // slot(pos) = table[pos / bs] * bs + pos % bs, block by block
                    const int64_t blk = p / bs, stop = std::min(end, (blk + 1) * bs);
                    const int32_t base = (int32_t)((int64_t)s.block_table[blk] * bs - blk * bs);
                    for (int64_t q = p; q < stop; ++q) slots[T + (q - c0)] = base + (int32_t)q;
                    p = stop;
                }
                T += n;
            }
            cu[b + 1] = (int32_t)T;
            max_ctx = std::max(max_ctx, end);
            if (prefill_paged) {
                if ((int64_t)s.block_table.size() > max_blocks_per_seq) return nvr::fail(NVR_ERR_INVARIANT, "block table longer than max_model_len allows");
                std::memcpy(bt + b * max_blocks_per_seq, s.block_table.data(), s.block_table.size() * 4);
            }
            if (flash_ok) {                                              // longest-context tiles of a sequence first
                const int64_t nq = end - c0;
                for (int64_t q0 = (nq - 1) / qb * qb; q0 >= 0; q0 -= qb)
                    tl[n_tiles++] = k::FlashTile{(int32_t)(cu[b] + q0), (int32_t)std::min<int64_t>(qb, nq - q0), (int32_t)(c0 + q0),
                                                 prefill_paged ? (int32_t)b : prefill_kv_cache ? (int32_t)((int64_t)s.block_table[0] * bs) : cu[b]};
            }
        }
        if (flash_ok) {
            // Dispatch order = array order.  Sequences are taken in groups of 4 and, inside a group, the tiles with the
            // longest key ranges go first: the ~64 workgroups an XCD runs at a time (kv head g = blockIdx % KVH lands on
            // XCD g) then share the K/V of 4 sequences (~2 MiB per head: L2-resident), and the short tiles fill the tail.
            // (Longest-first over the WHOLE batch made every XCD touch all sequences at once: K/V re-streamed from HBM.)
            {
                size_t t0 = 0;
                for (size_t b0 = 0; b0 < nseq; b0 += 4) {
                    const int32_t row_end = cu[std::min(nseq, b0 + 4)];
                    size_t t1 = t0;
                    while (t1 < (size_t)n_tiles && tl[t1].q_row0 < row_end) ++t1;
                    std::stable_sort(tl + t0, tl + t1, [](const k::FlashTile &a, const k::FlashTile &b) { return a.pos0 + a.nq > b.pos0 + b.nq; });
                    t0 = t1;
                    // two micro-batches (tp_overlap = 2): cut at the boundary between two groups of four sequences that is closest to half the rows
                    if (b0 + 4 < nseq && (mb_rows == 0 || std::llabs(2 * (int64_t)row_end - total) < std::llabs(2 * mb_rows - total))) {
                        mb_rows = row_end; mb_tiles = (int64_t)t1;
                    }
                }
            }
            // (every GEMM of a part must run the kernel the WHOLE step would run — 256-row or 128-row tiles, by linear.hip's cost rule for the whole
            //  step's shape — so the parts call that kernel directly; the plain graph only: see forward_prefill_two)
            {
                const int64_t rest = total - mb_rows;
                bool ok = tp_overlap == 2 && comm.active() && comm_stream && !mc.qk_norm && !mc.use_bias && !chunked && mb_rows >= 512 && rest >= 512;
                // kinds: 0 qkv (+RoPE), 1 o_proj, 2 gate_up (+SiLU), 3 down_proj
                const int64_t Ks[4] = {Hd, H * D, Hd, I}, Ns[4] = {QKV, Hd, 2 * I, Hd};
                for (int kd = 0; kd < 4 && ok; ++kd) {
                    auto ok256 = [&](int64_t rows) { return kd == 0 ? KD(gemm256_rope_ok(rows, Hd, H, KVH, D, Hd)) : kd == 2 ? KD(gemm256_silu_ok(rows, Hd, I, Hd))
                                                                                                                             : KD(gemm256_ok(rows, Ks[kd], Ns[kd], Ks[kd])); };
                    auto ok128 = [&](int64_t rows) { return KD(gemm_tiled_ok(rows, Ks[kd], kd == 2 ? I : Ns[kd], Ks[kd])) && (kd != 0 || 128 % D == 0) && (kd != 2 || I % 64 == 0); };
                    if (ok256(total) && KD(gemm256_preferred(total, Ks[kd], Ns[kd], Ks[kd]))) { mb_route[kd] = 0; ok = ok256(mb_rows) && ok256(rest); }
                    else if (ok128(total)) { mb_route[kd] = 1; ok = ok128(mb_rows) && ok128(rest); }
                    else ok = false;
                }
                if (!ok) mb_rows = mb_tiles = 0;
            }
        }
    } else {
        for (size_t b = 0; b < nseq; ++b) {
            const nvr_seq &s = *seqs[b];
            const int64_t len = (int64_t)s.len();
            if (len > max_pos) return nvr::fail(NVR_ERR_INVALID_ARG, "sequence of %ld tokens exceeds max_model_len %ld", (long)len, (long)max_pos);
            if ((int64_t)s.block_table.size() * bs < len) return nvr::fail(NVR_ERR_NOT_ALLOCATED, "Sequence has no allocated blocks");
            if ((uint64_t)s.last_token >= (uint64_t)V)
                return nvr::fail(NVR_ERR_INVALID_ARG, "token id %ld is outside the vocabulary [0, %ld)", (long)s.last_token, (long)V);
            ids[b] = s.last_token; pos[b] = len - 1;                                      // :201-202
            slots[b] = (int32_t)((int64_t)s.block_table[(len - 1) / bs] * bs + (len - 1) % bs);
            ctx[b] = (int32_t)len;                                                        // :275
            int32_t *row = bt + b * max_blocks_per_seq;                                   // -1 padded, :283-290
            const size_t nb = s.block_table.size();
            std::memcpy(row, s.block_table.data(), nb * 4);
            for (int64_t j = (int64_t)nb; j < max_blocks_per_seq; ++j) row[j] = -1;
            max_ctx = std::max(max_ctx, len); sum_ctx += len;
        }
        T = (int64_t)nseq;
        decode_shared_len = shared_prefix_plan(seqs, nseq, (int32_t *)(hd + dof_skv0), (int32_t *)(hd + dof_srows), (int32_t *)(hd + dof_scount),
                                               &decode_shared_rows);
        decode_ragged = ragged_batch(nseq, sum_ctx, (cfg.enforce_eager || graphs_disabled) ? max_ctx : (max_ctx + 255) / 256 * 256);
    }
    // one H2D per array actually used this step (K19)
    auto up = [&](size_t off, size_t bytes) { return hipMemcpyAsync(in_dev + off, in_host + off, bytes, hipMemcpyHostToDevice, stream); };
    if (is_prefill) {
        NVR_HIP_CHECK(up(0, n_tiles ? off_tiles + (size_t)n_tiles * sizeof(k::FlashTile) : prefill_bytes));   // ids | pos | slots | ctx | kv base | cu | tiles
        if (prefill_paged) NVR_HIP_CHECK(up(off_dec + dof_bt, nseq * max_blocks_per_seq * 4));
    } else NVR_HIP_CHECK(up(off_dec, dof_bt + nseq * max_blocks_per_seq * 4));

    last_rows = nseq; last_prefill = is_prefill; last_tokens = T;
    // arg-max partials come with the logits when the whole batch goes through one lm_head launch (a pure function of
    // the shapes, so a replayed graph and this bookkeeping always agree)
    lm_parts = lm_fused ? KD(lm_head_parts((int64_t)nseq, Hd, Vl, Hd)) : 0;
    // a batch that samples greedily everywhere takes its tokens from the arg-max partials: the f32 logits are then written
    // only on demand (ensure_logits: execute_model callers that ask for them, copy_logits)
    want_logits = !lazy_logits || lm_parts == 0;
    for (size_t i = 0; i < nseq && !want_logits; ++i) want_logits = seqs[i]->sampling.temperature != 0.0f;
    logits_valid = want_logits;
    lm_input = is_prefill ? nlast : n;                                   // the rows the LM head reads this step (forward())
    if (is_prefill || cfg.enforce_eager || graphs_disabled) {
        const int rc = forward(T, (int64_t)nseq, is_prefill, max_ctx);
        if (rc && comm_stream) {
            // a tensor-parallel prefill step that failed half way (row_parallel_norm's chunks, forward_prefill_two) leaves exchanges and events
            // queued on the communication stream that the compute stream never joined: drain both, so that the next step's collectives are
            // ordered against nothing left over (the engine aborts the batch, nvr_engine_abort_last_batch)
            (void)hipStreamSynchronize(comm_stream);
            (void)hipStreamSynchronize(stream);
            (void)hipGetLastError();
        }
        if (rc) (void)rearm_tickets();                                   // (a launch that never ran its last arriver would leave a counter behind: ADVICE r05)
        return rc;
    }

    // decode: replay a hipGraph captured per (batch size, context bucket) — execute_with_cuda_graph :303-326
    const int64_t bucket = (max_ctx + 255) / 256 * 256;
    const uint64_t key = graph_key(want_logits, nseq, bucket, decode_shared_len, decode_shared_len > 0 && decode_shared_rows < (int64_t)nseq, decode_ragged);
    auto it = graphs.find(key);
    if (it == graphs.end()) {
        if (graphs.size() >= (size_t)env.max_graphs) {   // a long-lived engine sees many (batch size, bucket) pairs: bound the cache
            NVR_HIP_CHECK(hipStreamSynchronize(stream));
            for (auto &kv : graphs) hipGraphExecDestroy(kv.second);
            graphs.clear();
        }
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        NVR_HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        int rc = forward(T, T, false, bucket);
        hipError_t e = hipStreamEndCapture(stream, &g);
        if (!rc && e != hipSuccess) rc = nvr::fail(NVR_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        if (!rc && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) rc = nvr::fail(NVR_ERR_HIP, "hipGraphInstantiate failed");
        if (g) hipGraphDestroy(g);
        if (rc && comm.active()) {            // a graph holding collective nodes could not be built on this stack: run eagerly
            graphs_disabled = true;
            (void)hipGetLastError();
            return forward(T, (int64_t)nseq, false, max_ctx);
        }
        if (rc) return rc;
        it = graphs.emplace(key, ge).first;
    }
    last_decode_graph = it->second;
    NVR_HIP_CHECK(hipGraphLaunch(it->second, stream));
    return NVR_OK;
}

// ---- launch-ahead of greedy decode steps (nvr_config.async_decode) ---------------------------------------------------------
// The step whose tokens the host has not seen yet is already followed on the stream by the next one: its input ids were
// written on the device by the previous step's arg-max merge (sample_launch), positions / slots / context lengths / block
// tables follow from the sequence lengths alone.  Nothing here synchronises the stream.
int nvr_model_runner::execute_decode_ahead(nvr_seq *const *seqs, size_t nseq, int parity) {
    NVR_HIP_CHECK(hipSetDevice(device));
    if (tp > 1) RC(comm.prepare());
    if (!ahead_ok(nseq)) return nvr::fail(NVR_ERR_UNSUPPORTED, "execute_decode_ahead: runner not set up for launch-ahead");
    if (tiled_dirty) return nvr::fail(NVR_ERR_UNSUPPORTED, "execute_decode_ahead: parameters changed, the tiled copies are rebuilt by a synchronous step");
    if (nseq == 0 || (int64_t)nseq > max_seqs) return nvr::fail(NVR_ERR_INVALID_ARG, "execute_decode_ahead: %zu sequences", nseq);
    char *hd = ahead_host[parity & 1];
    int64_t *pos = (int64_t *)(hd + dof_pos);
    int32_t *slots = (int32_t *)(hd + dof_slots), *ctx = (int32_t *)(hd + dof_ctx), *bt = (int32_t *)(hd + dof_bt);
    const int64_t bs = block_size;
    int64_t max_ctx = 0, sum_ctx = 0;
    for (size_t b = 0; b < nseq; ++b) {
        const nvr_seq &s = *seqs[b];
        const int64_t len = (int64_t)s.len();                             // includes the token the host has not seen yet
        if (len > max_pos) return nvr::fail(NVR_ERR_INVALID_ARG, "sequence of %ld tokens exceeds max_model_len %ld", (long)len, (long)max_pos);
        if ((int64_t)s.block_table.size() * bs < len) return nvr::fail(NVR_ERR_NOT_ALLOCATED, "Sequence has no allocated blocks");
        pos[b] = len - 1;
        slots[b] = (int32_t)((int64_t)s.block_table[(len - 1) / bs] * bs + (len - 1) % bs);
        ctx[b] = (int32_t)len;
        int32_t *row = bt + b * max_blocks_per_seq;
        const size_t nb = s.block_table.size();
        std::memcpy(row, s.block_table.data(), nb * 4);
        for (int64_t j = (int64_t)nb; j < max_blocks_per_seq; ++j) row[j] = -1;
        max_ctx = std::max(max_ctx, len); sum_ctx += len;
    }
    decode_shared_len = shared_prefix_plan(seqs, nseq, (int32_t *)(hd + dof_skv0), (int32_t *)(hd + dof_srows), (int32_t *)(hd + dof_scount),
                                           &decode_shared_rows);
    decode_ragged = ragged_batch(nseq, sum_ctx, (cfg.enforce_eager || graphs_disabled) ? max_ctx : (max_ctx + 255) / 256 * 256);
    NVR_HIP_CHECK(hipMemcpyAsync(in_dev + off_dec + dof_pos, hd + dof_pos, dof_bt + nseq * max_blocks_per_seq * 4 - dof_pos, hipMemcpyHostToDevice, stream));
    last_rows = nseq; last_prefill = false; last_tokens = (int64_t)nseq;
    lm_parts = KD(lm_head_parts((int64_t)nseq, Hd, Vl, Hd));
    if (lm_parts <= 0) return nvr::fail(NVR_ERR_UNSUPPORTED, "execute_decode_ahead: the fused LM head does not take this batch");
    want_logits = !lazy_logits; logits_valid = want_logits; lm_input = n;
    const int64_t T = (int64_t)nseq;
    if (cfg.enforce_eager || graphs_disabled) return forward(T, T, false, max_ctx);
    const int64_t bucket = (max_ctx + 255) / 256 * 256;
    const uint64_t key = graph_key(want_logits, nseq, bucket, decode_shared_len, decode_shared_len > 0 && decode_shared_rows < (int64_t)nseq, decode_ragged);
    auto it = graphs.find(key);
    if (it == graphs.end()) {
        // flushing the cache needs an idle stream, and the current step is still running: decline — the engine rolls the
        // speculative schedule back and the synchronous path of the next call flushes and captures (engine.cpp)
        if (graphs.size() >= (size_t)env.max_graphs) return nvr::fail(NVR_ERR_UNSUPPORTED, "execute_decode_ahead: graph cache full");
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        NVR_HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        int rc = forward(T, T, false, bucket);
        hipError_t e = hipStreamEndCapture(stream, &g);
        if (!rc && e != hipSuccess) rc = nvr::fail(NVR_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        if (!rc && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) rc = nvr::fail(NVR_ERR_HIP, "hipGraphInstantiate failed");
        if (g) hipGraphDestroy(g);
        if (rc) return rc;
        it = graphs.emplace(key, ge).first;
    }
    last_decode_graph = it->second;
    NVR_HIP_CHECK(hipGraphLaunch(it->second, stream));
    return NVR_OK;
}

int nvr_model_runner::sample_launch(nvr_seq *const *seqs, size_t nseq, int parity) {
    NVR_HIP_CHECK(hipSetDevice(device));
    if (!ahead_capable() || lm_parts <= 0) return nvr::fail(NVR_ERR_UNSUPPORTED, "sample_launch: runner not set up for launch-ahead");
    if (nseq != last_rows) return nvr::fail(NVR_ERR_LEN_MISMATCH, "sample_tokens: %zu sequences but logits hold %zu rows", nseq, last_rows);
    int64_t *ht = ahead_tok[parity & 1];
    facts_kept[parity & 1] = facts_now();                                 // (called right behind the step's execute / execute_decode_ahead)
    for (size_t i = 0; i < nseq; ++i) ht[i] = INT64_MIN;                  // (the kernel below has not been enqueued yet)
    ht[max_seqs] = 0;
    // greedy_sample, sampler.rs:109-112: token ids to the pinned host buffer (device-visible mapping) AND to the next decode
    // step's input ids on the device
    // ... and (same launch) the rows this step's LM head read to lm_snap[parity]: the logits of this step stay reproducible while the
    // next one, launched ahead, overwrites the hidden rows (present_step / ensure_logits)
    const int64_t row_bytes = Hd * 2 * em;
    if (tp == 1) return KD(argmax_partials(d_lm_pval, d_lm_pidx, lm_parts, (int64_t)nseq, ahead_tok_dev[parity & 1], nullptr, 0, stream, dd_ids, nullptr,
                                           lm_input, row_bytes, lm_snap[parity & 1], row_bytes));
    // vocabulary shards (embed_head.rs:321-336): this rank's (max, global arg-max) records -> all-gather through the peer arenas ->
    // the same rank-ordered merge on every rank, all stream-ordered (no host round trip: the next step can be enqueued behind it)
    if (!ahead_ok(nseq)) return nvr::fail(NVR_ERR_UNSUPPORTED, "sample_launch: %zu rows do not fit the peer-to-peer all-gather", nseq);
    RC(KD(argmax_partials(d_lm_pval, d_lm_pidx, lm_parts, (int64_t)nseq, d_tok, d_maxval, vocab_start, stream, nullptr, d_rec,
                          lm_input, row_bytes, lm_snap[parity & 1], row_bytes)));
    RC(comm.all_gather_bytes(d_rec, d_gather_rec, nseq * sizeof(k::TpArgmaxRec), stream));
    return KD(tp_argmax_merge(d_gather_rec, (int)tp, (int64_t)nseq, ahead_tok_dev[parity & 1], dd_ids, comm.p2p_words ? comm.p2p_words + 2 : nullptr,
                              ahead_tok_dev[parity & 1] + max_seqs, stream));
}

// The step the engine hands back to its caller is not always the step the runner executed last (launch-ahead: its successor is already
// enqueued, and has reused the hidden rows and, unless lazy, the logits buffer).  The logits accessors must refer to the step just returned
// (ModelRunner::execute_model returns that step's logits, model_runner.rs:105-128; LLMEngine::step, llm_engine.rs:155-197): they are
// recomputed on demand from the rows sample_launch kept for that step — same kernel, same rows, same bits as the step's own LM head.
void nvr_model_runner::present_step(int parity, size_t rows) {
    if (!lm_snap[parity & 1]) return;
    lm_input = lm_snap[parity & 1]; logits_valid = false; last_rows = rows;
    facts_shown = facts_kept[parity & 1]; facts_shown_valid = true;
}

int nvr_model_runner::sample_wait(size_t nseq, int parity, int64_t *out) {
    const volatile int64_t *ht = ahead_tok[parity & 1];
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < nseq; ++i) {
        unsigned spins = 0;
        while (ht[i] == INT64_MIN) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
                hipError_t e = hipStreamQuery(stream);
                return nvr::fail(NVR_ERR_HIP, "sample_wait: no tokens after 30 s (stream: %s)", hipGetErrorString(e));
            }
        }
        out[i] = ht[i];
    }
    if (tp > 1 && ht[max_seqs] != 0) {                                   // a collective of this step gave up on a peer: zeros stood in for its sums
        (void)hipMemsetAsync(comm.p2p_words + 2, 0, 4, stream);
        return nvr::fail(NVR_ERR_RCCL, "peer-to-peer all-reduce: a peer did not arrive (epoch %ld)", (long)ht[max_seqs]);
    }
    return NVR_OK;
}

// Diagnostic: the decode graph of the last step launched n more times back to back on the runner's stream, with no host round
// trip in between (same inputs: the step is recomputed in place, K/V rows are rewritten with the same values).  Timing this
// against n engine steps separates the GPU's launch chain from the host gap between steps.
int nvr_model_runner::replay_last_decode_graph(int n) {
    NVR_HIP_CHECK(hipSetDevice(device));
    if (!last_decode_graph) return nvr::fail(NVR_ERR_INVALID_ARG, "replay_last_decode_graph: no captured decode step yet");
    for (auto &kv : graphs) if (kv.second == (hipGraphExec_t)last_decode_graph) {
        for (int i = 0; i < n; ++i) NVR_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)last_decode_graph, stream));
        return NVR_OK;
    }
    return nvr::fail(NVR_ERR_INVALID_ARG, "replay_last_decode_graph: the graph was evicted");
}

int nvr_model_runner::set_p2p_fenced(bool on) {
    NVR_HIP_CHECK(hipSetDevice(device));
    if (comm.p2p_fenced == on) return NVR_OK;
    NVR_HIP_CHECK(hipStreamSynchronize(stream));
    for (auto &kv : graphs) hipGraphExecDestroy(kv.second);               // their kernel nodes carry P2PArgs::fenced of the old setting
    graphs.clear(); last_decode_graph = nullptr;
    comm.p2p_fenced = on;
    return rearm_tickets();
}

// The split-KV attention's arrival counters must read zero between launches; each launch's last arriver re-arms its own.  After anything that may
// have cut a launch short (a failed step, a reset of the collectives after a timed-out peer, a protocol switch) they are zeroed wholesale.
int nvr_model_runner::rearm_tickets() {
    if (!attn_tickets) return NVR_OK;
    NVR_HIP_CHECK(hipMemsetAsync(attn_tickets, 0, (size_t)(max_seqs * KVH) * sizeof(unsigned int), stream));
    return NVR_OK;
}

// all-reduce and all-gather of a known pattern on this runner's communicator (tests; collective over all ranks)
int nvr_model_runner::comm_selftest() {
    NVR_HIP_CHECK(hipSetDevice(device));
    if (tp > 1) RC(comm.prepare());
    if (!comm.comm && !comm.local && !comm.p2p_ready) return nvr::fail(NVR_ERR_RCCL, "comm_selftest: communicator not initialised");
    // (test hook: the collectives below still run, every rank alike; counted on runners whose peer arenas are attached — the self-test inside
    //  nvr_runner_init_comm, before the arenas exist, is not one of the calls)
    const bool inject = comm.p2p_ready && comm_selftest_calls++ < env.selftest_inject;
    const int n = 4096;
    std::vector<uint16_t> hbuf(n);
    for (int i = 0; i < n; ++i) hbuf[i] = bf16 ? 0x3F80 : 0x3C00; // 1.0 in the runner's 16-bit type
    NVR_HIP_CHECK(hipMemcpyAsync(proj, hbuf.data(), n * 2, hipMemcpyHostToDevice, stream));
    RC(comm.all_reduce_sum_f16(proj, n, stream));
    NVR_HIP_CHECK(hipMemcpyAsync(hbuf.data(), proj, n * 2, hipMemcpyDeviceToHost, stream));
    NVR_HIP_CHECK(hipStreamSynchronize(stream));
    const uint16_t want_h[9] = {0, 0x3C00, 0x4000, 0x4200, 0x4400, 0x4500, 0x4600, 0x4700, 0x4800};   // fp16 of 0..8
    const uint16_t want_b[9] = {0, 0x3F80, 0x4000, 0x4040, 0x4080, 0x40A0, 0x40C0, 0x40E0, 0x4100};   // bf16 of 0..8
    const uint16_t *want = bf16 ? want_b : want_h;
    for (int i = 0; i < n; ++i)
        if (hbuf[i] != want[comm.nranks]) return nvr::fail(NVR_ERR_RCCL, "comm_selftest: all-reduce gave 0x%04x at %d, want 0x%04x", hbuf[i], i, want[comm.nranks]);
    if (tp > 1) {
        // ... and 48 more rounds whose payload CHANGES every round and differs per rank (small integers: exact in fp16 and bf16): the one-shot
        // exchange reuses its two slot parities, so a rank that read a peer's slot before the peer's write-through stores had landed (the failure a
        // fence-free protocol would show on links that do not order payload and flag) returns the round-before-last's value and fails here, at
        // communicator set-up, not as diverging token streams later
        auto enc = [&](int v) -> uint16_t { return bf16 ? f32_to_bf16_bits((float)v) : f32_to_f16_bits((float)v); };
        for (int round = 0; round < 48; ++round) {
            for (int i = 0; i < n; ++i) hbuf[i] = enc((round * 5 + (int)rank * 3 + i % 7) % 29);
            NVR_HIP_CHECK(hipMemcpyAsync(proj, hbuf.data(), n * 2, hipMemcpyHostToDevice, stream));
            RC(comm.all_reduce_sum_f16(proj, n, stream));
            NVR_HIP_CHECK(hipMemcpyAsync(hbuf.data(), proj, n * 2, hipMemcpyDeviceToHost, stream));
            NVR_HIP_CHECK(hipStreamSynchronize(stream));
            for (int i = 0; i < n; ++i) {
                int sum = 0;
                for (int64_t r = 0; r < tp; ++r) sum += (round * 5 + (int)r * 3 + i % 7) % 29;
                if (hbuf[i] != enc(sum))
                    return nvr::fail(NVR_ERR_RCCL, "comm_selftest: round %d: all-reduce gave 0x%04x at %d, want %d (a peer's slot was read before its data had landed?)", round, hbuf[i], i, sum);
            }
        }
    }
    if (tp > 1) {
        const int64_t nr = std::min<int64_t>(8, max_seqs);                 // the token buffers hold max_num_seqs entries per rank
        std::vector<int64_t> mine(nr, rank), all(nr * tp, -1);
        NVR_HIP_CHECK(hipMemcpyAsync(d_tok, mine.data(), nr * 8, hipMemcpyHostToDevice, stream));
        RC(comm.all_gather_bytes(d_tok, d_gather_idx, (size_t)nr * 8, stream));
        NVR_HIP_CHECK(hipMemcpyAsync(all.data(), d_gather_idx, nr * 8 * tp, hipMemcpyDeviceToHost, stream));
        NVR_HIP_CHECK(hipStreamSynchronize(stream));
        for (int64_t r = 0; r < tp; ++r)
            for (int64_t j = 0; j < nr; ++j)
                if (all[r * nr + j] != r) return nvr::fail(NVR_ERR_RCCL, "comm_selftest: all-gather slot %ld holds %ld", (long)r, (long)all[r * nr + j]);
    }
    RC(comm.p2p_check_error(stream));
    if (tp > 1 && comm.p2p_ready && !f32) {
        // ... and what a decode step does to the arenas: its LARGEST message (max_num_seqs rows x hidden), R collectives BACK TO BACK with
        // no host synchronisation in between (consecutive epochs: both slot parities, a peer up to two collectives ahead), alternating the fused
        // all-reduce + residual + RMSNorm launch of the decode graph and the plain all-reduce, every round's payload its own (small integers: sums
        // exact in fp16 and bf16), all results checked after ONE synchronisation at the end.  A protocol that lets a rank read a slot before the
        // peer's payload has landed, or lets a peer overwrite a slot that is still being read, returns another round's values here.
        const int64_t rows = std::min<int64_t>(max_seqs, (int64_t)(nvr::Comm::kP2PSlotBytes / 2) / Hd);
        const int64_t R = std::min<int64_t>(32, max_tokens / std::max<int64_t>(rows, 1));
        if (rows >= 1 && R >= 2 && Hd % 4 == 0) {
            const size_t ne = (size_t)(rows * Hd);
            auto enc = [&](int v) -> uint16_t { return bf16 ? f32_to_bf16_bits((float)v) : f32_to_f16_bits((float)v); };
            auto val = [&](int64_t round, int64_t rk, size_t i) { return (int)((round * 5 + rk * 3 + (int64_t)(i % 7) + (int64_t)(i / (size_t)Hd)) % 29); };
            std::vector<uint16_t> hin((size_t)R * ne), hout((size_t)R * ne), ones((size_t)Hd, enc(1));
            for (int64_t r2 = 0; r2 < R; ++r2) for (size_t i = 0; i < ne; ++i) hin[(size_t)r2 * ne + i] = enc(val(r2, rank, i));
            // inputs in proj, the residual rows of the fused rounds in h (zero: h <- 0 + sum), the norm output in n, norm weights in act (all activations: idle now)
            NVR_HIP_CHECK(hipMemcpyAsync(proj, hin.data(), hin.size() * 2, hipMemcpyHostToDevice, stream));
            NVR_HIP_CHECK(hipMemsetAsync(h, 0, (size_t)R * ne * 2, stream));
            NVR_HIP_CHECK(hipMemcpyAsync(act, ones.data(), ones.size() * 2, hipMemcpyHostToDevice, stream));
            for (int64_t r2 = 0; r2 < R; ++r2) {
                uint16_t *in_r = proj + (size_t)r2 * ne, *h_r = h + (size_t)r2 * ne;
                if (r2 & 1) RC(comm.all_reduce_add_rmsnorm(in_r, h_r, act, mc.rms_norm_eps, (int)rows, (int)Hd, this->n, stream));
                else RC(comm.all_reduce_sum_f16(in_r, ne, stream));
            }
            // plain rounds leave their sums in proj, fused rounds in h
            NVR_HIP_CHECK(hipMemcpyAsync(hin.data(), proj, hin.size() * 2, hipMemcpyDeviceToHost, stream));
            NVR_HIP_CHECK(hipMemcpyAsync(hout.data(), h, hout.size() * 2, hipMemcpyDeviceToHost, stream));
            NVR_HIP_CHECK(hipStreamSynchronize(stream));
            RC(comm.p2p_check_error(stream));
            for (int64_t r2 = 0; r2 < R; ++r2)
                for (size_t i = 0; i < ne; ++i) {
                    int sum = 0;
                    for (int64_t rk = 0; rk < tp; ++rk) sum += val(r2, rk, i);
                    const uint16_t got = (r2 & 1) ? hout[(size_t)r2 * ne + i] : hin[(size_t)r2 * ne + i];
                    if (got != enc(sum))
                        return nvr::fail(NVR_ERR_RCCL, "comm_selftest: back-to-back round %ld of %ld (%s, %s protocol, %ld rows x %ld): element %zu is 0x%04x, want %d — a slot was read "
                                         "before its payload had landed, or overwritten while it was read", (long)r2, (long)R, (r2 & 1) ? "all-reduce + residual + norm" : "plain all-reduce",
                                         comm.p2p_fenced ? "fenced" : "fence-free", (long)rows, (long)Hd, i, got, sum);
                }
        }
    }
    if (inject) return nvr::fail(NVR_ERR_RCCL, "comm_selftest: failure injected by NVR_SELFTEST_INJECT (call %d)", comm_selftest_calls);
    return NVR_OK;
}

// sample_tokens :131-156 -> Sampler::batch_sample, src/layers/sampler.rs:221-254
// the f32 logits of the last step, if that step skipped their stores: the LM-head launch is repeated on the same
// (still resident) hidden rows with the stores on — same kernel, same bits
int nvr_model_runner::ensure_logits() {
    if (logits_valid) return NVR_OK;
    NVR_HIP_CHECK(hipSetDevice(device));
    int32_t np = 0;
    RC(KD(lm_head(lm_input, Hd, lm_head, (int64_t)last_rows, Hd, Vl, logits, d_lm_pval, d_lm_pidx, &np, stream, true,
                  (tiled_weights && last_rows <= 32) ? lm_head_t : nullptr)));
    logits_valid = true;
    return NVR_OK;
}

int nvr_model_runner::sample(nvr_seq *const *seqs, size_t nseq, int64_t *out) {
    NVR_HIP_CHECK(hipSetDevice(device));
    if (nseq != last_rows) return nvr::fail(NVR_ERR_LEN_MISMATCH, "sample_tokens: %zu sequences but logits hold %zu rows", nseq, last_rows);
    bool all_greedy = true;
    for (size_t i = 0; i < nseq; ++i) all_greedy &= (seqs[i]->sampling.temperature == 0.0f);
    const int64_t B = (int64_t)nseq;
    if (all_greedy) {
        if (!comm.active()) {
            if (lm_parts > 0 && h_tok_dev) {
                // the merge kernel writes the token ids straight into the pinned host buffer (device-visible mapping): no
                // device-to-host copy (a blit kernel of its own) between the last kernel of the step and the host's wake-up
                RC(KD(argmax_partials(d_lm_pval, d_lm_pidx, lm_parts, B, h_tok_dev, nullptr, 0, stream)));
                NVR_HIP_CHECK(hipStreamSynchronize(stream));
                std::memcpy(out, h_tok, B * 8);
                return NVR_OK;
            }
            if (lm_parts > 0) RC(KD(argmax_partials(d_lm_pval, d_lm_pidx, lm_parts, B, d_tok, nullptr, 0, stream)));
            else RC(KD(argmax(logits, B, Vl, d_tok, nullptr, 0, stream)));
        } else {
            // vocab-sharded greedy (embed_head.rs:321-336): all-gather (max, argmax) pairs, then lowest index wins
            if (lm_parts > 0) RC(KD(argmax_partials(d_lm_pval, d_lm_pidx, lm_parts, B, d_tok, d_maxval, vocab_start, stream)));
            else RC(KD(argmax(logits, B, Vl, d_tok, d_maxval, vocab_start, stream)));
            RC(comm.all_gather_bytes(d_maxval, d_gather_val, (size_t)B * 4, stream));
            RC(comm.all_gather_bytes(d_tok, d_gather_idx, (size_t)B * 8, stream));
            std::vector<float> gv(tp * B); std::vector<int64_t> gi(tp * B);
            NVR_HIP_CHECK(hipMemcpyAsync(gv.data(), d_gather_val, gv.size() * 4, hipMemcpyDeviceToHost, stream));
            NVR_HIP_CHECK(hipMemcpyAsync(gi.data(), d_gather_idx, gi.size() * 8, hipMemcpyDeviceToHost, stream));
            RC(comm.p2p_check_error(stream));                            // (synchronises) a peer that never arrived this step
            NVR_HIP_CHECK(hipStreamSynchronize(stream));
            for (int64_t b = 0; b < B; ++b) {
                float bv = gv[b]; int64_t bi = gi[b];
                for (int64_t r = 1; r < tp; ++r) {
                    float v = gv[r * B + b]; int64_t i = gi[r * B + b];
                    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
                }
                out[b] = bi;
            }
            return NVR_OK;
        }
    } else {
        RC(ensure_logits());
        const float *lg = logits; int64_t Vs = Vl; void *ws = sample_ws;
        if (comm.active()) {
            // gather the vocab shards on every rank (embed_head.rs:321-336) and sample identically everywhere (counter RNG)
            if (V % tp) return nvr::fail(NVR_ERR_UNSUPPORTED, "stochastic sampling under tensor parallelism needs vocab %% tp == 0");
            if (!d_full_logits) {
                RC(dmalloc(&d_gather_logits, tp * max_seqs * Vl)); RC(dmalloc(&d_full_logits, max_seqs * V));
                NVR_HIP_CHECK(hipMalloc(&sample_ws_full, KD(sample_workspace_bytes(max_seqs, V))));
            }
            RC(comm.all_gather_bytes(logits, d_gather_logits, (size_t)(B * Vl) * 4, stream));
            RC(comm.p2p_check_error(stream));                            // (synchronises) a peer that never arrived this step
            RC(KD(concat_vocab_shards(d_gather_logits, tp, B, Vl, d_full_logits, stream)));
            lg = d_full_logits; Vs = V; ws = sample_ws_full;
        }
        int64_t *tk = (int64_t *)samp_host; uint64_t *ky = (uint64_t *)(samp_host + max_seqs * 8);      // 8-byte arrays first
        float *t = (float *)(samp_host + max_seqs * 16), *tpv = (float *)(samp_host + max_seqs * 20);
        for (size_t i = 0; i < nseq; ++i) {
            const nvr_sampling_params &sp = seqs[i]->sampling;
            t[i] = sp.temperature;
            tk[i] = sp.has_top_k ? (int64_t)sp.top_k : 0;                                   // A-18
            // sampler.rs:233-240: in a batch where any row sets top_p the others get unwrap_or(1.0) and are filtered with p = 1.0,
            // which keeps every token in exact arithmetic: p >= 1.0 is "no filter" here (decision A-26, tested against the
            // oracle's literal restatement)
            tpv[i] = (sp.has_top_p && sp.top_p < 1.0f) ? sp.top_p : -1.0f;
            ky[i] = nvr_sample_key(cfg.sample_seed, seqs[i]->seq_id, seqs[i]->num_completion_tokens());   // A-20
        }
        NVR_HIP_CHECK(hipMemcpyAsync(d_temp, t, B * 4, hipMemcpyHostToDevice, stream));
        NVR_HIP_CHECK(hipMemcpyAsync(d_topk, tk, B * 8, hipMemcpyHostToDevice, stream));
        NVR_HIP_CHECK(hipMemcpyAsync(d_topp, tpv, B * 4, hipMemcpyHostToDevice, stream));
        NVR_HIP_CHECK(hipMemcpyAsync(d_keys, ky, B * 8, hipMemcpyHostToDevice, stream));
        RC(KD(sample(lg, B, Vs, d_temp, d_topk, d_topp, d_keys, d_tok, ws, stream, false)));
    }
    NVR_HIP_CHECK(hipMemcpyAsync(h_tok, d_tok, B * 8, hipMemcpyDeviceToHost, stream));      // to_vec1 :152
    NVR_HIP_CHECK(hipStreamSynchronize(stream));
    std::memcpy(out, h_tok, B * 8);
    return NVR_OK;
}

// kernel_decls.h — the launchers of one dtype build of the kernels (included by kernels.h once per namespace: nvr::k = fp16,
// nvr::kb = bfloat16; NO include guard on purpose).
namespace nvr { namespace NVR_KDECL_NS {
using namespace kt;

int embedding(const int64_t *ids, int64_t T, const half_bits *E, int64_t Hd, half_bits *out, hipStream_t s);
// decode-sized steps (T <= 64): embedding + the first layer's input RMSNorm in one launch (same arithmetic as the two kernels)
bool embedding_rmsnorm_ok(int64_t T, int64_t Hd);
int embedding_rmsnorm(const int64_t *ids, int64_t T, const half_bits *E, const half_bits *w, float eps, int64_t Hd, half_bits *h,
                      half_bits *out, hipStream_t s);
int rmsnorm(const half_bits *x, const half_bits *w, float eps, int64_t T, int64_t Hd, half_bits *out, hipStream_t s);
int add_rmsnorm(half_bits *h, const half_bits *y, const half_bits *w, float eps, int64_t T, int64_t Hd,
                half_bits *out, hipStream_t s);
int silu_and_mul(const half_bits *x, int64_t T, int64_t I, half_bits *out, hipStream_t s);
// Activation::forward (activation.rs:147-159): kind 0 silu, 1 gelu (tanh form), 2 relu: [T, cols] -> [T, cols]; 3 SiluAndMul, 4 GeluAndMul: [T, cols] -> [T, cols / 2]
int activation(int kind, const half_bits *x, int64_t T, int64_t cols, half_bits *out, hipStream_t s);
int add_bias(half_bits *y, const half_bits *b, int64_t T, int64_t N, hipStream_t s);   // y[T, N] <- 16bit(y + b[N]) (use_bias, A-30)
int select_last_tokens(const half_bits *h, const int32_t *cu, int64_t B, int64_t Hd, half_bits *out, hipStream_t s);
int rope_store_kv(half_bits *qkv, const int64_t *positions, const int32_t *slots, int64_t T, int64_t H, int64_t KVH,
                  int64_t D, const float *cos_t, const float *sin_t, half_bits *k_cache, half_bits *v_cache,
                  hipStream_t s, const half_bits *q_norm_w = nullptr, const half_bits *k_norm_w = nullptr, float eps = 0.f);
// argmax over f32 rows; out_val (nullable) receives the row maxima, idx_offset is added to indices
int argmax(const float *logits, int64_t B, int64_t V, int64_t *out_idx, float *out_val, int64_t idx_offset,
           hipStream_t s);
int fill_weight(half_bits *dst, int64_t rows, int64_t cols, int64_t ld, int64_t global_cols, int64_t row0,
                int64_t col0, uint64_t key, float scale, hipStream_t s);
int fill_const(half_bits *dst, int64_t n, float v, hipStream_t s);
// tiled copy of a row-major [N][K] weight for the decode kernels: [N/16][K/32][16][32]; mode 1 = the qkv row order of the RoPE epilogue
int retile_weight(const half_bits *src, half_bits *dst, int64_t N, int64_t K, int mode, int64_t H, int64_t KVH, int64_t D, hipStream_t s);
int concat_vocab_shards(const float *gathered, int64_t tp, int64_t B, int64_t Vl, float *full, hipStream_t s);

// y[T,N] = x[T,K] (row stride ldx) · W[N,K]^T, f32 accumulate on MFMA; y fp16 or f32
// Wt (optional): the tiled copy of W (retile_weight mode 0 / 1): read instead of W when the weight-streaming kernel takes the shape
int linear(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, void *y,
           bool y_f32, hipStream_t s, const half_bits *Wt = nullptr);

// LM head for decode-sized batches (T <= 32, K <= 2048): f32 logits + per-workgroup greedy arg-max partials
// ([*nparts][T] values and vocabulary indices, *nparts <= LM_HEAD_MAX_PARTS), finished by argmax_partials
// (lowest index wins ties; idx_offset is added; out_val nullable)
bool lm_head_ok(int64_t T, int64_t K, int64_t N, int64_t ldx);
int32_t lm_head_parts(int64_t T, int64_t K, int64_t N, int64_t ldx);      // partials lm_head will write (0: unsupported shape)
// store_logits = false: only the partials are written (a greedy batch never reads its 4·T·N logit bytes)
int lm_head(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, float *logits,
            float *part_val, int32_t *part_idx, int32_t *nparts, hipStream_t s, bool store_logits = true, const half_bits *Wt = nullptr);
// snap_dst (nullable): row m of snap_src (snap_ld_bytes apart) is copied to snap_dst + m * snap_row_bytes by the same launch (16-byte granular)
int argmax_partials(const float *part_val, const int32_t *part_idx, int32_t nparts, int64_t T, int64_t *out_idx, float *out_val,
                    int64_t idx_offset, hipStream_t s, int64_t *out_idx2 = nullptr, TpArgmaxRec *out_rec = nullptr, const void *snap_src = nullptr,
                    int64_t snap_ld_bytes = 0, void *snap_dst = nullptr, int64_t snap_row_bytes = 0);
// cross-rank merge of gathered records recs[tp][B] (largest value, lowest index on ties, rank order): token ids to out_host (and out_dev),
// *err (the collectives' error word, nullable) to *err_out (nullable) before token 0
int tp_argmax_merge(const TpArgmaxRec *recs, int tp, int64_t B, int64_t *out_host, int64_t *out_dev, const unsigned int *err, int64_t *err_out,
                    hipStream_t s);

int64_t stream_row_limit();    // rows up to which the weight-streaming kernels are preferred over the LDS-tiled GEMM (linear.hip)
bool splitk_prefers_tiles(int64_t T, int64_t K, int64_t N);   // linear_splitk takes the LDS-tiled kernel (gemm_tiled_splitk) for this shape
int linear_splitk(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, int64_t S,
                  float *slabs, hipStream_t s, const half_bits *Wt = nullptr);
// h = fp16(h + fp16(sum_z slabs[z])), out = rmsnorm(h)*w
int add_rmsnorm_slabs(half_bits *h, const float *slabs, int64_t S, const half_bits *w, float eps, int64_t T, int64_t Hd,
                      half_bits *out, hipStream_t s);

// fused decode epilogues (same GEMM kernel): gate_up -> SiluAndMul, W [2I,K] -> out [T,I];
// qkv -> RoPE(q,k) + KV store, W [(H+2KVH)D, K] -> qkv [T,(H+2KVH)D] (roped q,k; v) and cache rows at slots
int linear_silu_mul(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t I,
                    half_bits *out, hipStream_t s, const half_bits *Wt = nullptr);
int linear_qkv_rope_store(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t H,
                          int64_t KVH, int64_t D, const int64_t *positions, const int32_t *slots, const float *cos_t,
                          const float *sin_t, half_bits *qkv, half_bits *k_cache, half_bits *v_cache, hipStream_t s,
                          const half_bits *Wt = nullptr, bool kv_cache_only = false);
// kv_cache_only: the caller's attention reads K / V from the caches only, so the k and v columns of qkv need not be written (the 256^2
// prefill kernel then skips those stores: a third of its output bytes; the other routes write them anyway)

int decode_splitk_slices(int64_t T, int64_t K, int64_t N);     // k-slices of linear_splitk for the N = hidden GEMMs of a decode-sized step

// decode GEMMs over large weights (T <= 32, K >= 2048, >= 24 MiB of weights): activation block in LDS, persistent workgroups
// (linear_stream.hip); linear / linear_silu_mul / linear_qkv_rope_store route here when the shape test passes
int linear_stream_prepare();                                                // LDS opt-in of every instance (call outside captures)
bool linear_stream_ok(int64_t T, int64_t K, int64_t N, int64_t ldx);
bool linear_stream_silu_ok(int64_t T, int64_t K, int64_t I, int64_t ldx);
bool linear_stream_rope_ok(int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D, int64_t ldx);
int linear_stream(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, half_bits *y, hipStream_t s,
                  const half_bits *Wt = nullptr);
int linear_stream_silu_mul(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t I, half_bits *out, hipStream_t s,
                           const half_bits *Wt = nullptr);
int linear_stream_qkv_rope_store(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D,
                                 const int64_t *positions, const int32_t *slots, const float *cos_t, const float *sin_t, half_bits *qkv,
                                 half_bits *k_cache, half_bits *v_cache, hipStream_t s, const half_bits *Wt = nullptr);

// LM head over more than 32 rows: the 128x128 kernel with the logits / arg-max epilogue (one partial per 128-column tile)
bool gemm_tiled_splitk_ok(int64_t T, int64_t K, int64_t N, int64_t S, int64_t ldx);
int gemm_tiled_splitk(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, int64_t S, float *slabs, hipStream_t s);
int gemm_tiled_prepare();                                                   // LDS opt-in of the ring instances (call outside captures)
bool gemm_tiled_lm_head_ok(int64_t T, int64_t K, int64_t N, int64_t ldx);
int gemm_tiled_lm_head(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, float *logits,
                       float *part_val, int32_t *part_idx, int32_t *nparts, hipStream_t s);

// LDS-tiled MFMA GEMM for the prefill regime (T >= 128): same results layout and epilogues as the kernels above
bool gemm_tiled_ok(int64_t T, int64_t K, int64_t N, int64_t ldx);
int gemm_tiled(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, half_bits *y, hipStream_t s);
int gemm_tiled_silu_mul(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t I, half_bits *out,
                        hipStream_t s);
int gemm_tiled_qkv_rope_store(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t H, int64_t KVH,
                              int64_t D, const int64_t *positions, const int32_t *slots, const float *cos_t, const float *sin_t,
                              half_bits *qkv, half_bits *k_cache, half_bits *v_cache, hipStream_t s);

// 256x256x64 eight-wave MFMA GEMM (T >= 256, N % 256 == 0 / I % 128 == 0 / whole heads per 256 rows): same results
// layout and epilogues; preferred over gemm_tiled when its shape test passes
bool gemm256_ok(int64_t T, int64_t K, int64_t N, int64_t ldx);
bool gemm256_silu_ok(int64_t T, int64_t K, int64_t I, int64_t ldx);
bool gemm256_rope_ok(int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D, int64_t ldx);
int gemm256(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, half_bits *y, hipStream_t s);
int gemm256_resid(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, half_bits *h, hipStream_t s);
// LM head over >= 192 rows on the 256x256 tiles (f32 logits on demand + one arg-max partial per 256-column tile; any N % 16 == 0)
bool gemm256_lm_head_ok(int64_t T, int64_t K, int64_t N, int64_t ldx);
int gemm256_lm_head(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, float *logits, float *part_val,
                    int32_t *part_idx, int32_t *nparts, hipStream_t s);
bool gemm256_preferred(int64_t T, int64_t K, int64_t N, int64_t ldx);     // linear() would take the 256x256 kernel for this shape
int gemm256_silu_mul(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t I, half_bits *out, hipStream_t s);
int gemm256_qkv_rope_store(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D,
                           const int64_t *positions, const int32_t *slots, const float *cos_t, const float *sin_t, half_bits *qkv,
                           half_bits *k_cache, half_bits *v_cache, hipStream_t s, bool kv_cache_only = false);

size_t attn_workspace_bytes(int64_t nq, int64_t H, int64_t D, int64_t max_ctx);
int attention(const AttnArgs &a, bool paged, hipStream_t s);

bool flash_prefill_ok(int D, int H, int KVH);
int flash_tile_positions(int H, int KVH);
int flash_prefill(const FlashArgs &a, bool paged, hipStream_t s);
int flash_shared_prefix(const half_bits *q, int64_t ldq, const half_bits *k_cache, const half_bits *v_cache, const int32_t *block_tables,
                        int32_t max_blocks, int32_t block_size, int32_t nq, int32_t H, int32_t KVH, int32_t D, float scale,
                        int32_t part_len, int32_t sparts, int32_t num_parts, float *part_o, float *part_ml, hipStream_t s,
                        const int32_t *rows = nullptr, const int32_t *count = nullptr);

// sampler (top-k / top-p / gumbel)
size_t sample_workspace_bytes(int64_t B, int64_t V);
int sample(const float *logits, int64_t B, int64_t V, const float *temperature, const int64_t *top_k,
           const float *top_p, const uint64_t *keys, int64_t *out_ids, void *workspace, hipStream_t s,
           bool store_filtered = true);   // store_filtered: write the filtered, temperature-scaled rows to the workspace (tests)


}}  // namespace nvr::k / nvr::kb

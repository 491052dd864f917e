/*
 * nvr_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the arithmetic on the reference's paged-attention
 * prefill/decode hot path (ssvgopal/nano-vllm-rs @ 2025-07-18).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (nano-vllm-rs_amd/csrc) never links or calls it.
 *
 * Pinning status: the reference cannot be built or run here
 * (no Rust toolchain; manifest + source errors; all arithmetic delegated to the
 * un-vendored candle-core/candle-nn "0.8" and xxhash-rust "0.8" crates, no
 * Cargo.lock).  The oracle is therefore pinned by
 *   (i)  every known-answer value in the reference's own in-file unit tests
 *        (tests/test_oracle_kat.py),
 *   (ii) the public XXH64 specification vectors (xxhash-rust 0.8 implements
 *        XXH64 of the xxHash spec; cross-checked against python-xxhash),
 *   (iii) independent numpy/torch restatements of each float op.
 * Whole-model logits/token ids are *parity unpinned* by the reference (its LM
 * head re-randomises its weight per call, src/layers/embed_head.rs:309-318).
 *
 * Every function cites the reference file:line it follows.  Ambiguities are
 * resolved per SURVEY.md Appendix A (A-n tags below).
 *
 * Numeric convention: all tensors are f32 arrays.  "fp16-faithful" mode is
 * obtained by the caller rounding op outputs through nvo_round_f16_array(),
 * mirroring the rounding points of the fp16 GPU path (f32 accumulate inside
 * an op, fp16 storage between ops).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NVO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* fp16 helpers (IEEE binary16, round-to-nearest-even)                        */
/* ------------------------------------------------------------------------- */
static inline uint16_t f32_to_f16_bits(float f) {
    return (uint16_t)_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
}
static inline float f16_bits_to_f32(uint16_t h) { return _cvtsh_ss(h); }

NVO_API float nvo_round_f16(float x) { return f16_bits_to_f32(f32_to_f16_bits(x)); }

NVO_API void nvo_round_f16_copy(const float *x, float *y, size_t n) {     /* y = f32(f16_rne(x)); x == y allowed */
    const size_t n8 = n / 8;
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n8; ++i)
        _mm256_storeu_ps(y + i * 8, _mm256_cvtph_ps(_mm256_cvtps_ph(_mm256_loadu_ps(x + i * 8), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)));
    for (size_t i = n8 * 8; i < n; ++i) y[i] = f16_bits_to_f32(f32_to_f16_bits(x[i]));
}
NVO_API void nvo_round_f16_array(float *x, size_t n) { nvo_round_f16_copy(x, x, n); }
/* y = f32(bf16_rne(x)): round to nearest even at 8 significant bits (the 16-bit type of Config.dtype = "bfloat16", reference
 * src/config.rs:51,113-116); finite inputs; x == y allowed.  Integer form: add 0x7FFF + the lowest kept bit, clear the low half. */
NVO_API void nvo_round_bf16_copy(const float *x, float *y, size_t n) {
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; ++i) {
        uint32_t u; memcpy(&u, x + i, 4);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
        memcpy(y + i, &u, 4);
    }
}
NVO_API void nvo_f32_to_f16(const float *x, uint16_t *y, size_t n) {
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; ++i) y[i] = f32_to_f16_bits(x[i]);
}
NVO_API void nvo_f16_to_f32(const uint16_t *x, float *y, size_t n) {
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; ++i) y[i] = f16_bits_to_f32(x[i]);
}

NVO_API int nvo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
NVO_API void nvo_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------- */
/* XXH64 (public xxHash specification; the algorithm behind                   */
/* xxhash_rust::xxh64::xxh64 used at src/engine/block_manager.rs:7,122)       */
/* ------------------------------------------------------------------------- */
#define XP1 0x9E3779B185EBCA87ULL
#define XP2 0xC2B2AE3D27D4EB4FULL
#define XP3 0x165667B19E3779F9ULL
#define XP4 0x85EBCA77C2B2AE63ULL
#define XP5 0x27D4EB2F165667C5ULL
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xround(uint64_t acc, uint64_t in) {
    acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1;
}
static inline uint64_t xmerge(uint64_t acc, uint64_t v) {
    acc ^= xround(0, v); return acc * XP1 + XP4;
}
NVO_API uint64_t nvo_xxh64(const void *data, size_t len, uint64_t seed) {
    const uint8_t *p = (const uint8_t *)data, *end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        const uint8_t *lim = end - 32;
        do {
            v1 = xround(v1, rd64(p)); v2 = xround(v2, rd64(p + 8));
            v3 = xround(v3, rd64(p + 16)); v4 = xround(v4, rd64(p + 24));
            p += 32;
        } while (p <= lim);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = seed + XP5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rotl64(h, 27) * XP1 + XP4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * XP1; h = rotl64(h, 23) * XP2 + XP3; p += 4; }
    while (p < end) { h ^= (*p) * XP5; h = rotl64(h, 11) * XP1; ++p; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

/* BlockManager::compute_hash, src/engine/block_manager.rs:109-123 (A-3):
 * bytes = [prefix_hash as 8 B LE]? ++ each token as i64 8 B LE; xxh64 seed 0.
 * (x86-64 is little-endian, so the in-memory i64 array is the byte string.) */
NVO_API uint64_t nvo_block_hash(const int64_t *tokens, size_t n, int has_prefix, uint64_t prefix) {
    size_t len = (has_prefix ? 8 : 0) + 8 * n;
    uint8_t *buf = (uint8_t *)malloc(len ? len : 1);
    size_t off = 0;
    if (has_prefix) { memcpy(buf, &prefix, 8); off = 8; }
    memcpy(buf + off, tokens, 8 * n);
    uint64_t h = nvo_xxh64(buf, len, 0);
    free(buf);
    return h;
}

/* ------------------------------------------------------------------------- */
/* Synthetic weights (counter-based PRNG, N(0, std^2)-like,    */
/* identical bits on host and device: integer hash + one f32 multiply).       */
/* value(key, idx) = (sum of the four u16 lanes of splitmix64(key ^ idx)      */
/*                    - 131070) * scale,  scale = std / 37837.227...          */
/* (Irwin–Hall(4): variance of the lane sum = 4*(65536^2-1)/12.)              */
/* ------------------------------------------------------------------------- */
NVO_API uint64_t nvo_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
NVO_API uint64_t nvo_weight_key(uint64_t seed, uint64_t tensor_id) {
    return nvo_splitmix64(nvo_splitmix64(seed) + tensor_id * 0xD1B54A32D192ED03ULL);
}
NVO_API float nvo_weight_scale(double std) {
    return (float)(std / sqrt(4.0 * (65536.0 * 65536.0 - 1.0) / 12.0));
}
static inline float weight_value(uint64_t key, uint64_t idx, float scale) {
    uint64_t r = nvo_splitmix64(key ^ idx);
    int32_t s = (int32_t)(r & 0xFFFF) + (int32_t)((r >> 16) & 0xFFFF) +
                (int32_t)((r >> 32) & 0xFFFF) + (int32_t)((r >> 48) & 0xFFFF) - 131070;
    return (float)s * scale;
}
/* dst[i*ld + j] = value(key, (row0+i)*global_cols + (col0+j)), optionally fp16-rounded. */
NVO_API void nvo_fill_weight(float *dst, int64_t rows, int64_t cols, int64_t ld,
                             int64_t global_cols, int64_t row0, int64_t col0,
                             uint64_t key, float scale, int round_f16) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; ++i)
        for (int64_t j = 0; j < cols; ++j) {
            float v = weight_value(key, (uint64_t)((row0 + i) * global_cols + (col0 + j)), scale);
            dst[i * ld + j] = round_f16 ? f16_bits_to_f32(f32_to_f16_bits(v)) : v;
        }
}
/* Deterministic token ids uniform in [0, vocab): SURVEY.md §8d prompts. */
NVO_API void nvo_fill_tokens(int64_t *dst, int64_t n, uint64_t seed, uint64_t stream, int64_t vocab) {
    uint64_t key = nvo_weight_key(seed, stream);
    for (int64_t i = 0; i < n; ++i) dst[i] = (int64_t)(nvo_splitmix64(key ^ (uint64_t)i) % (uint64_t)vocab);
}

/* ------------------------------------------------------------------------- */
/* K1 embedding gather — src/layers/embed_head.rs:77-97                       */
/* ------------------------------------------------------------------------- */
NVO_API void nvo_embedding(const int64_t *ids, int64_t T, const float *E, int64_t Hd, float *out) {
#pragma omp parallel for schedule(static) if (T > 16)
    for (int64_t t = 0; t < T; ++t) memcpy(out + t * Hd, E + ids[t] * Hd, sizeof(float) * Hd);
}
/* Vocab-parallel mask + local index, embed_head.rs:100-127:
 * mask = (id >= start) & (id < end); local = mask ? id - start : 0 */
NVO_API void nvo_vocab_mask_local(const int64_t *ids, int64_t T, int64_t start, int64_t end,
                                  int32_t *mask, int64_t *local) {
    for (int64_t t = 0; t < T; ++t) {
        int m = ids[t] >= start && ids[t] < end;
        mask[t] = m; local[t] = m ? ids[t] - start : 0;
    }
}

/* ------------------------------------------------------------------------- */
/* K2 RMSNorm — src/layers/layernorm.rs:58-75 (forward_simple), A-13          */
/*   rms = sqrt(mean(x^2) + eps); out = (x / rms) * w, all in f32             */
/* ------------------------------------------------------------------------- */
NVO_API void nvo_rmsnorm(const float *x, const float *w, float eps, int64_t T, int64_t Hd, float *out) {
#pragma omp parallel for schedule(static) if (T > 8)
    for (int64_t t = 0; t < T; ++t) {
        const float *xr = x + t * Hd; float *o = out + t * Hd;
        double ss = 0.0;
        for (int64_t i = 0; i < Hd; ++i) ss += (double)xr[i] * (double)xr[i];
        float rms = sqrtf((float)(ss / (double)Hd) + eps);
        for (int64_t i = 0; i < Hd; ++i) o[i] = (xr[i] / rms) * w[i];
    }
}
/* residual add — src/models/qwen3.rs:382,389.  round_f16: fp16 add semantics. */
NVO_API void nvo_add(const float *a, const float *b, size_t n, float *out, int round_f16) {
#pragma omp parallel for schedule(static) if (n > 65536)
    for (size_t i = 0; i < n; ++i) {
        float v = a[i] + b[i];
        out[i] = round_f16 ? f16_bits_to_f32(f32_to_f16_bits(v)) : v;
    }
}

/* ------------------------------------------------------------------------- */
/* K3/K10/K12/K14/K16 Linear y = x · Wᵀ (+b) — src/layers/linear.rs:54,143,   */
/* 228-239,354-356,437-439; src/layers/embed_head.rs:292-306.                  */
/* x [T,K], W [N,K] (row-major, K contiguous), y [T,N]; f32 accumulate.       */
/* ------------------------------------------------------------------------- */
static inline float dot_f32(const float *a, const float *b, int64_t K) {
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
    __m256 acc2 = _mm256_setzero_ps(), acc3 = _mm256_setzero_ps();
    int64_t k = 0;
    for (; k + 32 <= K; k += 32) {
        acc0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k), _mm256_loadu_ps(b + k), acc0);
        acc1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k + 8), _mm256_loadu_ps(b + k + 8), acc1);
        acc2 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k + 16), _mm256_loadu_ps(b + k + 16), acc2);
        acc3 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k + 24), _mm256_loadu_ps(b + k + 24), acc3);
    }
    for (; k + 8 <= K; k += 8)
        acc0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + k), _mm256_loadu_ps(b + k), acc0);
    acc0 = _mm256_add_ps(_mm256_add_ps(acc0, acc1), _mm256_add_ps(acc2, acc3));
    float tmp[8]; _mm256_storeu_ps(tmp, acc0);
    float s = ((tmp[0] + tmp[4]) + (tmp[1] + tmp[5])) + ((tmp[2] + tmp[6]) + (tmp[3] + tmp[7]));
    for (; k < K; ++k) s += a[k] * b[k];
    return s;
}
NVO_API void nvo_linear(const float *x, const float *W, const float *bias,
                        int64_t T, int64_t K, int64_t N, float *y) {
    /* Every output is one 8-lane fmadd chain over k (in k order) followed by a fixed reduction tree, whatever
     * the blocking: 4 rows of x x 2 rows of W per register block; (t, n) blocks sized so that a block of x rows
     * stays in the core's L2 while the W rows of the block stream past it (BASELINE-size prefills, T = 32768). */
    if (T < 4) {
#pragma omp parallel for schedule(static)
        for (int64_t n = 0; n < N; ++n)
            for (int64_t t = 0; t < T; ++t) y[t * N + n] = dot_f32(x + t * K, W + n * K, K) + (bias ? bias[n] : 0.0f);
        return;
    }
    int64_t TBLK = (256 * 1024) / (K * 4); TBLK = TBLK < 4 ? 4 : TBLK / 4 * 4;
    const int64_t NBLK = 32;
    const int64_t ntb = (T + TBLK - 1) / TBLK, nnb = (N + NBLK - 1) / NBLK;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int64_t tb = 0; tb < ntb; ++tb)
      for (int64_t nb = 0; nb < nnb; ++nb) {
        const int64_t t_end = (tb + 1) * TBLK < T ? (tb + 1) * TBLK : T, n_end = (nb + 1) * NBLK < N ? (nb + 1) * NBLK : N;
        for (int64_t n = nb * NBLK; n < n_end; n += 2) {
            const int two = n + 1 < n_end;
            const float *w0 = W + n * K, *w1 = two ? w0 + K : w0;
            const float b0 = bias ? bias[n] : 0.0f, b1 = (bias && two) ? bias[n + 1] : 0.0f;
            int64_t t = tb * TBLK;
            for (; t + 4 <= t_end; t += 4) {
                __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0, c0 = a0, c1 = a0, c2 = a0, c3 = a0;
                const float *x0 = x + t * K, *x1 = x0 + K, *x2 = x1 + K, *x3 = x2 + K;
                int64_t k = 0;
                for (; k + 8 <= K; k += 8) {
                    const __m256 wv = _mm256_loadu_ps(w0 + k), wu = _mm256_loadu_ps(w1 + k);
                    const __m256 v0 = _mm256_loadu_ps(x0 + k), v1 = _mm256_loadu_ps(x1 + k);
                    const __m256 v2 = _mm256_loadu_ps(x2 + k), v3 = _mm256_loadu_ps(x3 + k);
                    a0 = _mm256_fmadd_ps(v0, wv, a0); a1 = _mm256_fmadd_ps(v1, wv, a1);
                    a2 = _mm256_fmadd_ps(v2, wv, a2); a3 = _mm256_fmadd_ps(v3, wv, a3);
                    c0 = _mm256_fmadd_ps(v0, wu, c0); c1 = _mm256_fmadd_ps(v1, wu, c1);
                    c2 = _mm256_fmadd_ps(v2, wu, c2); c3 = _mm256_fmadd_ps(v3, wu, c3);
                }
                float r[8][8];
                _mm256_storeu_ps(r[0], a0); _mm256_storeu_ps(r[1], a1); _mm256_storeu_ps(r[2], a2); _mm256_storeu_ps(r[3], a3);
                _mm256_storeu_ps(r[4], c0); _mm256_storeu_ps(r[5], c1); _mm256_storeu_ps(r[6], c2); _mm256_storeu_ps(r[7], c3);
                for (int i = 0; i < (two ? 8 : 4); ++i) {
                    float s = ((r[i][0] + r[i][4]) + (r[i][1] + r[i][5])) + ((r[i][2] + r[i][6]) + (r[i][3] + r[i][7]));
                    const float *xi = x + (t + (i & 3)) * K, *w = i < 4 ? w0 : w1;
                    for (int64_t kk = k; kk < K; ++kk) s += xi[kk] * w[kk];
                    y[(t + (i & 3)) * N + n + (i >> 2)] = s + (i < 4 ? b0 : b1);
                }
            }
            for (; t < t_end; ++t) {
                y[t * N + n] = dot_f32(x + t * K, w0, K) + b0;
                if (two) y[t * N + n + 1] = dot_f32(x + t * K, w1, K) + b1;
            }
        }
      }
}

/* ------------------------------------------------------------------------- */
/* K5 RoPE — src/layers/rotary_embedding.rs:23-48,74-158 (A-14)               */
/* inv_freq[j] = 1 / theta^(2j/D) in f64 -> f32; angle = (f32)pos * inv_freq  */
/* (f32 multiply); cos/sin evaluated in f64 on the f32 angle, rounded to f32. */
/* Tables [max_pos, D/2].                                                     */
/* ------------------------------------------------------------------------- */
NVO_API void nvo_rope_table(int64_t D, int64_t max_pos, double theta, float *cos_t, float *sin_t) {
    int64_t half = D / 2;
#pragma omp parallel for schedule(static) if (max_pos > 256)
    for (int64_t p = 0; p < max_pos; ++p)
        for (int64_t j = 0; j < half; ++j) {
            float inv = (float)(1.0 / pow(theta, (double)(2 * j) / (double)D));
            float ang = (float)p * inv;
            cos_t[p * half + j] = (float)cos((double)ang);
            sin_t[p * half + j] = (float)sin((double)ang);
        }
}
/* apply_rotary_emb_single, rotary_embedding.rs:23-48: halves split at D/2
 * (NeoX style): out1 = x1*c - x2*s ; out2 = x2*c + x1*s.  x [T, nh, D] in place. */
NVO_API void nvo_rope_apply(float *x, const int64_t *pos, int64_t T, int64_t nh, int64_t D,
                            const float *cos_t, const float *sin_t) {
    int64_t half = D / 2;
#pragma omp parallel for schedule(static) if (T > 16)
    for (int64_t t = 0; t < T; ++t) {
        const float *c = cos_t + pos[t] * half, *s = sin_t + pos[t] * half;
        for (int64_t h = 0; h < nh; ++h) {
            float *v = x + (t * nh + h) * D;
            for (int64_t j = 0; j < half; ++j) {
                float x1 = v[j], x2 = v[j + half];
                v[j] = x1 * c[j] - x2 * s[j];
                v[j + half] = x2 * c[j] + x1 * s[j];
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* K6 KV store — src/layers/attention.rs:150-174 (A-6):                       */
/* cache [NB, bs, KVH, D] viewed as [NB*bs, KVH*D]; cache[slot[t]] = kv[t].   */
/* slot < 0 skips the token.                                                  */
/* ------------------------------------------------------------------------- */
NVO_API void nvo_kv_store(const float *k, const float *v, const int32_t *slots, int64_t T,
                          int64_t row /* KVH*D */, float *k_cache, float *v_cache) {
    for (int64_t t = 0; t < T; ++t) {
        if (slots[t] < 0) continue;
        memcpy(k_cache + (int64_t)slots[t] * row, k + t * row, sizeof(float) * row);
        memcpy(v_cache + (int64_t)slots[t] * row, v + t * row, sizeof(float) * row);
    }
}

/* ------------------------------------------------------------------------- */
/* Attention core for one (query row, head): softmax_f32(q·K^T*scale)·V        */
/* compute_attention, src/layers/attention.rs:238-261 (A-9); GQA mapping       */
/* g(h) = h / (H/KVH), attention.rs:419-435.                                   */
/* Keys are fetched through a callback-free "row pointer" convention:          */
/* key j of the sequence lives at kbase + krow_off[j] (floats).                */
/* ------------------------------------------------------------------------- */
static void attn_row(const float *q, int64_t D, float scale, int64_t nkeys,
                     const float *kbase, const float *vbase, const int64_t *row_off,
                     float *scores /* scratch nkeys */, float *out) {
    float m = -INFINITY;
    for (int64_t j = 0; j < nkeys; ++j) {
        float s = dot_f32(q, kbase + row_off[j], D) * scale;
        scores[j] = s; if (s > m) m = s;
    }
    double l = 0.0;
    for (int64_t j = 0; j < nkeys; ++j) { float e = expf(scores[j] - m); scores[j] = e; l += e; }
    float inv = (float)(1.0 / l);
    if (D % 64 == 0) {
        /* same arithmetic per element as the scalar loop below (out[d] = out[d] + p*vr[d], one f32 multiply and
         * one f32 add, keys in order), 64 output columns at a time held in registers */
        for (int64_t d0 = 0; d0 < D; d0 += 64) {
            __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
            for (int64_t j = 0; j < nkeys; ++j) {
                const __m256 p = _mm256_set1_ps(scores[j] * inv);
                const float *vr = vbase + row_off[j] + d0;
                a0 = _mm256_add_ps(a0, _mm256_mul_ps(p, _mm256_loadu_ps(vr)));
                a1 = _mm256_add_ps(a1, _mm256_mul_ps(p, _mm256_loadu_ps(vr + 8)));
                a2 = _mm256_add_ps(a2, _mm256_mul_ps(p, _mm256_loadu_ps(vr + 16)));
                a3 = _mm256_add_ps(a3, _mm256_mul_ps(p, _mm256_loadu_ps(vr + 24)));
                a4 = _mm256_add_ps(a4, _mm256_mul_ps(p, _mm256_loadu_ps(vr + 32)));
                a5 = _mm256_add_ps(a5, _mm256_mul_ps(p, _mm256_loadu_ps(vr + 40)));
                a6 = _mm256_add_ps(a6, _mm256_mul_ps(p, _mm256_loadu_ps(vr + 48)));
                a7 = _mm256_add_ps(a7, _mm256_mul_ps(p, _mm256_loadu_ps(vr + 56)));
            }
            _mm256_storeu_ps(out + d0, a0); _mm256_storeu_ps(out + d0 + 8, a1); _mm256_storeu_ps(out + d0 + 16, a2);
            _mm256_storeu_ps(out + d0 + 24, a3); _mm256_storeu_ps(out + d0 + 32, a4); _mm256_storeu_ps(out + d0 + 40, a5);
            _mm256_storeu_ps(out + d0 + 48, a6); _mm256_storeu_ps(out + d0 + 56, a7);
        }
        return;
    }
    for (int64_t d = 0; d < D; ++d) out[d] = 0.0f;
    for (int64_t j = 0; j < nkeys; ++j) {
        float p = scores[j] * inv;
        const float *vr = vbase + row_off[j];
        for (int64_t d = 0; d < D; ++d) out[d] += p * vr[d];
    }
}

/* K7 varlen causal prefill — flash_attention_varlen, attention.rs:177-208 with
 * the causal mask of :321-339.  q [T,H,D], k,v [T,KVH,D], cu_seqlens [B+1]. */
NVO_API void nvo_attn_prefill_varlen(const float *q, const float *k, const float *v,
                                     const int32_t *cu_seqlens, int64_t B, int64_t H, int64_t KVH,
                                     int64_t D, float scale, float *out) {
    int64_t group = H / KVH, max_len = 0;
    for (int64_t b = 0; b < B; ++b) {
        int64_t L = cu_seqlens[b + 1] - cu_seqlens[b]; if (L > max_len) max_len = L;
    }
#pragma omp parallel
    {
        float *scores = (float *)malloc(sizeof(float) * (max_len ? max_len : 1));
        int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (max_len ? max_len : 1));
        for (int64_t b = 0; b < B; ++b) {
            int64_t s0 = cu_seqlens[b], L = cu_seqlens[b + 1] - s0;
#pragma omp for collapse(2) schedule(dynamic, 4) nowait
            for (int64_t i = 0; i < L; ++i)
                for (int64_t h = 0; h < H; ++h) {
                    int64_t g = h / group;
                    for (int64_t j = 0; j <= i; ++j) off[j] = ((s0 + j) * KVH + g) * D;
                    attn_row(q + ((s0 + i) * H + h) * D, D, scale, i + 1, k, v, off, scores,
                             out + ((s0 + i) * H + h) * D);
                }
        }
        free(scores); free(off);
    }
}

/* K8/K9 paged attention — flash_attention_varlen_with_cache / _decode,
 * attention.rs:211-235,264-318, with A-8: attend over exactly context_lens[b]
 * keys read through the block table.  Query i of sequence b (i in [0,nq_b))
 * sits at position context_lens[b]-nq_b+i and sees keys 0..=that position
 * (decode: nq_b = 1 -> all context_lens[b] keys, non-causal window, A-8).
 * q [Tq,H,D] with cu_seqlens_q [B+1]; caches [NB, bs, KVH, D];
 * block_tables [B, max_blocks] (-1 padded, model_runner.rs:283-290). */
NVO_API void nvo_attn_paged(const float *q, const int32_t *cu_seqlens_q,
                            const float *k_cache, const float *v_cache,
                            const int32_t *block_tables, int64_t max_blocks,
                            const int32_t *context_lens, int64_t B, int64_t H, int64_t KVH,
                            int64_t D, int64_t bs, float scale, float *out) {
    int64_t group = H / KVH, max_ctx = 0;
    for (int64_t b = 0; b < B; ++b) if (context_lens[b] > max_ctx) max_ctx = context_lens[b];
#pragma omp parallel
    {
        float *scores = (float *)malloc(sizeof(float) * (max_ctx ? max_ctx : 1));
        int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (max_ctx ? max_ctx : 1));
        for (int64_t b = 0; b < B; ++b) {
            int64_t q0 = cu_seqlens_q[b], nq = cu_seqlens_q[b + 1] - q0, ctx = context_lens[b];
#pragma omp for collapse(2) schedule(dynamic, 1) nowait
            for (int64_t i = 0; i < nq; ++i)
                for (int64_t h = 0; h < H; ++h) {
                    int64_t g = h / group, nkeys = ctx - nq + i + 1;
                    for (int64_t j = 0; j < nkeys; ++j) {
                        int64_t blk = block_tables[b * max_blocks + j / bs];
                        off[j] = ((blk * bs + j % bs) * KVH + g) * D;
                    }
                    attn_row(q + ((q0 + i) * H + h) * D, D, scale, nkeys, k_cache, v_cache, off,
                             scores, out + ((q0 + i) * H + h) * D);
                }
        }
        free(scores); free(off);
    }
}

/* ------------------------------------------------------------------------- */
/* K13 SiluAndMul — src/layers/activation.rs:12-15,46-63                      */
/* x [T, 2I] -> out [T, I]: silu(x[:, :I]) * x[:, I:], silu(x) = x*sigmoid(x) */
/* ------------------------------------------------------------------------- */
NVO_API void nvo_silu_and_mul(const float *x, int64_t T, int64_t I, float *out) {
#pragma omp parallel for schedule(static) if (T > 8)
    for (int64_t t = 0; t < T; ++t)
        for (int64_t i = 0; i < I; ++i) {
            float g = x[t * 2 * I + i], u = x[t * 2 * I + I + i];
            float sg = 1.0f / (1.0f + expf(-g));
            out[t * I + i] = (g * sg) * u;
        }
}

/* ------------------------------------------------------------------------- */
/* The rest of src/layers/activation.rs: silu :12-15, gelu :20-22 (candle's   */
/* tanh form), relu :25-27, GeluAndMul :74-100, Activation::forward :147-159. */
/* kind 0 silu, 1 gelu, 2 relu: out [T, cols]; 3 SiluAndMul, 4 GeluAndMul:    */
/* out [T, cols / 2] = act(x[:, :cols/2]) * x[:, cols/2:]                     */
/* ------------------------------------------------------------------------- */
static float nvo_act_one(int kind, float g) {
    if (kind == 1 || kind == 4) { const float inner = 0.7978845608028654f * g * (1.0f + 0.044715f * g * g); return 0.5f * g * (1.0f + tanhf(inner)); }
    if (kind == 2) return g > 0.0f ? g : 0.0f;
    return g * (1.0f / (1.0f + expf(-g)));
}
NVO_API void nvo_activation(int kind, const float *x, int64_t T, int64_t cols, float *out) {
    const int64_t co = kind >= 3 ? cols / 2 : cols;
    for (int64_t t = 0; t < T; ++t)
        for (int64_t i = 0; i < co; ++i) {
            const float a = nvo_act_one(kind, x[t * cols + i]);
            out[t * co + i] = kind >= 3 ? a * x[t * cols + co + i] : a;
        }
}

/* ------------------------------------------------------------------------- */
/* K17 greedy argmax — src/layers/sampler.rs:109-112, A-12 lowest index wins  */
/* ------------------------------------------------------------------------- */
NVO_API int64_t nvo_argmax(const float *x, int64_t n) {
    int64_t best = 0; float bv = x[0];
    for (int64_t i = 1; i < n; ++i) if (x[i] > bv) { bv = x[i]; best = i; }
    return best;
}

/* K18 — src/layers/sampler.rs:115-148 apply_top_k.  Stable descending sort,
 * keep first k (ties at the threshold beyond the k-th sorted entry are
 * dropped because only .take(k) entries are visited).  k==0 => disabled (A-18). */
typedef struct { float v; int64_t i; } vi_t;
static int cmp_desc_stable(const void *a, const void *b) {
    const vi_t *x = (const vi_t *)a, *y = (const vi_t *)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return (x->i > y->i) - (x->i < y->i); /* stable: lower index first */
}
NVO_API void nvo_top_k(const float *logits, int64_t n, int64_t k, float *out) {
    if (k <= 0) { memcpy(out, logits, sizeof(float) * n); return; }
    if (k > n) k = n;
    vi_t *s = (vi_t *)malloc(sizeof(vi_t) * n);
    for (int64_t i = 0; i < n; ++i) { s[i].v = logits[i]; s[i].i = i; }
    qsort(s, n, sizeof(vi_t), cmp_desc_stable);
    for (int64_t i = 0; i < n; ++i) out[i] = -INFINITY;
    for (int64_t i = 0; i < k; ++i) out[s[i].i] = s[i].v;
    free(s);
}
/* src/layers/sampler.rs:151-188 apply_top_p (A-19): softmax, stable sort desc,
 * keep the prefix up to and including the first index where cumsum >= p. */
NVO_API void nvo_top_p(const float *logits, int64_t n, float p, float *out) {
    float m = -INFINITY;
    for (int64_t i = 0; i < n; ++i) if (logits[i] > m) m = logits[i];
    vi_t *s = (vi_t *)malloc(sizeof(vi_t) * n);
    double l = 0.0;
    for (int64_t i = 0; i < n; ++i) { float e = expf(logits[i] - m); s[i].v = e; s[i].i = i; l += e; }
    float inv = (float)(1.0 / l);
    for (int64_t i = 0; i < n; ++i) s[i].v *= inv;
    qsort(s, n, sizeof(vi_t), cmp_desc_stable);
    float cum = 0.0f; int64_t cutoff = n;
    for (int64_t i = 0; i < n; ++i) { cum += s[i].v; if (cum >= p) { cutoff = i + 1; break; } }
    for (int64_t i = 0; i < n; ++i) out[i] = -INFINITY;
    for (int64_t i = 0; i < cutoff; ++i) out[s[i].i] = logits[s[i].i];
    free(s);
}
/* src/layers/sampler.rs:191-218 Gumbel-max with A-20's counter RNG:
 * u_v = (top 24 bits of splitmix64(key ^ v) + 0.5) / 2^24 clamped to
 * [1e-8, 1-1e-8]; g = -log(-log(u)); token = argmax(logits + g). */
NVO_API uint64_t nvo_sample_key(uint64_t seed, uint64_t seq_id, uint64_t step) {
    return nvo_splitmix64(nvo_weight_key(seed, seq_id) + step * 0xA24BAED4963EE407ULL);
}
NVO_API float nvo_gumbel(uint64_t key, int64_t v) {
    uint64_t r = nvo_splitmix64(key ^ (uint64_t)v);
    float u = ((float)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);
    const float eps = 1e-8f;
    if (u < eps) u = eps;
    if (u > 1.0f - eps) u = 1.0f - eps;
    return -logf(-logf(u));
}
/* sample_single, src/layers/sampler.rs:71-106 */
NVO_API int64_t nvo_sample(const float *logits, int64_t n, float temperature, int64_t top_k,
                           float top_p, int has_top_p, uint64_t key) {
    if (temperature == 0.0f) return nvo_argmax(logits, n);
    float *a = (float *)malloc(sizeof(float) * n), *b = (float *)malloc(sizeof(float) * n);
    for (int64_t i = 0; i < n; ++i) a[i] = (temperature != 1.0f) ? logits[i] / temperature : logits[i];
    if (top_k > 0) { nvo_top_k(a, n, top_k, b); float *t = a; a = b; b = t; }
    if (has_top_p) { nvo_top_p(a, n, top_p, b); float *t = a; a = b; b = t; }
    int64_t best = -1; float bv = -INFINITY;
    for (int64_t i = 0; i < n; ++i) {
        if (a[i] == -INFINITY) continue;
        float g = a[i] + nvo_gumbel(key, i);
        if (best < 0 || g > bv) { bv = g; best = i; }
    }
    free(a); free(b);
    return best < 0 ? 0 : best;
}

// kernels.h — host-side launchers of the gfx950 kernels (one per kernel-shaped op site of the
// hot path).  All functions enqueue on `stream` and return a hipError_t-like int
// (0 = ok, NVR_ERR_UNSUPPORTED for shapes outside the kernels' contract).
//
// Every kernel source is compiled twice (kernels/device_utils.h): nvr::k holds the fp16 build, nvr::kb the bfloat16 build of the SAME
// launchers (kernel_decls.h, included once per namespace below); argument structs and constants are shared (nvr::kt).  Activations,
// weights and cache rows travel as raw 16-bit words (half_bits) in both.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvr { namespace kt {

typedef uint16_t half_bits;

constexpr int LM_HEAD_MAX_PARTS = 2048;

struct TpArgmaxRec { float val; int32_t pad; int64_t idx; };           // 16 B: a rank's (max, global arg-max) of one row

// Attention over rows of keys addressed either through a block table (paged) or contiguously.
struct AttnArgs {
    const half_bits *q; int64_t ldq;          // q[t] at q + t*ldq, heads contiguous [H, D]
    const half_bits *k, *v;                    // paged: caches [NB, bs, KVH, D]; contiguous: rows with stride ldkv
    int64_t ldkv;                              // contiguous only
    const int32_t *ctx_lens;                   // [nq] keys visible to query t
    const int32_t *seq_of_q;                   // paged: [nq] row of block_tables (nullptr = identity)
    const int32_t *kv_base;                    // contiguous: [nq] first key row of query t
    const int32_t *block_tables; int32_t max_blocks; int32_t block_size;
    int32_t nq, H, KVH, D;
    float scale;
    int32_t max_ctx;                           // upper bound of ctx_lens (grid sizing)
    half_bits *out;                            // [nq, H, D] fp16
    void *workspace;                           // split-KV partials (paged decode)
    size_t workspace_bytes;                    // size of `workspace` (0 = not checked)
    int32_t shared_len;                        // paged decode: > 0 = every query's first shared_len keys sit in the blocks of
                                               // block-table row 0 (a multiple of block_size): that prefix goes through the MFMA
                                               // kernel once for the whole batch (flash_shared_prefix), the rest per sequence
    // ... or only SOME queries' (device arrays, all three or none): shared_rows[0 .. *shared_count) are the member rows (the
    // first one's block table names the shared blocks), shared_kv0[t] = shared_len for members, 0 for the others
    const int32_t *shared_rows, *shared_kv0, *shared_count;
    // paged decode, split over partitions: [nq * KVH] arrival counters, ZERO between launches (the kernel re-arms them): the merge of a
    // (query, kv head)'s partitions then rides on the last partition workgroup to finish instead of a second launch; null = merge launch.
    // A caller that passes them also allows the other launch-free merge: behind a shared-prefix pass that covers the WHOLE batch, with one
    // own partition per pair, that partition's workgroup merges the pair (it is the last arriver by stream order; no counter is touched)
    unsigned int *tickets;
    // paged decode with `tickets`: > 0 = the caller knows the contexts are RAGGED (their sum is well below nq * max_ctx) and asks for the work-balanced form
    // (attn_share_kernel) with this many equal shares of all pairs' keys, wherever it can run — not only where the pair count asks for it
    int32_t balance_hint;
};

// MFMA flash prefill attention.  A tile = up to 64/G consecutive query positions of one sequence.
struct FlashTile { int32_t q_row0, nq, pos0, kv_ref; };   // first q row, #queries, absolute position of the first query,
                                                          // contiguous: first key row of the sequence / paged: block-table row
struct FlashArgs {
    const half_bits *q; int64_t ldq;
    const half_bits *k, *v; int64_t ldkv;                  // contiguous rows (stride ldkv) or paged caches [NB, bs, KVH, D]
    const int32_t *block_tables; int32_t max_blocks, block_size;
    const FlashTile *tiles; int32_t ntiles;                // device array
    int32_t H, KVH, D; float scale;
    half_bits *out;                                        // [rows, H, D]
};

// Config.dtype = "float32": attention arguments of the f32 path (kernels/f32_path.hip)
struct AttnArgsF {
    const float *q; int64_t ldq;
    const float *k, *v; int64_t ldkv;             // contiguous rows (stride ldkv) or paged caches [NB, bs, KVH, D]
    const int32_t *ctx_lens, *seq_of_q, *kv_base, *block_tables; int32_t max_blocks, block_size;
    int32_t nq, H, KVH, D; float scale; int32_t max_ctx;
    float *out;                                   // [nq, H, D]
};

}}  // namespace nvr::kt

// the f32 path (one build: kernels/f32_path.hip)
namespace nvr { namespace kf {
using kt::AttnArgsF;
int fill_weight(float *dst, int64_t rows, int64_t cols, int64_t ld, int64_t gcols, int64_t row0, int64_t col0, uint64_t key, float scale, hipStream_t s);
int fill_const(float *dst, int64_t n, float v, hipStream_t s);
int embedding(const int64_t *ids, int64_t T, const float *E, int64_t Hd, float *out, hipStream_t s);
int select_last_tokens(const float *h, const int32_t *cu, int64_t B, int64_t Hd, float *out, hipStream_t s);
int rmsnorm(const float *x, const float *w, float eps, int64_t T, int64_t Hd, float *out, hipStream_t s);
int add_rmsnorm(float *h, const float *y, const float *w, float eps, int64_t T, int64_t Hd, float *out, hipStream_t s);
int sum_ranks_add_rmsnorm(float *h, const float *parts, int nranks, int64_t stride, const float *w, float eps, int64_t T, int64_t Hd, float *out, hipStream_t s);
int linear(const float *x, int64_t ldx, const float *W, int64_t T, int64_t K, int64_t N, const float *bias, float *y, hipStream_t s);
int rope_store_kv(float *qkv, const int64_t *pos, const int32_t *slots, int64_t T, int64_t H, int64_t KVH, int64_t D, const float *cos_t,
                  const float *sin_t, float *kc, float *vc, const float *q_norm, const float *k_norm, float eps, hipStream_t s);
int silu_and_mul(const float *gu, int64_t T, int64_t I, float *out, hipStream_t s);
int activation(int kind, const float *x, int64_t T, int64_t cols, float *out, hipStream_t s);
bool linear_qkv_rope_ok(int64_t T, int64_t K, int64_t D, int64_t ldx);   // decode-sized steps, no q / k head norms: qkv projection + RoPE + KV store in one launch (the same bits as the two)
int linear_qkv_rope_store(const float *x, int64_t ldx, const float *W, int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D, const float *bias,
                          const int64_t *pos, const int32_t *slots, const float *cos_t, const float *sin_t, float *qkv, float *kc, float *vc, hipStream_t s);
// r05: the residual add + RMSNorm in front of a decode-sized consumer GEMV done by the GEMV's workgroups themselves (h_out: the new residual stream, a buffer
// other than h_in; the bits of add_rmsnorm followed by the consumer)
bool fused_norm_ok(int64_t T, int64_t K);
int add_norm_linear_silu_mul(const float *h_in, const float *y, const float *nw, float eps, float *h_out, const float *W, int64_t T, int64_t K, int64_t I,
                             const float *bias, float *act, hipStream_t s);
int add_norm_linear_qkv_rope_store(const float *h_in, const float *y, const float *nw, float eps, float *h_out, const float *W, int64_t T, int64_t K, int64_t H,
                                   int64_t KVH, int64_t D, const float *bias, const int64_t *pos, const int32_t *slots, const float *cos_t, const float *sin_t,
                                   float *qkv, float *kc, float *vc, hipStream_t s);
bool linear_silu_ok(int64_t T, int64_t K, int64_t ldx);       // decode-sized steps: gate_up projection + SiluAndMul in one launch (the same bits as the two)
int linear_silu_mul(const float *x, int64_t ldx, const float *W, int64_t T, int64_t K, int64_t I, const float *bias, float *act, hipStream_t s);
int attention(const AttnArgsF &a, bool paged, hipStream_t s);
int prepare();
}}

#define NVR_KDECL_NS k
#include "kernel_decls.h"
#undef NVR_KDECL_NS
#define NVR_KDECL_NS kb
#include "kernel_decls.h"
#undef NVR_KDECL_NS

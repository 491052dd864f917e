// lm_head.hip — logits = h_last · W_lmᵀ (K16) for decode-sized batches, with the greedy arg-max folded into the
// epilogue (K17's greedy branch).
// reference: ParallelLMHead::compute_logits src/layers/embed_head.rs:292-306 (f32 logits per SURVEY A-21);
// Sampler::sample greedy branch src/layers/sampler.rs:126-151 (argmax, lowest index on ties, A-17).
//
// Roofline: HBM.  Algorithmic bytes = 2·N·K (every weight byte once; 311 MB for Qwen3-0.6B) + 4·T·N logits out.
// The weight matrix is the largest single stream of a decode step, so the kernel is persistent and shaped like the
// decode attention: one workgroup per CU slot keeps the T x K activation block in LDS (XOR-swizzled 16-byte
// chunks: ds_read_b128 of 16 rows at one k offset is conflict-free) for its whole life; each WAVE owns whole
// 16-row weight tiles (tile j -> wave j % all_waves: neighbouring waves stream neighbouring 16·K·2-byte regions) and
// runs the full k loop alone: U row-chunk loads of 1 KiB in flight per wave (A operand straight from HBM to VGPRs,
// non-temporal), B operand from LDS, one v_mfma_f32_16x16x32_f16 per 16 tokens and k-step, no cross-wave
// reduction and no barrier inside the stream.
// Greedy arg-max: every lane keeps (best value, lowest index) of the logits it produced for its token; lanes, waves and
// finally workgroups are merged with "greater value, else lower index", the workgroup results go to a small
// partials array and argmax_partials finishes the job, instead of re-reading all 4·T·N logit bytes.
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"
#include <cstdio>
#include <cstdlib>

namespace nvr { namespace NVR_DT_NS {

__device__ __forceinline__ void take_better(float &bv, int &bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

template <int MT, int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64) void lm_head_kernel(const half_t *__restrict__ x, int64_t ldx,
                                                             const half_t *__restrict__ W, int T, int K, int N,
                                                             float *__restrict__ y, float *__restrict__ pval,
                                                             int32_t *__restrict__ pidx, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = MT * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int cpr = K / 8;                                       // 16-byte chunks per activation row
    fill_x_image<ROWS, WAVES * 64, 8>(smem, x, ldx, 0, cpr, T, tid);
    __syncthreads();

    const int ntiles = N / 16, nw = gridDim.x * WAVES;
    float best[MT]; int besti[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) { best[j] = -INFINITY; besti[j] = 0x7fffffff; }

    for (int tile = blockIdx.x * WAVES + wave; tile < ntiles; tile += nw) {
        // flags bit 1: W is the tiled copy [N/16][K/32][16][32] (retile_weight): 1 KiB contiguous per wave-instruction
        const bool tiled = flags & 2;
        const half_t *wr = tiled ? W + (int64_t)tile * (K / 32) * 512 + r * 32 + q * 8 : W + ((int64_t)tile * 16 + r) * K + q * 8;
        const int kmul = tiled ? 16 : 1;
        float4_t acc[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < K; k0 += 32 * U) {
            half8_t a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = __builtin_nontemporal_load(reinterpret_cast<const half8_t *>(wr + (int64_t)(k0 + u * 32) * kmul));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ch = (k0 >> 3) + u * 4 + q;
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int row = j * 16 + r;
                    const half8_t b = *reinterpret_cast<const half8_t *>(smem + ((int64_t)row * cpr + (ch ^ (r & 7))) * 16);
                    acc[j] = mfma16(a[u], b, acc[j]);
                }
            }
        }
        // C layout: rows n = q*4 + e (4 consecutive vocabulary entries), column = token r
        const int n0 = tile * 16 + q * 4;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = j * 16 + r;
            if (m < T && !(flags & 1)) *reinterpret_cast<float4_t *>(y + (int64_t)m * N + n0) = acc[j];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (acc[j][e] > best[j]) { best[j] = acc[j][e]; besti[j] = n0 + e; }     // tiles ascend: first maximum kept
        }
    }

    // lanes r, r+16, r+32, r+48 hold the same token; then the waves through LDS (the activation image is dead)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            const float v = __shfl_xor(best[j], o, 64);
            const int i = __shfl_xor(besti[j], o, 64);
            take_better(best[j], besti[j], v, i);
        }
    __syncthreads();
    float *sv = reinterpret_cast<float *>(smem);
    int *si = reinterpret_cast<int *>(smem + WAVES * ROWS * 4);
    if (q == 0) {
#pragma unroll
        for (int j = 0; j < MT; ++j) { sv[wave * ROWS + j * 16 + r] = best[j]; si[wave * ROWS + j * 16 + r] = besti[j]; }
    }
    __syncthreads();
    if (tid < T) {
        float bv = sv[tid]; int bi = si[tid];
        for (int w2 = 1; w2 < WAVES; ++w2) take_better(bv, bi, sv[w2 * ROWS + tid], si[w2 * ROWS + tid]);
        pval[(int64_t)blockIdx.x * T + tid] = bv;
        pidx[(int64_t)blockIdx.x * T + tid] = bi;
    }
}

// K > 2048 (hidden 4096-class models: 32 x K fp16 no longer fits the LDS): the activation block passes through LDS in chunks of KC
// columns and every wave keeps the f32 accumulators of ALL its tiles (at most TPW: tiles w, w + nw, ...) in registers across the
// chunks — the weights are still streamed exactly once, tile by tile inside a chunk, and the epilogue is the one above (logits on
// demand, arg-max partials, tiles ascending so that the first maximum is kept).
template <int MT, int WAVES, int U, int TPW>
__global__ __launch_bounds__(WAVES * 64) void lm_head_kchunk_kernel(const half_t *__restrict__ x, int64_t ldx,
                                                                    const half_t *__restrict__ W, int T, int K, int N, int KC,
                                                                    float *__restrict__ y, float *__restrict__ pval,
                                                                    int32_t *__restrict__ pidx, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = MT * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int cpr = KC / 8;
    const int ntiles = N / 16, nw = gridDim.x * WAVES, w0 = blockIdx.x * WAVES + wave;
    const bool tiled = flags & 2;
    const int kmul = tiled ? 16 : 1;
    float4_t acc[TPW][MT];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    for (int kc0 = 0; kc0 < K; kc0 += KC) {
        __syncthreads();                                                     // the previous chunk's readers are done
        fill_x_image<ROWS, WAVES * 64, 8>(smem, x, ldx, kc0, cpr, T, tid);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int tile = w0 + i * nw;
            if (tile >= ntiles) break;                                       // wave-uniform
            const half_t *wr = tiled ? W + ((int64_t)tile * (K / 32) + kc0 / 32) * 512 + r * 32 + q * 8 : W + ((int64_t)tile * 16 + r) * K + kc0 + q * 8;
            for (int k0 = 0; k0 < KC; k0 += 32 * U) {
                half8_t a[U];
#pragma unroll
                for (int u = 0; u < U; ++u) a[u] = __builtin_nontemporal_load(reinterpret_cast<const half8_t *>(wr + (int64_t)(k0 + u * 32) * kmul));
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ch = (k0 >> 3) + u * 4 + q;
#pragma unroll
                    for (int j = 0; j < MT; ++j) {
                        const int row = j * 16 + r;
                        const half8_t b = *reinterpret_cast<const half8_t *>(smem + ((int64_t)row * cpr + (ch ^ (r & 7))) * 16);
                        acc[i][j] = mfma16(a[u], b, acc[i][j]);
                    }
                }
            }
        }
    }
    float best[MT]; int besti[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) { best[j] = -INFINITY; besti[j] = 0x7fffffff; }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = w0 + i * nw;
        if (tile >= ntiles) break;
        const int n0 = tile * 16 + q * 4;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = j * 16 + r;
            if (m < T && !(flags & 1)) *reinterpret_cast<float4_t *>(y + (int64_t)m * N + n0) = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (acc[i][j][e] > best[j]) { best[j] = acc[i][j][e]; besti[j] = n0 + e; }
        }
    }
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            const float v = __shfl_xor(best[j], o, 64);
            const int i2 = __shfl_xor(besti[j], o, 64);
            take_better(best[j], besti[j], v, i2);
        }
    __syncthreads();
    float *sv = reinterpret_cast<float *>(smem);
    int *si = reinterpret_cast<int *>(smem + WAVES * ROWS * 4);
    if (q == 0) {
#pragma unroll
        for (int j = 0; j < MT; ++j) { sv[wave * ROWS + j * 16 + r] = best[j]; si[wave * ROWS + j * 16 + r] = besti[j]; }
    }
    __syncthreads();
    if (tid < T) {
        float bv = sv[tid]; int bi = si[tid];
        for (int w2 = 1; w2 < WAVES; ++w2) take_better(bv, bi, sv[w2 * ROWS + tid], si[w2 * ROWS + tid]);
        pval[(int64_t)blockIdx.x * T + tid] = bv;
        pidx[(int64_t)blockIdx.x * T + tid] = bi;
    }
}

__global__ __launch_bounds__(64) void argmax_partials_kernel(const float *__restrict__ pval, const int32_t *__restrict__ pidx,
                                                             int nparts, int T, int64_t *__restrict__ out_idx,
                                                             float *__restrict__ out_val, int64_t idx_offset, int64_t *__restrict__ out_idx2,
                                                             TpArgmaxRec *__restrict__ out_rec, const uint4 *__restrict__ snap_src, int64_t snap_ld16,
                                                             uint4 *__restrict__ snap_dst, int snap_chunks) {
    const int m = blockIdx.x, lane = threadIdx.x;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int p0 = lane; p0 < nparts; p0 += 256) {                 // 4 partials per lane requested together (256 partials: one round trip)
        float pv[4]; int pi[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int p = p0 + f * 64, pp = p < nparts ? p : nparts - 1;
            pv[f] = pval[(int64_t)pp * T + m]; pi[f] = pidx[(int64_t)pp * T + m];
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
            if (p0 + f * 64 < nparts) take_better(bv, bi, pv[f], pi[f]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v = __shfl_xor(bv, o, 64);
        const int i = __shfl_xor(bi, o, 64);
        take_better(bv, bi, v, i);
    }
    if (lane == 0) {
        const int64_t tok = (bi == 0x7fffffff ? 0 : (int64_t)bi) + idx_offset;
        out_idx[m] = tok;
        if (out_idx2) out_idx2[m] = tok;                         // e.g. the NEXT decode step's input ids, already on the device
        if (out_val) out_val[m] = bv;
        if (out_rec) { TpArgmaxRec rc; rc.val = bv; rc.pad = 0; rc.idx = tok; out_rec[m] = rc; }   // one record per row for the cross-rank merge
    }
    // launch-ahead (engine.cpp): row m of the LM head's input is kept per step in flight, so that the logits of the step the engine has
    // just handed back can still be produced after the next step has overwritten the hidden rows (ModelRunner::execute_model returns THAT
    // step's logits, model_runner.rs:105-128).  Behind the token store: the host's wake-up does not wait for it.
    if (snap_dst)
        for (int c = lane; c < snap_chunks; c += 64) snap_dst[(int64_t)m * snap_chunks + c] = snap_src[(int64_t)m * snap_ld16 + c];
}

// Vocabulary-sharded greedy sampling (ParallelLMHead::gather_logits + Sampler::greedy, reference src/layers/embed_head.rs:321-336,
// src/layers/sampler.rs:109-112): every rank holds every rank's (max, global arg-max) record of every row (all-gather); the merge takes
// the largest value, lowest index on ties, in RANK ORDER — the same decision on every rank — and writes the token to the host-visible
// buffer, to the next decode step's device-side input ids (launch-ahead) and the collectives' error word next to the tokens.
__global__ __launch_bounds__(256) void tp_argmax_merge_kernel(const TpArgmaxRec *__restrict__ recs, int tp, int B, int64_t *__restrict__ out_host,
                                                              int64_t *__restrict__ out_dev, const unsigned int *__restrict__ err,
                                                              int64_t *__restrict__ err_out) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b == 0 && err_out) {                                      // before token 0: a reader that has seen token 0 sees this word
        *err_out = err ? (int64_t)*err : 0;
        __threadfence_system();
    }
    if (b >= B) return;
    float bv = recs[b].val; int64_t bi = recs[b].idx;
    for (int r = 1; r < tp; ++r) {
        const float v = recs[(int64_t)r * B + b].val; const int64_t i = recs[(int64_t)r * B + b].idx;
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    if (out_dev) out_dev[b] = bi;
    out_host[b] = bi;
}

constexpr int LM_KC = 2048, LM_TPW = 8;                             // K-chunked form: columns per chunk (at most), tiles per wave held in registers
static bool lm_kchunk(int64_t K) { return K > LM_KC; }
// chunk width: K in the fewest equal chunks of <= 2048 columns that are multiples of 256 (4096 -> 2048, 5120 -> 1280, 6144 -> 2048); 0: none
static int lm_kc(int64_t K) {
    for (int64_t n = (K + LM_KC - 1) / LM_KC; n * 256 <= K; ++n)
        if (K % n == 0 && (K / n) % 256 == 0) return (int)(K / n);
    return 0;
}
bool lm_head_ok(int64_t T, int64_t K, int64_t N, int64_t ldx) {
    if (!(T >= 1 && T <= 32 && K % 256 == 0 && N % 16 == 0 && N >= 16 && ldx % 8 == 0 && N < (1ll << 31))) return false;
    if (!lm_kchunk(K)) return true;
    return K <= 16384 && lm_kc(K) > 0 && N / 16 <= 256ll * 8 * LM_TPW;        // every wave's tiles fit its accumulator registers
}

struct LmPlan { int mt, waves, U; int64_t nwg; };
static LmPlan lm_plan(int64_t T, int64_t K, int64_t N) {
    LmPlan pl;
    pl.mt = T <= 16 ? 1 : 2;
    const size_t lds = (size_t)pl.mt * 16 * (lm_kchunk(K) ? lm_kc(K) : K) * 2;
    int per_cu = 1;
    if ((size_t)per_cu * lds > 160 * 1024) per_cu = (int)(160 * 1024 / lds);
    pl.waves = 8; pl.U = 4;
    pl.nwg = 256ll * per_cu;
    const int64_t ntiles = N / 16;
    if (pl.nwg * pl.waves > ntiles) pl.nwg = (ntiles + pl.waves - 1) / pl.waves;
    if (pl.nwg > LM_HEAD_MAX_PARTS) pl.nwg = LM_HEAD_MAX_PARTS;
    return pl;
}
int32_t lm_head_parts(int64_t T, int64_t K, int64_t N, int64_t ldx) {
    if (lm_head_ok(T, K, N, ldx)) return (int32_t)lm_plan(T, K, N).nwg;
    if (gemm256_lm_head_ok(T, K, N, ldx)) return (int32_t)((N + 255) / 256);              // one partial per 256-column tile (>= 256 rows)
    if (gemm_tiled_lm_head_ok(T, K, N, ldx)) return (int32_t)((N + 127) / 128);          // one partial per 128-column tile
    return 0;
}

// every compiled (MT, WAVES, U) instance, as X(mt, waves, U)
#define NVR_LM_INSTANCES(X) X(1, 8, 4) X(2, 8, 4)

// > 64 KiB of dynamic LDS needs an opt-in per kernel; done for all instances on the first (never captured: a
// sequence is prefilled eagerly before any decode graph exists) call
static int lm_allow_big_lds() {
    static bool done = false;
    if (done) return 0;
#define NVR_LM_ATTR(MT_, WV_, U_)                                                                                     \
    {                                                                                                                 \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lm_head_kernel<MT_, WV_, U_>),             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                   \
        if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "lm_head: hipFuncSetAttribute: %s", hipGetErrorString(e)); \
    }
    NVR_LM_INSTANCES(NVR_LM_ATTR)
#undef NVR_LM_ATTR
    for (const void *f : {reinterpret_cast<const void *>(&lm_head_kchunk_kernel<1, 8, 4, LM_TPW>), reinterpret_cast<const void *>(&lm_head_kchunk_kernel<2, 8, 4, LM_TPW>)})
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return nvr::fail(NVR_ERR_HIP, "lm_head: hipFuncSetAttribute (K-chunked form)");
    done = true;
    return 0;
}

// logits[T,N] (f32) = x·Wᵀ and per-workgroup arg-max partials: part_val/part_idx [*nparts][T], *nparts <= LM_HEAD_MAX_PARTS
int lm_head(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, float *logits,
            float *part_val, int32_t *part_idx, int32_t *nparts, hipStream_t s, bool store_logits, const half_bits *Wt) {
    if (!lm_head_ok(T, K, N, ldx)) {
        if (gemm256_lm_head_ok(T, K, N, ldx))                          // >= 256 rows: 256x256 tiles (r06), same outputs
            return gemm256_lm_head(x, ldx, W, T, K, N, store_logits ? logits : nullptr, part_val, part_idx, nparts, s);
        if (gemm_tiled_lm_head_ok(T, K, N, ldx))                       // more than 32 rows: 128x128 tiles, same outputs
            return gemm_tiled_lm_head(x, ldx, W, T, K, N, store_logits ? logits : nullptr, part_val, part_idx, nparts, s);
        return nvr::fail(NVR_ERR_UNSUPPORTED, "lm_head: T=%ld K=%ld N=%ld (T <= 32: K multiple of 256 (above 2048: in equal chunks of <= 2048 that are multiples of 256); T > 32: K multiple of 64, "
                         "N <= 128 * %d; N multiple of 16)", (long)T, (long)K, (long)N, LM_HEAD_MAX_PARTS);
    }
    const LmPlan pl = lm_plan(T, K, N);
    const int mt = pl.mt, waves = pl.waves, U = pl.U;
    const int64_t nwg = pl.nwg;
    *nparts = (int32_t)nwg;
    const half_t *xx = (const half_t *)x, *ww = (const half_t *)(Wt ? Wt : W);
    if (int rc0 = lm_allow_big_lds()) return rc0;
    const size_t lds = (size_t)mt * 16 * (lm_kchunk(K) ? lm_kc(K) : K) * 2;
    bool launched = false;
    const int dbg_flags = (store_logits ? 0 : 1) | (Wt ? 2 : 0);   // bit 0: skip the f32 logit stores (arg-max partials only); bit 1: tiled W
    if (lm_kchunk(K)) {
        if (nwg * waves * LM_TPW < N / 16) return nvr::fail(NVR_ERR_INVARIANT, "lm_head: %ld tiles on %ld waves", (long)(N / 16), (long)(nwg * waves));
        if (mt == 1) lm_head_kchunk_kernel<1, 8, 4, LM_TPW><<<dim3((unsigned)nwg), dim3(512), lds, s>>>(xx, ldx, ww, (int)T, (int)K, (int)N, lm_kc(K), logits, part_val, part_idx, dbg_flags);
        else lm_head_kchunk_kernel<2, 8, 4, LM_TPW><<<dim3((unsigned)nwg), dim3(512), lds, s>>>(xx, ldx, ww, (int)T, (int)K, (int)N, lm_kc(K), logits, part_val, part_idx, dbg_flags);
        launched = true;
    }
#define NVR_LM(MT_, WV_, U_)                                                                                          \
    if (!launched && mt == MT_ && waves == WV_ && U == U_) {                                                          \
        lm_head_kernel<MT_, WV_, U_><<<dim3((unsigned)nwg), dim3(WV_ * 64), lds, s>>>(xx, ldx, ww, (int)T, (int)K, (int)N, logits, \
                                                                                     part_val, part_idx, dbg_flags);  \
        launched = true;                                                                                              \
    }
    NVR_LM_INSTANCES(NVR_LM)
#undef NVR_LM
    if (!launched) return nvr::fail(NVR_ERR_UNSUPPORTED, "lm_head: variant U=%d waves=%d is not compiled in", U, waves);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "lm_head launch failed: %s", hipGetErrorString(e));
    return 0;
}

int tp_argmax_merge(const TpArgmaxRec *recs, int tp, int64_t B, int64_t *out_host, int64_t *out_dev, const unsigned int *err, int64_t *err_out,
                    hipStream_t s) {
    if (B == 0) return 0;
    tp_argmax_merge_kernel<<<dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s>>>(recs, tp, (int)B, out_host, out_dev, err, err_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "tp_argmax_merge launch failed: %s", hipGetErrorString(e));
    return 0;
}

int argmax_partials(const float *part_val, const int32_t *part_idx, int32_t nparts, int64_t T, int64_t *out_idx, float *out_val,
                    int64_t idx_offset, hipStream_t s, int64_t *out_idx2, TpArgmaxRec *out_rec, const void *snap_src, int64_t snap_ld_bytes,
                    void *snap_dst, int64_t snap_row_bytes) {
    if (T == 0) return 0;
    if (nparts < 1) return nvr::fail(NVR_ERR_INVALID_ARG, "argmax_partials: nparts=%d", nparts);
    if (snap_dst && (!snap_src || snap_row_bytes % 16 || snap_ld_bytes % 16 || ((uintptr_t)snap_src | (uintptr_t)snap_dst) % 16))
        return nvr::fail(NVR_ERR_INVALID_ARG, "argmax_partials: row snapshot of %ld bytes (stride %ld) is not 16-byte granular", (long)snap_row_bytes, (long)snap_ld_bytes);
    argmax_partials_kernel<<<dim3((unsigned)T), dim3(64), 0, s>>>(part_val, part_idx, nparts, (int)T, out_idx, out_val, idx_offset, out_idx2, out_rec,
                                                                  (const uint4 *)snap_src, snap_ld_bytes / 16, (uint4 *)snap_dst, (int)(snap_row_bytes / 16));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "argmax_partials launch failed: %s", hipGetErrorString(e));
    return 0;
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
